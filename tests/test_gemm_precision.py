"""GPU tests of the split-fp16 decoder contractions (recoder_amd/csrc/gemm.hip, PREC_H3).

decode (O = Z . W_de[T]^T, reference nn.py:271-280) and its backward dZ = dO . W_de[T]
multiply fp16 hi+lo pairs of the fp32 operands on the f16 MFMA pipe.  These tests pin
  * the accuracy of that arithmetic against a float64 product over the magnitudes the
    power-of-two operand scales are specified for (include/recoder_hip.h), next to the error
    an fp32 matmul of the same operands makes,
  * the fp32-MFMA fallback (RK_GEMM_PREC=f32): same golden parity in a sub-process.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _block(B, n_items, n_t, dev, seed):
  """A collated block whose item set has n_t items (one user row touching them all)."""
  import scipy.sparse as sp
  from recoder_amd.device import Block, DeviceCSR
  rng = np.random.RandomState(seed)
  items = np.sort(rng.choice(n_items, size=n_t, replace=False))
  rows = np.concatenate([np.zeros(n_t, np.int64), rng.randint(0, B, size=4 * B)])
  cols = np.concatenate([items, rng.choice(items, size=4 * B)])
  csr = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(B, n_items)).tocsr()
  csr.sum_duplicates()
  csr.data[:] = 1.0
  csr.sort_indices()
  dcsr = DeviceCSR(csr)
  blk = Block(B, int(csr.nnz), n_items, dev)
  blk.collate(dcsr, torch.arange(B, dtype=torch.int64, device=dev))
  n_b = blk.counts_host()[0]
  assert n_b == n_t
  return blk, items


# (top magnitude of Z, of W): typical training values, the top of the documented range
# (|Z| < 2048, |W| < 512), and small operands -- below |z| ~ 4e-3 / |w| ~ 1e-3 the lo halves go
# subnormal and the split keeps an ABSOLUTE error of 2^-25 / scale per element instead
@pytest.mark.parametrize("zmag,wmag,tol", [(1.0, 0.05, 6e-7), (1.0, 0.5, 6e-7), (1500.0, 0.3, 6e-7),
                                           (30.0, 400.0, 6e-7), (0.05, 0.02, 6e-7),
                                           (1e-3, 1e-3, 3e-5)])
@pytest.mark.parametrize("B,h,n_t", [(96, 200, 333), (500, 64, 1000)])
def test_decode_logits_match_float64(zmag, wmag, tol, B, h, n_t):
  from recoder_amd import _lib
  from recoder_amd._lib import LOSS_NONE, check, ptr
  from recoder_amd.device import current_stream
  lib = _lib.load()
  dev = torch.device("cuda")
  n_items = 4000
  blk, items = _block(B, n_items, n_t, dev, seed=B + n_t)
  g = torch.Generator(device="cpu").manual_seed(7)
  # a spread of magnitudes inside each operand (log-uniform over 2 decades below the top)
  def spread(shape, top):
    mag = top * 10.0 ** (-2.0 * torch.rand(shape, generator=g, dtype=torch.float64))
    sign = torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0).double()
    return (mag * sign).float()
  Z = spread((B, h), zmag)
  W = spread((n_items, h), wmag)
  b = torch.zeros(n_items)
  ld = blk.ld_cap
  out = torch.zeros(B * ld, device=dev)
  Zd, Wd, bd = Z.to(dev), W.to(dev), b.to(dev)
  check(lib.rk_decode_loss(ptr(Zd), B, h, blk.ref, 0, ptr(Wd), ptr(bd), LOSS_NONE, 0.0, 1.0, ptr(out),
                           ld, None, None, None, current_stream()), "rk_decode_loss")
  got = out.view(B, ld)[:, :n_t].cpu().double()
  Wt = W[torch.from_numpy(items)].double()
  exact = Z.double() @ Wt.t()
  scale = Z.double().abs() @ Wt.abs().t()            # sum of |products|: what rounding scales with
  err = ((got - exact).abs() / scale).max().item()
  f32 = (Z @ W[torch.from_numpy(items)].t()).double()
  err32 = ((f32 - exact).abs() / scale).max().item()
  print("zmag %g wmag %g: split-fp16 %.2e   fp32 matmul %.2e" % (zmag, wmag, err, err32))
  assert torch.isfinite(got).all()
  # 3 . 2^-22 per product at worst, far less after summation; fp32 fma chains sit at ~1e-7 here
  assert err < tol, (err, err32)


@pytest.mark.parametrize("gtop", [2e-3, 40.0, 1e-7])
def test_dz_matches_float64(gtop):
  """dO's split scale comes from the maximum the loss kernels publish: any magnitude works."""
  import struct
  from recoder_amd import _lib
  from recoder_amd._lib import check, ptr
  from recoder_amd.device import current_stream
  lib = _lib.load()
  dev = torch.device("cuda")
  B, h, n_t, n_items = 300, 200, 2500, 6000
  blk, items = _block(B, n_items, n_t, dev, seed=3)
  g = torch.Generator(device="cpu").manual_seed(11)
  ld = blk.counts_host()[2]
  dO = torch.zeros(B, blk.ld_cap)
  # gradients as the loss produces them: ~1e-3 and smaller, many decades of spread
  dO[:, :n_t] = (gtop * 10.0 ** (-3.0 * torch.rand((B, n_t), generator=g))
                 * torch.where(torch.rand((B, n_t), generator=g) < 0.5, -1.0, 1.0))
  # publish the maximum the way rk_decode_loss does (fp32 bit pattern in counts[8..71])
  amax = float(dO.abs().max())
  blk.counts[8] = struct.unpack("<i", struct.pack("<f", amax))[0]
  dO_dev = torch.zeros(B * blk.ld_cap, device=dev)
  dO_dev.view(-1)[:B * ld].view(B, ld)[:, :n_t] = dO[:, :n_t].to(dev)
  W = (torch.randn(n_items, h, generator=g) * 0.05)
  Wd = W.to(dev)
  dZ = torch.zeros(B * h, device=dev)
  ws = torch.zeros(lib.rk_dz_workspace_bytes(B, h) // 4, device=dev)
  check(lib.rk_decode_bwd_dz(ptr(dO_dev), B, h, blk.ref, ptr(Wd), None, 0, ptr(dZ), ptr(ws), None,
                             current_stream()), "rk_decode_bwd_dz")
  Wt = W[torch.from_numpy(items)].double()
  exact = dO[:, :n_t].double() @ Wt
  scale = dO[:, :n_t].double().abs() @ Wt.abs()
  err = ((dZ.view(B, h).cpu().double() - exact).abs() / scale).max().item()
  f32 = (dO[:, :n_t] @ W[torch.from_numpy(items)]).double()
  err32 = ((f32 - exact).abs() / scale).max().item()
  print("dZ gtop %g: split-fp16 %.2e   fp32 matmul %.2e" % (gtop, err, err32))
  assert err < 6e-7, (err, err32)


def test_fp32_mfma_fallback_keeps_golden_parity():
  """RK_GEMM_PREC=f32 (read once per process) routes the same entry points to the fp32 MFMA."""
  env = dict(os.environ, RK_GEMM_PREC="f32")
  r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                      os.path.join(ROOT, "tests", "test_hip_parity.py"), "-k",
                      "replays_reference_golden or steps_match_oracle"],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
  assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]


def test_recommend_under_plain_bf16_never_takes_the_pair_filter():
  """RK_GEMM_PREC=bf16: rk_split_image writes plain bf16 images; the fused top-k filter multiplies fp16
  pairs, so recommend() must fall back to the strips there (it returned wrong ids with
  status 0) -- same ids as the strip path, and the fused launch is never counted."""
  env = dict(os.environ, RK_GEMM_PREC="bf16")
  r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                      os.path.join(ROOT, "tests", "test_hip_parity.py"), "-k",
                      "recommend_fused_filter_equals_the_strips"],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
  assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("loss_name", ["mse", "bce", "mnll"])
def test_loss_kernels_publish_max_gradient(loss_name):
  """rk_decode_loss (MSE / BCE epilogue) and rk_mnll_finish publish max |dLoss/dLogit| into
  rk_block_t.counts[8..71] (fp32 bit patterns); rk_collate resets the slots."""
  from recoder_amd import _lib
  from recoder_amd._lib import LOSS_BCE, LOSS_MNLL, LOSS_MSE, check, ptr
  from recoder_amd.device import current_stream
  lib = _lib.load()
  dev = torch.device("cuda")
  B, h, n_t, n_items = 200, 64, 700, 3000
  blk, items = _block(B, n_items, n_t, dev, seed=9)
  assert int(blk.counts[8:72].abs().max()) == 0          # fresh from rk_collate
  g = torch.Generator(device="cpu").manual_seed(3)
  Z = torch.tanh(torch.randn(B, h, generator=g)).to(dev)
  W = (torch.randn(n_items, h, generator=g) * 0.2).to(dev)
  b = (torch.randn(n_items, generator=g) * 0.1).to(dev)
  ld = blk.counts_host()[2]
  dO = torch.zeros(B * blk.ld_cap, device=dev)
  part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), device=dev)
  kind = {"mse": LOSS_MSE, "bce": LOSS_BCE, "mnll": LOSS_MNLL}[loss_name]
  st = current_stream()
  check(lib.rk_decode_loss(ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(b), kind, 0.5, 1.0 / B, ptr(dO), 0,
                           ptr(part), None, None, st), "rk_decode_loss")
  if loss_name == "mnll":
    check(lib.rk_mnll_finish(ptr(dO), B, blk.ref, 0, 1.0 / B, None, None, None, ptr(part), st), "rk_mnll_finish")
  torch.cuda.synchronize()
  want = dO[:B * ld].view(B, ld)[:, :n_t].abs().max().cpu().numpy().astype(np.float32)
  slots = blk.counts[8:72].cpu().numpy().astype(np.int32).view(np.float32)
  assert slots.max() == want, (slots.max(), want)
  assert want > 0


# ---------------------------------------------------------------------------------------------
# bf16-pipe dW (csrc/dw3.hip): three bf16 pieces per fp32 operand, six products, no operand range
# ---------------------------------------------------------------------------------------------
def _dw3_case(B, h, n_t, gtop, ztop, give_G, seed=5, n_items=None, pairs=False):
  from recoder_amd import _lib
  from recoder_amd._lib import check, ptr
  from recoder_amd.device import current_stream
  lib = _lib.load()
  dev = torch.device("cuda")
  n_items = n_items or max(2 * n_t, 4000)
  blk, items = _block(B, n_items, n_t, dev, seed=seed)
  g = torch.Generator(device="cpu").manual_seed(seed)
  ld = blk.counts_host()[2]

  def spread(shape, top, decades):
    mag = top * 10.0 ** (-decades * torch.rand(shape, generator=g, dtype=torch.float64))
    sign = torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0).double()
    return (mag * sign).float()
  dO = spread((B, n_t), gtop, 4.0)
  Z = spread((B, h), ztop, 3.0)
  # everything outside the live region is poisoned: the kernel must not let it reach the result
  dO_dev = torch.full((B * blk.ld_cap + 64,), float("nan"), device=dev)
  live = dO_dev[:B * ld].view(B, ld)
  live[:, :] = 0.0                     # the decode epilogue writes finite values up to ld
  live[:, :n_t] = dO.to(dev)
  Zd = Z.to(dev)
  nbytes = lib.rk_dw3_workspace_bytes(B, h, blk.n_cap)
  ws = torch.full((nbytes // 4 + 64,), float("nan"), device=dev)
  G = torch.full((blk.n_cap * h,), float("nan"), device=dev) if give_G else None
  if pairs:
    # fp16 pairs (rk_decode_bwd_dw2): the operand maxima are published the way the training step
    # does it -- max |dO| in counts[8..71] (the loss kernels), a bound of |Z| in ranges[0..63] (rk_amax)
    blk.counts[8:72].zero_()
    blk.counts[8:9].copy_(dO.abs().max().reshape(1).to(dev).view(torch.int32))
    ranges = torch.zeros(128, dtype=torch.int32, device=dev)
    check(lib.rk_amax(ptr(Zd), B * h, ptr(ranges), current_stream()), "rk_amax")
    check(lib.rk_decode_bwd_dw2(ptr(dO_dev), ptr(Zd), B, h, blk.ref, ptr(G), None, ptr(ws), None, ptr(ranges),
                                current_stream()), "rk_decode_bwd_dw2")
  else:
    check(lib.rk_decode_bwd_dw3(ptr(dO_dev), ptr(Zd), B, h, blk.ref, ptr(G), None, ptr(ws), None,
                                current_stream()), "rk_decode_bwd_dw3")
  torch.cuda.synchronize()
  ns = int(blk.counts[4].item())
  assert 1 <= ns <= lib.rk_dw3_max_splits()
  if give_G:
    got = G[:n_t * h].view(n_t, h)
  else:
    off = (lib.rk_dw3_slabs(ptr(ws), B, h) - ws.data_ptr()) // 4
    stride = blk.n_cap * h
    got = sum(ws[off + k * stride:off + k * stride + n_t * h].view(n_t, h) for k in range(ns))
  got = got.cpu().double()
  exact = dO.double().t() @ Z.double()
  scale = dO.double().abs().t() @ Z.double().abs()
  err = ((got - exact).abs() / scale).max().item()
  f32 = (dO.t() @ Z).double()
  err32 = ((f32 - exact).abs() / scale).max().item()
  assert torch.isfinite(got).all()
  return err, err32, ns


@pytest.mark.parametrize("B,h,n_t", [(500, 200, 7842), (96, 64, 333), (37, 200, 65), (512, 512, 3000),
                                     (1000, 128, 129), (64, 256, 4097), (300, 20, 1000)])
@pytest.mark.parametrize("give_G", [True, False])
def test_dw3_matches_float64(B, h, n_t, give_G):
  err, err32, ns = _dw3_case(B, h, n_t, 2e-3, 1.0, give_G)
  print("B %d h %d n_t %d slabs %d: bf16x3 %.2e   fp32 matmul %.2e" % (B, h, n_t, ns, err, err32))
  # error / sum |products|: an fp32 matmul of the same operands sits at 2-6e-7 on this data
  assert err < max(5e-7, 1.05 * err32), (err, err32)


@pytest.mark.parametrize("B,h,n_t", [(500, 200, 7842), (96, 64, 333), (37, 200, 65), (512, 512, 3000),
                                     (1000, 128, 129), (64, 256, 4097), (300, 20, 1000)])
@pytest.mark.parametrize("give_G", [True, False])
def test_dw2_fp16_pairs_match_float64(B, h, n_t, give_G):
  """rk_decode_bwd_dw2 (the training step's default since round 3): fp16 pairs, three products --
  the accuracy of an fp32 matmul of the same operands, as the bf16 triples."""
  err, err32, ns = _dw3_case(B, h, n_t, 2e-3, 1.0, give_G, pairs=True)
  print("B %d h %d n_t %d slabs %d: fp16x2 %.2e   fp32 matmul %.2e" % (B, h, n_t, ns, err, err32))
  assert err < max(5e-7, 1.05 * err32), (err, err32)


@pytest.mark.parametrize("gtop,ztop", [(1e-7, 1.0), (40.0, 1e4), (3e4, 3e4), (1e-12, 1e-3), (1e15, 1e15)])
def test_dw2_scales_follow_the_published_maxima(gtop, ztop):
  """The pair split takes its power-of-two scales from the published maxima of BOTH operands:
  magnitudes from 1e-12 to 1e15 come out with the same relative accuracy."""
  err, err32, ns = _dw3_case(200, 200, 1500, gtop, ztop, True, seed=9, pairs=True)
  print("gtop %g ztop %g: fp16x2 %.2e   fp32 matmul %.2e" % (gtop, ztop, err, err32))
  assert err < max(5e-7, 1.05 * err32), (err, err32)


@pytest.mark.parametrize("gtop,ztop", [(1e-7, 1.0), (40.0, 1e4), (3e4, 3e4), (1e-12, 1e-3), (1e15, 1e15)])
def test_dw3_has_no_operand_range(gtop, ztop):
  """bf16 carries fp32's exponent: magnitudes that overflow an fp16 split (|x| >= 65504 / scale)
  and magnitudes far below it come out with the same relative accuracy."""
  err, err32, ns = _dw3_case(200, 200, 1500, gtop, ztop, True, seed=9)
  print("gtop %g ztop %g: bf16x3 %.2e   fp32 matmul %.2e" % (gtop, ztop, err, err32))
  assert err < max(5e-7, 1.05 * err32), (err, err32)




# ---------------------------------------------------------------------------------------------
# RK_GEMM_PREC=bf16 on the CURRENT kernel family (round 6): fdec_kernel<.., PLAIN> + dw_encbwd_kernel<.., PLAIN>
# ---------------------------------------------------------------------------------------------
@pytest.mark.skipif(os.environ.get("RK_GEMM_PREC", "")[:1].lower() != "b", reason="runs in the RK_GEMM_PREC=bf16 sub-process")
@pytest.mark.parametrize("h,loss,conf", [(200, "mse", 0.0), (40, "mse", 2.5), (200, "logistic", 0.0)])
def test_plain_bf16_steps_on_the_fused_decode(h, loss, conf):
  """Whole single-process steps with plain bf16 operands run the register-resident fused decode and the dW tiles of
  csrc/pgemm.h with ONE product each (rk_ae_step_uses_pg == 3, bias gradient as the tiles' output column h) and stay
  within bf16's error of the fp32 oracle: losses to 2e-3 relative (8 mantissa bits on every operand of three chained
  contractions), the run bitwise reproducible."""
  from oracle import recoder_oracle as orc
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  from tests.test_hip_parity import synth_csr
  csr = synth_csr(1500, 3000, 30, seed=5, ratings=False)
  B = 500
  order = np.random.RandomState(3).permutation(csr.shape[0]).astype(np.int64)
  lp = {"confidence": conf} if conf else None

  def run():
    torch.manual_seed(11)
    model = DynamicAutoencoder([h], activation_type="tanh", noise_prob=0.0, sparse=False)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss, loss_params=lp)
    rec.user_order_hook = lambda epoch, n: order
    ds = RecommendationDataset(csr)
    rec._Recoder__init_training(ds, 1e-3, 2e-5)
    init = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    rec.train(ds, batch_size=B, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True)
    eng = rec._engine()
    assert int(getattr(eng, "_step_mode", 0)) == 3 and int(getattr(eng, "_step_flags", 0)) & 16
    return init, np.concatenate(rec.loss_history), {k: v.detach().cpu().clone() for k, v in model.named_parameters()}

  init, got, pars = run()
  _, got2, pars2 = run()
  assert np.array_equal(got, got2)
  for k in pars:
    assert torch.equal(pars[k], pars2[k]), k
  o = orc.OracleRecoder("ae", init, hidden_layers=[h], activation_type="tanh", loss=loss, loss_params=lp,
                        lr=1e-3, weight_decay=2e-5)
  want = []
  for ep in range(2):
    for off in range(0, csr.shape[0], B):
      users = order[off:off + B]
      b = orc.collate(orc.extract_rows(csr, users), users, B, True)[0]
      want.append(o.train_step(b))
  want = np.asarray(want, dtype=np.float64)
  rel = np.abs(got - want) / np.abs(want)
  print("plain bf16, h = %d %s: max relative loss error %.2e over %d steps" % (h, loss, rel.max(), len(want)))
  assert rel.max() < 2e-3, rel
  assert rel.max() > 1e-7          # (it IS the one-product path, not the split one)


def test_plain_bf16_runs_the_current_kernel_family():
  """RK_GEMM_PREC=bf16 (read once per process): the cases above in a sub-process."""
  env = dict(os.environ, RK_GEMM_PREC="bf16")
  r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-s",
                      os.path.join(ROOT, "tests", "test_gemm_precision.py"), "-k", "plain_bf16_steps_on_the_fused_decode"],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
  assert "3 passed" in r.stdout, r.stdout[-800:]
