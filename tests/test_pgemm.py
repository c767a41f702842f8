"""GPU: the pipelined pair-plane contractions (csrc/pgemm.h, rk_pg_*) -- decode + loss with dLoss/dLogits
as a plane image, dZ and dW from that image -- against the round-3 plane kernels (same arithmetic: the
fp32 gradients must agree BIT FOR BIT) and against float64 products."""
import ctypes

import numpy as np
import pytest
import torch

from recoder_amd._lib import LOSS_BCE, LOSS_MSE, check, ptr
from recoder_amd.device import current_stream
from tests.test_planes import _setup

pytestmark = pytest.mark.gpu

ACT_TANH = 1


def _image_to_f32(img, scales, B, ld, gr, gc, pitch):
  """fp32 matrix [B, ld] of a dO image (uint16 view [rows][ld / 32][2][32]) and its scale table."""
  rows = img.shape[0]
  v = img.view(rows, ld // 32, 2, 32).view(torch.float16).float()
  x = (v[:, :, 0, :] + v[:, :, 1, :]).reshape(rows, ld)
  r = torch.arange(rows, device=img.device) // gr
  c = torch.arange(ld, device=img.device) // gc
  s = scales[(r[:, None] * pitch + c[None, :]).reshape(-1)].view(rows, ld)
  return (x / s)[:B]


CASES = [(500, 200, 3000, LOSS_MSE, False), (37, 20, 400, LOSS_BCE, False), (1, 8, 97, LOSS_MSE, False),
         (300, 512, 2000, LOSS_MSE, False), (64, 36, 333, LOSS_BCE, True), (1100, 64, 1500, LOSS_MSE, True),
         (1030, 136, 700, LOSS_BCE, False), (130, 260, 5000, LOSS_MSE, False)]


@pytest.mark.parametrize("B,h,n_items,loss,ratings", CASES)
def test_pg_decode_dz_dw(B, h, n_items, loss, ratings):
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=3 * B + h, ratings=ratings)
  st = current_stream()
  dev = Z.device
  f = dict(dtype=torch.float32, device=dev)
  n_b, nnz, ld, S = blk.counts_host()
  npart = lib.rk_loss_partials(B, blk.n_cap)
  ntile = -(-B // lib.rk_decode_row_tile())
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  # reference: the round-3 plane decode (fp32 dO)
  dO = torch.zeros(B * blk.ld_cap, **f)
  part = torch.zeros(npart, **f)
  gbp = torch.zeros(ntile * blk.ld_cap, **f)
  check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B,
                                  ptr(dO), blk.ld_cap, ptr(part), ptr(gbp), st))
  dO_ref = dO[:B * ld].view(B, ld).clone()
  # the pipelined decode: image + scale table (+ the fp32 matrix for this comparison)
  rows_img = -(-B // 32) * 32
  img = torch.full(((rows_img + 256) * blk.ld_cap * 2,), 0x7e00, dtype=torch.int16, device=dev)    # (fp16 NaNs: every read element must have been written)
  sc = torch.full((lib.rk_pg_scale_floats(B, blk.n_cap),), float("nan"), **f)
  dO2 = torch.zeros(B * blk.ld_cap, **f)
  part2 = torch.zeros(npart, **f)
  gbp2 = torch.zeros(ntile * blk.ld_cap, **f)
  blk.counts[8:72].zero_()
  check(lib.rk_pg_decode_loss(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(img), rows_img,
                              ptr(sc), ptr(dO2), ptr(part2), ptr(gbp2), st))
  torch.cuda.synchronize()
  got = dO2[:B * ld].view(B, ld)
  assert torch.equal(got[:, :n_b], dO_ref[:, :n_b])                        # same arithmetic, same k order
  assert float(got[:, n_b:].abs().max()) == 0.0 if ld > n_b else True
  s1, s2 = part.double().sum().item(), part2.double().sum().item()
  assert abs(s1 - s2) <= 1e-6 * abs(s1)
  assert torch.allclose(gbp[:ntile * ld].view(ntile, ld)[:, :n_b].sum(0), gbp2[:ntile * ld].view(ntile, ld)[:, :n_b].sum(0),
                        rtol=1e-5, atol=1e-7)
  gr, gc = ctypes.c_int32(), ctypes.c_int32()
  lib.rk_pg_decode_granule(B, blk.n_cap, ctypes.byref(gr), ctypes.byref(gc))
  gr, gc = gr.value, gc.value
  pitch = -(-blk.n_cap // gc)
  rec = _image_to_f32(img[:rows_img * ld * 2].view(rows_img, ld * 2), sc, rows_img, ld, gr, gc, pitch)
  assert torch.isfinite(rec).all()
  assert float(rec[B:].abs().max()) == 0.0 if rows_img > B else True        # the K tail of dW: zeros
  err = (rec[:B] - got).abs().max().item()
  assert err <= 2.0 ** -20 * max(got.abs().max().item(), 1e-30), err
  # published maximum (consumers of an fp32 dO)
  pub = blk.counts[8:72].view(torch.float32).max().item()
  assert pub == got.abs().max().item()

  # ---- dZ = dO . W[items] (* act'(Z)) ----
  items = blk.items[:n_b].long()
  Wt = W[items].double()
  g64 = got[:, :n_b].double()
  ws = torch.zeros(lib.rk_pg_dz_workspace_bytes(B, h) // 4 + 64, **f)
  dZ = torch.full((B * h,), float("nan"), **f)
  check(lib.rk_pg_dz(ptr(img), ptr(sc), gr, gc, B, ctypes.byref(pl), blk.ref, ptr(Z), ACT_TANH, ptr(dZ), ptr(ws), st))
  torch.cuda.synchronize()
  exact = (g64 @ Wt) * (1.0 - Z.double() ** 2)
  scale = (g64.abs() @ Wt.abs()) * (1.0 - Z.double() ** 2).abs() + 1e-300
  e_dz = ((dZ.view(B, h).double() - exact).abs() / scale).max().item()
  assert e_dz < 6e-7, e_dz

  # ---- dW = dO^T . Z as K slabs ----
  ns = lib.rk_pg_dw_splits(B, h, blk.n_cap)
  slabs = torch.full((lib.rk_pg_dw_workspace_bytes(B, h, blk.n_cap) // 4,), float("nan"), **f)
  check(lib.rk_pg_dw(ptr(img), ptr(sc), gr, gc, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st))
  torch.cuda.synchronize()
  live = int(blk.counts[4].item())
  assert 1 <= live <= ns
  G = slabs.view(ns, blk.n_cap, h)[:live, :n_b].double().sum(0)
  exact = g64.t() @ Z.double()
  scale = g64.abs().t() @ Z.double().abs() + 1e-300
  e_dw = ((G - exact).abs() / scale).max().item()
  assert e_dw < 6e-7, e_dw
  # the deep DMA ring of the 64 x 128 dW tiles (RK_TUNE_DW_RING = 13: 3 / 4 / 6 LDS stages, counted waits, raw
  # barriers, asm transpose reads): the same MFMAs in the same order per accumulator -- bit for bit
  if B < 1024:
    for ring in (2, 4):
      slabs2 = torch.full_like(slabs, float("nan"))
      lib.rk_tune(13, ring)
      try:
        for _ in range(3):          # (a race in the ring would come and go: three runs)
          check(lib.rk_pg_dw(ptr(img), ptr(sc), gr, gc, B, ctypes.byref(pl), blk.ref, ptr(slabs2), None, st))
          torch.cuda.synchronize()
          assert int(blk.counts[4].item()) == live
          assert torch.equal(slabs2.view(ns, blk.n_cap, h)[:live, :n_b], slabs.view(ns, blk.n_cap, h)[:live, :n_b]), ring
      finally:
        lib.rk_tune(13, 0)
  print("B=%d h=%d n_b=%d: image err %.2e of max, dZ %.2e, dW %.2e (of sum |products|), %d dW slabs" % (
      B, h, n_b, err / max(got.abs().max().item(), 1e-30), e_dz, e_dw, ns))


def test_pg_scales_follow_the_data():
  """Tiles of very different magnitude (a few rows with 1e4 x larger targets): every granule carries its
  own scale and the consumers rescale their accumulators at the granule boundaries exactly."""
  B, h, n_items = 300, 64, 1200
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, 600, n_items, 14, seed=77, ratings=True)
  st = current_stream()
  dev = Z.device
  f = dict(dtype=torch.float32, device=dev)
  # rows 64..127 of the batch: ratings 1e4 x larger
  ip = blk.indptr[:B + 1].long()
  blk.vals[int(ip[64]):int(ip[128])] *= 1e4
  n_b, nnz, ld, S = blk.counts_host()
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  rows_img = -(-B // 32) * 32
  img = torch.zeros((rows_img + 256) * blk.ld_cap * 2, dtype=torch.int16, device=dev)
  sc = torch.zeros(lib.rk_pg_scale_floats(B, blk.n_cap), **f)
  dO2 = torch.zeros(B * blk.ld_cap, **f)
  part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
  check(lib.rk_pg_decode_loss(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(img), rows_img,
                              ptr(sc), ptr(dO2), ptr(part), None, st))
  torch.cuda.synchronize()
  got = dO2[:B * ld].view(B, ld)[:, :n_b].double()
  live = sc[:(-(-B // 64)) * (-(-blk.n_cap // 32))].view(-(-B // 64), -1)[:, :-(-n_b // 32)]
  assert live[1].min().item() < live[0].max().item() / 64       # the heavy rows' granules have smaller scales
  items = blk.items[:n_b].long()
  ws = torch.zeros(lib.rk_pg_dz_workspace_bytes(B, h) // 4 + 64, **f)
  dZ = torch.zeros(B * h, **f)
  check(lib.rk_pg_dz(ptr(img), ptr(sc), 64, 32, B, ctypes.byref(pl), blk.ref, None, 0, ptr(dZ), ptr(ws), st))
  ns = lib.rk_pg_dw_splits(B, h, blk.n_cap)
  slabs = torch.zeros(lib.rk_pg_dw_workspace_bytes(B, h, blk.n_cap) // 4, **f)
  check(lib.rk_pg_dw(ptr(img), ptr(sc), 64, 32, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st))
  torch.cuda.synchronize()
  Wt = W[items].double()
  e1 = ((dZ.view(B, h).double() - got @ Wt).abs() / (got.abs() @ Wt.abs() + 1e-300)).max().item()
  G = slabs.view(ns, blk.n_cap, h)[:int(blk.counts[4].item()), :n_b].double().sum(0)
  e2 = ((G - got.t() @ Z.double()).abs() / (got.abs().t() @ Z.double().abs() + 1e-300)).max().item()
  # (rows 1e4 x apart in one fp32 accumulator: the bound of an fp32 matmul of the same operands applies)
  f32 = ((got.float().t() @ Z).double() - got.t() @ Z.double()).abs() / (got.abs().t() @ Z.double().abs() + 1e-300)
  assert e1 < 6e-7 and e2 < max(6e-7, 2 * f32.max().item()), (e1, e2, f32.max().item())


@pytest.mark.parametrize("B,h,n_items,ratings", [(500, 200, 3000, False), (37, 20, 400, True), (300, 64, 5000, False),
                                                 (1, 8, 97, False), (260, 512, 900, True)])
def test_pg_decode_mnll_matches_the_two_launch_form(B, h, n_items, ratings):
  """rk_pg_decode_mnll (statistics pass + decode / loss pass, nothing but 8 bytes per row and column tile in
  between) against rk_decode_loss_planes + rk_mnll_finish (logits matrix written, re-read, rewritten)."""
  from recoder_amd._lib import LOSS_MNLL
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=5 * B + h, ratings=ratings)
  st = current_stream()
  dev = Z.device
  f = dict(dtype=torch.float32, device=dev)
  n_b, nnz, ld, S = blk.counts_host()
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  dO = torch.zeros(B * blk.ld_cap, **f)
  part = torch.zeros(max(B, lib.rk_loss_partials(B, blk.n_cap)), **f)
  check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MNLL, 0.0, 1.0 / B,
                                  ptr(dO), 0, ptr(part), None, st))
  check(lib.rk_mnll_finish(ptr(dO), B, blk.ref, 0, 1.0 / B, None, None, None, ptr(part), st))
  torch.cuda.synchronize()
  ref = dO[:B * ld].view(B, ld)[:, :n_b].clone()
  ref_loss = part[:B].double().sum().item()
  rows_img = -(-B // 32) * 32
  img = torch.full(((rows_img + 256) * blk.ld_cap * 2,), 0x7e00, dtype=torch.int16, device=dev)
  sc = torch.full((lib.rk_pg_scale_floats(B, blk.n_cap),), float("nan"), **f)
  ws = torch.full((lib.rk_pg_mnll_workspace_floats(B, blk.n_cap),), float("nan"), **f)
  dO2 = torch.zeros(B * blk.ld_cap, **f)
  part2 = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
  ntile = -(-B // lib.rk_decode_row_tile())
  gbp2 = torch.zeros(ntile * blk.ld_cap, **f)
  check(lib.rk_pg_decode_mnll(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), 1.0 / B, ptr(ws), ptr(img), rows_img,
                              ptr(sc), ptr(dO2), ptr(part2), ptr(gbp2), st))
  torch.cuda.synchronize()
  got = dO2[:B * ld].view(B, ld)[:, :n_b]
  scale = ref.abs().max().item()
  assert (got - ref).abs().max().item() <= 2e-6 * max(scale, 1e-30), ((got - ref).abs().max().item(), scale)
  loss = part2.double().sum().item()
  assert abs(loss - ref_loss) <= 2e-6 * abs(ref_loss), (loss, ref_loss)
  # decoder bias gradient = column sums of dO (rk_colsum in the two-launch form)
  gb = gbp2[:ntile * ld].view(ntile, ld)[:, :n_b].sum(0)
  assert torch.allclose(gb, ref.sum(0), rtol=1e-4, atol=2e-6 * max(scale, 1e-30) * B)
  rec = _image_to_f32(img[:rows_img * ld * 2].view(rows_img, ld * 2), sc, rows_img, ld, 64, 32, -(-blk.n_cap // 32))
  assert torch.isfinite(rec).all()
  assert (rec[:B, :n_b] - got).abs().max().item() <= 2.0 ** -20 * max(got.abs().max().item(), 1e-30)


@pytest.mark.parametrize("B,h,n_items,loss,ratings", [(500, 200, 3000, LOSS_MSE, False), (37, 20, 400, LOSS_BCE, False),
                                                      (1, 8, 97, LOSS_MSE, False), (130, 128, 2000, LOSS_MSE, True),
                                                      (300, 64, 5000, LOSS_BCE, True), (513, 224, 700, LOSS_MSE, False),
                                                      (64, 36, 333, LOSS_BCE, True)])
def test_fdec_matches_the_lds_fused_decode(B, h, n_items, loss, ratings):
  _fdec_case(B, h, n_items, loss, ratings)


def _fdec_case(B, h, n_items, loss, ratings):
  """rk_fdec_loss_dz (register-resident fused decode: transposed tile, permlane32 fragments, resident W
  rows) against rk_decode_loss_dz_planes: the same logits bit for bit (so the same dO up to the image's
  2^-21), the same loss, dZ against float64."""
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=7 * B + h, ratings=ratings)
  if not lib.rk_fdec_ok(B, h, blk.n_cap, loss):
    pytest.skip("outside the fused decode's domain")
  st = current_stream()
  dev = Z.device
  f = dict(dtype=torch.float32, device=dev)
  n_b, nnz, ld, S = blk.counts_host()
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  npart = lib.rk_loss_partials(B, blk.n_cap)
  ntile = -(-B // lib.rk_decode_row_tile())
  dO = torch.zeros(B * blk.ld_cap, **f)
  part = torch.zeros(npart, **f)
  gbp = torch.zeros(ntile * blk.ld_cap, **f)
  ws = torch.zeros(lib.rk_dz_fused_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  check(lib.rk_decode_loss_dz_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                                     ptr(part), ptr(gbp), ptr(ws), st))
  dZ_ref = torch.zeros(B * h, **f)
  check(lib.rk_decode_dz_reduce(ptr(ws), B, h, blk.ref, ptr(Z), ACT_TANH, ptr(dZ_ref), st))
  torch.cuda.synchronize()
  ref = dO[:B * ld].view(B, ld)[:, :n_b].clone()
  rows_img = -(-B // 32) * 32
  img = torch.full(((rows_img + 256) * blk.ld_cap * 2,), 0x7e00, dtype=torch.int16, device=dev)
  sc = torch.full((lib.rk_pg_scale_floats(B, blk.n_cap),), float("nan"), **f)
  part2 = torch.zeros(npart, **f)
  ws2 = torch.full((lib.rk_fdec_workspace_bytes(B, h, blk.n_cap) // 4 + 64,), float("nan"), **f)
  blk.counts[8:72].zero_()
  check(lib.rk_fdec_loss_dz(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(img), rows_img,
                            ptr(sc), ptr(part2), ptr(ws2), st))
  dZ = torch.zeros(B * h, **f)
  check(lib.rk_fdec_dz_reduce(ptr(ws2), B, h, blk.ref, ptr(Z), ACT_TANH, ptr(dZ), st))
  torch.cuda.synchronize()
  pitch = -(-blk.n_cap // 64)
  rec = _image_to_f32(img[:rows_img * ld * 2].view(rows_img, ld * 2), sc, rows_img, ld, 32, 64, pitch)
  assert torch.isfinite(rec).all()
  assert float(rec[B:].abs().max()) == 0.0 if rows_img > B else True
  assert float(rec[:B, n_b:].abs().max()) == 0.0 if ld > n_b else True
  scale = max(ref.abs().max().item(), 1e-30)
  err = (rec[:B, :n_b] - ref).abs().max().item()
  assert err <= 2.0 ** -20 * scale, (err, scale)
  s1, s2 = part.double().sum().item(), part2.double().sum().item()
  assert abs(s1 - s2) <= 1e-6 * abs(s1)
  assert blk.counts[8:72].view(torch.float32).max().item() == ref.abs().max().item()
  items = blk.items[:n_b].long()
  exact = (ref.double() @ W[items].double()) * (1.0 - Z.double() ** 2)
  den = (ref.double().abs() @ W[items].double().abs()) * (1.0 - Z.double() ** 2).abs() + 1e-300
  e_new = ((dZ.view(B, h).double() - exact).abs() / den).max().item()
  e_old = ((dZ_ref.view(B, h).double() - exact).abs() / den).max().item()
  print("B=%d h=%d n_b=%d: image err %.2e of max; dZ err new %.2e old %.2e" % (B, h, n_b, err / scale, e_new, e_old))
  assert e_new < 6e-7, (e_new, e_old)
  # ... and rk_pg_dw reads that image (granule 32 x 64)
  ns = lib.rk_pg_dw_splits(B, h, blk.n_cap)
  slabs = torch.full((lib.rk_pg_dw_workspace_bytes(B, h, blk.n_cap) // 4,), float("nan"), **f)
  gbd = torch.full((blk.n_cap,), 7.0, **f)
  check(lib.rk_pg_dw(ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), ptr(gbd), st))
  torch.cuda.synchronize()
  live = int(blk.counts[4].item())
  G = slabs.view(ns, blk.n_cap, h)[:live, :n_b].double().sum(0)
  ex = ref.double().t() @ Z.double()
  e_dw = ((G - ex).abs() / (ref.double().abs().t() @ Z.double().abs() + 1e-300)).max().item()
  assert e_dw < 6e-7, e_dw
  # ... and sums its columns (the decoder bias gradient) in a second workgroup range
  cs = ref.double().sum(0)
  assert (gbd[:n_b].double() - cs).abs().max().item() <= 2e-6 * max(ref.double().abs().sum(0).max().item(), 1e-30)
  # ... and rk_pg_dw_dz_reduce does all of it -- dW, the column sums, the dZ slab reduce -- in ONE launch: the same
  # numbers bit for bit, as K slabs (single process) and as one dense array (dense = 1: the users-DP exchange)
  if B < 1024:
    for dense in (0, 1):
      out = torch.full_like(slabs, float("nan"))
      gb2 = torch.full((blk.n_cap,), 7.0, **f)
      dZ2 = torch.full((B * h,), float("nan"), **f)
      check(lib.rk_pg_dw_dz_reduce(ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(out), ptr(gb2), ptr(ws2),
                                   ptr(Z), ACT_TANH, ptr(dZ2), dense, st))
      torch.cuda.synchronize()
      # (the slabs are added up by 4 waves per output group here, by 16 in rk_fdec_dz_reduce: another order)
      assert ((dZ2.view(B, h).double() - exact).abs() / den).max().item() < 6e-7
      assert torch.equal(gb2[:n_b], gbd[:n_b])
      if dense:
        G2 = out[:blk.n_cap * h].view(blk.n_cap, h)[:n_b].double()
        assert ((G2 - ex).abs() / (ref.double().abs().t() @ Z.double().abs() + 1e-300)).max().item() < 6e-7
        if live == 1:
          assert torch.equal(out[:blk.n_cap * h].view(blk.n_cap, h)[:n_b], slabs.view(ns, blk.n_cap, h)[0, :n_b])
      else:
        assert int(blk.counts[4].item()) == live
        assert torch.equal(out.view(ns, blk.n_cap, h)[:live, :n_b], slabs.view(ns, blk.n_cap, h)[:live, :n_b])


def test_fdec_scale_table_has_no_write_past_the_capacity():
  """A granule half past the item capacity has no slot in the scale table: its write used to land in the next
  row's first column and won or lost a race against that granule's own scale (intermittent: 34 of 40 runs)."""
  for _ in range(12):
    test_fdec_matches_the_lds_fused_decode(513, 224, 700, LOSS_MSE, False)
