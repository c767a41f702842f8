"""GPU: the pre-split operand planes (csrc/planes.h, csrc/decode16.hip) against the in-loop split
kernels of csrc/gemm.hip -- the same arithmetic (s.x = hi + lo in fp16, lo.hi + hi.lo + hi.hi in
fp32, same k order), so with equal tile shapes every output must agree BIT FOR BIT; the 128 x 128
decode tile only regroups the loss / bias partial sums.  Plus the batched collation against the
per-block one."""
import ctypes

import numpy as np
import pytest
import torch

from recoder_amd import _lib
from recoder_amd._lib import LOSS_BCE, LOSS_MSE, LOSS_NONE, RkBlock, RkPlanes, check, ptr
from recoder_amd.device import Block, DeviceCSR, current_stream
from tests.test_hip_parity import synth_csr

pytestmark = pytest.mark.gpu


def _setup(B, h, n_users, n_items, deg, seed, ratings=False):
  lib = _lib.load()
  dev = torch.device("cuda")
  csr = synth_csr(n_users, n_items, deg, seed=seed, ratings=ratings)
  dcsr = DeviceCSR(csr)
  f = dict(dtype=torch.float32, device=dev)
  g = torch.Generator(device=dev)
  g.manual_seed(seed)
  W = torch.randn(n_items, h, generator=g, **f) * 0.07
  bias = torch.randn(n_items, generator=g, **f) * 0.02
  users = torch.arange(B, dtype=torch.int64, device=dev)
  blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev)
  blk.collate(dcsr, users)
  Z = torch.tanh(torch.randn(B, h, generator=g, **f))
  ranges = torch.zeros(128, dtype=torch.int32, device=dev)
  ranges[64:65].copy_(W.abs().max().reshape(1).view(torch.int32))
  buf = torch.zeros(lib.rk_planes_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  pl = RkPlanes()
  check(lib.rk_planes_layout(ptr(buf), B, h, blk.n_cap, ctypes.byref(pl)))
  return lib, blk, W, bias, Z, ranges, pl, buf


@pytest.mark.parametrize("B,h,n_items,loss", [(500, 200, 3000, LOSS_MSE), (37, 20, 400, LOSS_BCE),
                                              (1, 8, 97, LOSS_MSE), (130, 64, 900, LOSS_NONE),
                                              (300, 512, 2000, LOSS_MSE), (64, 36, 333, LOSS_BCE)])
def test_decode_and_dz_on_planes_equal_the_in_loop_split_bit_for_bit(B, h, n_items, loss):
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=B + h,
                                                 ratings=(loss == LOSS_MSE and h == 200))
  st = current_stream()
  f = dict(dtype=torch.float32, device=Z.device)
  n_b, nnz, ld, S = blk.counts_host()
  npart = lib.rk_loss_partials(B, blk.n_cap)
  ntile = -(-B // lib.rk_decode_row_tile())

  def decode(planes, tile=0):
    dO = torch.zeros(B * blk.ld_cap, **f)
    part = torch.zeros(npart, **f)
    gbp = torch.zeros(ntile * blk.ld_cap, **f)
    blk.counts[8:72].zero_()
    if planes:
      lib.rk_planes_tile(tile)
      check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
      check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B,
                                      ptr(dO), blk.ld_cap, ptr(part), ptr(gbp), st))
      lib.rk_planes_tile(0)
    else:
      check(lib.rk_decode_loss(ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                               blk.ld_cap, ptr(part), ptr(gbp), ptr(ranges), st))
    ldo = blk.ld_cap if loss == LOSS_NONE else ld
    return (dO[:B * ldo].view(B, ldo)[:, :n_b].clone(), part.clone(),
            gbp[:ntile * ld].view(ntile, ld)[:, :n_b].clone(), blk.counts[8:72].clone())

  o = decode(False)
  n64 = decode(True, 64)
  n128 = decode(True, 128)
  assert torch.equal(o[0], n64[0]) and torch.equal(o[0], n128[0])            # logits / dLoss/dLogits
  if loss != LOSS_NONE:
    assert torch.equal(o[1], n64[1]) and torch.equal(o[2], n64[2])           # loss and bias partials
    assert abs(o[1].double().sum().item() - n128[1].double().sum().item()) <= 1e-6 * abs(o[1].double().sum().item())
    assert torch.allclose(o[2].sum(0), n128[2].sum(0), rtol=1e-5, atol=1e-7)
    assert o[3].view(torch.float32).max() == n64[3].view(torch.float32).max() == n128[3].view(torch.float32).max()
    # the padding columns of every dO row are zeros (the dZ contraction's K tail)
    if ld > n_b:
      dO = torch.zeros(B * blk.ld_cap, **f).fill_(7.0)
      part = torch.zeros(npart, **f)
      check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B,
                                      ptr(dO), blk.ld_cap, ptr(part), None, st))
      assert float(dO[:B * ld].view(B, ld)[:, n_b:].abs().max()) == 0.0
    # dZ: planes == in-loop split, also with garbage in dO's padding columns (masked in the kernel)
    dO = torch.zeros(B * blk.ld_cap, **f)
    part = torch.zeros(npart, **f)
    blk.counts[8:72].zero_()
    check(lib.rk_decode_loss(ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                             blk.ld_cap, ptr(part), None, ptr(ranges), st))
    ws = torch.zeros(lib.rk_dz_workspace_bytes(B, h) // 4 + 64, **f)
    dz0 = torch.zeros(B, h, **f)
    dz1 = torch.zeros(B, h, **f)
    check(lib.rk_decode_bwd_dz(ptr(dO), B, h, blk.ref, ptr(W), ptr(Z), 1, ptr(dz0), ptr(ws), ptr(ranges), st))
    if ld > n_b:
      dO[:B * ld].view(B, ld)[:, n_b:] = 3.0e30          # would overflow the fp16 split: must be masked
    ws.zero_()
    check(lib.rk_decode_bwd_dz_planes(ptr(dO), B, ctypes.byref(pl), blk.ref, ptr(Z), 1, ptr(dz1), ptr(ws), st))
    assert torch.equal(dz0, dz1)
    ref = (dO[:B * ld].view(B, ld)[:, :n_b].double() @ W[blk.items[:n_b].long()].double()) * (1 - Z.double() ** 2)
    assert (dz1.double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), 1e-30)


@pytest.mark.parametrize("B,h,n_items,loss", [(500, 200, 3000, LOSS_MSE), (37, 20, 400, LOSS_BCE),
                                              (1, 8, 97, LOSS_MSE), (130, 64, 900, LOSS_BCE),
                                              (300, 256, 2000, LOSS_MSE), (64, 36, 333, LOSS_BCE)])
def test_decode_with_fused_dz_equals_the_two_launches(B, h, n_items, loss):
  """rk_decode_loss_dz_planes + rk_decode_dz_reduce against rk_decode_loss_planes +
  rk_decode_bwd_dz_planes: dO, loss and bias partials, published maxima bit for bit (the same decode
  tiles and epilogue); dZ to 1e-6 of its maximum (the fused form cuts every dO tile with the tile's
  own scale and sums per 128-item tile instead of per K chunk) and against a float64 product."""
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=B + h,
                                                 ratings=(loss == LOSS_MSE and h == 200))
  if lib.rk_decode_dz_fused_ok(B, h, blk.n_cap, loss) != 1:
    pytest.skip("the fused decode + dZ launch is switched off (plain bf16 operands)")
  st = current_stream()
  f = dict(dtype=torch.float32, device=Z.device)
  n_b, nnz, ld, S = blk.counts_host()
  npart = lib.rk_loss_partials(B, blk.n_cap)
  ntile = -(-B // lib.rk_decode_row_tile())
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  lib.rk_planes_tile(64)

  def run(fused):
    dO = torch.full((B * blk.ld_cap,), 5.0, **f)
    part = torch.zeros(npart, **f)
    gbp = torch.zeros(ntile * blk.ld_cap, **f)
    dz = torch.zeros(B, h, **f)
    blk.counts[8:72].zero_()
    if fused:
      ws = torch.full((lib.rk_dz_fused_workspace_bytes(B, h, blk.n_cap) // 4 + 64,), float("nan"), **f)
      check(lib.rk_decode_loss_dz_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                                         ptr(part), ptr(gbp), ptr(ws), st))
      check(lib.rk_decode_dz_reduce(ptr(ws), B, h, blk.ref, ptr(Z), 1, ptr(dz), st))
    else:
      ws = torch.zeros(lib.rk_dz_workspace_bytes(B, h) // 4 + 64, **f)
      check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                                      blk.ld_cap, ptr(part), ptr(gbp), st))
      check(lib.rk_decode_bwd_dz_planes(ptr(dO), B, ctypes.byref(pl), blk.ref, ptr(Z), 1, ptr(dz), ptr(ws), st))
    torch.cuda.synchronize()
    return dO[:B * ld].view(B, ld).clone(), part.clone(), gbp.clone(), blk.counts[8:72].clone(), dz
  a, b = run(False), run(True)
  lib.rk_planes_tile(0)
  for x, y in zip(a[:4], b[:4]):
    assert torch.equal(x, y)
  ref = (a[0][:, :n_b].double() @ W[blk.items[:n_b].long()].double()) * (1 - Z.double() ** 2)
  top = max(ref.abs().max().item(), 1e-30)
  assert (b[4].double() - ref).abs().max().item() <= 2e-6 * top
  assert (b[4].double() - a[4].double()).abs().max().item() <= 2e-6 * top
  assert bool(torch.isfinite(b[4]).all())


def test_plane_images_hold_the_split_operands():
  """hi + lo of every image entry reproduces s.x to 2^-22 (fp16 pair), the K padding is zero, the W^T
  image is the transpose of the W image (k-tile major) with zeros behind the live items."""
  B, h, n_items = 70, 40, 500
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, 600, n_items, 12, seed=5)
  st = current_stream()
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  torch.cuda.synchronize()
  n_b = blk.counts_host()[0]
  KT = -(-h // 32)
  raw = buf.view(torch.uint8)
  base = buf.data_ptr()
  sc = buf[:4].cpu().numpy()

  def image(p, rows, kt):
    off = p - base
    x = raw[off:off + rows * kt * 128].view(torch.float16).view(rows, kt, 2, 32).float()
    return x[:, :, 0, :].reshape(rows, kt * 32), x[:, :, 1, :].reshape(rows, kt * 32)
  zh, zl = image(pl.z, B, KT)
  assert torch.allclose((zh + zl)[:, :h] / float(sc[0]), Z, rtol=3e-7, atol=1e-9)
  assert float((zh + zl)[:, h:].abs().max()) == 0.0
  wh, wl = image(pl.w, n_b, KT)
  Wg = W[blk.items[:n_b].long()]
  assert torch.allclose((wh + wl)[:, :h] / float(sc[1]), Wg, rtol=3e-7, atol=1e-9)
  Hp = KT * 32
  nkt = -(-n_b // 32)
  off = pl.wt - base
  t = raw[off:off + nkt * Hp * 128].view(torch.float16).view(nkt, Hp, 2, 32).float()
  wt = (t[:, :, 0, :] + t[:, :, 1, :]).permute(1, 0, 2).reshape(Hp, nkt * 32)      # [j][item]
  assert torch.allclose(wt[:h, :n_b] / float(sc[1]), Wg.t(), rtol=3e-7, atol=1e-9)
  assert float(wt[:, n_b:].abs().max()) == 0.0 and float(wt[h:].abs().max()) == 0.0


def test_batched_collation_equals_per_block_collation():
  """rk_collate_at_multi (the G blocks of a group in one set of launches) against rk_collate_at per
  block: every array of every block bit-equal."""
  lib = _lib.load()
  dev = torch.device("cuda")
  csr = synth_csr(900, 700, 15, seed=3, ratings=True)
  dcsr = DeviceCSR(csr)
  S, G = 100, 4
  order = torch.from_numpy(np.random.RandomState(1).permutation(900).astype(np.int64)).to(dev)
  cursor = torch.tensor([3, 0], dtype=torch.int64, device=dev)      # step 3 of the epoch
  st = current_stream()
  cap = int(np.sort(dcsr.degrees)[-S:].sum())
  a = [Block(S, cap, 700, dev) for _ in range(G)]
  b = [Block(S, cap, 700, dev) for _ in range(G)]
  for g, blk in enumerate(a):
    blk.c.implicit = 0
    check(lib.rk_collate_at(ptr(dcsr.indptr), ptr(dcsr.indices), ptr(dcsr.data), ptr(order), S, 1,
                            ptr(cursor), g, blk.ref, st))
  arr = (ctypes.POINTER(RkBlock) * G)()
  for g, blk in enumerate(b):
    blk.c.implicit = 0
    arr[g] = ctypes.pointer(blk.c)
  check(lib.rk_collate_at_multi(ptr(dcsr.indptr), ptr(dcsr.indices), ptr(dcsr.data), ptr(order), S, 1,
                                ptr(cursor), 0, arr, G, 0, st))
  torch.cuda.synchronize()
  for x, y in zip(a, b):
    n_b, nnz = int(x.counts[0]), int(x.counts[1])
    assert torch.equal(x.counts[:8], y.counts[:8]) and n_b > 0
    assert torch.equal(x.items[:n_b], y.items[:n_b]) and torch.equal(x.pos, y.pos)
    assert torch.equal(x.indptr[:S + 1], y.indptr[:S + 1])
    assert torch.equal(x.cols[:nnz], y.cols[:nnz]) and torch.equal(x.vals[:nnz], y.vals[:nnz])
    wr = (n_b + 31) // 32
    assert torch.equal(x.bits_rc.view(S, -1)[:, :wr], y.bits_rc.view(S, -1)[:, :wr])
    assert torch.equal(x.bits_cr.view(x.n_cap, -1)[:n_b], y.bits_cr.view(y.n_cap, -1)[:n_b])


def test_device_csr_from_arrays_rejects_inconsistent_indptr():
  """a malformed CSR (indptr decreasing, or pointing past the index array) must be
  refused before the collation kernels index through it."""
  dev = torch.device("cuda")
  indptr = np.array([0, 2, 4, 6], dtype=np.int64)
  indices = np.array([0, 1, 0, 2, 1, 3], dtype=np.int32)
  DeviceCSR.from_arrays((3, 4), indptr, indices, None, dev)          # well formed
  with pytest.raises(ValueError, match="non-decreasing"):
    DeviceCSR.from_arrays((3, 4), np.array([0, 4, 2, 6], dtype=np.int64), indices, None, dev)
  with pytest.raises(ValueError, match="exceeds"):
    DeviceCSR.from_arrays((3, 4), np.array([0, 2, 4, 9], dtype=np.int64), indices, None, dev)
  with pytest.raises(ValueError, match="exceeds"):
    DeviceCSR.from_arrays((3, 4), indptr, indices, np.ones(4, dtype=np.float32), dev)


@pytest.mark.parametrize("B,h,n_items,row_off", [(500, 200, 3000, 0), (130, 64, 900, 0), (33, 20, 400, 0),
                                                 (300, 260, 2000, 0), (200, 128, 1500, 56)])
def test_dw_and_encoder_backward_in_one_launch_equal_the_two_launches(B, h, n_items, row_off):
  """rk_decode_bwd_dw2_encode_bwd (dw3.hip dw_encbwd_kernel: dW tiles || encoder-backward columns)
  against rk_decode_bwd_dw2 + rk_ae_encode_bwd: the same workgroup bodies, so the K slabs of dW, their
  count, G_en and the encoder-bias gradient agree bit for bit."""
  S = B + row_off
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(S, h, max(S, 600), n_items, 14, seed=B + h + 1)
  if lib.rk_dw_encode_bwd_fused_ok(row_off, B) != 1:
    pytest.skip("the fused dW || encoder-backward launch is switched off")
  st = current_stream()
  f = dict(dtype=torch.float32, device=Z.device)
  n_b, nnz, ld, _ = blk.counts_host()
  g = torch.Generator(device=Z.device)
  g.manual_seed(B)
  dO = torch.zeros(B * blk.ld_cap, **f)
  dO[:B * ld].view(B, ld)[:, :n_b] = torch.randn(B, n_b, generator=g, **f) * 1e-3
  blk.counts[8:72].zero_()
  blk.counts[8:9].copy_(dO.abs().max().reshape(1).view(torch.int32))
  Zb = Z[:B].contiguous()
  dZ0 = torch.randn(B, h, generator=g, **f) * 1e-2
  # svals: what the encoder forward leaves (any values do for the comparison)
  blk.svals[:nnz].copy_(torch.rand(nnz, generator=g, **f))
  wsz = lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64

  def run(fused):
    ws = torch.zeros(wsz, **f)
    G_en = torch.full((blk.n_cap * h,), 3.0, **f)
    gb = torch.full((h * 8,), 3.0, **f)
    blk.counts[4:5].zero_()
    if fused:
      check(lib.rk_decode_bwd_dw2_encode_bwd(ptr(dO), ptr(Zb), B, h, blk.ref, ptr(ws), None, ptr(ranges),
                                             row_off, ptr(dZ0), ptr(G_en), ptr(gb), st))
    else:
      check(lib.rk_decode_bwd_dw2(ptr(dO), ptr(Zb), B, h, blk.ref, None, None, ptr(ws), None, ptr(ranges), st))
      check(lib.rk_ae_encode_bwd(blk.ref, row_off, B, ptr(dZ0), h, ptr(G_en), 0, ptr(gb), st))
    torch.cuda.synchronize()
    ns = int(blk.counts[4].item())
    off = (lib.rk_dw3_planes_bytes(B, h) + 255) // 256 * 256 // 4
    slabs = ws[off:off + ns * blk.n_cap * h].view(ns, blk.n_cap, h)[:, :n_b].clone()
    return ns, slabs, G_en[:n_b * h].clone(), gb[:h].clone()
  a, b = run(False), run(True)
  assert a[0] == b[0] and a[0] >= 1
  assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
  assert float(a[1].abs().max()) > 0 and float(a[2].abs().max()) > 0
  # ... and with dO's column sums (the multinomial loss's decoder bias gradient) as a third workgroup
  # range: everything else unchanged, the sums equal to rk_colsum's up to the order of 8 row slices
  ws = torch.zeros(wsz, **f)
  G_en = torch.full((blk.n_cap * h,), 3.0, **f)
  gb = torch.full((h * 8,), 3.0, **f)
  gb_de = torch.full((blk.ld_cap,), 7.0, **f)
  want = torch.empty(blk.ld_cap, **f)
  blk.counts[4:5].zero_()
  check(lib.rk_decode_bwd_dw2_encode_bwd_colsum(ptr(dO), ptr(Zb), B, h, blk.ref, ptr(ws), None, ptr(ranges),
                                                row_off, ptr(dZ0), ptr(G_en), ptr(gb), ptr(gb_de), st))
  check(lib.rk_colsum(ptr(dO), B, blk.n_cap, 0, ptr(blk.counts), ptr(want), st))
  torch.cuda.synchronize()
  off = (lib.rk_dw3_planes_bytes(B, h) + 255) // 256 * 256 // 4
  assert int(blk.counts[4].item()) == b[0]
  assert torch.equal(ws[off:off + b[0] * blk.n_cap * h].view(b[0], blk.n_cap, h)[:, :n_b], b[1])
  assert torch.equal(G_en[:n_b * h], b[2]) and torch.equal(gb[:h], b[3])
  ref64 = dO[:B * ld].view(B, ld)[:, :n_b].double().sum(0)
  assert torch.allclose(gb_de[:n_b].double(), ref64, rtol=1e-5, atol=1e-8)
  assert torch.allclose(gb_de[:n_b], want[:n_b], rtol=1e-5, atol=1e-8)
  assert bool((gb_de[n_b:] == 7.0).all())          # nothing past the live columns is written


@pytest.mark.parametrize("B,h,n_items", [(500, 128, 3000), (37, 20, 400), (1, 8, 97), (300, 512, 2000)])
def test_split_wz_is_the_two_split_launches(B, h, n_items):
  """rk_split_wz (one launch) leaves the images rk_split_w + rk_split_z leave, byte for byte, also with
  a published (non-static) range of Z."""
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=7 * B + h)
  st = current_stream()
  Zu = Z * 37.5                                   # unbounded activation: the range comes from rk_amax
  check(lib.rk_amax(ptr(Zu), B * h, ptr(ranges), st))
  buf.zero_()
  check(lib.rk_split_wz(ptr(W), ptr(Zu), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  torch.cuda.synchronize()
  two = buf.clone()
  buf.zero_()
  check(lib.rk_split_wz(ptr(W), ptr(Zu), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  torch.cuda.synchronize()
  assert torch.equal(two.view(torch.int32), buf.view(torch.int32))


@pytest.mark.parametrize("B,d,act", [(500, 128, 0), (37, 20, 3), (1, 8, 4), (1000, 64, 1)])
def test_gather_rows_amax_is_gather_plus_amax(B, d, act):
  """rk_gather_rows_amax == a gather (torch) followed by rk_amax: the same rows, and the maximum over the
  64 published slots (what the split kernels use) is max |out| exactly."""
  lib = _lib.load()
  dev = torch.device("cuda")
  g = torch.Generator(device=dev)
  g.manual_seed(B + d)
  E = torch.randn(5000, d, generator=g, device=dev) * 3.0
  rows = torch.randint(0, 5000, (B,), generator=g, device=dev, dtype=torch.int64)
  st = current_stream()
  out0 = torch.empty(B * d, device=dev)
  out1 = torch.empty(B * d, device=dev)
  r0 = torch.full((128,), 7, dtype=torch.int32, device=dev)
  r1 = torch.full((128,), 7, dtype=torch.int32, device=dev)
  out0 = E[rows].contiguous()
  out0 = {0: out0, 1: torch.tanh(out0), 3: torch.relu(out0),
          4: torch.nn.functional.selu(out0)}[act].reshape(-1).contiguous()
  check(lib.rk_amax(ptr(out0), B * d, ptr(r0), st))
  r32 = torch.full((B + 3,), -5, dtype=torch.int32, device=dev)
  check(lib.rk_gather_rows_amax(ptr(E), ptr(rows), B, d, act, ptr(out1), ptr(r1), ptr(r32), st))
  out2 = torch.empty(B * d, device=dev)
  check(lib.rk_gather_rows_amax(ptr(E), ptr(rows), B, d, act, ptr(out2), None, None, st))   # plain gather
  torch.cuda.synchronize()
  assert torch.equal(out1, out2)
  assert torch.allclose(out0, out1, rtol=2e-6, atol=1e-7)            # (tanh / selu: torch's vs the library's)
  # the rows as an int32 index array behind their count (a SparseAdam job's rows / n_dev)
  assert int(r32[0]) == B and torch.equal(r32[1:B + 1].long(), rows) and bool((r32[B + 1:] == -5).all())
  assert int(r1[:64].max()) == int(out1.abs().max().view(torch.int32))
  assert abs(int(r0[:64].max()) - int(r1[:64].max())) <= 4           # (bit patterns: a few ulp between the two tanh)
  assert torch.equal(r0[64:], r1[64:])            # the W half is not touched


@pytest.mark.parametrize("B,h,n_items,scale", [(500, 128, 3000, 37.5), (130, 64, 900, 1.0), (33, 20, 400, 1e-3),
                                               (300, 260, 2000, 5.0)])
def test_split_wz_zt_leaves_the_planes_dw2_would_make(B, h, n_items, scale):
  """rk_split_wz (dw_workspace given) writes Z^T as the dW kernel's fp16 pair planes (+ their scale) at the head of the dW
  workspace: rk_decode_bwd_dw2 called with zt_planes == workspace then produces the K slabs it produces
  when it makes the planes itself (zt_planes == NULL), bit for bit -- bounded and unbounded ranges of Z."""
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=3 * B + h)
  if lib.rk_split_zt_ok() != 1:
    pytest.skip("dW is not on fp16 pairs")
  st = current_stream()
  f = dict(dtype=torch.float32, device=Z.device)
  n_b, nnz, ld, _ = blk.counts_host()
  g = torch.Generator(device=Z.device)
  g.manual_seed(B)
  dO = torch.zeros(B * blk.ld_cap, **f)
  dO[:B * ld].view(B, ld)[:, :n_b] = torch.randn(B, n_b, generator=g, **f) * 1e-3
  blk.counts[8:72].zero_()
  blk.counts[8:9].copy_(dO.abs().max().reshape(1).view(torch.int32))
  Zu = (Z * scale).contiguous()
  if scale != 1.0:
    check(lib.rk_amax(ptr(Zu), B * h, ptr(ranges), st))       # (scale == 1: |Z| <= 1, the static range)
  wsz = lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64
  off = (lib.rk_dw3_planes_bytes(B, h) + 255) // 256 * 256 // 4

  def run(pre):
    ws = torch.full((wsz,), 3.0, **f)
    blk.counts[4:5].zero_()
    if pre:
      check(lib.rk_split_wz(ptr(W), ptr(Zu), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), ptr(ws), st))
    check(lib.rk_decode_bwd_dw2(ptr(dO), ptr(Zu), B, h, blk.ref, None, None, ptr(ws), ptr(ws) if pre else None,
                                ptr(ranges), st))
    torch.cuda.synchronize()
    ns = int(blk.counts[4].item())
    return ns, ws[:off].clone(), ws[off:off + ns * blk.n_cap * h].view(ns, blk.n_cap, h)[:, :n_b].clone()
  a, b = run(False), run(True)
  rp, cp = lib.rk_dw3_rows_pad(B), lib.rk_dw3_cols_pad(h)
  used = 2 * rp * cp * 2 // 4 + 1                     # two 16-bit planes + the scale
  assert a[0] == b[0] >= 1
  assert torch.equal(a[1][:used].view(torch.int32), b[1][:used].view(torch.int32))
  assert torch.equal(a[2], b[2]) and float(a[2].abs().max()) > 0


@pytest.mark.parametrize("B,h,n_items,act", [(500, 128, 3000, 0), (130, 64, 900, 1), (33, 20, 400, 3)])
def test_dw_and_dz_reduce_in_one_launch_equal_the_two_launches(B, h, n_items, act):
  """rk_decode_bwd_dw2_dz_reduce (dw3.hip dw_reduce_kernel: dW tiles || the slab reduce of the fused
  decode's dZ partials) against rk_decode_bwd_dw2 + rk_decode_dz_reduce: the same workgroup bodies, so
  the K slabs of dW and dZ agree bit for bit (with and without act' folded in)."""
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(B, h, max(B, 600), n_items, 14, seed=5 * B + h)
  if lib.rk_split_zt_ok() != 1:
    pytest.skip("dW is not on fp16 pairs")
  st = current_stream()
  f = dict(dtype=torch.float32, device=Z.device)
  n_b, nnz, ld, _ = blk.counts_host()
  g = torch.Generator(device=Z.device)
  g.manual_seed(B + 1)
  dO = torch.zeros(B * blk.ld_cap, **f)
  dO[:B * ld].view(B, ld)[:, :n_b] = torch.randn(B, n_b, generator=g, **f) * 1e-3
  blk.counts[8:72].zero_()
  blk.counts[8:9].copy_(dO.abs().max().reshape(1).view(torch.int32))
  Zb = Z[:B].contiguous()
  n_tiles = -(-blk.n_cap // 128)
  dz_ws = torch.randn(n_tiles * B * h, generator=g, **f)          # one slab per 128-item column tile
  wsz = lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64
  off = (lib.rk_dw3_planes_bytes(B, h) + 255) // 256 * 256 // 4

  def run(fused, zact):
    ws = torch.zeros(wsz, **f)
    dZ = torch.full((B * h,), 9.0, **f)
    blk.counts[4:5].zero_()
    if fused:
      check(lib.rk_decode_bwd_dw2_dz_reduce(ptr(dO), ptr(Zb), B, h, blk.ref, ptr(ws), None, ptr(ranges),
                                            ptr(dz_ws), ptr(Zb) if zact else None, act, ptr(dZ), st))
    else:
      check(lib.rk_decode_bwd_dw2(ptr(dO), ptr(Zb), B, h, blk.ref, None, None, ptr(ws), None, ptr(ranges), st))
      check(lib.rk_decode_dz_reduce(ptr(dz_ws), B, h, blk.ref, ptr(Zb) if zact else None, act, ptr(dZ), st))
    torch.cuda.synchronize()
    ns = int(blk.counts[4].item())
    return ns, ws[off:off + ns * blk.n_cap * h].view(ns, blk.n_cap, h)[:, :n_b].clone(), dZ
  for zact in (False, True):
    a, b = run(False, zact), run(True, zact)
    assert a[0] == b[0] >= 1
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    live = -(-n_b // 128)
    want = dz_ws.view(n_tiles, B * h)[:live].double().sum(0)
    if not zact:
      assert torch.allclose(a[2].double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,h,n_items,row_off", [(2500, 64, 9000, 0), (2100, 32, 6000, 37), (700, 64, 5000, 0)])
def test_encoder_backward_over_row_windows(B, h, n_items, row_off):
  """rk_ae_encode_bwd on a row window of more than 2048 rows over a long item set: the wave-per-column
  kernel once per window of <= 2016 rows, accumulating (csrc/encoder.hip) -- G_en[c] = sum_r svals[r, c] dZ[r]
  and gb_en = column sums of dZ against float64."""
  S = B + row_off
  lib, blk, W, bias, Z, ranges, pl, buf = _setup(S, h, S, n_items, 6, seed=B + h)
  st = current_stream()
  dev = Z.device
  f = dict(dtype=torch.float32, device=dev)
  n_b, nnz, ld, _ = blk.counts_host()
  assert blk.n_cap >= 4096 or B <= 2048
  g = torch.Generator(device=dev)
  g.manual_seed(B)
  dZ0 = torch.randn(B, h, generator=g, **f) * 1e-2
  blk.svals[:nnz].copy_(torch.rand(nnz, generator=g, **f))
  G_en = torch.full((blk.n_cap * h,), 3.0, **f)
  gb = torch.full((h * 8,), 3.0, **f)
  check(lib.rk_ae_encode_bwd(blk.ref, row_off, B, ptr(dZ0), h, ptr(G_en), 0, ptr(gb), st))
  torch.cuda.synchronize()
  indptr = blk.indptr[:S + 1].cpu().numpy()
  cols = blk.cols[:nnz].cpu().numpy()
  sv = blk.svals[:nnz].double().cpu().numpy()
  dz = dZ0.double().cpu().numpy()
  want = np.zeros((n_b, h))
  for r in range(B):
    lo, hi = indptr[row_off + r], indptr[row_off + r + 1]
    np.add.at(want, cols[lo:hi], sv[lo:hi, None] * dz[r][None, :])
  got = G_en[:n_b * h].view(n_b, h).double().cpu().numpy()
  assert np.abs(got - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-30)
  assert np.abs(gb[:h].double().cpu().numpy() - dz.sum(0)).max() <= 1e-5 * np.abs(dz).sum(0).max()


def test_zero_tail_rows_clears_exactly_the_rows_past_the_live_count():
  """rk_zero_tail_rows: rows [n_b, top) of up to four compact [n_cap, h] arrays <- 0 with n_b read on the
  device -- what a replayed data-parallel step does in front of its capacity-sized, in-place summed exchange.  Without a
  high-water mark top = n_cap; with one, top = max(mark, n_b) and the mark moves up."""
  lib = _lib.load()
  dev = torch.device("cuda")
  st = current_stream()
  n_cap = 777
  widths = [200, 1, 64, 8]
  for n_b in (0, 1, 300, 776, 777, 900):
    counts = torch.tensor([n_b, 0, 0, 0], dtype=torch.int32, device=dev)
    arrs = [torch.full((n_cap * w + 5,), 3.0, device=dev) for w in widths]       # (+5: nothing past the array is touched)
    X = (ctypes.c_void_p * 4)(*[a.data_ptr() for a in arrs])
    H = (ctypes.c_int32 * 4)(*widths)
    check(lib.rk_zero_tail_rows(X, H, 4, ptr(counts), n_cap, None, st))
    torch.cuda.synchronize()
    live = min(max(n_b, 0), n_cap)
    for a, w in zip(arrs, widths):
      assert bool((a[:live * w] == 3.0).all()) and bool((a[live * w:n_cap * w] == 0.0).all())
      assert bool((a[n_cap * w:] == 3.0).all())
  # with the mark: a sequence of steps whose live rows are rewritten with "gradients" and whose tails must read zero
  hwm = torch.zeros(1, dtype=torch.int32, device=dev)
  arrs = [torch.zeros(n_cap * w, device=dev) for w in widths]
  X = (ctypes.c_void_p * 4)(*[a.data_ptr() for a in arrs])
  H = (ctypes.c_int32 * 4)(*widths)
  top = 0
  for step, n_b in enumerate((400, 380, 500, 120, 777, 300, 301)):
    for a, w in zip(arrs, widths):
      a[:n_b * w] = float(step + 1)                       # the step's kernels rewrite the live rows only
    counts = torch.tensor([n_b, 0, 0, 0], dtype=torch.int32, device=dev)
    check(lib.rk_zero_tail_rows(X, H, 4, ptr(counts), n_cap, ptr(hwm), st))
    torch.cuda.synchronize()
    top = max(top, n_b)
    assert int(hwm[0]) == top
    for a, w in zip(arrs, widths):
      assert bool((a[:n_b * w] == float(step + 1)).all()) and bool((a[n_b * w:] == 0.0).all())
