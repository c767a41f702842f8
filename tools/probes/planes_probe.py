#!/usr/bin/env python
"""Pre-split operand planes (csrc/decode16.hip) against the in-loop split kernels (csrc/gemm.hip) on
a collated C2-like block: bitwise comparison of dO / loss partials / bias partials / dZ, and timings.

    python tools/probes/planes_probe.py [B ...]      (H=200 NUSERS=20000 env)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from recoder_amd import _lib, synthetic  # noqa: E402
from recoder_amd._lib import LOSS_BCE, LOSS_MSE, LOSS_NONE, RkPlanes, check, ptr  # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream  # noqa: E402


def timeit(fn, n=40, warm=5):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  evs = []
  for _ in range(n):
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    evs.append((s, e))
  torch.cuda.synchronize()
  t = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
  return t[len(t) // 2]


def main():
  Bs = [int(x) for x in sys.argv[1:]] or [500]
  lib = _lib.load()
  dev = torch.device("cuda")
  h = int(os.environ.get("H", "200"))
  act = int(os.environ.get("ACT", "1"))
  if os.environ.get("DATA", "ml20m") == "ml20m":
    csr = synthetic.ml20m_like(seed=0, n_users=int(os.environ.get("NUSERS", "20000")))
  else:
    csr = synthetic.uniform(int(os.environ.get("NUSERS", "20000")), int(os.environ.get("NITEMS", "1000000")), 100)
  dcsr = DeviceCSR(csr)
  n_items = csr.shape[1]
  f = dict(dtype=torch.float32, device=dev)
  torch.manual_seed(0)
  W = torch.randn(n_items, h, **f) * 0.05
  bias = torch.randn(n_items, **f) * 0.01
  st = current_stream()
  for B in Bs:
    users = torch.arange(B, dtype=torch.int64, device=dev)
    blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev)
    blk.collate(dcsr, users)
    n_b, nnz, ld, S = blk.counts_host()
    Z = torch.tanh(torch.randn(B, h, **f))
    ranges = torch.zeros(128, dtype=torch.int32, device=dev)
    ranges[64:65].copy_(W.abs().max().reshape(1).view(torch.int32))
    buf = torch.zeros(lib.rk_planes_bytes(B, h, blk.n_cap) // 4 + 64, **f)
    pl = RkPlanes()
    check(lib.rk_planes_layout(ptr(buf), B, h, blk.n_cap, ctypes.byref(pl)))
    npart = lib.rk_loss_partials(B, blk.n_cap)
    rt = lib.rk_decode_row_tile()
    ntile = -(-B // rt)

    def run_old(loss):
      dO = torch.zeros(B * blk.ld_cap, **f)
      part = torch.zeros(npart, **f)
      gbp = torch.zeros(ntile * blk.ld_cap, **f)
      blk.counts[8:72].zero_()
      check(lib.rk_decode_loss(ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                               blk.ld_cap, ptr(part), ptr(gbp), ptr(ranges), st))
      return dO, part, gbp, blk.counts[8:72].clone()

    def run_new(loss):
      dO = torch.zeros(B * blk.ld_cap, **f)
      part = torch.zeros(npart, **f)
      gbp = torch.zeros(ntile * blk.ld_cap, **f)
      blk.counts[8:72].zero_()
      check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
      check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.5, 1.0 / B, ptr(dO),
                                      blk.ld_cap, ptr(part), ptr(gbp), st))
      return dO, part, gbp, blk.counts[8:72].clone()

    for loss, name in ((LOSS_MSE, "mse"), (LOSS_BCE, "bce"), (LOSS_NONE, "store")):
      o = run_old(loss)
      for tile in (64, 128):
        lib.rk_planes_tile(tile)
        nw = run_new(loss)
        v_old = o[0][:B * ld].view(B, ld)[:, :n_b]
        v_new = nw[0][:B * ld].view(B, ld)[:, :n_b]
        same = torch.equal(v_old, v_new)
        md = (v_old - v_new).abs().max().item()
        l_old, l_new = o[1].double().sum().item(), nw[1].double().sum().item()
        gb_old = o[2][:ntile * ld].view(ntile, ld)[:, :n_b].sum(0)
        gb_new = nw[2][:ntile * ld].view(ntile, ld)[:, :n_b].sum(0)
        print("B=%d n_b=%d %s tile %d: dO bit-equal %s (max diff %.3g) loss %.9g vs %.9g  gb max diff %.3g  amax %g vs %g"
              % (B, n_b, name, tile, same, md, l_old, l_new, (gb_old - gb_new).abs().max().item(),
                 o[3].view(torch.float32).max().item(), nw[3].view(torch.float32).max().item()), flush=True)
    # dZ
    dO = torch.zeros(B * blk.ld_cap, **f)
    part = torch.zeros(npart, **f)
    gbp = torch.zeros(ntile * blk.ld_cap, **f)
    blk.counts[8:72].zero_()
    lib.rk_planes_tile(128)
    check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
    check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.5, 1.0 / B, ptr(dO),
                                    blk.ld_cap, ptr(part), ptr(gbp), st))
    ws = torch.zeros(lib.rk_dz_workspace_bytes(B, h) // 4 + 64, **f)
    dz_old = torch.zeros(B, h, **f)
    dz_new = torch.zeros(B, h, **f)
    check(lib.rk_decode_bwd_dz(ptr(dO), B, h, blk.ref, ptr(W), ptr(Z), act, ptr(dz_old), ptr(ws), ptr(ranges), st))
    ws.zero_()
    check(lib.rk_decode_bwd_dz_planes(ptr(dO), B, ctypes.byref(pl), blk.ref, ptr(Z), act, ptr(dz_new), ptr(ws), st))
    ref = (dO[:B * ld].view(B, ld)[:, :n_b].double() @ W[blk.items[:n_b].long()].double()) * (1 - Z.double() ** 2)
    print("dZ bit-equal %s max diff %.3g; vs float64: old %.3g new %.3g (scale %.3g)" % (
        torch.equal(dz_old, dz_new), (dz_old - dz_new).abs().max().item(),
        (dz_old.double() - ref).abs().max().item(), (dz_new.double() - ref).abs().max().item(),
        ref.abs().max().item()), flush=True)
    # timings
    r = {}
    r["split_w"] = timeit(lambda: check(lib.rk_split_wz(ptr(W), None, 0, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st)))
    r["split_z"] = timeit(lambda: check(lib.rk_split_wz(None, ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st)))
    r["dec_old"] = timeit(lambda: check(lib.rk_decode_loss(
        ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), LOSS_MSE, 0.5, 1.0 / B, ptr(dO), blk.ld_cap, ptr(part),
        ptr(gbp), ptr(ranges), st)))
    for tile in (64, 128):
      lib.rk_planes_tile(tile)
      r["dec_new%d" % tile] = timeit(lambda: check(lib.rk_decode_loss_planes(
          ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.5, 1.0 / B, ptr(dO), blk.ld_cap, ptr(part),
          ptr(gbp), st)))
      r["store_new%d" % tile] = timeit(lambda: check(lib.rk_decode_loss_planes(
          ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_NONE, 0.5, 1.0 / B, ptr(dO), blk.ld_cap, None,
          None, st)))
    check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.5, 1.0 / B, ptr(dO),
                                    blk.ld_cap, ptr(part), ptr(gbp), st))
    r["dz_old"] = timeit(lambda: check(lib.rk_decode_bwd_dz(
        ptr(dO), B, h, blk.ref, ptr(W), ptr(Z), act, ptr(dz_old), ptr(ws), ptr(ranges), st)))
    r["dz_new"] = timeit(lambda: check(lib.rk_decode_bwd_dz_planes(
        ptr(dO), B, ctypes.byref(pl), blk.ref, ptr(Z), act, ptr(dz_new), ptr(ws), st)))
    ws3 = torch.zeros(lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
    r["dw_bf16x3"] = timeit(lambda: check(lib.rk_decode_bwd_dw3(
        ptr(dO), ptr(Z), B, h, blk.ref, None, None, ptr(ws3), None, st)), n=10, warm=2)
    r["dw_fp16x2"] = timeit(lambda: check(lib.rk_decode_bwd_dw2(
        ptr(dO), ptr(Z), B, h, blk.ref, None, None, ptr(ws3), None, ptr(ranges), st)), n=10, warm=2)
    gf = 2.0 * B * h * n_b / 1e9
    print("B=%5d n_b=%6d h=%d | " % (B, n_b, h) + " ".join("%s %.1fus" % kv for kv in r.items()) +
          " | GEMM %.2f GF" % gf, flush=True)


if __name__ == "__main__":
  main()
