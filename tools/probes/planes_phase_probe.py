#!/usr/bin/env python
"""Per-workgroup phase timeline of the plane kernels (rk_planes_probe): when each workgroup starts,
how long its prologue / k-loop / epilogue take.   python tools/probes/planes_phase_probe.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from recoder_amd import _lib, synthetic  # noqa: E402
from recoder_amd._lib import LOSS_MSE, LOSS_NONE, RkPlanes, check, ptr  # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream  # noqa: E402


def report(name, buf, tick_us=0.01):
  a = buf.cpu().numpy().reshape(-1, 8)
  live = a[a[:, 4] > 0]
  t0 = live[:, 0].min()
  rel = (live[:, :4].astype(np.float64) - t0) * tick_us
  start, pro, loop, epi = rel[:, 0], rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 1], rel[:, 3] - rel[:, 2]
  q = lambda x: "min %5.1f med %5.1f p90 %5.1f max %5.1f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
  print("%s: %d live workgroups, last end %.1f us" % (name, len(live), rel[:, 3].max()))
  for lab, x in (("start", start), ("prologue", pro), ("k-loop", loop), ("epilogue", epi)):
    print("   %-9s %s" % (lab, q(x)))
  if (live[:, 5] > 0).all():          # the fused decode + dZ launch: where its dZ part starts
    mid = (live[:, 5].astype(np.float64) - t0) * tick_us
    print("   %-9s %s" % ("  loss ep.", q(mid - rel[:, 2])))
    print("   %-9s %s" % ("  dZ part", q(rel[:, 3] - mid)))


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
  lib = _lib.load()
  dev = torch.device("cuda")
  h = int(os.environ.get("H", "200"))
  csr = synthetic.ml20m_like(seed=0, n_users=20000)
  dcsr = DeviceCSR(csr)
  n_items = csr.shape[1]
  f = dict(dtype=torch.float32, device=dev)
  W = torch.randn(n_items, h, **f) * 0.05
  bias = torch.zeros(n_items, **f)
  st = current_stream()
  users = torch.arange(B, dtype=torch.int64, device=dev)
  blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev)
  blk.collate(dcsr, users)
  Z = torch.tanh(torch.randn(B, h, **f))
  dZ = torch.empty(B, h, **f)
  dO = torch.zeros(B * blk.ld_cap, **f)
  ws = torch.empty(lib.rk_dz_workspace_bytes(B, h) // 4, **f)
  part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
  gbp = torch.empty((B // 32 + 1) * blk.ld_cap, **f)
  ranges = torch.zeros(128, dtype=torch.int32, device=dev)
  ranges[64:65].copy_(W.abs().max().reshape(1).view(torch.int32))
  buf = torch.zeros(lib.rk_planes_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  pl = RkPlanes()
  check(lib.rk_planes_layout(ptr(buf), B, h, blk.n_cap, ctypes.byref(pl)))
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  probe = torch.zeros(8 * 200000, dtype=torch.int64, device=dev)

  def dec(loss):
    return lambda: lib.rk_decode_loss_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), loss, 0.0, 1.0 / B,
                                             ptr(dO), blk.ld_cap, ptr(part), ptr(gbp), st)
  calls = []
  for tile in (64, 128):
    calls.append(("decode+mse tile %d" % tile, tile, dec(LOSS_MSE)))
    calls.append(("decode store tile %d" % tile, tile, dec(LOSS_NONE)))
  wsf = torch.empty(lib.rk_dz_fused_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  calls.append(("decode+mse+dZ partials (fused) tile 64", 64, lambda: lib.rk_decode_loss_dz_planes(
      ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(dO), ptr(part), ptr(gbp), ptr(wsf), st)))
  calls.append(("dz (GEMM + reduce)", 64, lambda: lib.rk_decode_bwd_dz_planes(
      ptr(dO), B, ctypes.byref(pl), blk.ref, None, 0, ptr(dZ), ptr(ws), st)))
  for name, tile, fn in calls:
    lib.rk_planes_tile(tile)
    for _ in range(3):
      check(fn())
    torch.cuda.synchronize()
    probe.zero_()
    lib.rk_planes_probe(ptr(probe))
    check(fn())
    torch.cuda.synchronize()
    lib.rk_planes_probe(None)
    report(name, probe)


if __name__ == "__main__":
  main()
