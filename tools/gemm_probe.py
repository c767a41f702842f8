#!/usr/bin/env python
"""Per-workgroup phase timeline of the three GEMMs (rk_gemm_probe): when each workgroup
starts, how long its prologue / k-loop / epilogue take.   python tools/gemm_probe.py [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recoder_amd import _lib, synthetic  # noqa: E402
from recoder_amd._lib import LOSS_MSE, check, ptr  # noqa: E402
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


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
  lib = _lib.load()
  dev = torch.device("cuda")
  h = 200
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
  Z = torch.randn(B, h, **f)
  dZ = torch.empty(B, h, **f)
  dO = torch.empty(B * blk.ld_cap, **f)
  G = torch.empty(blk.n_cap * h, **f)
  ws = torch.empty(lib.rk_dz_workspace_bytes(B, h) // 4, **f)
  part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
  gbp = torch.empty((B // 32 + 1) * blk.ld_cap, **f)
  probe = torch.zeros(8 * 200000, dtype=torch.int64, device=dev)
  calls = {
    "decode+loss": lambda: lib.rk_decode_loss(ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), LOSS_MSE, 0.0,
                                              1.0 / B, ptr(dO), 0, ptr(part), ptr(gbp), None, st),
    "dz (split-K GEMM only)": lambda: lib.rk_decode_bwd_dz(ptr(dO), B, h, blk.ref, ptr(W), None, 0,
                                                           ptr(dZ), ptr(ws), None, st),
    "dw": lambda: lib.rk_decode_bwd_dw(ptr(dO), ptr(Z), B, h, blk.ref, ptr(G), None, st),
  }
  for name, fn in calls.items():
    for _ in range(3):
      check(fn())
    torch.cuda.synchronize()
    probe.zero_()
    lib.rk_gemm_probe(ptr(probe))
    check(fn())
    torch.cuda.synchronize()
    lib.rk_gemm_probe(None)
    report(name, probe)


if __name__ == "__main__":
  main()
