#!/usr/bin/env python
"""Per-workgroup phase timeline of the bf16-pipe dW kernel (csrc/dw3.hip, rk_dw3_probe):
    python tools/dw3_probe.py [B] [h]
prints start / first tile landed / per-k-tile / epilogue times (us, wall_clock64 at 100 MHz)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recoder_amd import _lib, synthetic  # noqa: E402
from recoder_amd._lib import check, ptr  # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream  # noqa: E402


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
  h = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  lib = _lib.load()
  dev = torch.device("cuda")
  csr = synthetic.ml20m_like(seed=0, n_users=20000)
  dcsr = DeviceCSR(csr)
  users = torch.arange(B, dtype=torch.int64, device=dev)
  blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), csr.shape[1], dev)
  blk.collate(dcsr, users)
  n_b = blk.counts_host()[0]
  f = dict(dtype=torch.float32, device=dev)
  Z = torch.randn(B, h, **f)
  dO = torch.randn(B * blk.ld_cap + 64, **f)
  ws = torch.zeros(lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  st = current_stream()
  n_wg = 4096
  probe = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)
  for it in range(3):
    probe.zero_()
    lib.rk_dw3_probe(probe.data_ptr())
    check(lib.rk_decode_bwd_dw3(ptr(dO), ptr(Z), B, h, blk.ref, None, None, ptr(ws), None, st))
    torch.cuda.synchronize()
  lib.rk_dw3_probe(None)
  p = probe.cpu().numpy().reshape(n_wg, 16)
  live = p[p[:, 0] != 0]
  t0 = live[:, 0].min()
  us = lambda x: (x - t0) / 100.0
  nk = int(live[:, 15].max())
  print("B %d h %d n_b %d: %d live workgroups, %d k-tiles, slabs %d" %
        (B, h, n_b, len(live), nk, int(blk.counts[4].item())))
  def stat(name, v):
    v = np.sort(v)
    print("   %-14s min %5.1f med %5.1f p90 %5.1f max %5.1f" % (name, v[0], v[len(v) // 2], v[int(len(v) * .9)], v[-1]))
  stat("start", us(live[:, 0]))
  stat("prologue", (live[:, 1] - live[:, 0]) / 100.0)
  stat("k-loop", (live[:, 13] - live[:, 1]) / 100.0)
  stat("per k-tile", (live[:, 13] - live[:, 1]) / 100.0 / nk)
  stat("epilogue", (live[:, 14] - live[:, 13]) / 100.0)
  stat("end", us(live[:, 14]))


if __name__ == "__main__":
  main()
