#!/usr/bin/env python
"""Race screen of the ring k-loop (csrc/pgemm.h: counted vmcnt waits, raw s_barrier, asm transpose reads): the dW
tiles of a C2-shaped block, thousands of launches per ring depth against the compiler-scheduled two-stage loop's
result, bit for bit, alone and next to a stream that keeps the memory system busy.
    python tools/probes/dw_race_screen.py [launches]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recoder_amd import _lib, synthetic                                   # noqa: E402
from recoder_amd._lib import LOSS_MSE, RkPlanes, check, ptr               # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream          # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
  lib = _lib.load()
  dev = torch.device("cuda")
  csr = synthetic.ml20m_like(seed=0)
  dcsr = DeviceCSR(csr)
  f = dict(dtype=torch.float32, device=dev)
  g = torch.Generator(device=dev); g.manual_seed(1)
  st = current_stream()
  bad = 0
  for B, h in ((500, 200), (500, 128), (1100, 200), (4000, 200)):
    n_items = csr.shape[1]
    W = torch.randn(n_items, h, generator=g, **f) * 0.07
    bias = torch.randn(n_items, generator=g, **f) * 0.02
    users = torch.from_numpy(np.random.RandomState(B).permutation(csr.shape[0])[:B]).to(dev)
    blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev)
    blk.collate(dcsr, users)
    Z = torch.tanh(torch.randn(B, h, generator=g, **f))
    ranges = torch.zeros(128, dtype=torch.int32, device=dev)
    ranges[64:65].copy_(W.abs().max().reshape(1).view(torch.int32))
    buf = torch.zeros(lib.rk_planes_bytes(B, h, blk.n_cap) // 4 + 64, **f)
    pl = RkPlanes()
    check(lib.rk_planes_layout(ptr(buf), B, h, blk.n_cap, ctypes.byref(pl)))
    check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
    rows_img = -(-B // 32) * 32
    img = torch.zeros((rows_img + 256) * blk.ld_cap * 2, dtype=torch.int16, device=dev)
    sc = torch.ones(lib.rk_pg_scale_floats(B, blk.n_cap), **f)
    part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
    gbp = torch.zeros(-(-B // 64) * blk.ld_cap, **f)
    gr, gc = ctypes.c_int32(), ctypes.c_int32()
    lib.rk_pg_decode_granule(B, blk.n_cap, ctypes.byref(gr), ctypes.byref(gc))
    check(lib.rk_pg_decode_loss(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(img), rows_img,
                                ptr(sc), None, ptr(part), ptr(gbp), st))
    nw = lib.rk_pg_dw_workspace_bytes(B, h, blk.n_cap) // 4
    ws = torch.zeros(lib.rk_pg_dz_workspace_bytes(B, h) // 4 + 64, **f)
    n_b = blk.counts_host()[0]

    def dw(out):
      check(lib.rk_pg_dw(ptr(img), ptr(sc), gr.value, gc.value, B, ctypes.byref(pl), blk.ref, ptr(out), None, st))

    def dz(out):
      check(lib.rk_pg_dz(ptr(img), ptr(sc), gr.value, gc.value, B, ctypes.byref(pl), blk.ref, ptr(Z), 1, ptr(out), ptr(ws), st))
    lib.rk_tune(13, 0)
    ref = torch.full((nw,), float("nan"), **f); dw(ref)
    refz = torch.full((B * h,), float("nan"), **f); dz(refz)
    torch.cuda.synchronize()
    live = int(blk.counts[4].item())
    side = torch.cuda.Stream()
    noise = torch.zeros(48 << 20, **f)
    for ring in ((2, 4) if B < 1024 else (2,)):
      lib.rk_tune(13, ring)
      for load in (False, True):
        wrong = wrongz = 0
        out = torch.full((nw,), float("nan"), **f)
        outz = torch.full((B * h,), float("nan"), **f)
        for it in range(n if B < 1024 else max(50, n // 20)):
          if load:
            with torch.cuda.stream(side):
              noise.add_(1.0)
          dw(out); dz(outz)
          if it % 8 == 7 or not load:
            torch.cuda.synchronize()
            wrong += int(not torch.equal(out.view(-1, blk.n_cap, h)[:live, :n_b], ref.view(-1, blk.n_cap, h)[:live, :n_b]))
            wrongz += int(not torch.equal(outz, refz))
        torch.cuda.synchronize()
        print("B=%d h=%d n_b=%d ring=%d %s: dW mismatches %d, dZ mismatches %d" % (
            B, h, n_b, ring, "under load" if load else "alone", wrong, wrongz))
        bad += wrong + wrongz
    lib.rk_tune(13, 2)
  print("RACE SCREEN", "FAILED" if bad else "clean")
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
