#!/usr/bin/env python
"""The register-resident fused decode (csrc/fdecode.hip) at the C2 shape against the LDS-staged fused
decode of decode16.hip: HIP-event medians with the caches flushed before every launch (the event pair
itself costs ~6 us: an empty launch of the same grid).
    python tools/probes/fdec_probe.py [B h n_items]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recoder_amd import _lib                                              # noqa: E402
from recoder_amd._lib import LOSS_MSE, RkPlanes, check, ptr               # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream          # noqa: E402
from recoder_amd import synthetic                                         # noqa: E402


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
  h = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  lib = _lib.load()
  dev = torch.device("cuda")
  csr = synthetic.ml20m_like(seed=0)
  n_items = csr.shape[1]
  dcsr = DeviceCSR(csr)
  f = dict(dtype=torch.float32, device=dev)
  g = torch.Generator(device=dev); g.manual_seed(1)
  W = torch.randn(n_items, h, generator=g, **f) * 0.07
  bias = torch.randn(n_items, generator=g, **f) * 0.02
  # argv[3]: users collated into the block (default B); more than B = the union item set of a data-parallel
  # step (B rows of this rank against the items of all ranks' users)
  S = int(sys.argv[3]) if len(sys.argv) > 3 else B
  users = torch.from_numpy(np.random.RandomState(0).permutation(csr.shape[0])[:S]).to(dev)
  blk = Block(S, int(np.sort(dcsr.degrees)[-S:].sum()), n_items, dev)
  blk.collate(dcsr, users)
  Z = torch.tanh(torch.randn(B, h, generator=g, **f))
  ranges = torch.zeros(128, dtype=torch.int32, device=dev)
  ranges[64:65].copy_(W.abs().max().reshape(1).view(torch.int32))
  buf = torch.zeros(lib.rk_planes_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  pl = RkPlanes()
  check(lib.rk_planes_layout(ptr(buf), B, h, blk.n_cap, ctypes.byref(pl)))
  st = current_stream()
  check(lib.rk_split_wz(ptr(W), ptr(Z), B, h, blk.ref, ptr(ranges), ctypes.byref(pl), None, st))
  n_b = blk.counts_host()[0]
  rows_img = -(-B // 32) * 32
  img = torch.zeros((rows_img + 256) * blk.ld_cap * 2, dtype=torch.int16, device=dev)
  sc = torch.ones(lib.rk_pg_scale_floats(B, blk.n_cap), **f)
  part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
  ws = torch.zeros(lib.rk_fdec_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  dO = torch.zeros(B * blk.ld_cap, **f)
  gbp = torch.zeros(-(-B // 64) * blk.ld_cap, **f)
  flush = torch.zeros(64 << 20, **f)

  def timeit(fn, n=30, flush_first=True):
    ts = []
    for _ in range(n):
      if flush_first:
        flush.add_(1.0)                                 # (evict: the step's other launches do that)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); fn(); e1.record(); e1.synchronize()
      ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))

  def fdec():
    check(lib.rk_fdec_loss_dz(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(img), rows_img,
                              ptr(sc), ptr(part), ptr(ws), st))

  def old():
    check(lib.rk_decode_loss_dz_planes(ctypes.byref(pl), B, blk.ref, 0, ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(dO),
                                       ptr(part), ptr(gbp), ptr(ws), st))
  print("C2-shaped block: B = %d, h = %d, n_b = %d" % (B, h, n_b))
  print("decode16 fused (LDS dO tile, W^T stage): %.1f us" % timeit(old))
  print("fdec (register resident, W rows resident in LDS): %.1f us" % timeit(fdec))
  # the launches behind it in the C2 step, each alone (same flush in front of every launch)
  dZ = torch.zeros(B * h, **f)
  slabs = torch.zeros(lib.rk_pg_dw_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
  G_en = torch.zeros(blk.n_cap * h, **f)
  gb_en = torch.zeros(h * 8, **f)
  gb_de = torch.zeros(blk.n_cap, **f)
  fdec()
  print("rk_fdec_dz_reduce alone: %.1f us" % timeit(lambda: check(lib.rk_fdec_dz_reduce(
      ptr(ws), B, h, blk.ref, ptr(Z), 1, ptr(dZ), st))))
  print("rk_pg_dw alone (dW tiles from the image): %.1f us, hot %.1f us" % (timeit(lambda: check(lib.rk_pg_dw(
      ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st))), timeit(lambda: check(lib.rk_pg_dw(
      ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st)), flush_first=False)))
  # RK_TUNE_DW_RING (13): LDS stages of the dW tiles' ring loop (2 / 3 / 4 / 6)
  for v in (2, 4):
    lib.rk_tune(13, v)
    print("rk_pg_dw alone, RK_TUNE_DW_RING = %d: %.1f us, hot %.1f us" % (v, timeit(lambda: check(lib.rk_pg_dw(
        ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st))), timeit(lambda: check(lib.rk_pg_dw(
        ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), None, st)), flush_first=False)))
  lib.rk_tune(13, 0)
  print("rk_ae_encode_bwd alone: %.1f us" % timeit(lambda: check(lib.rk_ae_encode_bwd(
      blk.ref, 0, B, ptr(dZ), h, ptr(G_en), 0, ptr(gb_en), st))))
  print("rk_pg_dw_encode_bwd (dW || encoder backward): %.1f us" % timeit(lambda: check(lib.rk_pg_dw_encode_bwd(
      ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), 0, ptr(dZ), ptr(G_en), ptr(gb_en), None, st))))
  print("rk_pg_dw_encode_bwd (dW || encoder backward || image column sums): %.1f us" % timeit(lambda: check(
      lib.rk_pg_dw_encode_bwd(ptr(img), ptr(sc), 32, 64, B, ctypes.byref(pl), blk.ref, ptr(slabs), 0, ptr(dZ),
                              ptr(G_en), ptr(gb_en), ptr(gb_de), st))))


if __name__ == "__main__":
  main()
