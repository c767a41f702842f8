#!/usr/bin/env python
"""Micro-benchmark of single C-ABI entry points on a collated block (GPU only).

    python tools/microbench.py [B ...]

Times rk_decode_loss (MSE epilogue vs plain store), rk_decode_bwd_dz,
rk_decode_bwd_dw, rk_ae_encode_fwd/bwd and a dense-Adam table job of rk_adam_multi with HIP events, for the
C2 shape (ML-20M-like items, h = 200) at several batch sizes.
"""
import os
import sys

import numpy as np
import ctypes

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recoder_amd import _lib, synthetic  # noqa: E402
from recoder_amd._lib import LOSS_MSE, LOSS_NONE, check, ptr  # noqa: E402
from recoder_amd.device import Block, DeviceCSR, current_stream  # noqa: E402


def timeit(fn, n=30, warm=5):
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
  Bs = [int(x) for x in sys.argv[1:]] or [64, 128, 256, 500, 1000, 2000]
  lib = _lib.load()
  dev = torch.device("cuda")
  h = int(os.environ.get("H", "200"))
  csr = synthetic.ml20m_like(seed=0, n_users=20000)
  shard = int(os.environ.get("SHARD", "1"))       # item-parallel shard of rank 0 (items % SHARD == 0)
  if shard > 1:
    from recoder_amd.parallel import ItemParallel
    csr = ItemParallel(rank=0, world=shard, allreduce_fn=lambda t: t).shard_csr(csr)
  dcsr = DeviceCSR(csr)
  n_items = csr.shape[1]
  f = dict(dtype=torch.float32, device=dev)
  W = torch.randn(n_items, h, **f) * 0.05
  bias = torch.zeros(n_items, **f)
  m = torch.zeros_like(W)
  v = torch.zeros_like(W)
  st = current_stream()
  for B in Bs:
    users = torch.arange(B, dtype=torch.int64, device=dev)
    blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), n_items, dev,
                n_cap=(int(os.environ["NCAP"]) if "NCAP" in os.environ else
                       -(-n_items // shard) if shard > 1 else None))
    blk.collate(dcsr, users)
    n_b, nnz, ld, S = blk.counts_host()
    Z = torch.randn(B, h, **f)
    Z0 = torch.empty(B, h, **f)
    dZ = torch.empty(B, h, **f)
    dO = torch.empty(B * blk.ld_cap, **f)
    out = torch.empty(B * blk.ld_cap, **f)
    G = torch.empty(blk.n_cap * h, **f)
    gbp = torch.empty((B // 64 + 1) * blk.ld_cap, **f)
    ws = torch.empty(lib.rk_dz_workspace_bytes(B, h) // 4, **f)
    part = torch.zeros(lib.rk_loss_partials(B, blk.n_cap), **f)
    gbp = torch.empty((B // 32 + 1) * blk.ld_cap, **f)
    r = {}
    r["collate"] = timeit(lambda: blk.collate(dcsr, users))
    r["dec_mse"] = timeit(lambda: check(lib.rk_decode_loss(
        ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), LOSS_MSE, 0.0, 1.0 / B, ptr(dO), 0, ptr(part),
        ptr(gbp), None, st)))
    r["dec_store"] = timeit(lambda: check(lib.rk_decode_loss(
        ptr(Z), B, h, blk.ref, 0, ptr(W), ptr(bias), LOSS_NONE, 0.0, 1.0, ptr(out), blk.ld_cap, None,
        None, None, st)))
    r["dz"] = timeit(lambda: check(lib.rk_decode_bwd_dz(
        ptr(dO), B, h, blk.ref, ptr(W), None, 0, ptr(dZ), ptr(ws), None, st)))
    r["dw"] = timeit(lambda: check(lib.rk_decode_bwd_dw(
        ptr(dO), ptr(Z), B, h, blk.ref, ptr(G), None, st)))
    ws3 = torch.zeros(lib.rk_dw3_workspace_bytes(B, h, blk.n_cap) // 4 + 64, **f)
    dO.normal_()
    r["dw3_slabs"] = timeit(lambda: check(lib.rk_decode_bwd_dw3(
        ptr(dO), ptr(Z), B, h, blk.ref, None, None, ptr(ws3), None, st)))
    r["dw3_G"] = timeit(lambda: check(lib.rk_decode_bwd_dw3(
        ptr(dO), ptr(Z), B, h, blk.ref, ptr(G), None, ptr(ws3), None, st)))
    r["enc_fwd"] = timeit(lambda: check(lib.rk_ae_encode_fwd(
        blk.ref, 0, B, ptr(W), ptr(bias), h, None, 0.5, 1, 1, ptr(users), 1, ptr(Z0), st)))
    r["enc_bwd"] = timeit(lambda: check(lib.rk_ae_encode_bwd(blk.ref, 0, B, ptr(dZ), h, ptr(G), 0, None, st)))
    from recoder_amd._lib import RkAdamJob
    job = RkAdamJob()
    job.par.p, job.par.m, job.par.v = ptr(W), ptr(m), ptr(v)
    job.par.lr, job.par.beta1, job.par.beta2, job.par.eps, job.par.weight_decay, job.par.step = 1e-3, 0.9, 0.999, 1e-8, 2e-5, 1
    job.n_rows, job.h, job.pos, job.g, job.g_parts = n_items, h, ptr(blk.pos), ptr(G), 1
    r["adam_tab"] = timeit(lambda: check(lib.rk_adam_multi(ctypes.byref(job), 1, None, 0, 1.0, None, st)))
    gf = 2.0 * B * h * n_b / 1e9
    print("B=%5d n_b=%6d nnz=%7d | " % (B, n_b, nnz) +
          " ".join("%s %.1fus" % (k, t) for k, t in r.items()) +
          " | GEMM %.2f GFLOP -> dec %.1f dz %.1f dw %.1f TF/s" %
          (gf, gf / r["dec_mse"] * 1e3, gf / r["dz"] * 1e3, gf / r["dw"] * 1e3), flush=True)


if __name__ == "__main__":
  main()
