#!/usr/bin/env python
"""Per-term model of the users-DP step at N = 1 / 2 / 4 / 8 for C2 (dense Adam), C4 and C5 (SparseAdam):
union item-set sizes MEASURED on the bench's synthetic matrices, kernel terms scaled from this round's
single-GPU times, exchange priced at a stated link bandwidth -- a PREDICTION (nothing here has run on more
than one GPU).  DESIGN.md section 6 quotes this output.
    python tools/dp_model.py [--bw 300] [--steps 6]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def union_sizes(csr, B, world, steps, seed=0):
  n = csr.shape[0]
  per = n // world
  out = []
  for k in range(steps):
    rows = []
    for r in range(world):
      o = np.random.RandomState(seed + 1000 * r).permutation(per)[k * B:(k + 1) * B] + r * per
      rows.append(o)
    sub = csr[np.concatenate(rows)]
    out.append(len(np.unique(sub.indices)))
  return float(np.mean(out))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--bw", type=float, default=300.0, help="GB/s a rank moves over its 7 xGMI links in a direct RS / AG")
  ap.add_argument("--ring", type=float, default=150.0, help="GB/s of a one-link ring (the round-3 pricing)")
  ap.add_argument("--lat", type=float, default=15.0, help="us per collective")
  ap.add_argument("--steps", type=int, default=4)
  a = ap.parse_args()
  import bench
  HBM = 5.3e6          # bytes / us the Adam sweeps reach (profiles: 5.2-5.5 TB/s)
  rows = []
  for name, h, kind in (("c2", 200, "dense"), ("c4", 128, "sparse"), ("c5u", 512, "sparse")):
    cfg = bench.CONFIGS[name]
    csr = bench.make_csr(cfg)
    n_items = csr.shape[1]
    B = cfg["batch_size"]
    tables = 1 if cfg["kind"] == "mf" else 2
    nb1 = union_sizes(csr, B, 1, a.steps)
    # single-GPU kernel terms (us) at n_b = nb1: {fixed, scales with n_b} from this round's profiles
    if name == "c2":
      fixed, scaled, adam1 = 14.0, 28.0 + 6.5 + 23.5, 38.0          # enc fwd | decode+dZ, reduce, dW||enc bwd | dense sweep
    elif name == "c4":
      fixed, scaled, adam1 = 7.0 + 8.0, 33.0 + 21.0, 18.0            # gather, split | decode+dZ, dW||reduce | SparseAdam
    else:
      fixed, scaled, adam1 = 90.0, 117.0 + 100.0 + 100.0 + 62.0, 280.0   # enc fwd | decode, dZ, dW, enc bwd | SparseAdam
    for N in (1, 2, 4, 8):
      nb = union_sizes(csr, B, N, a.steps) if N > 1 else nb1
      bytes_g = tables * nb * h * 4 + nb * 4
      kern = fixed + scaled * nb / nb1
      if kind == "dense":
        adam_rep = adam1                                   # the sweep covers the whole table either way
        adam_own = adam1
      else:
        adam_rep = adam1 * nb / nb1
        adam_own = adam_rep / N + (tables * nb * h * 4 * 2 * (N - 1) / N) / HBM * 0.5   # + scatter of the others' rows
      if N == 1:
        ex_ring = ex_direct = 0.0
      else:
        ex_ring = 2 * (N - 1) / N * bytes_g / (a.ring * 1e3) + 2 * a.lat
        ex_direct = 2 * (N - 1) / N * bytes_g / (a.bw * 1e3) + 2 * a.lat
      sync = 0.0 if N == 1 else 25.0
      t_rep_ring = kern + adam_rep + ex_ring + sync
      t_rep_dir = kern + adam_rep + ex_direct + sync
      t_own_dir = kern + adam_own + ex_direct + sync
      # NOT the reference's shared-set semantics (each user would only see its own rank's negatives):
      # every rank decodes its OWN item set, only the gradient rows are unioned -- for comparison
      t_ownset = (fixed + scaled) + min(adam_rep, adam_own) + ex_direct + sync
      rows.append((name, N, nb, bytes_g / 1e6, kern, adam_rep, adam_own, ex_ring, ex_direct, t_rep_ring, t_rep_dir,
                   t_own_dir, t_ownset))
  print("config N   union n_b  exch MB | kernels  Adam(repl)  Adam(owned) | exch ring@%g  exch direct@%g | step: repl+ring  repl+direct  owned+direct | users/s (best)  x vs N=1 | per-rank item sets (other semantics): step  x" % (a.ring, a.bw))
  base = {}
  for r in rows:
    name, N = r[0], r[1]
    best = min(r[9], r[10], r[11]) if N > 1 else r[9]
    B = bench.CONFIGS[name]["batch_size"]
    ups = N * B / best * 1e6
    if N == 1:
      base[name] = ups
    print("%-5s %2d  %9.0f  %7.1f | %7.0f  %9.0f  %10.0f | %12.0f  %13.0f | %14.0f  %11.0f  %12.0f | %10.2f M  %5.2f | %8.0f  %5.2f" % (
        name, N, r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], ups / 1e6, ups / base[name],
        r[12], (N * B / r[12] * 1e6) / base[name]))


if __name__ == "__main__":
  main()
