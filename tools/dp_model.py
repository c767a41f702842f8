#!/usr/bin/env python
"""Per-term model of the users-DP step at N = 1 / 2 / 4 / 8 for C2 and C3 (dense Adam: replicated vs sharded, ZeRO-1),
C4 and C5 (SparseAdam: replicated + graph replay vs owned rows, host-sequenced, with the measured host term):
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


def swept_frac(csr, B, world, steps, period=16, seed=0):
  """Fraction of a table's rows a LAZY dense-Adam sweep brings up to date (round 6): the union item sets of two
  consecutive global batches + the round-robin chunk."""
  n, n_items = csr.shape
  per = n // world
  sets = []
  for k in range(steps + 1):
    rows = [np.random.RandomState(seed + 1000 * r).permutation(per)[k * B:(k + 1) * B] + r * per for r in range(world)]
    sets.append(np.unique(csr[np.concatenate(rows)].indices))
  out = []
  for k in range(steps):
    m = np.zeros(n_items, dtype=bool)
    m[sets[k]] = True
    m[sets[k + 1]] = True
    c = k % period
    m[c * n_items // period:(c + 1) * n_items // period] = True
    out.append(m.mean())
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
  # host terms, MEASURED at one forced RCCL rank (profiles/r0[56]_bench_*_dp1_*.json): a replayed step costs the
  # host nothing the GPU waits for; the owned-row SparseAdam step is sequenced from the host (a device -> host
  # read of the row offsets, ~25 launches enqueued one by one): C4 0.266 vs 0.129 ms, C5-shaped 1.43 vs 0.79
  HOST_OWNED = {"c4": 137.0, "c5u": 150.0}
  rows = []
  for name, h, kind in (("c2", 200, "dense"), ("c3", 200, "dense"), ("c4", 128, "sparse"), ("c5u", 512, "sparse")):
    cfg = bench.CONFIGS[name]
    csr = bench.make_csr(cfg)
    n_items = csr.shape[1]
    B = cfg["batch_size"]
    tables = 1 if cfg["kind"] == "mf" else 2
    nb1 = union_sizes(csr, B, 1, a.steps)
    # single-GPU kernel terms (us) at n_b = nb1: {fixed, scales with n_b} from the round-4/5 profiles
    # round 6: adam1 = the dense sweep with the explicit-fma update (every row), lazy1 = (fraction of rows swept, us) of
    # the LAZY sweep at one rank -- between the two the model interpolates by the fraction of rows the ranks' union item
    # sets of two consecutive steps + the chunk cover (it reaches the dense sweep where the union is the catalogue)
    lazy1 = None
    if name == "c2":
      fixed, scaled, adam1 = 12.5, 23.6 + 9.2 + 17.9 + 14.3, 35.0   # enc fwd | fdec, reduce, dW (dense) + encoder backward of the PHASED step (r06 one-rank line) | dense sweep
      lazy1 = 29.0
    elif name == "c3":
      fixed, scaled, adam1 = 10.5 + 2 * 6.6 + 2 * 8.9, 18.7 + 15.2 + 16.8 + 6.3 + 28.9 + 5.0, 68.0   # enc fwd, Linears | decode, mnll, dZ, reduce, dW||enc bwd, split | dense sweep
      lazy1 = 48.0
    elif name == "c4":
      fixed, scaled, adam1 = 6.5 + 7.6 + 8.4 + 6.5, 28.0 + 20.0, 17.0   # gather, split, loss reduce, user rows | fused decode, dW || colsum || reduce | SparseAdam
    else:
      fixed, scaled, adam1 = 88.0, 118.0 + 115.0 + 100.0 + 62.0, 276.0   # enc fwd | decode, dZ, dW, enc bwd | SparseAdam
    f1 = swept_frac(csr, B, 1, a.steps) if lazy1 is not None else None
    for N in (1, 2, 4, 8):
      nb = union_sizes(csr, B, N, a.steps) if N > 1 else nb1
      bytes_g = tables * nb * h * 4 + nb * 4
      bytes_dense = tables * n_items * h * 4            # the dense layout of the sharded dense Adam (= capacity rows)
      kern = fixed + scaled * nb / nb1
      adam_zero = host_own = float("nan")
      if kind == "dense":
        # the replicated sweep is LAZY (round 6): its rows grow with the ranks' union item sets
        fN = swept_frac(csr, B, N, a.steps) if N > 1 else f1
        adam_rep = lazy1 + (adam1 - lazy1) * max(0.0, fN - f1) / max(1e-9, 1.0 - f1)
        adam_own = adam_rep
        # ZeRO-1: 1/N of the sweep (p, m, v and the DENSE gradient shard) + laying the compact rows out by item id
        # (two staging launches: reads n_b rows, writes n_items rows; the decoder half runs beside the chain)
        adam_zero = (tables * n_items * h * 28 / N) / HBM + (tables * (n_items + nb) * h * 4) / HBM * 0.5
      else:
        adam_rep = adam1 * nb / nb1
        adam_own = adam_rep / N + (tables * nb * h * 4 * 2 * (N - 1) / N) / HBM * 0.5   # + scatter of the others' rows
        host_own = HOST_OWNED.get(name, 150.0)
      if N == 1:
        ex_ring = ex_direct = ex_dense = 0.0
      else:
        ex_ring = 2 * (N - 1) / N * bytes_g / (a.ring * 1e3) + 2 * a.lat
        ex_direct = 2 * (N - 1) / N * bytes_g / (a.bw * 1e3) + 2 * a.lat
        ex_dense = 2 * (N - 1) / N * bytes_dense / (a.bw * 1e3) + 3 * a.lat
      sync = 0.0 if N == 1 else 25.0
      t_rep_ring = kern + adam_rep + ex_ring + sync
      t_rep_dir = kern + adam_rep + ex_direct + sync
      t_alt = (kern + adam_zero + ex_dense + sync) if kind == "dense" else (kern + adam_own + ex_direct + sync + host_own)
      if N == 1:
        t_alt = t_rep_dir
      if kind == "dense":
        # BUILT (round 5, RK_DP_ITEMSETS=local + RK_DP_ZERO=1): kernels at the rank's own item set, the gradients laid
        # out by item id (staging), reduce-scatter + all-gather of the dense layout, 1/N of the sweep
        adam_loc = (tables * n_items * h * 28 / N) / HBM + (tables * (n_items + nb1) * h * 4) / HBM * 0.5
        t_ownset = (fixed + scaled) + (adam_loc if N > 1 else adam_rep) + ex_dense + sync
      else:
        t_ownset = (fixed + scaled) + min(adam_rep, adam_own) + ex_direct + sync      # (SparseAdam tables: not built)
      rows.append((name, N, nb, bytes_g / 1e6, bytes_dense / 1e6 if kind == "dense" else float("nan"), kern, adam_rep,
                   adam_zero if kind == "dense" else adam_own, host_own, ex_ring, ex_direct, ex_dense if kind == "dense" else float("nan"),
                   t_rep_ring, t_rep_dir, t_alt, t_ownset, kind))
  print("A PREDICTION: nothing here has run on more than one GPU.  us per step; exchange at %g GB/s per rank (ring: %g)." % (a.bw, a.ring))
  print("dense-Adam configs (c2, c3): alt = sharded dense Adam (ZeRO-1, graph replay); SparseAdam configs (c4, c5u): alt = owned-row "
        "Adam, host-sequenced (host = measured at one forced rank)")
  print("config N   union n_b  exch MB (compact | dense) | kernels  Adam repl  Adam alt  host alt | exch ring  direct  dense | "
        "step: repl+ring  repl+direct (graph)  alt | best users/s  x vs N=1 | per-rank item sets (RK_DP_ITEMSETS=local, the reference under DDP -- another estimator; dense-Adam configs: built): step  x")
  base = {}
  for r in rows:
    name, N = r[0], r[1]
    best = min(r[12], r[13], r[14]) if N > 1 else r[13]
    B = bench.CONFIGS[name]["batch_size"]
    ups = N * B / best * 1e6
    if N == 1:
      base[name] = ups
    f = lambda x, w: ("%" + str(w) + ".0f") % x if x == x else " " * (w - 1) + "-"
    print("%-5s %2d  %9.0f  %7.1f | %s | %7.0f  %9.0f  %s  %s | %9.0f  %6.0f  %s | %14.0f  %19.0f  %s | %9.2f M  %5.2f | %8.0f  %5.2f" % (
        name, N, r[2], r[3], f(r[4], 7), r[5], r[6], f(r[7], 8), f(r[8], 8), r[9], r[10], f(r[11], 5), r[12], r[13], f(r[14], 5),
        ups / 1e6, ups / base[name], r[15], (N * B / r[15] * 1e6) / base[name]))


if __name__ == "__main__":
  main()
