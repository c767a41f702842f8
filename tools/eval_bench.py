#!/usr/bin/env python
"""Recoder.recommend / evaluate throughput at C5's catalogue (1 M items, AE[512]) on one GPU:
the fused form (top-k filter in the decode epilogue, include/recoder_hip.h) against the strip-by-strip
path (RK_EVAL_FUSED=0: decode a 65 536-item strip -> scores -> radix-select top-k -> merge).

    python tools/eval_bench.py [n_items] [h] [k]

Synthetic: 20 000 users x n_items uniform, 100 interactions per user; random-init weights (scores of
an untrained model: a harder case for the sample bound than a trained one, whose scores are peaked)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recoder_amd import synthetic  # noqa: E402
from recoder_amd.data import RecommendationDataset, UsersInteractions  # noqa: E402
from recoder_amd.metrics import Recall  # noqa: E402
from recoder_amd.model import Recoder  # noqa: E402
from recoder_amd.nn import DynamicAutoencoder  # noqa: E402


def main():
  n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
  h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
  k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
  n_users = 20000
  csr = synthetic.uniform(n_users, n_items, 100, seed=5)
  torch.manual_seed(1)
  model = DynamicAutoencoder([h], activation_type="tanh", sparse=True)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
  # one short epoch on 2 000 users: initialises the engine and moves the weights off their init
  rec.train(RecommendationDataset(csr[:2000]), batch_size=500, lr=1e-3, num_epochs=1, negative_sampling=True)
  print("catalogue %d items, h = %d, k = %d" % (n_items, h, k))
  for B in (500, 2000):
    users = np.arange(2000, 2000 + B)
    ui = UsersInteractions(users=users, interactions_matrix=csr[users])
    res = {}
    for mode in ("fused", "strips"):
      os.environ["RK_EVAL_FUSED"] = "1" if mode == "fused" else "0"
      rec.eval_fused_batches = 0
      out = rec.recommend_array(ui, k)          # warm-up (images, workspaces)
      torch.cuda.synchronize()
      reps = 5
      t0 = time.perf_counter()
      for _ in range(reps):
        out = rec.recommend_array(ui, k)
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / reps
      res[mode] = out
      flops = 2.0 * B * n_items * h
      print("  B = %4d  %-6s  %8.2f ms / batch  %9.0f users/s  %6.1f algorithmic TFLOP/s%s" %
            (B, mode, dt * 1e3, B / dt, flops / dt * 1e-12,
             "  (fused batches: %d of %d)" % (rec.eval_fused_batches, reps + 1) if mode == "fused" else ""))
    print("  B = %4d  identical recommendations: %s" % (B, bool(np.array_equal(res["fused"], res["strips"]))))
  # Recoder.evaluate end to end (host side included: collation of the users' rows, metrics)
  held = csr[2000:6000]
  for mode in ("fused", "strips"):
    os.environ["RK_EVAL_FUSED"] = "1" if mode == "fused" else "0"
    t0 = time.perf_counter()
    r = rec.evaluate(RecommendationDataset(held, held), num_recommendations=k, metrics=[Recall(k)], batch_size=500)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("  Recoder.evaluate 4000 users, batch 500, %-6s: %6.2f s  %7.0f users/s  (%s)" %
          (mode, dt, 4000 / dt, {str(m): float(np.mean(v)) for m, v in r.items()}))


if __name__ == "__main__":
  main()
