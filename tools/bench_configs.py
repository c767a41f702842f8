#!/usr/bin/env python
"""Throughput of the other BASELINE configurations through the public API (Recoder.train) on one
GPU: users/s over one whole epoch after a warm one (synthetic data of SURVEY 8d's shapes; the
headline C2 number comes from bench.py).   python tools/bench_configs.py [c2 c2s c3 c4 c5u]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recoder_amd import synthetic  # noqa: E402
from recoder_amd.data import RecommendationDataset  # noqa: E402
from recoder_amd.model import Recoder  # noqa: E402
from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization  # noqa: E402

CONFIGS = {
  "c2": dict(data=lambda: synthetic.ml20m_like(seed=0),
             model=lambda: DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5, sparse=False),
             loss="mse", wd=2e-5, note="ML-20M-like AE[200] MSE dense Adam"),
  "c2s": dict(data=lambda: synthetic.ml20m_like(seed=0),
              model=lambda: DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5, sparse=True),
              loss="mse", wd=0.0, note="ML-20M-like AE[200] MSE SparseAdam"),
  "c3": dict(data=lambda: synthetic.lognormal_zipf(200000, 41140, 59, seed=1),
             model=lambda: DynamicAutoencoder([200, 200], activation_type="tanh", noise_prob=0.5, sparse=False),
             loss="logloss", wd=2e-5, note="MSD-like (200k of 471k users) AE[200,200] multinomial NLL"),
  "c4": dict(data=lambda: synthetic.lognormal_zipf(300000, 250000, 50, seed=2),
             model=lambda: MatrixFactorization(128, activation_type="none", sparse=True),
             loss="mse", wd=0.0, note="MSD-big stand-in 300k x 250k MF d=128 SparseAdam"),
  "c5u": dict(data=lambda: synthetic.uniform(100000, 1000000, 100, seed=3),
              model=lambda: DynamicAutoencoder([512], activation_type="tanh", noise_prob=0.0, sparse=True),
              loss="mse", wd=0.0, note="C5-shaped 1M items (100k-user shard) AE[512] SparseAdam"),
  "c5u4k": dict(data=lambda: synthetic.uniform(100000, 1000000, 100, seed=3), B=4096,
                model=lambda: DynamicAutoencoder([512], activation_type="tanh", noise_prob=0.0, sparse=True),
                loss="mse", wd=0.0, note="the same at B = 4096 (SURVEY 8d's second C5 batch size)"),
}


def run(name, B=500):
  """One warm epoch, then RK_EPOCHS (default 5) timed ones -- whole epochs, i.e. what Recoder.train
  does by default (single-layer autoencoders replay HIP graphs then; the per-epoch setup -- user
  order, Adam constants, loss read-back -- is inside the timing).  Two timed epochs right behind
  the start-up were what round 2 first reported: 70 ms that the container's CPU throttling (set off
  by the start-up's OpenMP teams) hit or missed at random."""
  c = CONFIGS[name]
  B = c.get("B", B)
  csr = c["data"]()
  torch.manual_seed(0)
  rec = Recoder(model=c["model"](), use_cuda=True, optimizer_type="adam", loss=c["loss"])
  ds = RecommendationDataset(csr)
  kw = dict(batch_size=B, lr=1e-3, weight_decay=c["wd"], negative_sampling=True)
  rec.train(ds, num_epochs=1, **kw)          # warm-up (allocations, first touches, graph capture)
  torch.cuda.synchronize()
  k0 = len(rec.loss_history)
  t0 = time.perf_counter()
  # (resuming repeats the last epoch, as the reference does: num_epochs = E trains E epochs here)
  rec.train(ds, num_epochs=int(os.environ.get("RK_EPOCHS", "5")), **kw)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  n = sum(len(x) for x in rec.loss_history[k0:])
  print("%-4s %-62s %8.0f users/s  %.3f ms/step  (%d steps in %d epochs, loss %.4g -> %.4g)"
        % (name, c["note"], n * B / dt, dt / n * 1e3, n, len(rec.loss_history) - k0,
           rec.loss_history[k0][0], rec.last_epoch_losses[-1]), flush=True)


if __name__ == "__main__":
  for name in (sys.argv[1:] or list(CONFIGS)):
    run(name)
