"""Evaluation throughput (Recoder.evaluate: strip-wise decode + masked top-k + metrics on the host)
on the C2 shape: users per second for Recall@20 / NDCG@100."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from recoder_amd import synthetic
from recoder_amd.data import RecommendationDataset
from recoder_amd.metrics import NDCG, Recall
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder

csr = synthetic.ml20m_like(seed=0)
n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.RandomState(1)
ev = csr[:n_eval].tocsr()
# 80 / 20 split of every user's items
coo = ev.tocoo()
keep = rng.rand(coo.nnz) < 0.8
import scipy.sparse as sp
x = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=ev.shape)
y = sp.csr_matrix((coo.data[~keep], (coo.row[~keep], coo.col[~keep])), shape=ev.shape)
torch.manual_seed(0)
rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5), use_cuda=True,
              optimizer_type="adam", loss="mse", num_items=csr.shape[1], num_users=csr.shape[0])
rec.train(RecommendationDataset(csr[n_eval:]), batch_size=500, lr=1e-3, weight_decay=2e-5, num_epochs=1,
          negative_sampling=True)
ds = RecommendationDataset(x, y)
for bs in (500, 2000):
  for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = rec.evaluate(ds, num_recommendations=100, metrics=[Recall(20), NDCG(100)], batch_size=bs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print("batch %4d: %.0f users/s (%.1f ms per batch)  %s" % (bs, n_eval / dt, dt / (n_eval / bs) * 1e3,
        {str(k): round(float(np.mean(v)), 4) for k, v in res.items()}))
if os.environ.get("PROFILE"):
  import cProfile, pstats
  pr = cProfile.Profile(); pr.enable()
  rec.evaluate(ds, num_recommendations=100, metrics=[Recall(20), NDCG(100)], batch_size=500)
  torch.cuda.synchronize(); pr.disable()
  pstats.Stats(pr).sort_stats("tottime").print_stats(18)
