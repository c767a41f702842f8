"""Wall-clock phases of one epoch of Recoder.train on the graph path (C2): monkey-patched timers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from recoder_amd import synthetic, model as M, graph as Gm, data as D
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
T = {}
def timed(obj, name, key=None):
  f = getattr(obj, name)
  key = key or name
  def w(*a, **k):
    t0 = time.perf_counter()
    try:
      return f(*a, **k)
    finally:
      T[key] = T.get(key, 0.0) + time.perf_counter() - t0
  setattr(obj, name, w)
timed(M, "epoch_user_order")
timed(Gm.GraphStepper, "begin_epoch")
timed(Gm.GraphStepper, "run")
timed(M.Recoder, "_epoch_end")
timed(M.Recoder, "_run_epoch_graph")
timed(M.Recoder, "_sync_ranges")
csr = synthetic.ml20m_like(seed=0)
torch.manual_seed(0)
rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5, sparse=os.environ.get("SPARSE") == "1"), use_cuda=True,
              optimizer_type="adam", loss="mse")
ds = RecommendationDataset(csr)
kw = dict(batch_size=500, lr=1e-3, weight_decay=0.0 if os.environ.get("SPARSE") == "1" else 2e-5, negative_sampling=True)
rec.train(ds, num_epochs=2, **kw)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
rec.train(ds, num_epochs=8, **kw)          # epochs 2..8 = 7 epochs
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = 7
print("per epoch: total %.2f ms" % (dt / n * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
  print("   %-22s %.2f ms" % (k, v / n * 1e3))
