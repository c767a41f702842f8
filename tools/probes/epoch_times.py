"""Wall-clock time of every epoch of a long Recoder.train run on the C2 workload (graph path):
does a step cost the same in epoch 12 as in epoch 1?   [SPARSE=1] python tools/probes/epoch_times.py [epochs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from recoder_amd import synthetic, model as M
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
sparse = os.environ.get("SPARSE") == "1"
n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 12
csr = synthetic.ml20m_like(seed=0)
torch.manual_seed(0)
rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5, sparse=sparse),
              use_cuda=True, optimizer_type="adam", loss="mse")
stamps = []
import gc
if os.environ.get('GCOFF') == '1':
  gc.disable()
if os.environ.get('GCFREEZE') == '1':
  gc.freeze()
gc.callbacks.append(lambda phase, info: phase == 'stop' and info['generation'] == 2 and print('gen-2 collection', info, 'at epoch', len(stamps)))
orig = M.Recoder._epoch_end
def ep_end(self, *a, **k):
  r = orig(self, *a, **k)
  torch.cuda.synchronize()
  stamps.append(time.perf_counter())
  return r
M.Recoder._epoch_end = ep_end
rec.train(RecommendationDataset(csr), batch_size=500, lr=1e-3, weight_decay=0.0 if sparse else 2e-5,
          negative_sampling=True, num_epochs=n_ep)
d = np.diff(stamps) * 1e3
steps = -(-csr.shape[0] // 500)
print("epoch ms:", " ".join("%.1f" % x for x in d))
print("ms/step :", " ".join("%.3f" % (x / steps) for x in d))
m, v = [rec.optimizer.state[p]["exp_avg"] for p in rec.optimizer.state][0], None
for p, st in rec.optimizer.state.items():
  if st["exp_avg"].numel() > 1e6:
    a = st["exp_avg"].abs()
    tiny = float(((a > 0) & (a < 1.1754944e-38)).float().mean())
    a2 = st["exp_avg_sq"]
    tiny2 = float(((a2 > 0) & (a2 < 1.1754944e-38)).float().mean())
    print("state of a %s tensor: subnormal fraction exp_avg %.3f exp_avg_sq %.3f" % (tuple(p.shape), tiny, tiny2))
