"""Where the per-epoch overhead of Recoder.train goes (C2, whole epochs on the graph path)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from recoder_amd import synthetic
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
csr = synthetic.ml20m_like(seed=0)
torch.manual_seed(0)
rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5), use_cuda=True,
              optimizer_type="adam", loss="mse")
ds = RecommendationDataset(csr)
kw = dict(batch_size=500, lr=1e-3, weight_decay=2e-5, negative_sampling=True)
rec.train(ds, num_epochs=1, **kw)
torch.cuda.synchronize()
n_ep = 6
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
rec.train(ds, num_epochs=n_ep, **kw)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
steps = sum(len(x) for x in rec.loss_history[1:])
print("%d epochs, %d steps: %.3f ms/step, %.2f ms per epoch" % (n_ep, steps, dt / steps * 1e3, dt / n_ep * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
