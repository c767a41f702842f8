"""Run-to-run determinism of hook-free training, eager and graph, with and without the dW branch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_hip_parity import synth_csr
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
csr = synth_csr(1024, 600, 12, seed=43)
def run():
  torch.manual_seed(37)
  model = DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.0, sparse=False)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
  rec.train(RecommendationDataset(csr), batch_size=128, lr=1e-3, weight_decay=1e-5, num_epochs=4,
            negative_sampling=True)
  return np.concatenate(rec.loss_history)
res = {}
for graph in ("0", "1"):
  for br in ("1",):                    # (the dW side branch is always on now)
    os.environ["RK_GRAPH"] = graph
    a, b = run(), run()
    res[(graph, br)] = a
    print("graph=%s branch=%s: run-to-run equal %s (max diff %.3g)" % (graph, br, np.array_equal(a, b), np.abs(a - b).max()))
base = res[("0", "1")]
for k, v in res.items():
  print(k, "vs eager/inline: equal", np.array_equal(v, base), "max diff %.3g" % np.abs(v - base).max(), "first diff at", int(np.argmax(v != base)) if not np.array_equal(v, base) else -1)
