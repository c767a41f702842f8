"""Soak: the full C2 workload for many epochs (noise on, validation + Recall@20 every few epochs, a
checkpoint written and resumed in the middle), once with graph replay and once eagerly enqueued:
losses, metrics and final parameters must be bit-identical.   python tools/probes/soak.py [epochs]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from recoder_amd import synthetic
from recoder_amd.data import RecommendationDataset
from recoder_amd.metrics import Recall
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
E = int(sys.argv[1]) if len(sys.argv) > 1 else 30
csr = synthetic.ml20m_like(seed=0)
val = csr[:5000]
train = csr[5000:]

def run(graph):
  os.environ["RK_GRAPH"] = "1" if graph else "0"
  torch.manual_seed(3)
  rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5), use_cuda=True,
                optimizer_type="adam", loss="mse")
  kw = dict(batch_size=500, lr=1e-3, weight_decay=2e-5, negative_sampling=True, lr_milestones=[E // 2],
            val_dataset=RecommendationDataset(val, val), eval_freq=7, metrics=[Recall(20)],
            eval_num_recommendations=20, eval_num_users=2000)
  d = tempfile.mkdtemp()
  t0 = time.perf_counter()
  rec.train(RecommendationDataset(train), num_epochs=E // 2, model_checkpoint_prefix=os.path.join(d, "m"), **kw)
  path = os.path.join(d, "m_epoch_%d.model" % (E // 2))
  rec2 = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5), use_cuda=True,
                 optimizer_type="adam", loss="mse")
  rec2.init_from_model_file(path)
  rec2.train(RecommendationDataset(train), num_epochs=E, **kw)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  losses = np.concatenate(rec.loss_history + rec2.loss_history)
  pars = {k: v.detach().cpu().clone() for k, v in rec2.model.named_parameters()}
  print("graph=%s: %d steps in %.1f s, loss %.4f -> %.4f, last summary %s" %
        (graph, len(losses), dt, losses[0], losses[-1], rec2.last_epoch_summary), flush=True)
  return losses, pars

l0, p0 = run(False)
l1, p1 = run(True)
print("losses bit-identical:", np.array_equal(l0, l1), "finite:", bool(np.isfinite(l1).all()))
print("parameters bit-identical:", all(torch.equal(p0[k], p1[k]) for k in p0))
