import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_hip_parity import synth_csr
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
sampling = os.environ.get("SAMPLING", "0") == "1"
marks = [int(x) for x in os.environ.get("MARKS", "2").split(",") if x]
loss = os.environ.get("LOSS", "logistic")
noise = float(os.environ.get("NOISE", "0.3"))
os.environ["RK_GRAPH_GROUP"] = os.environ.get("GG", "8")
N = int(os.environ.get("N", "600")); BB = int(os.environ.get("B", "100")); H = int(os.environ.get("H", "24")); EP = int(os.environ.get("EPOCHS", "3")); NI = int(os.environ.get("NI", "400"))
csr = synth_csr(N, NI, 9, seed=515)
torch.manual_seed(56)
model = DynamicAutoencoder([H], activation_type=os.environ.get("ACT", "selu"), noise_prob=noise, sparse=False)
rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
rec.step_marks = {m: (lambda: False) for m in marks}
import recoder_amd.device as D
_orig = D.Block.check
def chk(self):
  c = self.counts.cpu().numpy()
  print("block n_cap", self.n_cap, "ref.n_cap", self.c.n_cap, "counts[:8]", c[:8].tolist(), flush=True)
  return _orig(self)
D.Block.check = chk
rec.train(RecommendationDataset(csr), batch_size=BB, lr=1e-3, weight_decay=1e-5, num_epochs=EP,
          negative_sampling=sampling, lr_milestones=([2] if os.environ.get("MS", "1") == "1" else None))
torch.cuda.synchronize()
print("OK", np.concatenate(rec.loss_history)[:4])
