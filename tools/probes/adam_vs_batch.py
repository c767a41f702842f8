import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from recoder_amd import synthetic
from recoder_amd.data import RecommendationDataset
from recoder_amd.model import Recoder
from recoder_amd.nn import DynamicAutoencoder
B = int(sys.argv[1])
csr = synthetic.ml20m_like(seed=0)[:B * 300]
torch.manual_seed(0)
rec = Recoder(model=DynamicAutoencoder([200], activation_type="tanh", noise_prob=0.5, sparse=False), use_cuda=True, optimizer_type="adam", loss="mse")
rec.train(RecommendationDataset(csr), batch_size=B, lr=1e-3, weight_decay=2e-5, num_epochs=1, negative_sampling=True)
torch.cuda.synchronize()
