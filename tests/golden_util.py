"""Helpers shared by the golden-vector tests (test infrastructure)."""
import os
from collections import OrderedDict

import numpy as np
import scipy.sparse as sp
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# model / train hyper-parameters of each golden file (the data itself is in the
# .npz; these mirror tests/golden/make_golden.py CONFIGS)
CONFIGS = {
  "ae_mse_dense": dict(kind="ae", hidden_layers=[24], activation_type="tanh", noise_prob=0.5,
                       sparse=False, loss="mse", loss_params=None, batch_size=32, lr=1e-3,
                       weight_decay=2e-5, negative_sampling=True, lr_milestones=[3], evaluate=True),
  "ae_mse_conf_sparse": dict(kind="ae", hidden_layers=[24], activation_type="tanh", noise_prob=0.0,
                             sparse=True, loss="mse", loss_params=dict(confidence=3), batch_size=32,
                             lr=1e-3, weight_decay=2e-5, negative_sampling=True),
  "ae2_logloss_dense": dict(kind="ae", hidden_layers=[24, 16], activation_type="tanh", noise_prob=0.5,
                            dropout_prob=0.25, sparse=False, loss="logloss", loss_params=None,
                            batch_size=32, lr=1e-3, weight_decay=2e-5, negative_sampling=True,
                            evaluate=True),
  "ae2_constrained_bce": dict(kind="ae", hidden_layers=[24, 16], activation_type="sigmoid",
                              noise_prob=0.3, is_constrained=True, sparse=False, loss="logistic",
                              loss_params=None, batch_size=32, lr=2e-3, weight_decay=1e-5,
                              negative_sampling=True),
  "ae_mse_sampling2": dict(kind="ae", hidden_layers=[24], activation_type="relu", noise_prob=0.5,
                           sparse=True, loss="mse", loss_params=None, batch_size=32, lr=1e-3,
                           weight_decay=0.0, negative_sampling=True, num_sampling_users=64),
  "ae_mse_nosampling": dict(kind="ae", hidden_layers=[24], activation_type="tanh", noise_prob=0.0,
                            sparse=False, loss="mse", loss_params=None, batch_size=32, lr=1e-3,
                            weight_decay=2e-5, negative_sampling=False),
  "mf_mse_sparse": dict(kind="mf", embedding_size=16, activation_type="none", sparse=True,
                        loss="mse", loss_params=None, batch_size=32, lr=1e-3, weight_decay=2e-5,
                        negative_sampling=True),
  "mf_bce_dense": dict(kind="mf", embedding_size=16, activation_type="tanh", dropout_prob=0.3,
                       sparse=False, loss="logistic", loss_params=None, batch_size=32, lr=1e-3,
                       weight_decay=2e-5, negative_sampling=True, evaluate=True),
}


class Golden:
  def __init__(self, name):
    self.name = name
    self.cfg = CONFIGS[name]
    self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    shape = tuple(int(x) for x in self.z["csr/shape"])
    self.csr = sp.csr_matrix((self.z["csr/data"], self.z["csr/indices"], self.z["csr/indptr"]),
                             shape=shape)
    self.csr_te = sp.csr_matrix((self.z["csr_te/data"], self.z["csr_te/indices"],
                                 self.z["csr_te/indptr"]), shape=shape)
    self.nsteps = int(self.z["nsteps"])
    self.losses = self.z["losses"]

  def state(self, prefix):
    out = OrderedDict()
    pre = prefix + "/"
    for k in self.z.files:
      if k.startswith(pre):
        out[k[len(pre):]] = torch.from_numpy(self.z[k])
    return out

  def adam(self, prefix):
    """{param: (step, exp_avg, exp_avg_sq)}"""
    out = {}
    pre = prefix + "/"
    for k in self.z.files:
      if k.startswith(pre) and k.endswith("/step"):
        n = k[len(pre):-len("/step")]
        out[n] = (int(self.z[k]), self.z[pre + n + "/exp_avg"], self.z[pre + n + "/exp_avg_sq"])
    return out

  def step(self, i):
    g = lambda k: self.z["step%d/%s" % (i, k)] if ("step%d/%s" % (i, k)) in self.z.files else None
    return dict(users=g("users"), items=g("items"), indices=g("indices"), values=g("values"),
                size=tuple(int(x) for x in g("size")), noise_keep=g("noise_keep"),
                drop_keep=g("drop_keep"))

  def steps_per_epoch(self):
    return int(np.ceil(self.csr.shape[0] / self.cfg["batch_size"]))

  def lr_at(self, i):
    ms = self.cfg.get("lr_milestones")
    epoch = i // self.steps_per_epoch() + 1
    if not ms:
      return self.cfg["lr"]
    return self.cfg["lr"] * (0.1 ** sum(1 for m in ms if m <= epoch))

  def groups(self):
    """Yield (first_step, [step indices]) per sampling group (data.py:138-144)."""
    B = self.cfg["batch_size"]
    S = self.cfg.get("num_sampling_users", 0) or B
    spe = self.steps_per_epoch()
    i = 0
    while i < self.nsteps:
      epoch = i // spe
      j, n = i, 0
      while j < self.nsteps and n < S and j // spe == epoch:
        n += len(self.z["step%d/users" % j])
        j += 1
      yield list(range(i, j))
      i = j
