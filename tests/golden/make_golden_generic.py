"""Golden vectors for the GENERIC (non-fused) path from the REAL reference
(build container only; see make_golden.py for the import shims): user-defined
model, nn.Module loss, sgd / adagrad / rmsprop.  Records the CSR, the initial
parameters, the user order of every batch, the per-step losses, the validation
loss and the final parameters.  Only data is written (tests/golden/generic_*.npz).

    python tests/golden/make_golden_generic.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.golden.make_golden import import_reference, synth_csr  # noqa: E402
from tests.generic_configs import GENERIC_CONFIGS  # noqa: E402


def run(name, cfg):
  from recoder.data import RecommendationDataset
  from recoder.model import Recoder
  from recoder.nn import DynamicAutoencoder, FactorizationModel, MatrixFactorization
  from tests.custom_models import make_two_tower
  csr = synth_csr(**cfg["data"])
  d2 = dict(cfg["data"]); d2["seed"] += 500
  csr_te = synth_csr(**d2)
  torch.manual_seed(4321)
  if cfg["kind"] == "ae":
    model = DynamicAutoencoder(**cfg["model"])
  elif cfg["kind"] == "mf":
    model = MatrixFactorization(**cfg["model"])
  else:
    model = make_two_tower(FactorizationModel)(**cfg["model"])
  loss = cfg["loss"]
  if loss == "smooth_l1_sum":
    loss = torch.nn.SmoothL1Loss(reduction="sum")
  trainer = Recoder(model=model, use_cuda=False, optimizer_type=cfg["optimizer"], loss=loss)
  rec = dict(users=[], losses=[])
  holder = {}
  orig_init = model.init_model

  def init_model(num_items=None, num_users=None):
    orig_init(num_items, num_users)
    holder["init"] = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
  model.init_model = init_model
  orig_cl = trainer._Recoder__compute_loss

  def compute_loss(input, target):
    loss_t = orig_cl(input, target)
    if model.training:
      rec["users"].append(input.users.numpy().copy())
      rec["losses"].append(float(loss_t.item()))
    return loss_t
  trainer._Recoder__compute_loss = compute_loss
  trainer.train(train_dataset=RecommendationDataset(csr), **cfg["train"])
  gold = {"csr/indptr": csr.indptr.astype(np.int64), "csr/indices": csr.indices.astype(np.int32),
          "csr/data": csr.data.astype(np.float32), "csr/shape": np.asarray(csr.shape),
          "csr_te/indptr": csr_te.indptr.astype(np.int64),
          "csr_te/indices": csr_te.indices.astype(np.int32),
          "csr_te/data": csr_te.data.astype(np.float32),
          "losses": np.asarray(rec["losses"], dtype=np.float64),
          "order": np.concatenate(rec["users"]).astype(np.int64),
          "batch_sizes": np.asarray([len(u) for u in rec["users"]])}
  for k, v in holder["init"].items():
    gold["init/" + k] = v
  for k, v in model.named_parameters():
    gold["final/" + k] = v.detach().numpy().copy()
  # validation loss (eval mode, independently collated target)
  val = RecommendationDataset(csr, csr_te)
  from recoder.data import RecommendationDataLoader
  vl = RecommendationDataLoader(val, batch_size=cfg["train"]["batch_size"],
                                negative_sampling=cfg["train"].get("negative_sampling", False))
  vusers = []
  orig2 = trainer._Recoder__compute_loss

  def cl2(input, target):
    vusers.append(input.users.numpy().copy())
    return orig_cl(input, target)
  trainer._Recoder__compute_loss = cl2
  gold["val_loss"] = np.asarray(trainer._validate(vl))
  gold["val_order"] = np.concatenate(vusers).astype(np.int64)
  # full-catalogue scores of the first 8 users (predict)
  from recoder.data import UsersInteractions
  ui = UsersInteractions(users=np.arange(8), interactions_matrix=csr[:8])
  out, _ = trainer.predict(ui, return_input=True)
  gold["predict8"] = out.detach().numpy()
  np.savez_compressed(os.path.join(HERE, "generic_%s.npz" % name), **gold)
  print(name, "steps", len(rec["losses"]), "loss", rec["losses"][0], "->", rec["losses"][-1],
        "val", float(gold["val_loss"]))


if __name__ == "__main__":
  import_reference()
  for name, cfg in GENERIC_CONFIGS.items():
    run(name, cfg)
