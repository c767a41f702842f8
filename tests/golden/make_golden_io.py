"""Golden vectors of the rows either side of the hot path, from the REAL reference (build
container only; import shims as in make_golden.py):

  io_dataframe.npz        recoder.utils.dataframe_to_csr_matrix (utils.py:26-66) on a seeded
                          DataFrame (string user ids, integer item ids, duplicate pairs): the CSR
                          and both id maps; plus the call with the maps passed in on a subset.
  ref_ckpt_<name>.model   a checkpoint file written by the reference's Recoder.save_state
                          (model.py:193-224) -- a torch pickle of tensors / arrays / scalars: data.
  train_target_<name>.npz training on a dataset WITH a target matrix (data.py:60-62): CSRs, initial
                          parameters, user order, per-step losses, final parameters.
  ref_ckpt_<name>.npz     what produced it (CSR, initial parameters, every batch's users) and
                          what the reference does with it after init_from_model_file in a fresh
                          trainer: top-k recommendations, scores, the losses of one more epoch.

    python tests/golden/make_golden_io.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import import_reference, synth_csr  # noqa: E402

CKPT_CONFIGS = {
  "ae": dict(kind="ae", model=dict(hidden_layers=[20], activation_type="tanh", sparse=False),
             loss="mse", loss_params=None,
             train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True,
                        lr_milestones=[2]),
             data=dict(n_users=130, n_items=110, mean_deg=9, seed=41)),
  "mf_sparse": dict(kind="mf", model=dict(embedding_size=12, activation_type="tanh", sparse=True),
                    loss="logistic", loss_params=None,
                    train=dict(batch_size=32, lr=1e-3, weight_decay=0.0, num_epochs=2,
                               negative_sampling=True),
                    data=dict(n_users=130, n_items=110, mean_deg=9, seed=42)),
}


def make_dataframe_fixture():
  import pandas as pd
  from recoder.utils import dataframe_to_csr_matrix
  rng = np.random.RandomState(7)
  n = 900
  users = np.asarray(["u%03d" % u for u in rng.randint(0, 80, n)])
  items = rng.randint(1000, 1150, n).astype(np.int64)
  inter = rng.randint(1, 6, n).astype(np.float32)
  df = pd.DataFrame({"user": users, "item": items, "inter": inter})     # duplicate pairs stay in
  m, imap, umap = dataframe_to_csr_matrix(df, user_col="user", item_col="item", inter_col="inter")
  m.sort_indices()
  sub = df.iloc[100:400]
  m2, imap2, umap2 = dataframe_to_csr_matrix(sub, "user", "item", "inter", item_id_map=imap,
                                             user_id_map=umap)
  m2.sort_indices()
  assert imap2 is imap and umap2 is umap
  gold = {
    "users": users, "items": items, "inter": inter,
    "csr/indptr": m.indptr.astype(np.int64), "csr/indices": m.indices.astype(np.int64),
    "csr/data": np.asarray(m.data, dtype=np.float64), "csr/shape": np.asarray(m.shape),
    "item_keys": np.asarray(list(imap.keys()), dtype=np.int64),
    "item_vals": np.asarray(list(imap.values()), dtype=np.int64),
    "user_keys": np.asarray(list(umap.keys())),
    "user_vals": np.asarray(list(umap.values()), dtype=np.int64),
    "sub_lo": np.asarray(100), "sub_hi": np.asarray(400),
    "csr2/indptr": m2.indptr.astype(np.int64), "csr2/indices": m2.indices.astype(np.int64),
    "csr2/data": np.asarray(m2.data, dtype=np.float64), "csr2/shape": np.asarray(m2.shape),
  }
  path = os.path.join(HERE, "io_dataframe.npz")
  np.savez_compressed(path, **gold)
  print("wrote", path, m.shape, m.nnz)


def make_checkpoint_fixture(name, cfg):
  from recoder.data import RecommendationDataset, UsersInteractions
  from recoder.model import Recoder
  from recoder.nn import DynamicAutoencoder, MatrixFactorization
  csr = synth_csr(**cfg["data"])

  def new_trainer():
    model = DynamicAutoencoder(**cfg["model"]) if cfg["kind"] == "ae" else MatrixFactorization(**cfg["model"])
    return model, Recoder(model=model, use_cuda=False, optimizer_type="adam", loss=cfg["loss"],
                          loss_params=cfg["loss_params"])

  def record(trainer, model, sink):
    orig = trainer._Recoder__compute_loss

    def compute_loss(input, target):
      loss = orig(input, target)
      if model.training:
        sink["users"].append(input.users.numpy().copy())
        sink["losses"].append(float(loss.item()))
      return loss
    trainer._Recoder__compute_loss = compute_loss

  torch.manual_seed(2468)
  model, trainer = new_trainer()
  holder = {}
  orig_init = model.init_model

  def init_model(num_items=None, num_users=None):
    orig_init(num_items, num_users)
    holder["init"] = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
  model.init_model = init_model
  run1 = dict(users=[], losses=[])
  record(trainer, model, run1)
  prefix = os.path.join(HERE, "ref_ckpt_" + name)
  trainer.train(train_dataset=RecommendationDataset(csr), model_checkpoint_prefix=prefix, **cfg["train"])
  written = "%s_epoch_%d.model" % (prefix, cfg["train"]["num_epochs"])
  final = os.path.join(HERE, "ref_ckpt_%s.model" % name)
  os.replace(written, final)

  # a FRESH reference trainer loads the file: recommendations, scores, one more epoch
  model2, trainer2 = new_trainer()
  trainer2.init_from_model_file(final)
  users = np.arange(csr.shape[0])
  recs = []
  for off in range(0, len(users), 50):
    u = users[off:off + 50]
    recs += trainer2.recommend(UsersInteractions(users=u, interactions_matrix=csr[u]), 10)
  out, _ = trainer2.predict(UsersInteractions(users=users[:8], interactions_matrix=csr[users[:8]]))
  run2 = dict(users=[], losses=[])
  record(trainer2, model2, run2)
  t2 = dict(cfg["train"]); t2["num_epochs"] = cfg["train"]["num_epochs"] + 1
  torch.manual_seed(1357)
  trainer2.train(train_dataset=RecommendationDataset(csr), **t2)

  gold = {"csr/indptr": csr.indptr.astype(np.int64), "csr/indices": csr.indices.astype(np.int32),
          "csr/data": csr.data.astype(np.float32), "csr/shape": np.asarray(csr.shape),
          "order1": np.concatenate(run1["users"]).astype(np.int64),
          "losses1": np.asarray(run1["losses"], dtype=np.float64),
          "topk": np.asarray(recs, dtype=np.int64), "scores8": out.detach().numpy(),
          "order2": np.concatenate(run2["users"]).astype(np.int64),
          "losses2": np.asarray(run2["losses"], dtype=np.float64),
          "resumed_epoch": np.asarray(trainer2.current_epoch)}
  for k, v in holder["init"].items():
    gold["init/" + k] = v
  for k, v in model2.named_parameters():
    gold["final2/" + k] = v.detach().numpy().copy()
  np.savez_compressed(os.path.join(HERE, "ref_ckpt_%s.npz" % name), **gold)
  st = torch.load(final, map_location="cpu", weights_only=False)
  print("wrote", final, "%.1f KB" % (os.path.getsize(final) / 1024), "keys", sorted(st.keys()),
        "| run1", len(run1["losses"]), "steps, resumed run", len(run2["losses"]), "steps at epoch",
        int(gold["resumed_epoch"]))


TARGET_CONFIGS = {
  # datasets WITH a target matrix in training (reference data.py:60-62, model.py:464-472): the
  # decoder side runs over the target batch's item set, the encoder side over the input's
  "ae": dict(kind="ae", model=dict(hidden_layers=[20], activation_type="tanh", sparse=False),
             loss="mse", train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2,
                                    negative_sampling=True),
             data=dict(n_users=130, n_items=110, mean_deg=9, seed=51)),
  "ae2_sparse": dict(kind="ae", model=dict(hidden_layers=[24, 12], activation_type="sigmoid", sparse=True),
                     loss="logistic", train=dict(batch_size=32, lr=1e-3, weight_decay=0.0, num_epochs=2,
                                                 negative_sampling=True),
                     data=dict(n_users=130, n_items=110, mean_deg=9, seed=52)),
  # tied weights (nn.py:191-202, 224-226): the decoder reads the ENCODER table at the target items
  "ae_tied": dict(kind="ae", model=dict(hidden_layers=[20], activation_type="tanh", sparse=False,
                                        is_constrained=True),
                  loss="logistic", train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2,
                                              negative_sampling=True),
                  data=dict(n_users=130, n_items=110, mean_deg=9, seed=54)),
  "mf": dict(kind="mf", model=dict(embedding_size=12, activation_type="tanh", sparse=False),
             loss="logloss", train=dict(batch_size=32, lr=1e-3, weight_decay=1e-5, num_epochs=2,
                                        negative_sampling=True),
             data=dict(n_users=130, n_items=110, mean_deg=9, seed=53)),
}


def make_target_fixture(name, cfg):
  from recoder.data import RecommendationDataset
  from recoder.model import Recoder
  from recoder.nn import DynamicAutoencoder, MatrixFactorization
  csr = synth_csr(**cfg["data"])
  d2 = dict(cfg["data"]); d2["seed"] += 500
  csr_t = synth_csr(**d2)
  torch.manual_seed(8642)
  model = DynamicAutoencoder(**cfg["model"]) if cfg["kind"] == "ae" else MatrixFactorization(**cfg["model"])
  trainer = Recoder(model=model, use_cuda=False, optimizer_type="adam", loss=cfg["loss"])
  holder, rec = {}, dict(users=[], losses=[])
  orig_init = model.init_model

  def init_model(num_items=None, num_users=None):
    orig_init(num_items, num_users)
    holder["init"] = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
  model.init_model = init_model
  orig = trainer._Recoder__compute_loss

  def compute_loss(input, target):
    assert target is not None
    loss = orig(input, target)
    if model.training:
      rec["users"].append(input.users.numpy().copy())
      rec["losses"].append(float(loss.item()))
    return loss
  trainer._Recoder__compute_loss = compute_loss
  trainer.train(train_dataset=RecommendationDataset(csr, csr_t), **cfg["train"])
  gold = {"csr/indptr": csr.indptr.astype(np.int64), "csr/indices": csr.indices.astype(np.int32),
          "csr/data": csr.data.astype(np.float32), "csr/shape": np.asarray(csr.shape),
          "csr_t/indptr": csr_t.indptr.astype(np.int64), "csr_t/indices": csr_t.indices.astype(np.int32),
          "csr_t/data": csr_t.data.astype(np.float32),
          "order": np.concatenate(rec["users"]).astype(np.int64),
          "losses": np.asarray(rec["losses"], dtype=np.float64)}
  for k, v in holder["init"].items():
    gold["init/" + k] = v
  for k, v in model.named_parameters():
    gold["final/" + k] = v.detach().numpy().copy()
  np.savez_compressed(os.path.join(HERE, "train_target_%s.npz" % name), **gold)
  print("wrote train_target_%s.npz:" % name, len(rec["losses"]), "steps, loss", rec["losses"][0], "->",
        rec["losses"][-1])


if __name__ == "__main__":
  import_reference()
  # fifth shim (this torch, not the reference's 1.8.1): torch.load defaults to weights_only=True
  # since 2.6 and then refuses the numpy arrays of the reference's own checkpoint dict
  import functools
  _load = torch.load
  torch.load = functools.partial(_load, weights_only=False)
  only = sys.argv[1:]            # e.g. `target:ae_tied`: regenerate just that fixture
  if not only:
    make_dataframe_fixture()
  for name, cfg in CKPT_CONFIGS.items():
    if not only or "ckpt:" + name in only:
      make_checkpoint_fixture(name, cfg)
  for name, cfg in TARGET_CONFIGS.items():
    if not only or "target:" + name in only:
      make_target_fixture(name, cfg)
