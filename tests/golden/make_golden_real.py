"""Golden run of the REAL reference on REAL data: the ML-20M slice its own tests hold
(/root/reference/tests/data/val.csv: 10 000 users, 7 915 items, 142 514 interactions).

The reference's quality test (tests/test_model.py:14-84) cannot run here (its train.csv is not in
the mount), so this is the same recipe on the part that is: the autoencoder of scripts/ml-20m
(hidden [200], tanh, Adam lr 1e-3 wd 2e-5, B = 500, negative sampling) trained by the reference's
own ``Recoder.train`` on 80 % of every user's interactions, evaluated by its own ``_evaluate``
on the held-out 20 % (Recall@20, Recall@50, NDCG@100).  Input noise is off, so the run depends on
nothing but ``torch.manual_seed``: model initialisation and the per-epoch user orders come from the
global RNG, which recoder_amd consumes in the same sequence -- the GPU test
(tests/test_real_data_golden.py) gets NO hook, only the seed.

Written: tests/golden/real_ml20m_slice.npz -- the two CSR matrices (data: a data file the
reference's tests hold, re-encoded), the per-step losses of the reference, its metrics, and the
top-100 lists of the first 50 users.

    python tests/golden/make_golden_real.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import import_reference  # noqa: E402

SEED, EPOCHS, B = 20240917, 6, 500
LOSSES = ("logloss", "mse")
# third run: SparseAdam + validation loss and evaluation INSIDE training every 2 epochs (their
# loaders draw from the global RNG too: the training orders behind them depend on it), then the
# reference's save_state -> fresh trainer -> init_from_model_file -> evaluate (tests/test_model.py:64-82)


def split(csr, seed):
  """80 / 20 split of every user's interactions (users with < 5 keep everything as input)."""
  rng = np.random.RandomState(seed)
  coo = csr.tocoo()
  keep = np.ones(coo.nnz, dtype=bool)
  order = np.argsort(coo.row, kind="stable")
  rows = coo.row[order]
  starts = np.flatnonzero(np.r_[True, rows[1:] != rows[:-1]])
  ends = np.r_[starts[1:], len(rows)]
  for s, e in zip(starts, ends):
    n = e - s
    if n >= 5:
      held = rng.choice(n, size=max(1, n // 5), replace=False)
      keep[order[s + held]] = False
  x = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=csr.shape)
  y = sp.csr_matrix((coo.data[~keep], (coo.row[~keep], coo.col[~keep])), shape=csr.shape)
  for m in (x, y):
    m.sum_duplicates()
    m.sort_indices()
  return x, y


def main():
  import functools
  import pandas as pd
  import_reference()
  _load = torch.load
  torch.load = functools.partial(_load, weights_only=False)
  from recoder.data import RecommendationDataset
  from recoder.metrics import NDCG, Recall
  from recoder.model import Recoder
  from recoder.nn import DynamicAutoencoder
  from recoder.utils import dataframe_to_csr_matrix
  df = pd.read_csv("/root/reference/tests/data/val.csv")
  m, item_map, user_map = dataframe_to_csr_matrix(df, user_col="uid", item_col="sid", inter_col="watched")
  m = m.astype(np.float32).tocsr()
  m.sort_indices()
  x, y = split(m, 11)
  gold = {"x/indptr": x.indptr.astype(np.int64), "x/indices": x.indices.astype(np.int32),
          "x/data": x.data.astype(np.float32), "y/indptr": y.indptr.astype(np.int64),
          "y/indices": y.indices.astype(np.int32), "y/data": y.data.astype(np.float32),
          "shape": np.asarray(m.shape), "seed": np.asarray(SEED), "epochs": np.asarray(EPOCHS),
          "batch_size": np.asarray(B)}
  for loss in LOSSES:
    torch.manual_seed(SEED)
    model = DynamicAutoencoder(hidden_layers=[200], activation_type="tanh", noise_prob=0.0, sparse=False)
    trainer = Recoder(model=model, use_cuda=False, optimizer_type="adam", loss=loss)
    rec = []
    orig = trainer._Recoder__compute_loss

    def compute_loss(input, target, orig=orig, model=model, rec=rec):
      out = orig(input, target)
      if model.training:
        rec.append(float(out.item()))
      return out
    trainer._Recoder__compute_loss = compute_loss
    trainer.train(train_dataset=RecommendationDataset(x), batch_size=B, lr=1e-3, weight_decay=2e-5,
                  num_epochs=EPOCHS, negative_sampling=True)
    metrics = [Recall(k=20, normalize=True), Recall(k=50, normalize=True), NDCG(k=100)]
    res = trainer._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                            metrics=metrics, batch_size=500)
    from recoder.data import UsersInteractions
    top = trainer.recommend(UsersInteractions(users=np.arange(50), interactions_matrix=x[:50]), 100)
    gold[loss + "/losses"] = np.asarray(rec, dtype=np.float64)
    for mt in metrics:
      gold[loss + "/" + str(mt)] = np.asarray(res[mt], dtype=np.float64)
    gold[loss + "/top100"] = np.asarray(top, dtype=np.int64)
    print(loss, "steps", len(rec), "loss %.5f -> %.5f" % (rec[0], rec[-1]),
          {str(mt): round(float(np.nanmean(res[mt])), 6) for mt in metrics})
  # ---- sparse + validation / evaluation inside training + checkpoint round trip ----
  torch.manual_seed(SEED + 1)
  model = DynamicAutoencoder(hidden_layers=[200], activation_type="tanh", noise_prob=0.0, sparse=True)
  trainer = Recoder(model=model, use_cuda=False, optimizer_type="adam", loss="logloss")
  rec, vals = [], []
  orig = trainer._Recoder__compute_loss

  def compute_loss2(input, target):
    out = orig(input, target)
    (rec if model.training else vals).append(float(out.item()))
    return out
  trainer._Recoder__compute_loss = compute_loss2
  metrics = [Recall(k=20, normalize=True), NDCG(k=100)]
  prefix = "/tmp/rk_real_ckpt"
  trainer.train(train_dataset=RecommendationDataset(x), val_dataset=RecommendationDataset(x, y),
                batch_size=B, lr=1e-3, weight_decay=0, num_epochs=4, negative_sampling=True,
                eval_freq=2, metrics=metrics, eval_num_recommendations=100, eval_num_users=2000,
                model_checkpoint_prefix=prefix, checkpoint_freq=4)
  model2 = DynamicAutoencoder(sparse=True)
  trainer2 = Recoder(model=model2, use_cuda=False, optimizer_type="adam", loss="logloss")
  trainer2.init_from_model_file(prefix + "_epoch_4.model")
  res = trainer2._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                           metrics=metrics, batch_size=500)
  gold["sv/losses"] = np.asarray(rec, dtype=np.float64)
  gold["sv/val_losses"] = np.asarray(vals, dtype=np.float64)
  for mt in metrics:
    gold["sv/" + str(mt)] = np.asarray(res[mt], dtype=np.float64)
  print("sparse+val steps", len(rec), "val batches", len(vals), "loss %.5f -> %.5f" % (rec[0], rec[-1]),
        {str(mt): round(float(np.nanmean(res[mt])), 6) for mt in metrics})
  # ---- MatrixFactorization (the other model family north_star names) ----
  from recoder.nn import MatrixFactorization
  torch.manual_seed(SEED + 2)
  model = MatrixFactorization(embedding_size=64, activation_type="tanh", dropout_prob=0, sparse=False)
  trainer = Recoder(model=model, use_cuda=False, optimizer_type="adam", loss="logistic")
  rec = []
  orig_mf = trainer._Recoder__compute_loss

  def compute_loss3(input, target):
    out = orig_mf(input, target)
    if model.training:
      rec.append(float(out.item()))
    return out
  trainer._Recoder__compute_loss = compute_loss3
  trainer.train(train_dataset=RecommendationDataset(x), batch_size=B, lr=1e-3, weight_decay=2e-5,
                num_epochs=4, negative_sampling=True)
  metrics = [Recall(k=20, normalize=True), NDCG(k=100)]
  res = trainer._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                          metrics=metrics, batch_size=500)
  gold["mf/losses"] = np.asarray(rec, dtype=np.float64)
  for mt in metrics:
    gold["mf/" + str(mt)] = np.asarray(res[mt], dtype=np.float64)
  print("mf steps", len(rec), "loss %.5f -> %.5f" % (rec[0], rec[-1]),
        {str(mt): round(float(np.nanmean(res[mt])), 6) for mt in metrics})
  path = os.path.join(HERE, "real_ml20m_slice.npz")
  np.savez_compressed(path, **gold)
  print("wrote", path, "%.0f KB" % (os.path.getsize(path) / 1024), m.shape, m.nnz)


if __name__ == "__main__":
  main()
