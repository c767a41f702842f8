"""The REAL reference on REAL data vs this build, with nothing shared but the seed.

tests/golden/real_ml20m_slice.npz (tests/golden/make_golden_real.py) holds what the reference's own
Recoder.train / _evaluate produced on the ML-20M slice its tests ship (10 000 users x 7 915 items):
per-step losses of 6 epochs, per-user Recall@20 / Recall@50 / NDCG@100 on a held-out fifth of every
user's interactions, the top-100 lists of 50 users.  Here the same calls run on the GPU (default
path: HIP-graph replay) after the same torch.manual_seed -- model initialisation and every epoch's
user order come out of the global RNG in the reference's sequence, no hooks.
north_star: loss within 1e-5 relative, Recall@k matching the reference to 4 decimals."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load():
  z = np.load(os.path.join(HERE, "golden", "real_ml20m_slice.npz"))
  shape = tuple(int(v) for v in z["shape"])
  mk = lambda p: sp.csr_matrix((z[p + "/data"], z[p + "/indices"], z[p + "/indptr"]), shape=shape)
  return z, mk("x"), mk("y")


@pytest.mark.parametrize("loss", ["logloss", "mse"])
def test_reference_run_on_its_own_ml20m_slice(loss):
  from recoder_amd.data import RecommendationDataset, UsersInteractions
  from recoder_amd.metrics import NDCG, Recall
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  z, x, y = load()
  torch.manual_seed(int(z["seed"]))
  model = DynamicAutoencoder(hidden_layers=[200], activation_type="tanh", noise_prob=0.0, sparse=False)
  trainer = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
  trainer.train(train_dataset=RecommendationDataset(x), batch_size=int(z["batch_size"]), lr=1e-3,
                weight_decay=2e-5, num_epochs=int(z["epochs"]), negative_sampling=True)
  losses = np.concatenate(trainer.loss_history)
  ref = z[loss + "/losses"]
  assert len(losses) == len(ref) == 120
  rel = np.abs(losses - ref) / np.abs(ref)
  print(loss, "max rel loss error %.3g at step %d" % (rel.max(), int(rel.argmax())))
  assert rel.max() < 1e-5
  metrics = [Recall(k=20, normalize=True), Recall(k=50, normalize=True), NDCG(k=100)]
  res = trainer._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                          metrics=metrics, batch_size=500)
  for m in metrics:
    got, want = np.asarray(res[m], dtype=np.float64), z[loss + "/" + str(m)]
    assert got.shape == want.shape == (10000,)
    same = np.isclose(got, want, rtol=0, atol=1e-12) | (np.isnan(got) & np.isnan(want))
    print("   %-10s mean %.6f (reference %.6f), %d of 10000 users differ"
          % (m, np.nanmean(got), np.nanmean(want), int((~same).sum())))
    assert abs(np.nanmean(got) - np.nanmean(want)) < 5e-5          # 4 decimals
    assert (~same).sum() <= 20           # (a near-tie at the k-th place may swap for a few users)
  top = np.asarray(trainer.recommend(UsersInteractions(users=np.arange(50), interactions_matrix=x[:50]), 100))
  want = z[loss + "/top100"]
  assert (top == want).mean() > 0.995     # positions; swaps only between scores a rounding apart
  assert all(set(a) == set(b) or len(set(a) ^ set(b)) <= 2 for a, b in zip(top, want))


def test_reference_run_with_validation_evaluation_and_checkpoint(tmp_path):
  """SparseAdam + the validation loss and an evaluation INSIDE training every 2 epochs (their
  loaders draw from the global RNG as well: every training order behind them depends on the same
  consumption), then the reference test's own epilogue (tests/test_model.py:64-82): save_state,
  a fresh trainer, init_from_model_file, evaluate."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.metrics import NDCG, Recall
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  z, x, y = load()
  torch.manual_seed(int(z["seed"]) + 1)
  model = DynamicAutoencoder(hidden_layers=[200], activation_type="tanh", noise_prob=0.0, sparse=True)
  trainer = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="logloss")
  metrics = [Recall(k=20, normalize=True), NDCG(k=100)]
  prefix = str(tmp_path / "ck")
  trainer.train(train_dataset=RecommendationDataset(x), val_dataset=RecommendationDataset(x, y),
                batch_size=int(z["batch_size"]), lr=1e-3, weight_decay=0, num_epochs=4,
                negative_sampling=True, eval_freq=2, metrics=metrics, eval_num_recommendations=100,
                eval_num_users=2000, model_checkpoint_prefix=prefix, checkpoint_freq=4)
  losses, ref = np.concatenate(trainer.loss_history), z["sv/losses"]
  assert len(losses) == len(ref) == 80
  rel = np.abs(losses - ref) / np.abs(ref)
  print("max rel loss error %.3g at step %d" % (rel.max(), int(rel.argmax())))
  assert rel.max() < 1e-5
  model2 = DynamicAutoencoder(sparse=True)
  trainer2 = Recoder(model=model2, use_cuda=True, optimizer_type="adam", loss="logloss")
  trainer2.init_from_model_file(prefix + "_epoch_4.model")
  res = trainer2._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                           metrics=metrics, batch_size=500)
  for m in metrics:
    got, want = np.asarray(res[m], dtype=np.float64), z["sv/" + str(m)]
    same = np.isclose(got, want, rtol=0, atol=1e-12) | (np.isnan(got) & np.isnan(want))
    print("   %-10s mean %.6f (reference %.6f), %d of 10000 users differ"
          % (m, np.nanmean(got), np.nanmean(want), int((~same).sum())))
    assert abs(np.nanmean(got) - np.nanmean(want)) < 5e-5 and (~same).sum() <= 20


def test_reference_matrix_factorization_run():
  """MatrixFactorization (d = 64, tanh, logistic loss, dense Adam incl. the user table), 4 epochs."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.metrics import NDCG, Recall
  from recoder_amd.model import Recoder
  from recoder_amd.nn import MatrixFactorization
  z, x, y = load()
  torch.manual_seed(int(z["seed"]) + 2)
  model = MatrixFactorization(embedding_size=64, activation_type="tanh", dropout_prob=0, sparse=False)
  trainer = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="logistic")
  trainer.train(train_dataset=RecommendationDataset(x), batch_size=int(z["batch_size"]), lr=1e-3,
                weight_decay=2e-5, num_epochs=4, negative_sampling=True)
  losses, ref = np.concatenate(trainer.loss_history), z["mf/losses"]
  assert len(losses) == len(ref) == 80
  rel = np.abs(losses - ref) / np.abs(ref)
  print("mf max rel loss error %.3g at step %d" % (rel.max(), int(rel.argmax())))
  assert rel.max() < 1e-5
  metrics = [Recall(k=20, normalize=True), NDCG(k=100)]
  res = trainer._evaluate(eval_dataset=RecommendationDataset(x, y), num_recommendations=100,
                          metrics=metrics, batch_size=500)
  for m in metrics:
    got, want = np.asarray(res[m], dtype=np.float64), z["mf/" + str(m)]
    same = np.isclose(got, want, rtol=0, atol=1e-12) | (np.isnan(got) & np.isnan(want))
    print("   %-10s mean %.6f (reference %.6f), %d of 10000 users differ"
          % (m, np.nanmean(got), np.nanmean(want), int((~same).sum())))
    assert abs(np.nanmean(got) - np.nanmean(want)) < 5e-5 and (~same).sum() <= 40
