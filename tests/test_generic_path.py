"""GPU tests of the generic (non-fused) path (recoder_amd/generic.py) against golden
vectors produced by the real reference (tests/golden/make_golden_generic.py):
sgd / rmsprop / adagrad, an nn.Module loss, and a user-defined FactorizationModel
(tutorial.md "Your Own Factorization Model").  Same RNG seed + recorded user order ->
same initial parameters and batches; losses within 1e-5 relative, final parameters
within 1e-4."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.generic_configs import GENERIC_CONFIGS

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
  g = np.load(os.path.join(HERE, "golden", "generic_%s.npz" % name))
  shape = tuple(int(x) for x in g["csr/shape"])
  csr = sp.csr_matrix((g["csr/data"], g["csr/indices"], g["csr/indptr"]), shape=shape)
  csr_te = sp.csr_matrix((g["csr_te/data"], g["csr_te/indices"], g["csr_te/indptr"]), shape=shape)
  return g, csr, csr_te


def build(cfg):
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, FactorizationModel, MatrixFactorization
  from tests.custom_models import make_two_tower
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
  return model, Recoder(model=model, use_cuda=True, optimizer_type=cfg["optimizer"], loss=loss)


@pytest.mark.parametrize("name", list(GENERIC_CONFIGS))
def test_generic_path_replays_reference(name):
  from recoder_amd.data import RecommendationDataLoader, RecommendationDataset, UsersInteractions
  from recoder_amd.generic import GenericEngine
  cfg = GENERIC_CONFIGS[name]
  g, csr, csr_te = load(name)
  model, rec = build(cfg)
  n = csr.shape[0]
  rec.user_order_hook = lambda epoch, n_: g["order"][(epoch - 1) * n:epoch * n] if epoch > 0 \
      else g["val_order"]
  rec.train(RecommendationDataset(csr), **cfg["train"])
  assert isinstance(rec._engine(), GenericEngine)
  # same seed, same init_model draw order -> the reference's initial parameters
  losses = np.concatenate(rec.loss_history)
  assert len(losses) == len(g["losses"])
  rel = np.abs(losses - g["losses"]) / np.abs(g["losses"])
  print(name, "max rel loss err", rel.max())
  assert rel.max() < 1e-5, (rel.argmax(), rel.max(), losses[:3], g["losses"][:3])
  for k, p in model.named_parameters():
    want = g["final/" + k]
    got = p.detach().cpu().numpy()
    err = np.abs(got - want).max()
    assert err < 1e-4 * max(1.0, np.abs(want).max()), (k, err)
  # validation loss with an independently collated target
  val = RecommendationDataLoader(RecommendationDataset(csr, csr_te), batch_size=cfg["train"]["batch_size"],
                                 negative_sampling=cfg["train"].get("negative_sampling", False))
  got = rec._validate(val)
  assert abs(got - float(g["val_loss"])) / abs(float(g["val_loss"])) < 1e-5, (got, float(g["val_loss"]))
  # predict: full-catalogue scores
  out, _ = rec.predict(UsersInteractions(users=np.arange(8), interactions_matrix=csr[:8]),
                       return_input=True)
  want = g["predict8"]
  err = np.abs(out.cpu().numpy() - want).max()
  assert err < 1e-4 * max(1.0, np.abs(want).max()), err
  # recommend runs on the same scores (top-k kernel)
  recs = rec.recommend(UsersInteractions(users=np.arange(8), interactions_matrix=csr[:8]), 5)
  assert len(recs) == 8 and all(len(r) == 5 for r in recs)


def test_sparse_params_with_sgd_raise_like_the_reference():
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  _, csr, _ = load("ae_sgd")
  rec = Recoder(model=DynamicAutoencoder([16], sparse=True), use_cuda=True, optimizer_type="sgd",
                loss="mse")
  with pytest.raises(ValueError):
    rec.train(RecommendationDataset(csr), batch_size=32, num_epochs=1)
