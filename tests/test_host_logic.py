"""CPU: host-side logic that mirrors the reference's API behaviour (no kernels)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests.golden_util import Golden


def test_metrics_known_answers():
  """reference tests/test_metrics.py:12-54 (plain numbers, rtol 1e-9)."""
  from recoder_amd.metrics import AveragePrecision, NDCG, Recall
  x = np.arange(10)
  assert np.isclose(AveragePrecision(10, False).evaluate(x, [0, 2, 5, 8, 9]),
                    1 / 5 * (1 + 2 / 3 + 3 / 6 + 4 / 9 + 5 / 10), rtol=1e-9, atol=0)
  assert np.isclose(AveragePrecision(3, True).evaluate(x, [1, 4, 5, 6, 12]), 1 / 3 * (1 / 2), rtol=1e-9)
  assert np.isclose(Recall(10, False).evaluate(x, [1, 4, 5, 6, 12]), 4 / 5, rtol=1e-9)
  assert np.isclose(Recall(3, True).evaluate(x, [0, 2, 5, 8, 9]), 2 / 3, rtol=1e-9)
  assert np.isclose(NDCG(10).evaluate(x, [0, 2, 5, 8, 9]), 0.8296882915641869, rtol=1e-9)
  assert np.isclose(NDCG(10).evaluate(x, [1, 4, 5, 6, 12]), 0.5790560467042355, rtol=1e-9)
  assert np.isclose(NDCG(3).evaluate(x, [0, 2, 5, 8, 9]), 0.7039180890341347, rtol=1e-9)
  assert np.isclose(NDCG(3).evaluate(x, [1, 4, 5, 6, 12]), 0.2960819109658652, rtol=1e-9)
  assert str(Recall(20)) == "Recall@20" and hash(NDCG(100)) == hash("NDCG@100")


def test_epoch_user_order_reproduces_the_reference_sampler():
  """Same global-RNG consumption as torch DataLoader + RandomSampler
  (reference data.py:124-136): after the reference-ordered model init, the user
  orders of both epochs equal the golden recording."""
  from recoder_amd.data import epoch_user_order
  from recoder_amd.nn import DynamicAutoencoder
  g = Golden("ae_mse_conf_sparse")
  torch.manual_seed(1234)
  m = DynamicAutoencoder(hidden_layers=[24], activation_type="tanh", noise_prob=0.0, sparse=True)
  m.init_model(g.csr.shape[1])
  spe = g.steps_per_epoch()
  for ep in range(2):
    order = epoch_user_order(g.csr.shape[0])
    want = np.concatenate([g.step(i)["users"] for i in range(ep * spe, (ep + 1) * spe)])
    assert np.array_equal(order, want)


@pytest.mark.parametrize("name", ["ae2_logloss_dense", "ae2_constrained_bce", "mf_bce_dense"])
def test_init_model_matches_reference_state(name):
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  g = Golden(name)
  c = g.cfg
  torch.manual_seed(1234)
  if c["kind"] == "ae":
    m = DynamicAutoencoder(hidden_layers=c["hidden_layers"], activation_type=c["activation_type"],
                           is_constrained=c.get("is_constrained", False), sparse=c.get("sparse", False))
  else:
    m = MatrixFactorization(embedding_size=c["embedding_size"], activation_type=c["activation_type"])
  m.init_model(g.csr.shape[1], g.csr.shape[0])
  init = g.state("init")
  got = dict(m.named_parameters())
  assert list(got) == list(init)
  for k, v in init.items():
    assert torch.equal(got[k].detach(), v), k
  # state-dict keys of the reference, including the aliased, name-mangled entries
  keys = set(m.state_dict())
  if c["kind"] == "ae":
    assert "_DynamicAutoencoder__en_linear_embedding_layer.embedding_layer.weight" in keys
    assert "_DynamicAutoencoder__de_linear_embedding_layer.bias" in keys
  assert set(m.model_params()) == ({"hidden_layers", "activation_type", "is_constrained",
                                    "dropout_prob", "noise_prob"} if c["kind"] == "ae" else
                                   {"embedding_size", "activation_type", "dropout_prob"})


def test_dataset_and_loader_contracts():
  from recoder_amd.data import RecommendationDataLoader, RecommendationDataset
  rng = np.random.RandomState(0)
  m = sp.random(23, 40, density=0.2, random_state=rng, format="csr", dtype=np.float32)
  ds = RecommendationDataset(m, m)
  assert len(ds) == 23 and len(ds.items) == 40 and len(ds.users) == 23
  inp, tgt = ds[[3, 5, 7]]
  assert list(inp.users) == [3, 5, 7] and inp.interactions_matrix.shape == (3, 40)
  assert np.array_equal(inp.users, tgt.users)
  one, none = RecommendationDataset(m)[4]
  assert none is None and one.interactions_matrix.shape[0] == 1
  dl = RecommendationDataLoader(ds, batch_size=5, negative_sampling=True, num_sampling_users=10)
  assert len(dl) == 5
  with pytest.raises(AssertionError):
    RecommendationDataLoader(ds, batch_size=5, num_sampling_users=3)
  # identity collate_fn (the evaluator's use, metrics.py:167): groups of raw UsersInteractions
  torch.manual_seed(0)
  seen = []
  for a, b in RecommendationDataLoader(ds, batch_size=4, collate_fn=lambda _: _):
    assert np.array_equal(a.users, b.users)
    seen += list(a.users)
  assert sorted(seen) == list(range(23))


def test_recoder_error_behaviour_before_any_kernel():
  """Same exceptions as model.py:96-99,141-156,175,310-311 (raised on the host)."""
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  m = DynamicAutoencoder(hidden_layers=[8])
  r = Recoder(m, loss="nope", optimizer_type="adam")
  with pytest.raises(ValueError):
    r._Recoder__init_loss_module()
  r = Recoder(m, loss=None, optimizer_type="adam")
  with pytest.raises(ValueError):
    r._Recoder__init_loss_module()
  with pytest.raises(Exception):
    Recoder(m).init_from_model_file("/nonexistent/file.model")
  csr = sp.identity(8, dtype=np.float32, format="csr")
  from recoder_amd.data import RecommendationDataset
  with pytest.raises(AssertionError):
    Recoder(m, optimizer_type="adam").train(RecommendationDataset(csr), batch_size=4, num_sampling_users=6)


def test_batch_metrics_equal_the_per_user_functions():
  """metrics.batch_metrics (what RecommenderEvaluator uses for the built-in metrics) against the
  per-user functions of the reference's arithmetic, incl. users without targets (nan), k beyond the
  list length, zero-valued stored entries, un-normalised variants; a user-defined metric or ragged
  lists fall back to the per-user loop."""
  import scipy.sparse as sp
  from recoder_amd import metrics as M
  rng = np.random.RandomState(0)
  B, n_items, K = 57, 300, 25
  recs = np.stack([rng.permutation(n_items)[:K] for _ in range(B)])
  dens = rng.rand(B, n_items) < 0.06
  dens[5] = False                                   # a user without targets
  vals = np.where(dens, rng.choice([0.0, 1.0, 3.0], size=dens.shape, p=[0.1, 0.6, 0.3]), 0.0)
  tm = sp.csr_matrix(vals)
  tm_explicit_zeros = sp.csr_matrix((np.where(dens, vals, 0.0)[dens], np.nonzero(dens)), shape=dens.shape)
  ms = [M.Recall(20), M.Recall(5, normalize=False), M.NDCG(10), M.NDCG(100), M.AveragePrecision(20),
        M.AveragePrecision(7, normalize=False)]
  for t in (tm, tm_explicit_zeros):
    got = M.batch_metrics(recs.tolist(), t, ms)
    assert got is not None
    t = t.tocsr()
    with np.errstate(divide="ignore", invalid="ignore"):
      for i in range(B):
        lo, hi = t.indptr[i], t.indptr[i + 1]
        y = t.indices[lo:hi][t.data[lo:hi] != 0]
        for m in ms:
          want = m.evaluate(recs[i], y)
          have = got[m][i]
          assert (np.isnan(want) and np.isnan(have)) or abs(want - have) < 1e-12, (str(m), i, want, have)

  class Mine(M.Metric):
    def evaluate(self, x, y):
      return 1.0
  assert M.batch_metrics(recs.tolist(), tm, [M.Recall(5), Mine("mine")]) is None
  ragged = [list(r[:K - (i % 2)]) for i, r in enumerate(recs)]
  assert M.batch_metrics(ragged, tm, [M.Recall(5)]) is None


def test_epoch_order_is_drawn_on_one_thread_and_unchanged_by_it():
  """The permutation of an epoch is torch.randperm under a private generator (reference
  data.py:124-136); ours runs it with torch's intra-op team reduced to the calling thread (the
  team, sized by the host's cores, gets the whole process throttled inside a CPU-quota container)
  and must restore the setting and draw the same permutation."""
  import torch
  from recoder_amd.data import epoch_user_order
  n0 = torch.get_num_threads()
  torch.manual_seed(11)
  got = epoch_user_order(50000)
  assert torch.get_num_threads() == n0
  torch.manual_seed(11)
  torch.empty((), dtype=torch.int64).random_()
  seed = int(torch.empty((), dtype=torch.int64).random_().item())
  g = torch.Generator()
  g.manual_seed(seed)
  want = torch.randperm(50000, generator=g).numpy()
  assert np.array_equal(got, want)


def test_engine_choice_by_dataset_and_loss_params():
  """Combinations without a fused HIP step are routed to the generic (torch autograd, GPU) engine
  instead of raising: tied weights with a separate target matrix (reference model.py:464-476,
  nn.py:191-202), and named losses whose loss_params go beyond what the fused epilogues implement
  (model.py:87-99 builds BCEWithLogitsLoss(reduction='sum', **loss_params))."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  m = sp.random(30, 20, density=0.2, format="csr", random_state=0, dtype=np.float32)
  m.data[:] = 1.0
  tied = DynamicAutoencoder(hidden_layers=[8], is_constrained=True)
  rec = Recoder(model=tied, use_cuda=True, optimizer_type="adam", loss="mse")
  assert not rec._use_generic()
  rec._pick_engine_for(RecommendationDataset(m, m))
  assert rec._use_generic() and "target matrix" in rec._force_generic
  rec._pick_engine_for(RecommendationDataset(m))          # back to the fused engine
  assert not rec._use_generic()
  plain = DynamicAutoencoder(hidden_layers=[8])
  rec2 = Recoder(model=plain, use_cuda=True, optimizer_type="adam", loss="mse")
  rec2._pick_engine_for(RecommendationDataset(m, m))      # untied + target: fused
  assert not rec2._use_generic()
  assert not Recoder(model=plain, optimizer_type="adam", loss="mse", loss_params={"confidence": 2})._use_generic()
  assert Recoder(model=plain, optimizer_type="adam", loss="logistic",
                 loss_params={"pos_weight": torch.ones(20)})._use_generic()
  assert Recoder(model=plain, optimizer_type="adam", loss="logistic",
                 loss_params={"weight": torch.ones(20)})._use_generic()
