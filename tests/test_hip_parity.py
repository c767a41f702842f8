"""GPU parity tests: the HIP path (through the C ABI / ctypes) against
  (a) the committed golden vectors of the REAL reference (tests/golden/*.npz),
  (b) the CPU oracle (oracle/recoder_oracle.py) on seeded synthetic inputs.

Bars: integer / index work (collation) is bit-exact; floating point is within
1e-5 relative for losses (BASELINE.json north_star) -- gradients/parameters are
compared with rtol 1e-4 + a small atol because the reference accumulates in a
different (MKL) order.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import recoder_oracle as orc
from tests.golden_util import CONFIGS, Golden

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5


def dev():
  return torch.device("cuda")


def synth_csr(n_users, n_items, mean_deg, seed, ratings=False):
  rng = np.random.RandomState(seed)
  pop = 1.0 / np.arange(1, n_items + 1)
  pop /= pop.sum()
  deg = np.clip(rng.lognormal(np.log(mean_deg) - 0.5, 1.0, n_users).astype(int), 1, n_items // 4)
  rows = np.repeat(np.arange(n_users), deg)
  cols = rng.choice(n_items, size=int(deg.sum()), p=pop)
  vals = rng.randint(1, 6, size=len(cols)).astype(np.float32) if ratings else np.ones(len(cols), np.float32)
  m = sp.coo_matrix((vals, (rows, cols)), shape=(n_users, n_items)).tocsr()
  m.sum_duplicates()
  if not ratings:
    m.data[:] = 1.0
  m.sort_indices()
  return m


# --------------------------------------------------------------------------
# collation: bit-exact against np.unique semantics (oracle == reference)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n_users,n_items,S,ns", [
  (150, 120, 64, True), (150, 120, 150, True), (300, 5000, 128, True),
  (100, 90, 32, False), (2000, 20000, 500, True), (50, 3000, 1, True),
])
def test_collate_bit_exact(n_users, n_items, S, ns):
  from recoder_amd.device import Block, DeviceCSR
  csr = synth_csr(n_users, n_items, 12, seed=n_items + S, ratings=True)
  dcsr = DeviceCSR(csr)
  rng = np.random.RandomState(5)
  blk = Block(S, int(np.sort(np.diff(csr.indptr))[-S:].sum()), n_items, negative_sampling=ns)
  for rep in range(3):
    users = rng.permutation(n_users)[:S].astype(np.int64)
    blk.collate(dcsr, torch.from_numpy(users).to(dev()))
    h = blk.to_host()
    ref = orc.collate(orc.extract_rows(csr, users), users, S, ns)[0]
    assert h["S"] == len(users)
    assert h["nnz"] == ref.indices.shape[1]
    rows = np.repeat(np.arange(len(users)), np.diff(h["indptr"]))
    assert np.array_equal(rows, ref.indices[0])
    assert np.array_equal(h["cols"], ref.indices[1])
    assert np.array_equal(h["vals"], ref.values)
    if ns:
      assert np.array_equal(h["items"], ref.items)
      assert h["n_b"] == len(ref.items)
      pos = np.full(n_items, -1)
      pos[ref.items] = np.arange(len(ref.items))
      assert np.array_equal(h["pos"], pos)
    else:
      assert h["n_b"] == n_items
    # bitmaps: every stored (r,c) and nothing else
    n_b = h["n_b"]
    bits = blk.bits_rc.cpu().numpy().view(np.uint32).reshape(blk.S_cap, blk.ldw_rc)
    dense = np.zeros((len(users), n_b), dtype=bool)
    dense[ref.indices[0], ref.indices[1]] = True
    got = np.unpackbits(bits[:len(users)].view(np.uint8), axis=1, bitorder="little")[:, :n_b].astype(bool)
    assert np.array_equal(got, dense)
    bits_t = blk.bits_cr.cpu().numpy().view(np.uint32).reshape(-1, blk.ldw_cr)
    got_t = np.unpackbits(bits_t[:n_b].view(np.uint8), axis=1, bitorder="little")[:, :len(users)].astype(bool)
    assert np.array_equal(got_t, dense.T)


@pytest.mark.parametrize("n_items", [300, 40000])        # one-workgroup scan / chunked scan
def test_collate_overflow_is_clamped_in_bounds_and_raises(n_items):
  """A block sized below its item set (never by construction): rk_collate truncates
  in bounds, leaves the true count in counts[5], and the host raises when it looks."""
  from recoder_amd._lib import RecoderHipError
  from recoder_amd.device import Block, DeviceCSR
  csr = synth_csr(200, n_items, 12, seed=3, ratings=True)
  dcsr = DeviceCSR(csr)
  S = 64
  users = np.arange(S, dtype=np.int64)
  ref = orc.collate(orc.extract_rows(csr, users), users, S, True)[0]
  n_true = len(ref.items)
  nnz_cap = int(np.sort(np.diff(csr.indptr))[-S:].sum())
  blk = Block(S, nnz_cap, n_items, negative_sampling=True, n_cap=n_true - 17)
  guard = torch.full((4096,), 12345, dtype=torch.int32, device=dev())       # something to trample on
  blk.collate(dcsr, torch.from_numpy(users).to(dev()))
  torch.cuda.synchronize()
  c = blk.counts.cpu().numpy()
  assert c[0] == n_true - 17 and c[5] == n_true
  assert np.array_equal(blk.items[:c[0]].cpu().numpy(), ref.items[:c[0]])
  assert int(blk.cols[:c[1]].min()) >= 0 and int(blk.cols[:c[1]].max()) < c[0]
  assert bool((guard == 12345).all())
  with pytest.raises(RecoderHipError):
    blk.counts_host()
  with pytest.raises(RecoderHipError):
    blk.check()
  # too few slots for the stored interactions (B copies of one heavy user did this to the padding
  # blocks of the graph stepper): row ranges clamped into the arrays, flagged in counts[6]
  heavy = int(np.argmax(np.diff(csr.indptr)))
  users_h = np.full(S, heavy, dtype=np.int64)
  blk3 = Block(S, nnz_cap, n_items, negative_sampling=True)
  guard3 = torch.full((4096,), 777, dtype=torch.int32, device=dev())
  if int(np.diff(csr.indptr)[heavy]) * S > nnz_cap:
    blk3.collate(dcsr, torch.from_numpy(users_h).to(dev()))
    torch.cuda.synchronize()
    c3 = blk3.counts.cpu().numpy()
    assert c3[6] == int(np.diff(csr.indptr)[heavy]) * S and c3[1] == nnz_cap
    assert int(blk3.indptr[:S + 1].max()) <= nnz_cap and bool((guard3 == 777).all())
    with pytest.raises(RecoderHipError):
      blk3.check()
  # the same rows in a block that is large enough: clean
  blk2 = Block(S, nnz_cap, n_items, negative_sampling=True)
  blk2.collate(dcsr, torch.from_numpy(users).to(dev()))
  assert blk2.counts_host()[0] == n_true
  blk2.check()


def test_batch_collator_api_matches_oracle():
  """The reference-style host API (BatchCollator.collate -> list[Batch])."""
  from recoder_amd.data import BatchCollator, RecommendationDataset
  csr = synth_csr(100, 200, 10, seed=3, ratings=True)
  ds = RecommendationDataset(csr)
  for B in (1, 2, 5, 10, 13):
    big, _ = ds[np.arange(len(ds))]
    batches = BatchCollator(batch_size=B, negative_sampling=True).collate(big)
    ref = orc.collate(csr, np.arange(len(ds)), B, True)
    assert len(batches) == int(np.ceil(len(ds) / B)) == len(ref)
    for b, r in zip(batches, ref):
      assert np.array_equal(b.indices.numpy(), r.indices)
      assert np.array_equal(b.values.numpy(), r.values)
      assert np.array_equal(b.items.numpy(), r.items)
      assert tuple(b.size) == tuple(r.size)
      assert np.array_equal(b.users.numpy(), r.users)


# --------------------------------------------------------------------------
# helpers to stand a Recoder up on given state
# --------------------------------------------------------------------------
def make_model(c):
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  if c["kind"] == "ae":
    return DynamicAutoencoder(hidden_layers=c["hidden_layers"], activation_type=c["activation_type"],
                              is_constrained=c.get("is_constrained", False),
                              dropout_prob=c.get("dropout_prob", 0.0),
                              noise_prob=c.get("noise_prob", 0.0), sparse=c.get("sparse", False))
  return MatrixFactorization(embedding_size=c["embedding_size"], activation_type=c["activation_type"],
                             dropout_prob=c.get("dropout_prob", 0), sparse=c.get("sparse", False))


def make_oracle(c, state):
  return orc.OracleRecoder(
      c["kind"], state, hidden_layers=c.get("hidden_layers"),
      activation_type=c.get("activation_type"), is_constrained=c.get("is_constrained", False),
      noise_prob=c.get("noise_prob", 0.0), dropout_prob=c.get("dropout_prob", 0.0),
      sparse=c.get("sparse", False), loss=c["loss"], loss_params=c["loss_params"],
      lr=c["lr"], weight_decay=c["weight_decay"])


# Post-step parameters against the reference's: Adam turns a gradient g into a step of size
# lr * m / (sqrt(v) + eps) ~ lr * sign(g) -- for the handful of elements whose gradient is near
# zero (|g| ~ eps) a 1e-7 relative difference of the GEMM output moves m / sqrt(v) by O(1), i.e.
# the parameter by up to lr = 1e-3 ABSOLUTE, whatever its size; everywhere else the parameters
# agree to rounding.  So: all but TIGHT_FRAC of the elements within 1e-5 relative (+2e-7), and
# the worst relative error of the elements with |w| > 1e-3 bounded by TIGHT_REL.
# Measured over the eight golden replays (round 3): every bias, hidden layer, MF table and most
# embedding tables: ALL elements within 1e-5 (worst 3.3e-6 relative where |w| > 1e-3); two embedding
# tables have 3.5e-4 / 6.9e-4 of their elements beyond it, worst 1.6e-5 / 5.2e-5 relative.
# Round 4: the bounds are what was measured plus a small margin, not a round number above it -- the two
# autoencoder embedding tables get 8e-4 / 6e-5; EVERY other tensor (biases, hidden layers, the
# MatrixFactorization tables) must meet north_star's 1e-5 on all of its elements, so that a
# regression of the split arithmetic cannot hide inside the slack.
TIGHT_FRAC = float(os.environ.get("RK_TIGHT_FRAC", "8e-4"))
TIGHT_REL = float(os.environ.get("RK_TIGHT_REL", "6e-5"))
AE_TABLES = ("en_embedding_layer.weight", "de_embedding_layer.weight")


def tight_bounds(name, kind):
  """(max fraction of elements beyond 1e-5 relative, max relative error where |w| > 1e-3) of a tensor."""
  if kind == "ae" and any(name.endswith(t) for t in AE_TABLES):
    return TIGHT_FRAC, TIGHT_REL
  return 0.0, 1e-5


def close_stats(a, b, rtol, atol):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  err = np.abs(a - b)
  tol = atol + rtol * np.abs(b)
  bad = err > tol
  return float(bad.mean()), float(err.max()), float(np.abs(b).max())


def tight_stats(a, b):
  """north_star's "1e-5 relative" applied to post-step PARAMETERS: (fraction of elements beyond
  1e-5 relative + 2e-7 absolute, max relative error over the elements with |w| > 1e-3)."""
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  err = np.abs(a - b)
  bad = err > 2e-7 + 1e-5 * np.abs(b)
  big = np.abs(b) > 1e-3
  mx_rel = float((err[big] / np.abs(b[big])).max()) if big.any() else 0.0
  return float(bad.mean()), mx_rel


# --------------------------------------------------------------------------
# golden replay: full Recoder.train() with injected user order + masks
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CONFIGS))
def test_train_replays_reference_golden(name):
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  g = Golden(name)
  c = g.cfg
  torch.manual_seed(1234)
  model = make_model(c)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=c["loss"],
                loss_params=c["loss_params"])
  spe = g.steps_per_epoch()
  n_epochs = g.nsteps // spe

  def order_hook(epoch, n):
    return np.concatenate([g.step(i)["users"] for i in range((epoch - 1) * spe, epoch * spe)])

  def mask_hook(step, users):
    # the group starting at global step `step`
    grp = [gg for gg in g.groups() if gg[0] == step][0]
    nk = [g.step(i)["noise_keep"] for i in grp]
    dk = [g.step(i)["drop_keep"] for i in grp]
    nk = None if nk[0] is None else torch.from_numpy(np.concatenate(nk)).to(dev())
    dk = None if dk[0] is None else torch.from_numpy(np.concatenate(dk, axis=0)).to(dev())
    return nk, dk

  rec.user_order_hook = order_hook
  rec.mask_hook = mask_hook
  ds = RecommendationDataset(g.csr)
  # the model is initialised inside train(); it draws from the torch RNG like the
  # reference (seed 1234), so the initial state equals the golden init state
  rec.train(ds, batch_size=c["batch_size"], lr=c["lr"], weight_decay=c["weight_decay"],
            num_epochs=n_epochs, negative_sampling=c["negative_sampling"],
            num_sampling_users=c.get("num_sampling_users", 0),
            lr_milestones=c.get("lr_milestones"))
  losses = rec.loss_history
  losses = np.concatenate(losses)
  assert len(losses) == g.nsteps
  rel = np.abs(losses - g.losses) / np.abs(g.losses)
  print(name, "max rel loss err", rel.max())
  assert rel.max() < LOSS_RTOL, (rel.argmax(), rel.max())
  final = g.state("final")
  sd = {k: v.detach().cpu() for k, v in model.named_parameters()}
  for k, v in final.items():
    frac, mx, scale = close_stats(sd[k].numpy(), v.numpy(), 1e-4, 2e-6)
    tfrac, trel = tight_stats(sd[k].numpy(), v.numpy())
    print("  ", k, "bad frac %.2e max err %.3e (scale %.3e) | beyond 1e-5 rel: %.2e of elements, "
          "max rel err where |w| > 1e-3: %.2e" % (frac, mx, scale, tfrac, trel))
    assert frac < 2e-3, (k, frac, mx)
    assert mx < 5e-3 * max(1.0, scale), (k, mx)
    # north_star's tolerance on the parameters themselves (TIGHT_FRAC / TIGHT_REL: see the note there)
    bfrac, brel = tight_bounds(k, c["kind"])
    assert tfrac <= bfrac if bfrac == 0.0 else tfrac < bfrac, (k, tfrac, bfrac)
    assert trel < brel, (k, trel, brel)


def test_model_init_matches_reference_golden():
  """init_model consumes the torch RNG in the reference's order."""
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization  # noqa
  for name in ("ae2_logloss_dense", "ae2_constrained_bce", "mf_bce_dense"):
    g = Golden(name)
    torch.manual_seed(1234)
    m = make_model(g.cfg)
    m.init_model(g.csr.shape[1], g.csr.shape[0])
    init = g.state("init")
    got = dict(m.named_parameters())
    assert list(got.keys()) == list(init.keys())
    for k, v in init.items():
      assert torch.equal(got[k].detach().cpu(), v), (name, k)


# --------------------------------------------------------------------------
# evaluation path against the golden top-k / Recall / NDCG / scores
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", [n for n, c in CONFIGS.items() if c.get("evaluate")])
def test_eval_matches_reference_golden(name):
  from recoder_amd.data import RecommendationDataset, UsersInteractions
  from recoder_amd.metrics import NDCG, Recall
  from recoder_amd.model import Recoder
  g = Golden(name)
  c = g.cfg
  model = make_model(c)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=c["loss"],
                loss_params=c["loss_params"], num_items=g.csr.shape[1], num_users=g.csr.shape[0])
  rec._Recoder__init_model()
  model.load_state_dict({k: v for k, v in g.state("final").items()}, strict=False)
  users = np.arange(g.csr.shape[0])
  out, _ = rec.predict(UsersInteractions(users[:8], g.csr[users[:8]]))
  frac, mx, scale = close_stats(out.cpu().numpy(), g.z["eval/scores8"], 1e-4, 1e-6)
  assert frac == 0.0, (frac, mx)
  recs = []
  for off in range(0, len(users), 50):
    u = users[off:off + 50]
    recs += rec.recommend(UsersInteractions(u, g.csr[u]), 20)
  recs = np.asarray(recs)
  gold = g.z["eval/topk"]
  same = (recs == gold).mean()
  print(name, "top-k positions identical: %.4f" % same)
  assert same > 0.99
  # index work is exact given equal scores: a position may only differ from the reference's where
  # the two items' scores are tied to rounding (1e-5 relative)
  for i in np.nonzero((recs != gold).any(axis=1))[0]:
    sc = rec.predict(UsersInteractions(users[i:i + 1], g.csr[users[i:i + 1]]))[0][0].cpu().numpy()
    for p in np.nonzero(recs[i] != gold[i])[0]:
      a, b = sc[recs[i][p]], sc[gold[i][p]]
      assert abs(a - b) <= 1e-5 * max(abs(a), abs(b), 1e-30), (i, p, a, b)
  # the strip-wise path (catalogue decoded 40 items at a time, winners merged) gives the same lists
  rec.eval_strip_items = 40
  recs_s = []
  for off in range(0, len(users), 50):
    u = users[off:off + 50]
    recs_s += rec.recommend(UsersInteractions(u, g.csr[u]), 20)
  assert np.array_equal(np.asarray(recs_s), recs)
  r20, r5, n20 = Recall(20), Recall(5), NDCG(20)
  v = {"recall20": [], "recall5": [], "ndcg20": []}
  for i, u in enumerate(users):
    y = g.csr_te[u].nonzero()[1]
    v["recall20"].append(r20.evaluate(recs[i], y))
    v["recall5"].append(r5.evaluate(recs[i], y))
    v["ndcg20"].append(n20.evaluate(recs[i], y))
  for k in v:
    assert abs(np.mean(v[k]) - float(g.z["eval/" + k])) < 5e-5, (k, np.mean(v[k]), float(g.z["eval/" + k]))


# --------------------------------------------------------------------------
# one step vs the oracle on seeded synthetic data at larger shapes:
# loss, every gradient buffer, post-step parameters
# --------------------------------------------------------------------------
STEP_CASES = [
  # name, cfg, (n_users, n_items, mean_deg), B, S
  ("ae200_mse", dict(kind="ae", hidden_layers=[200], activation_type="tanh", noise_prob=0.5,
                     sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (1500, 3000, 30), 500, 500),
  ("ae200_mse_sparse_conf", dict(kind="ae", hidden_layers=[200], activation_type="tanh", noise_prob=0.0,
                                 sparse=True, loss="mse", loss_params=dict(confidence=2.5), lr=1e-3,
                                 weight_decay=2e-5),
   (900, 2500, 25), 300, 600),
  ("ae512_bce", dict(kind="ae", hidden_layers=[512], activation_type="sigmoid", noise_prob=0.3,
                     sparse=True, loss="logistic", loss_params=None, lr=1e-3, weight_decay=0.0),
   (700, 4000, 20), 257, 257),
  ("ae64_32_logloss", dict(kind="ae", hidden_layers=[64, 32], activation_type="tanh", noise_prob=0.5,
                           dropout_prob=0.2, sparse=False, loss="logloss", loss_params=None, lr=1e-3,
                           weight_decay=2e-5),
   (600, 1200, 20), 200, 200),
  ("ae128_72_40_constrained", dict(kind="ae", hidden_layers=[128, 72, 40], activation_type="selu",
                                   noise_prob=0.2, is_constrained=True, sparse=False, loss="mse",
                                   loss_params=None, lr=1e-3, weight_decay=1e-5),
   (500, 900, 15), 130, 130),
  ("ae200_relu_nosampling", dict(kind="ae", hidden_layers=[200], activation_type="relu", noise_prob=0.0,
                                 sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5,
                                 negative_sampling=False),
   (300, 700, 15), 100, 100),
  ("mf128_mse_sparse", dict(kind="mf", embedding_size=128, activation_type="none", sparse=True,
                            loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (1000, 2000, 25), 500, 500),
  ("mf64_bce_dense_drop", dict(kind="mf", embedding_size=64, activation_type="tanh", dropout_prob=0.3,
                               sparse=False, loss="logistic", loss_params=None, lr=1e-3,
                               weight_decay=2e-5),
   (800, 1500, 25), 256, 256),
  # operand ranges of the split-fp16 decoder GEMMs follow the data (include/recoder_hip.h rk_amax):
  # unbounded activation with |Z| ~ 1e4 and |W_de| ~ 1e3 -- far outside the static range
  # (|Z| < 2048, |W| < 512), no environment variable involved
  ("range_relu_big", dict(kind="ae", hidden_layers=[200], activation_type="relu", noise_prob=0.0,
                          sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5,
                          prescale=dict(z_max=1.0e4, w_max=1.0e3)),
   (900, 2500, 25), 300, 300),
  ("range_relu_big_sparse_2l", dict(kind="ae", hidden_layers=[64, 40], activation_type="relu",
                                    noise_prob=0.0, sparse=True, loss="mse", loss_params=None, lr=1e-3,
                                    weight_decay=0.0, prescale=dict(z_max=3.0e4, w_max=2.0e3)),
   (500, 1200, 20), 128, 256),
  # edge shapes (names starting with "edge": every 5th user has NO interactions): fewer than 32
  # sampled items, a ragged last batch (37 users in batches of 16), one-row batches
  ("edge_tiny_ae8", dict(kind="ae", hidden_layers=[8], activation_type="tanh", noise_prob=0.0,
                         sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (37, 40, 4), 16, 16),
  ("edge_tiny_ae8_bce_sparse", dict(kind="ae", hidden_layers=[8], activation_type="relu", noise_prob=0.0,
                                    sparse=True, loss="logistic", loss_params=None, lr=1e-3,
                                    weight_decay=0.0),
   (37, 40, 4), 7, 21),
  ("edge_b1_mf4", dict(kind="mf", embedding_size=4, activation_type="tanh", sparse=False,
                       loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (9, 33, 3), 1, 1),
  # a batch of >= 1024 rows (128-row decode tiles, stand-alone dZ kernel) followed by a ragged last one
  # below 1024 (the fused decode + dZ launch, in the workspace the engine sized for the big batch)
  ("big_then_ragged_ae64", dict(kind="ae", hidden_layers=[64], activation_type="tanh", noise_prob=0.3,
                                sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (1400, 900, 12), 1100, 1100),
  ("big_then_ragged_mf16", dict(kind="mf", embedding_size=16, activation_type="none", sparse=True,
                                loss="logistic", loss_params=None, lr=1e-3, weight_decay=0.0),
   (1300, 700, 10), 1030, 1030),
  # the dZ slab count is not monotone in the batch size -- a ragged last batch of 1024 rows takes
  # 64 slabs (65 536 slab rows) where the capacity batch of 1100 takes 51 (56 100); few items, so that no other
  # workspace happens to cover it (h = 512: rk_pg_dz; h = 64: the fused decode's domain ends at 1024 rows)
  ("ragged_1024_after_1100_ae512", dict(kind="ae", hidden_layers=[512], activation_type="tanh", noise_prob=0.0,
                                        sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (2124, 300, 8), 1100, 1100),
  # HEAVY rows (lognormal degrees around 150: rows of 300 ... 1500 stored interactions, several 64-entry rounds per
  # wave of the encoder forward; explicit ratings: the norm over the whole row; implicit + noise: the counter RNG)
  ("heavy_rows_ae200", dict(kind="ae", hidden_layers=[200], activation_type="tanh", noise_prob=0.5,
                            sparse=False, loss="mse", loss_params=None, lr=1e-3, weight_decay=2e-5),
   (600, 6000, 150), 200, 200),
  ("heavy_rows_ae64_bce", dict(kind="ae", hidden_layers=[64], activation_type="relu", noise_prob=0.3,
                               sparse=True, loss="logistic", loss_params=None, lr=1e-3, weight_decay=0.0),
   (500, 5000, 200), 128, 128),
  # h > 512 at >= 1024 rows: 256 x 256 dW tiles; the merged dW || encoder-backward instantiation spilled there
  #  -- two launches now
  ("big_ae640", dict(kind="ae", hidden_layers=[640], activation_type="tanh", noise_prob=0.0,
                     sparse=True, loss="mse", loss_params=None, lr=1e-3, weight_decay=0.0),
   (1300, 400, 10), 1100, 1100),
  ("ragged_1024_after_1100_ae64", dict(kind="ae", hidden_layers=[64], activation_type="tanh", noise_prob=0.0,
                                       sparse=True, loss="logistic", loss_params=None, lr=1e-3, weight_decay=0.0),
   (2124, 300, 8), 1100, 1100),
]


def _fuzz_cases(n=20, seed=2024):
  """Deterministic random shapes around the tiling boundaries (rows 64/128/512, items 32/64/128,
  h 4..512) x losses x Adam kinds x tied / hidden stacks."""
  rng = np.random.RandomState(seed)
  out = []
  for i in range(n):
    kind = "mf" if i % 5 == 4 else "ae"
    B = int(rng.choice([1, 3, 31, 64, 65, 127, 128, 129, 200, 255, 256, 300, 511, 512, 513, 700]))
    n_items = int(rng.choice([33, 64, 65, 127, 129, 300, 1000, 2500]))
    n_users = max(B + 7, int(B * rng.uniform(1.1, 2.5)))
    deg = int(rng.choice([2, 5, 12, 30]))
    loss = str(rng.choice(["mse", "logistic", "logloss"]))
    sparse = bool(rng.rand() < 0.4)
    if kind == "ae":
      h0 = int(rng.choice([4, 8, 36, 64, 100, 200, 256, 260, 512]))
      layers = [h0] if rng.rand() < 0.6 else [h0, int(rng.choice([8, 24, 40]))]
      c = dict(kind="ae", hidden_layers=layers, activation_type=str(rng.choice(["tanh", "relu", "sigmoid"])),
               noise_prob=float(rng.choice([0.0, 0.3])), dropout_prob=float(rng.choice([0.0, 0.0, 0.25])),
               is_constrained=bool(rng.rand() < 0.25), sparse=sparse, loss=loss,
               loss_params=(dict(confidence=2) if loss == "mse" and rng.rand() < 0.5 else None),
               lr=1e-3, weight_decay=(0.0 if sparse else 2e-5))
    else:
      c = dict(kind="mf", embedding_size=int(rng.choice([4, 16, 128])),
               activation_type=str(rng.choice(["none", "tanh"])), dropout_prob=float(rng.choice([0.0, 0.2])),
               sparse=sparse, loss=loss, loss_params=None, lr=1e-3, weight_decay=(0.0 if sparse else 2e-5))
    S = B * int(rng.choice([1, 1, 2]))
    out.append(("fuzz%02d_%s_%s_B%d_n%d" % (i, kind, loss, B, n_items), c, (n_users, n_items, deg), B, S))
  return out


STEP_CASES += _fuzz_cases(40)


@pytest.mark.parametrize("name,c,shape,B,S", STEP_CASES, ids=[x[0] for x in STEP_CASES])
def test_steps_match_oracle(name, c, shape, B, S):
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  n_users, n_items, deg = shape
  csr = synth_csr(n_users, n_items, deg, seed=len(name), ratings=(c["loss"] == "mse"))
  if name.startswith("edge"):
    keep = np.ones(n_users, dtype=bool)
    keep[::5] = False
    csr = sp.diags(keep.astype(np.float32)).dot(csr).tocsr()
    csr.eliminate_zeros()
    csr.sort_indices()
  ns = c.get("negative_sampling", True)
  torch.manual_seed(7)
  model = make_model(c)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=c["loss"],
                loss_params=c["loss_params"])
  ds = RecommendationDataset(csr)
  rng = np.random.RandomState(11)
  order = rng.permutation(n_users)[: 3 * S].astype(np.int64)
  n_steps = sum(int(np.ceil(len(order[o:o + S]) / B)) for o in range(0, len(order), S))
  masks = {}
  h_last = c["hidden_layers"][-1] if c["kind"] == "ae" else c["embedding_size"]

  def mask_hook(step, users):
    rows = csr[users]
    nk = (rng.rand(rows.nnz) >= c.get("noise_prob", 0.0)).astype(np.uint8) \
        if c.get("noise_prob", 0.0) > 0 else None
    dk = (rng.rand(len(users), h_last) >= c.get("dropout_prob", 0.0)).astype(np.uint8) \
        if c.get("dropout_prob", 0.0) > 0 else None
    masks[step] = (nk, dk)
    return (None if nk is None else torch.from_numpy(nk).to(dev()),
            None if dk is None else torch.from_numpy(dk).to(dev()))

  rec.user_order_hook = lambda epoch, n: order
  rec.mask_hook = mask_hook
  # capture the initial state right after init: train with 0 iterations is not
  # possible, so initialise explicitly
  rec._Recoder__init_training(ds, c["lr"], c["weight_decay"])
  if c.get("prescale"):
    # blow the operands of the decoder GEMMs up: scale the encoder table until the bottleneck
    # reaches z_max on the first batch, the decoder table until its largest entry is w_max
    ps = c["prescale"]
    with torch.no_grad():
      pr = dict(model.named_parameters())
      pr[orc.AE_DE_W].mul_(ps["w_max"] / pr[orc.AE_DE_W].abs().max().item())
      o0 = make_oracle(c, {k: v.detach().cpu().clone() for k, v in pr.items()})
      b0 = orc.collate(orc.extract_rows(csr, order[:S]), order[:S], B, ns)[0]
      z0 = o0.decoder_input(b0).abs().max().item()
      pr[orc.AE_EN_W].mul_(ps["z_max"] / z0)
      for k, v in pr.items():
        if k.endswith("bias"):
          v.mul_(ps["z_max"] / z0)
  init = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
  rec.train(ds, batch_size=B, lr=c["lr"], weight_decay=c["weight_decay"], num_epochs=1,
            negative_sampling=ns, num_sampling_users=S)
  losses = rec.last_epoch_losses
  assert len(losses) == n_steps
  if c.get("prescale"):
    rg = rec._engine().ranges.view(torch.float32).cpu().numpy()
    print("   ranges: |Z| <= %.4g, |W_de| <= %.4g" % (rg[:64].max(), rg[64:].max()))
    assert rg[:64].max() > 2048 and rg[64:].max() > 512     # really outside the static range
    assert np.all(np.isfinite(losses))

  o = make_oracle(c, init)
  ref_losses = []
  step = 0
  for goff in range(0, len(order), S):
    users = order[goff:goff + S]
    batches = orc.collate(orc.extract_rows(csr, users), users, B, ns)
    nk, dk = masks[step]
    nnz_off = 0
    for bi, b in enumerate(batches):
      nnz = b.indices.shape[1]
      bnk = None if nk is None else nk[nnz_off:nnz_off + nnz]
      bdk = None if dk is None else dk[bi * B:(bi + 1) * B]
      nnz_off += nnz
      ref_losses.append(o.train_step(b, None, bnk, bdk))
      step += 1
  ref_losses = np.asarray(ref_losses)
  rel = np.abs(losses - ref_losses) / np.abs(ref_losses)
  print(name, "losses", losses[:3], "ref", ref_losses[:3], "max rel", rel.max())
  assert rel.max() < LOSS_RTOL
  # gradients of the last step (buffers are still intact after the update)
  eng = rec._engine()
  grads = o.grads()
  items = batches[-1].items
  n_b = len(items) if items is not None else n_items
  items_idx = items if items is not None else np.arange(n_items)
  h0 = eng.h[0]

  def check(label, got, want, rtol=2e-4, atol=None):
    want = np.asarray(want)
    atol = 2e-6 * max(1e-30, np.abs(want).max()) if atol is None else atol
    frac, mx, scale = close_stats(got, want, rtol, atol)
    print("   grad %-28s bad %.2e maxerr %.3e scale %.3e" % (label, frac, mx, scale))
    assert frac < 1e-3, (label, frac, mx, scale)

  G_de = eng.decoder_row_grad(n_b).cpu().numpy()
  gb_de = eng.decoder_bias_grad(n_b).cpu().numpy()
  if c["kind"] == "ae":
    if c.get("is_constrained"):
      check("W_en(tied)[items]", G_de, grads[orc.AE_EN_W][items_idx].numpy())
    else:
      check("W_de[items]", G_de, grads[orc.AE_DE_W][items_idx].numpy())
      G_en = eng.encoder_row_grad(n_b).cpu().numpy()
      check("W_en[items]", G_en, grads[orc.AE_EN_W][items_idx].numpy())
    check("b_de[items]", gb_de, grads[orc.AE_DE_B][items_idx].numpy())
    check("b_en", eng.encoder_bias_grad().cpu().numpy(), grads[orc.AE_EN_B].numpy())
    for i in range(eng.nl):
      check("enc%d.W" % i, eng.g_enc_w[i].cpu().numpy(), grads["encoding_layers.%d.weight" % i].numpy())
      check("enc%d.b" % i, eng.g_enc_b[i].cpu().numpy(), grads["encoding_layers.%d.bias" % i].numpy())
      check("dec%d.b" % i, eng.g_dec_b[i].cpu().numpy(), grads["decoding_layers.%d.bias" % i].numpy())
      if not c.get("is_constrained"):
        check("dec%d.W" % i, eng.g_dec_w[i].cpu().numpy(), grads["decoding_layers.%d.weight" % i].numpy())
  else:
    check("E_i[items]", G_de, grads["item_embedding_layer.weight"][items_idx].numpy())
    check("bias[items]", gb_de, grads["bias"][items_idx].numpy())
    ub = batches[-1].users
    check("E_u[users]", eng.dbott[:len(ub) * h0].view(len(ub), h0).cpu().numpy(),
          grads["user_embedding_layer.weight"][ub].numpy())
  # parameters after the steps
  ost = o.state()
  for k, p in model.named_parameters():
    frac, mx, scale = close_stats(p.detach().cpu().numpy(), ost[k].numpy(), 1e-4, 2e-6)
    print("   param %-50s bad %.2e maxerr %.3e" % (k, frac, mx))
    assert frac < 2e-3, (k, frac, mx)
    assert mx < 2.5 * c["lr"] * n_steps + 1e-6, (k, mx)


def test_step_is_bitwise_deterministic():
  """No atomics in any floating-point reduction: two runs from the same state
  give bit-identical parameters."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  csr = synth_csr(900, 2500, 25, seed=2)
  c = STEP_CASES[0][1]
  outs = []
  for rep in range(2):
    torch.manual_seed(3)
    model = make_model(c)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
    rec.user_order_hook = lambda epoch, n: np.arange(n, dtype=np.int64)
    rec.train(RecommendationDataset(csr), batch_size=300, lr=1e-3, weight_decay=2e-5, num_epochs=1,
              negative_sampling=True)
    outs.append(({k: v.detach().cpu().clone() for k, v in model.named_parameters()},
                 rec.last_epoch_losses.copy()))
  assert np.array_equal(outs[0][1], outs[1][1])
  for k in outs[0][0]:
    assert torch.equal(outs[0][0][k], outs[1][0][k]), k


def test_nn_forward_shape_contract():
  """reference tests/test_nn.py:15-42: layer dims and output shapes for
  input_items != target_items and the full decode."""
  from recoder_amd.nn import DynamicAutoencoder
  ae = DynamicAutoencoder([300, 200])
  ae.init_model(num_items=500)
  ae = ae.to(dev())
  assert ae.en_embedding_layer.embedding_dim == 300
  assert ae.de_embedding_layer.embedding_dim == 300
  assert len(ae.encoding_layers) == 1 and len(ae.decoding_layers) == 1
  assert ae.encoding_layers[0].weight.size(0) == 200
  assert ae.decoding_layers[0].weight.size(1) == 200
  x = torch.rand(32, 5)
  items = torch.LongTensor([10, 126, 452, 29, 34])
  out = ae(x, input_items=items, target_items=items)
  assert tuple(out.shape) == (32, 5)
  t_items = torch.LongTensor([31, 14, 95, 49, 10, 36, 239])
  out = ae(x, input_items=items, target_items=t_items)
  assert tuple(out.shape) == (32, 7)
  out_full = ae(x, input_items=items)
  assert tuple(out_full.shape) == (32, 500)
  # values against the oracle forward
  st = {k: v.detach().cpu() for k, v in ae.named_parameters()}
  o = orc.OracleRecoder("ae", st, hidden_layers=[300, 200], activation_type="tanh")
  o.training = False
  with torch.no_grad():
    want = o._ae_forward(x, items, t_items, None, None)
  frac, mx, _ = close_stats(out.cpu().numpy(), want.numpy(), 1e-4, 1e-6)
  assert frac == 0.0, (frac, mx)


# --------------------------------------------------------------------------
# validation loss (model.py:439-452): input and target collated independently
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(CONFIGS))
def test_validation_loss_matches_reference_golden(name):
  from recoder_amd.data import RecommendationDataLoader, RecommendationDataset
  from recoder_amd.model import Recoder
  g = Golden(name)
  c = g.cfg
  model = make_model(c)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=c["loss"],
                loss_params=c["loss_params"], num_items=g.csr.shape[1], num_users=g.csr.shape[0])
  rec._Recoder__init_model()
  model.load_state_dict({k: v for k, v in g.state("final").items()}, strict=False)
  nb = int(g.z["val/nbatches"])
  order = np.concatenate([g.z["val%d/users" % i] for i in range(nb)])
  rec.user_order_hook = lambda epoch, n: order
  ds = RecommendationDataset(g.csr, g.csr_te)
  dl = RecommendationDataLoader(ds, batch_size=c["batch_size"], negative_sampling=c["negative_sampling"],
                                num_sampling_users=c.get("num_sampling_users", 0))
  val = rec._validate(dl)
  want = float(g.z["val/loss"])
  assert abs(val - want) / abs(want) < LOSS_RTOL, (val, want)


# --------------------------------------------------------------------------
# checkpoint round trip with the reference's dict layout (model.py:166-224)
# --------------------------------------------------------------------------
def test_checkpoint_roundtrip_reference_layout(tmp_path):
  from recoder_amd.data import RecommendationDataset, UsersInteractions
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  g = Golden("ae_mse_dense")
  c = g.cfg
  torch.manual_seed(5)
  model = make_model(c)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
  ds = RecommendationDataset(g.csr)
  rec.train(ds, batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True,
            model_checkpoint_prefix=str(tmp_path / "ck"))
  path = str(tmp_path / "ck_epoch_2.model")
  st = torch.load(path, map_location="cpu", weights_only=False)
  assert set(st) == {"recoder_version", "model_params", "last_epoch", "model", "optimizer_type",
                     "optimizer", "items", "users", "num_items", "num_users", "loss", "loss_params"}
  assert st["recoder_version"] == "0.4.0" and st["last_epoch"] == 2 and st["optimizer_type"] == "adam"
  assert set(st["model"]) == {
      "en_embedding_layer.weight", "_DynamicAutoencoder__en_linear_embedding_layer.bias",
      "_DynamicAutoencoder__en_linear_embedding_layer.embedding_layer.weight",
      "de_embedding_layer.weight", "_DynamicAutoencoder__de_linear_embedding_layer.bias",
      "_DynamicAutoencoder__de_linear_embedding_layer.embedding_layer.weight"}
  ost = st["optimizer"]["state"]
  assert len(ost) == 4 and all(set(v) >= {"step", "exp_avg", "exp_avg_sq"} for v in ost.values())
  steps = 2 * int(np.ceil(g.csr.shape[0] / 32))
  assert all(int(v["step"]) == steps for v in ost.values())
  users = np.arange(40)
  ui = UsersInteractions(users, g.csr[users])
  before = rec.recommend(ui, 10)
  # fresh instance restored from the file (tests/test_model.py:64-70 of the reference)
  rec2 = Recoder(model=DynamicAutoencoder(), use_cuda=True, optimizer_type="adam", loss="mse")
  rec2.init_from_model_file(path)
  assert rec2.current_epoch == 2 and rec2.num_items == g.csr.shape[1]
  assert rec2.recommend(ui, 10) == before
  # resume: the optimizer state is re-applied and training continues
  rec2.train(ds, batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=3, negative_sampling=True)
  assert len(rec2.loss_history) == 2      # the reference repeats epoch `last_epoch` on resume
  assert np.all(np.isfinite(np.concatenate(rec2.loss_history)))
  eng = rec2._engine()
  assert all(s.step > steps for s in eng.states.values())


def test_dataframe_to_csr_matrix_contract():
  """reference tests/test_data.py:30-58 (every interaction returned exactly once)."""
  import pandas as pd
  from recoder_amd.utils import dataframe_to_csr_matrix
  rng = np.random.RandomState(0)
  df = pd.DataFrame({"user": rng.randint(0, 100, 1000), "item": rng.randint(0, 200, 1000),
                     "inter": np.ones(1000)}).drop_duplicates(["user", "item"]).reset_index(drop=True)
  m, imap, umap = dataframe_to_csr_matrix(df, user_col="user", item_col="item", inter_col="inter")
  assert m.shape == (df.user.nunique(), df.item.nunique()) and m.nnz == len(df)
  for u, it in zip(df.user[:50], df.item[:50]):
    assert m[umap[u], imap[it]] == 1.0
  m2, _, _ = dataframe_to_csr_matrix(df[:100], "user", "item", "inter", item_id_map=imap, user_id_map=umap)
  assert m2.shape == m.shape


@pytest.mark.parametrize("kind", ["ae", "ae_overlap", "ae_eager", "ae_items", "mf", "mf_sparse", "ae_rsag",
                                  "ae_eager_rsag", "ae_sparse_owned", "mf_sparse_owned", "ae_stack", "ae_zero",
                                  "ae_eager_zero", "ae_local", "ae_eager_local"])
def test_recoder_data_parallel_one_rank_equals_single_process(monkeypatch, kind):
  """Recoder.train under an initialised torch.distributed group (RCCL, 1 rank):
  two-phase collation + all-reduced gradients must reproduce the plain run.  ae_overlap: the
  two-group exchange of a multi-rank run (decoder-side gradients on the communication stream while
  the step stream runs dZ -> encoder backward), forced on although one rank has nothing to hide.
  ae / ae_overlap replay the phased step, its collectives included, as HIP graphs
  (graph.GraphStepper with a DataParallel); ae_eager (RK_GRAPH=0) sequences it from the host."""
  import torch.distributed as dist
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  csr = synth_csr(1200, 2500, 25, seed=9)
  # *_zero: sharded dense Adam forced on with one rank (the "shard" is the whole table): the dense gradient
  # layout, ncclReduceScatter, the row-range sweep from the shard, the publish -- replayed or host-sequenced
  zero = kind.endswith("_zero")
  if zero:
    monkeypatch.setenv("RK_DP_ZERO", "force")
    kind = kind[:-5]
  # *_local: per-rank item sets (RK_DP_ITEMSETS=local; with one rank the rank's set IS the union): the gradients
  # laid out by item id, all-reduced in place, the update reading them row by row
  local = kind.endswith("_local")
  if local:
    monkeypatch.setenv("RK_DP_ITEMSETS", "local")
    kind = kind[:-6]
  # "ae" = users sharded (gradient all-reduce), "ae_items" = items sharded (parallel.ItemParallel)
  monkeypatch.setenv("RK_PARALLEL", "items" if kind == "ae_items" else "users")
  items_mode = kind == "ae_items"
  overlap = kind == "ae_overlap"
  eager_dp = kind in ("ae_eager", "ae_eager_rsag")
  # ae_rsag / ae_eager_rsag: the gradient buckets as ncclReduceScatter + ncclAllGather (RK_DP_EXCHANGE);
  # *_owned: SparseAdam tables under owned-row Adam, forced on with one rank -- the partial rows travel
  # through the grouped ncclSend / ncclRecv exchange to "their owner", the updated rows back
  rsag = kind in ("ae_rsag", "ae_eager_rsag")
  owned = kind.endswith("_owned")
  if rsag:
    monkeypatch.setenv("RK_DP_EXCHANGE", "rsag")
  if owned:
    monkeypatch.setenv("RK_DP_OWNED", "force")
  port_off = 40 * rsag + 60 * owned + 7 * (kind == "mf_sparse_owned") + 110 * zero + 130 * local
  ae_sparse = kind == "ae_sparse_owned"
  if kind == "mf_sparse_owned":
    kind = "mf_sparse"
  if rsag or ae_sparse:
    kind = "ae_eager" if kind == "ae_eager_rsag" else "ae"
  if overlap:
    monkeypatch.setenv("RK_DP_OVERLAP", "1")
  if eager_dp:
    monkeypatch.setenv("RK_GRAPH", "0")
  # (round 5: the entry-by-entry engines -- MatrixFactorization, hidden stacks -- replay under users-DP too,
  # their all-reduces captured with the step; owned-row SparseAdam stays host-sequenced)
  graph_dp = kind in ("ae", "ae_overlap", "mf", "mf_sparse", "ae_stack") and not owned
  stack = kind == "ae_stack"
  kind = "ae" if (items_mode or overlap or eager_dp or stack) else kind
  c = STEP_CASES[0][1] if kind == "ae" else dict(kind="mf", embedding_size=32,
                                                 activation_type="tanh", sparse=(kind == "mf_sparse"))
  if stack:
    c = dict(c, hidden_layers=[64, 40], loss="logloss")
  if ae_sparse:
    c = dict(c, sparse=True)

  def run(dp):
    torch.manual_seed(11)
    model = make_model(c)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam",
                  loss=("logloss" if stack else "mse") if kind == "ae" else "logistic")
    rec.user_order_hook = lambda epoch, n: np.arange(n, dtype=np.int64)
    rec.train(RecommendationDataset(csr), batch_size=300, lr=1e-3, weight_decay=2e-5, num_epochs=2,
              negative_sampling=True)
    assert ((rec._ip if items_mode else rec._dp) is not None) == dp
    gs = getattr(rec, "_graph_stepper", None)
    if dp:
      assert (gs is not None and gs.dp is rec._dp and gs.warmed) == graph_dp
      assert bool(getattr(rec._engine(), "owned_rows", False)) == owned
      assert bool(getattr(rec._engine(), "zero_adam", False)) == zero
      assert bool(getattr(rec._dp, "local_sets", False)) == local
      if rsag:
        assert rec._dp.exchange_mode == "rsag"
    return np.concatenate(rec.loss_history), {k: v.detach().cpu().clone() for k, v in model.named_parameters()}

  base_l, base_p = run(False)
  monkeypatch.setenv("RK_FORCE_DP", "1")
  monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
  monkeypatch.setenv("MASTER_PORT", str(29577 + ["ae", "mf", "mf_sparse"].index(kind) + 5 * items_mode +
                                        20 * overlap + 30 * eager_dp + port_off + 90 * stack))
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
  try:
    dp_l, dp_p = run(True)
  finally:
    dist.destroy_process_group()
  assert np.allclose(dp_l, base_l, rtol=1e-6, atol=0)
  # (the multi-GPU step variants run dW through different kernels than the single-GPU one-call
  # step -- fp32-MFMA tiles vs bf16 triples: gradients agree to ~1e-7, Adam's m / sqrt(v)
  # amplifies that on near-zero gradients)
  for k in base_p:
    frac, mx, scale = close_stats(dp_p[k].numpy(), base_p[k].numpy(), 1e-4, 2e-6)
    assert frac < 2e-3, (k, frac, mx, scale)


@pytest.mark.parametrize("case_name", ["ae200_mse", "mf128_mse_sparse", "mf64_bce_dense_drop", "stack_mse", "stack_logloss"])
def test_data_parallel_graph_replay_is_bitwise_equal_to_host_sequencing(case_name, monkeypatch):
  """Users-DP under one RCCL rank: the phased step replayed as HIP graphs (collectives inside, exchange
  over the blocks' capacity) against the same step sequenced from the host (exchange over the live
  rows): same losses, same parameters, to the bit.  The one-call autoencoder step and the
  entry-by-entry engines -- MatrixFactorization, hidden stacks -- whose replay captures train_step with its
  collectives, capacity-sized gradient views and the loss_dp -> loss-slot publish."""
  import torch.distributed as dist
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  csr = synth_csr(1300, 2500, 25, seed=19)
  extra = {"stack_mse": dict(kind="ae", hidden_layers=[64, 32], activation_type="tanh", noise_prob=0.3, sparse=False,
                             loss="mse"),
           "stack_logloss": dict(kind="ae", hidden_layers=[48, 24], activation_type="tanh", noise_prob=0.0,
                                 dropout_prob=0.2, sparse=False, loss="logloss")}
  c = extra.get(case_name) or next(cfg for name, cfg, *_ in STEP_CASES if name == case_name)
  monkeypatch.setenv("RK_PARALLEL", "users")
  monkeypatch.setenv("RK_FORCE_DP", "1")
  monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
  monkeypatch.setenv("MASTER_PORT", str(29643 + ["ae200_mse", "mf128_mse_sparse", "mf64_bce_dense_drop", "stack_mse",
                                                 "stack_logloss"].index(case_name)))
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
  try:
    out = {}
    for mode in ("1", "0"):
      monkeypatch.setenv("RK_GRAPH", mode)
      torch.manual_seed(13)
      model = make_model(c)
      rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=c["loss"])
      rec.user_order_hook = lambda epoch, n: np.random.RandomState(epoch).permutation(n)
      rec.train(RecommendationDataset(csr), batch_size=250, lr=1e-3, weight_decay=2e-5, num_epochs=2,
                negative_sampling=True)
      gs = getattr(rec, "_graph_stepper", None)
      assert (gs is not None and gs.dp is rec._dp and gs.warmed) == (mode == "1")
      if mode == "1":
        assert gs.c_step == (case_name == "ae200_mse")
      out[mode] = (np.concatenate(rec.loss_history), {k: v.detach().cpu().clone() for k, v in model.named_parameters()})
  finally:
    dist.destroy_process_group()
  assert np.array_equal(out["1"][0], out["0"][0])
  for k in out["1"][1]:
    assert torch.equal(out["1"][1][k], out["0"][1][k]), k


class _VirtualRanks:
  """In-process stand-in for the collectives of parallel.ItemParallel: `world` threads,
  one virtual rank each, on ONE GPU."""

  def __init__(self, world):
    import threading
    self.world = world
    self.barrier = threading.Barrier(world)
    self.slots = [None] * world

  def _exchange(self, rank, t):
    torch.cuda.synchronize()
    self.slots[rank] = t
    self.barrier.wait()
    parts = [self.slots[r].clone() for r in range(self.world)]
    torch.cuda.synchronize()
    self.barrier.wait()
    return parts

  def allreduce(self, rank):
    def fn(t):
      parts = self._exchange(rank, t)
      acc = parts[0]
      for q in parts[1:]:          # same order on every rank
        acc = acc + q
      t.copy_(acc)
      torch.cuda.synchronize()
      self.barrier.wait()
      return t
    return fn

  def allgather(self, rank):
    return lambda t: self._exchange(rank, t)

  def allreduce_max(self, rank):
    def fn(t):
      parts = self._exchange(rank, t)
      acc = parts[0]
      for q in parts[1:]:
        acc = torch.maximum(acc, q)
      t.copy_(acc)
      torch.cuda.synchronize()
      self.barrier.wait()
      return t
    return fn


@pytest.mark.parametrize("case", ["mse_dense", "bce_sparse_tied", "ae2_dropout", "ae_logloss", "mf_dense",
                                  "mf_sparse"])
def test_item_parallel_two_virtual_ranks_equal_single_process(case):
  """parallel.ItemParallel with N = 2 on one GPU (two threads, injected collectives):
  item i on rank i % 2, every rank sees all users; after training and the owners'
  publication both replicas equal the single-process run."""
  import threading
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  from recoder_amd.parallel import ItemParallel
  # mse_dense runs a global batch of 1600 rows: split-K dW (2 slabs) and 4 row segments in the
  # encoder backward; the other case stays below both thresholds
  big = case in ("mse_dense", "bce_sparse_tied")     # (the BCE epilogue's large-batch tiling too)
  csr = synth_csr(3300 if big else 900, 1500, 20, seed=21)
  gb = 1600 if big else 300
  if case == "mse_dense":
    mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.0, sparse=False)
    loss, wd = "mse", 2e-5
  elif case == "bce_sparse_tied":
    mk = lambda: DynamicAutoencoder([32], activation_type="sigmoid", noise_prob=0.0, sparse=True,
                                    is_constrained=True)
    loss, wd = "logistic", 0.0
  elif case == "ae2_dropout":
    # hidden stack + bottleneck dropout (counter RNG keyed on the batch row: identical on all
    # ranks): the per-entry Python sequencing of the step
    mk = lambda: DynamicAutoencoder([48, 24], activation_type="tanh", noise_prob=0.0, dropout_prob=0.3,
                                    sparse=False)
    loss, wd = "mse", 1e-5
  elif case == "ae_logloss":
    # multinomial loss: the softmax statistics are combined over the item shards
    mk = lambda: DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.0, sparse=False)
    loss, wd = "logloss", 2e-5
  else:
    mk = lambda: MatrixFactorization(24, activation_type="tanh", sparse=(case == "mf_sparse"))
    loss, wd = "logistic", (0.0 if case == "mf_sparse" else 2e-5)
  order = np.random.RandomState(5).permutation(csr.shape[0]).astype(np.int64)
  kw = dict(lr=1e-3, weight_decay=wd, num_epochs=2, negative_sampling=True)

  def new(batch):
    torch.manual_seed(17)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
    rec.user_order_hook = lambda epoch, n: order
    return model, rec

  model0, rec0 = new(gb)
  rec0.train(RecommendationDataset(csr), batch_size=gb, **kw)
  base_l = np.concatenate(rec0.loss_history)
  base_p = {k: v.detach().cpu().clone() for k, v in model0.named_parameters()}

  world = 2
  vr = _VirtualRanks(world)
  reps = []
  for r in range(world):
    model, rec = new(gb // world)
    rec._Recoder__init_training(RecommendationDataset(csr), kw["lr"], wd)   # same seed, main thread
    rec._ip_override = ItemParallel(rank=r, world=world, allreduce_fn=vr.allreduce(r),
                                    allgather_fn=vr.allgather(r))
    reps.append((model, rec))
  errs = []

  def run(r):
    try:
      torch.cuda.set_device(0)
      reps[r][1].train(RecommendationDataset(csr), batch_size=gb // world, **kw)
    except BaseException as e:       # noqa: B036 -- release the other thread
      errs.append(e)
      vr.barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  assert not errs, errs
  for model, rec in reps:
    got_l = np.concatenate(rec.loss_history)
    assert np.allclose(got_l, base_l, rtol=2e-5, atol=0), (got_l[:3], base_l[:3])
    for k, v in model.named_parameters():
      frac, mx, scale = close_stats(v.detach().cpu().numpy(), base_p[k].numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (k, frac, mx, scale)


@pytest.mark.parametrize("seed", list(range(16)))
def test_item_parallel_virtual_ranks_random_shapes(seed):
  """test_item_parallel_two_virtual_ranks_equal_single_process over random shapes: 2 or 3 virtual
  ranks (item i on rank i % world), ragged last batches, both model families, the three losses,
  hidden stacks / dropout (per-entry sequencing), dense / sparse / tied."""
  import threading
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  from recoder_amd.parallel import ItemParallel
  rng = np.random.RandomState(4000 + seed)
  world = int(rng.choice([2, 3]))
  b = int(rng.choice([8, 40, 75]))                 # per-rank batch; the global batch is world * b
  gb = world * b
  n = gb * int(rng.choice([1, 2, 4])) + (int(rng.randint(1, gb)) if rng.rand() < 0.5 else 0)
  n_items = int(rng.choice([97, 600]))
  kind = "mf" if rng.rand() < 0.25 else "ae"
  loss = str(rng.choice(["mse", "logistic", "logloss"] if kind == "ae" else ["mse", "logistic"]))
  sparse = bool(rng.rand() < 0.4)
  wd = 0.0 if sparse else 1e-5
  if kind == "ae":
    layers = [int(rng.choice([8, 32]))] if rng.rand() < 0.7 else [32, 16]
    tied = bool(rng.rand() < 0.25) and loss != "logloss"
    drop = 0.25 if (len(layers) > 1 and rng.rand() < 0.5) else 0.0
    act = str(rng.choice(["tanh", "sigmoid", "relu"]))
    mk = lambda: DynamicAutoencoder(layers, activation_type=act, noise_prob=0.0, dropout_prob=drop,
                                    sparse=sparse, is_constrained=tied)
    desc = dict(kind=kind, layers=layers, tied=tied, drop=drop, act=act)
  else:
    d = int(rng.choice([8, 24]))
    act = str(rng.choice(["none", "tanh"]))
    mk = lambda: MatrixFactorization(d, activation_type=act, sparse=sparse)
    desc = dict(kind=kind, d=d, act=act)
  desc.update(world=world, b=b, n=n, n_items=n_items, loss=loss, sparse=sparse)
  csr = synth_csr(n, n_items, 10, seed=4100 + seed, ratings=(loss == "mse" and rng.rand() < 0.5))
  order = rng.permutation(n).astype(np.int64)
  kw = dict(lr=1e-3, weight_decay=wd, num_epochs=2, negative_sampling=True)

  def new():
    torch.manual_seed(71 + seed)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
    rec.user_order_hook = lambda epoch, n_: order
    return model, rec
  model0, rec0 = new()
  rec0.train(RecommendationDataset(csr), batch_size=gb, **kw)
  base_l = np.concatenate(rec0.loss_history)
  base_p = {k: v.detach().cpu().clone() for k, v in model0.named_parameters()}
  vr = _VirtualRanks(world)
  reps = []
  for r in range(world):
    model, rec = new()
    rec._Recoder__init_training(RecommendationDataset(csr), kw["lr"], wd)
    rec._ip_override = ItemParallel(rank=r, world=world, allreduce_fn=vr.allreduce(r),
                                    allgather_fn=vr.allgather(r))
    reps.append((model, rec))
  errs = []

  def run(r):
    try:
      torch.cuda.set_device(0)
      ds_r = RecommendationDataset(csr)
      if seed % 3 == 1:
        # the matrix only exists in HBM: row statistics and the column shard are taken on the device
        from recoder_amd.data import DeviceDataset
        ds_r = DeviceDataset(ds_r.device_csr())
      elif seed % 3 == 2:
        ds_r.device_csr()              # host dataset, already resident
      reps[r][1].train(ds_r, batch_size=b, **kw)
    except BaseException as e:       # noqa: B036 -- release the other threads
      errs.append(e)
      vr.barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  assert not errs, (desc, errs)
  for model, rec in reps:
    got_l = np.concatenate(rec.loss_history)
    assert len(got_l) == len(base_l), desc
    assert np.allclose(got_l, base_l, rtol=2e-5, atol=0), (desc, got_l[:3], base_l[:3])
    for k, v in model.named_parameters():
      frac, mx, scale = close_stats(v.detach().cpu().numpy(), base_p[k].numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (desc, k, frac, mx, scale)


@pytest.mark.parametrize("case", ["mse_dense", "bce_sparse_tied", "ae2_dropout", "ae_logloss", "mse_sparse_owned",
                                  "mse_sparse_replicated", "mse_dense_replicated", "bce_dense_tied"])
def test_data_parallel_two_virtual_ranks_equal_single_process(case, monkeypatch):
  """parallel.DataParallel (users sharded -- north_star's partitioning, the multi-GPU default)
  with N = 2 on one GPU: two threads drive the REAL product class -- two-phase collation with the
  MAX-reduced item stamps, the engine's data-parallel phases of rk_ae_train_step, one group of
  SUM all-reduces per step -- with in-process collectives.  Both replicas must equal the
  single-process run with batch_size = 2 * B over the interleaved user order (the reference's own
  "one item set, several row blocks" semantics, data.py:216-223,231-249)."""
  import threading
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  from recoder_amd.parallel import DataParallel, shard_range
  csr = synth_csr(1200, 1500, 20, seed=23)
  B, world = 150, 2
  if case in ("mse_dense", "mse_dense_replicated"):
    # dense Adam on the one-call step: SHARDED over the ranks by default (parallel.DataParallel ZeRO-1: dense
    # gradient layout reduce-scattered, each rank sweeps its row range, updated rows all-gathered);
    # RK_DP_ZERO=0 keeps the replicated sweep
    mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.0, sparse=False)
    loss, wd = "mse", 2e-5
    monkeypatch.setenv("RK_DP_ZERO", "0" if case.endswith("replicated") else "1")
  elif case == "bce_dense_tied":
    monkeypatch.setenv("RK_DP_ZERO", "1")
    mk = lambda: DynamicAutoencoder([32], activation_type="sigmoid", noise_prob=0.0, sparse=False,
                                    is_constrained=True)
    loss, wd = "logistic", 2e-5
  elif case == "bce_sparse_tied":
    monkeypatch.setenv("RK_DP_OWNED", "1")     # (owned-row SparseAdam on a tied table)
    mk = lambda: DynamicAutoencoder([32], activation_type="sigmoid", noise_prob=0.0, sparse=True,
                                    is_constrained=True)
    loss, wd = "logistic", 0.0
  elif case == "ae2_dropout":
    mk = lambda: DynamicAutoencoder([48, 24], activation_type="tanh", noise_prob=0.0, dropout_prob=0.0,
                                    sparse=False)
    loss, wd = "mse", 1e-5
  elif case.startswith("mse_sparse"):
    # SparseAdam tables: every item's rows belong to one rank (owned-row Adam: the partial gradient rows
    # go to their owner, the updated rows come back) -- or, RK_DP_OWNED=0, the replicated update
    mk = lambda: DynamicAutoencoder([40], activation_type="tanh", noise_prob=0.0, sparse=True)
    loss, wd = "mse", 0.0
    # (RK_DP_OWNED=auto, the default since round 5, prices the two and would pick the replicated update here)
    monkeypatch.setenv("RK_DP_OWNED", "0" if case.endswith("replicated") else "1")
  else:
    monkeypatch.setenv("RK_DP_ZERO", "1")
    mk = lambda: DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.0, sparse=False)
    loss, wd = "logloss", 2e-5
  n = csr.shape[0]
  per = n // world
  rng = np.random.RandomState(6)
  shard_orders = [rng.permutation(shard_range(n, r, world)[1] - shard_range(n, r, world)[0])[:per]
                  .astype(np.int64) for r in range(world)]
  kw = dict(lr=1e-3, weight_decay=wd, num_epochs=2, negative_sampling=True)

  def new():
    torch.manual_seed(19)
    model = mk()
    return model, Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)

  # single process: global batch = rank 0's B users followed by rank 1's B users, step by step
  lo1 = shard_range(n, 1, world)[0]
  glob = []
  for off in range(0, per, B):
    glob += list(shard_orders[0][off:off + B]) + list(lo1 + shard_orders[1][off:off + B])
  glob = np.asarray(glob, dtype=np.int64)
  assert len(glob) == n
  model0, rec0 = new()
  rec0.user_order_hook = lambda epoch, n_: glob
  rec0.train(RecommendationDataset(csr), batch_size=world * B, **kw)
  base_l = np.concatenate(rec0.loss_history)
  base_p = {k: v.detach().cpu().clone() for k, v in model0.named_parameters()}

  vr = _VirtualRanks(world)
  reps = []
  for r in range(world):
    model, rec = new()
    rec._Recoder__init_training(RecommendationDataset(csr), kw["lr"], wd)   # same seed, main thread
    rec._dp_override = DataParallel(rank=r, world=world, allreduce_fn=vr.allreduce(r),
                                    allreduce_max_fn=vr.allreduce_max(r), allgather_fn=vr.allgather(r))
    rec.user_order_hook = (lambda rr: (lambda epoch, n_: shard_orders[rr]))(r)
    reps.append((model, rec))
  errs = []

  def run(r):
    try:
      torch.cuda.set_device(0)
      reps[r][1].train(RecommendationDataset(csr), batch_size=B, **kw)
    except BaseException as e:       # noqa: B036 -- release the other thread
      errs.append(e)
      vr.barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  assert not errs, errs
  for model, rec in reps:
    assert rec._dp is not None and rec._ip is None
    owned = case in ("bce_sparse_tied", "mse_sparse_owned")
    assert bool(rec._engine().owned_rows) == owned, (case, rec._engine().owned_rows)
    assert bool(getattr(rec._engine(), "zero_adam", False)) == (case in ("mse_dense", "bce_dense_tied", "ae_logloss")), case
    got_l = np.concatenate(rec.loss_history)
    assert len(got_l) == len(base_l)
    assert np.allclose(got_l, base_l, rtol=2e-5, atol=0), (got_l[:3], base_l[:3])
    for k, v in model.named_parameters():
      frac, mx, scale = close_stats(v.detach().cpu().numpy(), base_p[k].numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (k, frac, mx, scale)
  if case in ("mse_sparse_owned", "mse_dense", "bce_dense_tied"):
    # the replicas hold identical parameters AND (after train()'s final sync) identical Adam moments
    (m0, r0), (m1, r1) = reps
    for (k, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
      assert torch.equal(a, b), k
    for name in ("en_embedding_layer.weight", "de_embedding_layer.weight")[:1 if case == "bce_dense_tied" else 2]:
      s0, s1 = r0._engine().states[name], r1._engine().states[name]
      assert torch.equal(s0.m, s1.m) and torch.equal(s0.v, s1.v), name
      assert float(s0.v.abs().max()) > 0


@pytest.mark.parametrize("variant", ["allreduce", "sharded", "tied", "three_ranks"])
def test_data_parallel_local_item_sets_equal_the_ddp_oracle(variant, monkeypatch):
  """RK_DP_ITEMSETS=local (opt-in): every rank samples its negatives from ITS OWN users' item set -- the reference's
  trainer under conventional DDP, not its shared item set -- and the gradients travel laid out by item id (all-reduced
  in place, or reduce-scattered with the sharded dense update on top).  Virtual ranks on one GPU against
  oracle.train_step_ddp: gradient accumulation over the ranks' batches, one optimizer step."""
  import threading
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  from recoder_amd.parallel import DataParallel, shard_range
  monkeypatch.setenv("RK_DP_ITEMSETS", "local")
  monkeypatch.setenv("RK_DP_ZERO", "1" if variant in ("sharded", "three_ranks") else "0")
  world = 3 if variant == "three_ranks" else 2
  tied = variant == "tied"
  csr = synth_csr(606, 1501, 20, seed=41)
  B, epochs = 100, 2
  n = csr.shape[0]
  per = n // world
  mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.0, sparse=False, is_constrained=tied)
  loss, wd, lr = ("logistic" if tied else "mse"), 2e-5, 1e-3
  orders = {e: [np.random.RandomState(50 + 10 * e + r).permutation(shard_range(n, r, world)[1] - shard_range(n, r, world)[0])[:per]
                .astype(np.int64) for r in range(world)] for e in range(1, epochs + 1)}
  vr = _VirtualRanks(world)
  reps = []
  for r in range(world):
    torch.manual_seed(23)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
    rec._Recoder__init_training(RecommendationDataset(csr), lr, wd)
    rec._dp_override = DataParallel(rank=r, world=world, allreduce_fn=vr.allreduce(r),
                                    allreduce_max_fn=vr.allreduce_max(r), allgather_fn=vr.allgather(r))
    rec.user_order_hook = (lambda rr: (lambda epoch, n_: orders[epoch][rr]))(r)
    reps.append((model, rec))
  init = {k: v.detach().cpu().clone() for k, v in reps[0][0].named_parameters()}
  errs = []

  def run(r):
    try:
      torch.cuda.set_device(0)
      reps[r][1].train(RecommendationDataset(csr), batch_size=B, lr=lr, weight_decay=wd, num_epochs=epochs,
                       negative_sampling=True)
    except BaseException as e:       # noqa: B036 -- release the other threads
      errs.append(e)
      vr.barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  assert not errs, errs
  # the oracle: one model, per step one Batch PER RANK (collated on its own: its own item set), averaged loss
  o = orc.OracleRecoder("ae", init, hidden_layers=[64], activation_type="tanh", is_constrained=tied, loss=loss,
                        loss_params=None, lr=lr, weight_decay=wd)
  want = []
  for e in range(1, epochs + 1):
    for off in range(0, (per // B) * B, B):
      batches = []
      for r in range(world):
        users = shard_range(n, r, world)[0] + orders[e][r][off:off + B]
        batches.append(orc.collate(orc.extract_rows(csr, users), users, B, True)[0])
      want.append(o.train_step_ddp(batches))
    tail = per % B
    if tail:
      batches = []
      for r in range(world):
        users = shard_range(n, r, world)[0] + orders[e][r][(per // B) * B:per]
        batches.append(orc.collate(orc.extract_rows(csr, users), users, tail, True)[0])
      want.append(o.train_step_ddp(batches))
  want = np.asarray(want)
  ref_p = o.state()
  for model, rec in reps:
    assert rec._dp is not None and rec._dp.local_sets
    assert bool(getattr(rec._engine(), "zero_adam", False)) == (variant in ("sharded", "three_ranks"))
    got = np.concatenate(rec.loss_history)
    assert len(got) == len(want)
    assert np.allclose(got, want, rtol=2e-5, atol=0), (got[:3], want[:3])
    for k, v in model.named_parameters():
      frac, mx, scale = close_stats(v.detach().cpu().numpy(), ref_p[k].numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (k, frac, mx, scale)
  for (k, a), (_, b) in zip(reps[0][0].named_parameters(), reps[1][0].named_parameters()):
    assert torch.equal(a, b), k


@pytest.mark.parametrize("seed", list(range(24)))
def test_data_parallel_virtual_ranks_random_shapes(seed, monkeypatch):
  """test_data_parallel_two_virtual_ranks_equal_single_process over random shapes: 2 or 3 virtual
  ranks, shard sizes that leave ragged remainders, sampling on / off, dense / sparse / tied, the
  three losses.  Even seeds: dense-Adam tables of the one-call step sharded over the ranks (RK_DP_ZERO=1:
  item counts that are no multiple of the world size, 3 ranks), odd seeds: the replicated sweep."""
  import threading
  monkeypatch.setenv("RK_DP_ZERO", "1" if seed % 2 == 0 else "0")
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  from recoder_amd.parallel import DataParallel, shard_range
  rng = np.random.RandomState(3000 + seed)
  world = int(rng.choice([2, 3]))
  B = int(rng.choice([16, 50, 96]))
  steps = int(rng.choice([1, 2, 3, 5]))
  # (a remainder is dropped by every rank: then only the replicas can be compared with each other)
  n = world * steps * B + (int(rng.randint(1, world * B)) if rng.rand() < 0.4 else 0)
  n_items = int(rng.choice([120, 700]))
  loss = str(rng.choice(["mse", "logistic", "logloss"]))
  sparse = bool(rng.rand() < 0.4)
  tied = bool(rng.rand() < 0.25) and loss != "logloss"
  ns = bool(rng.rand() < 0.8)
  act = str(rng.choice(["tanh", "sigmoid", "relu"]))
  h = int(rng.choice([8, 32, 60]))
  wd = 0.0 if sparse else 1e-5
  csr = synth_csr(n, n_items, 10, seed=3100 + seed, ratings=(loss == "mse" and rng.rand() < 0.5))
  per = n // world
  n_st = per // B                                              # whole steps per rank and epoch
  if n_st == 0:
    pytest.skip("shard smaller than a batch")
  shard_orders = [rng.permutation(shard_range(n, r, world)[1] - shard_range(n, r, world)[0])[:n_st * B]
                  .astype(np.int64) for r in range(world)]
  kw = dict(lr=1e-3, weight_decay=wd, num_epochs=2, negative_sampling=ns)
  desc = dict(world=world, B=B, n=n, n_items=n_items, loss=loss, sparse=sparse, tied=tied, ns=ns, act=act, h=h)

  def new():
    torch.manual_seed(61 + seed)
    model = DynamicAutoencoder([h], activation_type=act, noise_prob=0.0, sparse=sparse, is_constrained=tied)
    return model, Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
  los = [shard_range(n, r, world)[0] for r in range(world)]
  glob = []
  for off in range(0, n_st * B, B):
    for r in range(world):
      glob += list(los[r] + shard_orders[r][off:off + B])
  glob = np.asarray(glob, dtype=np.int64)
  base_l = base_p = None
  if len(glob) == n:       # no remainder dropped: the single-process run with batch world * B is the same training
    model0, rec0 = new()
    rec0.user_order_hook = lambda epoch, n_: glob
    rec0.train(RecommendationDataset(csr), batch_size=world * B, **kw)
    base_l = np.concatenate(rec0.loss_history)
    base_p = {k: v.detach().cpu().clone() for k, v in model0.named_parameters()}
  vr = _VirtualRanks(world)
  reps = []
  for r in range(world):
    model, rec = new()
    rec._Recoder__init_training(RecommendationDataset(csr), kw["lr"], wd)
    rec._dp_override = DataParallel(rank=r, world=world, allreduce_fn=vr.allreduce(r),
                                    allreduce_max_fn=vr.allreduce_max(r), allgather_fn=vr.allgather(r))
    rec.user_order_hook = (lambda rr: (lambda epoch, n_: shard_orders[rr]))(r)
    reps.append((model, rec))
  errs = []

  def run(r):
    try:
      torch.cuda.set_device(0)
      ds_r = RecommendationDataset(csr)
      if seed % 2:
        # a host dataset that is already resident in HBM: the rank's rows are a device-side slice
        # (model.Recoder._setup_data_parallel) -- same training as the host-sliced shard
        ds_r.device_csr()
      reps[r][1].train(ds_r, batch_size=B, **kw)
    except BaseException as e:       # noqa: B036 -- release the other threads
      errs.append(e)
      vr.barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  assert not errs, (desc, errs)
  for model, rec in reps:
    got_l = np.concatenate(rec.loss_history)
    assert len(got_l) == 2 * n_st, desc
    if base_l is not None:
      assert np.allclose(got_l, base_l, rtol=2e-5, atol=0), (desc, got_l[:3], base_l[:3])
      for k, v in model.named_parameters():
        frac, mx, scale = close_stats(v.detach().cpu().numpy(), base_p[k].numpy(), 1e-4, 2e-6)
        assert frac < 2e-3, (desc, k, frac, mx, scale)
  # the replicas agree with each other bit for bit whatever the shapes
  p_ref = {k: v.detach().cpu() for k, v in reps[0][0].named_parameters()}
  for model, rec in reps[1:]:
    for k, v in model.named_parameters():
      assert torch.equal(v.detach().cpu(), p_ref[k]), (desc, k)


@pytest.mark.parametrize("case", ["dense_noise", "sparse_tied_bce", "logloss", "ratings_all_items",
                                  "ratings_relu_conf", "stack_dropout", "stack_logloss_tied", "mf_sparse",
                                  "mf_dense_dropout"])
def test_graph_replay_is_bitwise_equal_to_eager_steps(case, monkeypatch):
  """recoder_amd/graph.py: groups of steps replayed as HIP graphs (users, stamps, RNG step, Adam
  constants and loss slot derived on the device from a cursor) must reproduce the eagerly
  enqueued steps bit for bit -- over several epochs, with a ragged last batch, a tail that does not
  fill a group and a step mark that cuts a group."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  csr = synth_csr(1430, 900, 18, seed=31, ratings=case.startswith("ratings"))   # 1430 = 11 x 128 + 22: ragged tail
  ns, loss_params = True, None
  if case == "ratings_all_items":       # explicit values, NO negative sampling (the block is the catalogue)
    mk = lambda: DynamicAutoencoder([40], activation_type="tanh", noise_prob=0.2, sparse=False)
    loss, wd, ns = "mse", 2e-5, False
  elif case == "ratings_relu_conf":     # explicit values, unbounded activation (rk_amax in the graph)
    mk = lambda: DynamicAutoencoder([48], activation_type="relu", noise_prob=0.3, sparse=True)
    loss, wd, loss_params = "mse", 0.0, {"confidence": 3}
  elif case == "dense_noise":
    mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.4, sparse=False)
    loss, wd = "mse", 2e-5
  elif case == "sparse_tied_bce":
    mk = lambda: DynamicAutoencoder([32], activation_type="sigmoid", noise_prob=0.0, sparse=True,
                                    is_constrained=True)
    loss, wd = "logistic", 0.0
  elif case == "stack_dropout":
    # hidden stack + both dropouts: the entry-by-entry sequencing under the replay context
    # (rk_replay_t: RNG steps, users, Adam constants and the loss slot from the cursor)
    mk = lambda: DynamicAutoencoder([48, 24], activation_type="tanh", noise_prob=0.2, dropout_prob=0.3,
                                    sparse=False)
    loss, wd = "mse", 1e-5
  elif case == "stack_logloss_tied":
    mk = lambda: DynamicAutoencoder([32, 16], activation_type="sigmoid", noise_prob=0.0, sparse=True,
                                    is_constrained=True)
    loss, wd = "logloss", 0.0
  elif case == "mf_sparse":
    from recoder_amd.nn import MatrixFactorization
    mk = lambda: MatrixFactorization(24, activation_type="none", sparse=True)
    loss, wd = "mse", 0.0
  elif case == "mf_dense_dropout":
    from recoder_amd.nn import MatrixFactorization
    mk = lambda: MatrixFactorization(16, activation_type="tanh", dropout_prob=0.25, sparse=False)
    loss, wd = "logistic", 2e-5
  else:
    mk = lambda: DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.3, sparse=False)
    loss, wd = "logloss", 1e-5
  orders = [np.random.RandomState(40 + e).permutation(csr.shape[0]).astype(np.int64) for e in range(4)]

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    torch.manual_seed(23)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss, loss_params=loss_params)
    rec.user_order_hook = lambda epoch, n: orders[epoch]
    seen = []
    rec.step_marks = {7: lambda: seen.append(7) or False, 18: lambda: seen.append(18) or False}
    rec.train(RecommendationDataset(csr), batch_size=128, lr=1e-3, weight_decay=wd, num_epochs=3,
              negative_sampling=ns, lr_milestones=[2])
    assert seen == [7, 18]
    assert (getattr(rec, "_graph_stepper", None) is not None) == graph
    return (np.concatenate(rec.loss_history),
            {k: v.detach().cpu().clone() for k, v in model.named_parameters()},
            {k: (int(s.step), s.m.detach().cpu().clone()) for k, s in rec._engine().states.items()})
  l0, p0, a0 = run(False)
  l1, p1, a1 = run(True)
  assert len(l0) == 36 and np.array_equal(l0, l1), np.abs(l0 - l1).max()
  for k in p0:
    assert torch.equal(p0[k], p1[k]), k
  for k in a0:
    assert a0[k][0] == a1[k][0] and torch.equal(a0[k][1], a1[k][1]), k


def test_graph_replay_survives_workspace_growth_between_epochs(monkeypatch):
  """Validation / evaluation between two epochs with a LARGER batch and the whole catalogue as the
  item strip make the engine re-allocate its workspaces; the captured graphs hold the old
  addresses and must be dropped and captured again (GraphStepper.recaptures) -- results stay
  bit-identical to the eager run."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.metrics import Recall
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  csr = synth_csr(1100, 6000, 6, seed=37)           # few interactions per row: n_cap << n_items
  val = synth_csr(400, 6000, 6, seed=38)
  orders = [np.random.RandomState(50 + e).permutation(csr.shape[0]).astype(np.int64) for e in range(5)]

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    torch.manual_seed(29)
    model = DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.2, sparse=False)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
    rec.user_order_hook = lambda epoch, n: orders[epoch] if n == csr.shape[0] else None
    rec.train(RecommendationDataset(csr), val_dataset=RecommendationDataset(val, val), batch_size=64,
              lr=1e-3, weight_decay=1e-5, num_epochs=3, negative_sampling=True, eval_freq=1,
              metrics=[Recall(5)], eval_num_recommendations=5, eval_batch_size=300)
    gs = getattr(rec, "_graph_stepper", None)
    assert (gs is not None) == graph
    if graph:
      assert gs.recaptures >= 1, "the evaluation was expected to grow the engine's workspaces"
    return (np.concatenate(rec.loss_history),
            {k: v.detach().cpu().clone() for k, v in model.named_parameters()})
  l0, p0 = run(False)
  l1, p1 = run(True)
  assert np.array_equal(l0, l1), np.abs(l0 - l1).max()
  for k in p0:
    assert torch.equal(p0[k], p1[k]), k


def test_graph_path_through_a_users_session(monkeypatch, tmp_path):
  """What a user does with one trainer, on the default (graph replay) path and eagerly: train, evaluate
  with a larger batch, train on (resume repeats the last epoch, as the reference), checkpoint, load the
  file into a FRESH trainer and continue there.  Losses and parameters must be bit-identical."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.metrics import NDCG, Recall
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  csr = synth_csr(900, 700, 10, seed=41)
  held = synth_csr(200, 700, 10, seed=42)
  mk = lambda: DynamicAutoencoder([48], activation_type="tanh", noise_prob=0.0, sparse=False)

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    order = lambda epoch, n: np.random.RandomState(60 + epoch).permutation(n).astype(np.int64)
    kw = dict(batch_size=100, lr=1e-3, weight_decay=1e-5, negative_sampling=True)
    torch.manual_seed(31)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
    rec.user_order_hook = order
    ds = RecommendationDataset(csr)
    rec.train(ds, num_epochs=2, **kw)
    ev1 = rec.evaluate(RecommendationDataset(held, held), num_recommendations=10,
                       metrics=[Recall(5), NDCG(10)], batch_size=160)
    prefix = str(tmp_path / ("g" if graph else "e"))
    rec.train(ds, num_epochs=4, model_checkpoint_prefix=prefix, checkpoint_freq=4, **kw)
    losses = np.concatenate(rec.loss_history)
    torch.manual_seed(99)
    model2 = mk()
    rec2 = Recoder(model=model2, use_cuda=True, optimizer_type="adam", loss="mse")
    rec2.init_from_model_file(prefix + "_epoch_4.model")
    rec2.user_order_hook = order
    rec2.train(ds, num_epochs=5, **kw)
    ev2 = rec2.evaluate(RecommendationDataset(held, held), num_recommendations=10,
                        metrics=[Recall(5), NDCG(10)], batch_size=50)
    return (losses, np.concatenate(rec2.loss_history),
            {k: v.detach().cpu().clone() for k, v in model2.named_parameters()},
            {str(k): np.asarray(v) for k, v in list(ev1.items()) + list(ev2.items())})
  e = run(False)
  g = run(True)
  assert np.array_equal(e[0], g[0]) and np.array_equal(e[1], g[1])
  assert len(e[0]) == 9 * 5 and len(e[1]) == 9 * 2        # epochs 1-2, then 2-4; the fresh trainer: 4-5
  for k in e[2]:
    assert torch.equal(e[2][k], g[2][k]), k
  for k in e[3]:
    assert np.array_equal(e[3][k], g[3][k], equal_nan=True), k


@pytest.mark.parametrize("with_val", [False, True])
def test_user_order_drawn_ahead_keeps_the_sequence_of_rng_draws(with_val, monkeypatch):
  """No hooks: the user orders come from the global torch RNG exactly as the reference's
  RandomSampler draws them.  The graph path draws the NEXT epoch's order at the end of an epoch
  (while the GPU works) unless a validation pass -- which draws too -- sits in between; the eager
  path draws at the start of every epoch.  Same sequence of draws <=> bit-identical training."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  csr = synth_csr(1024, 600, 12, seed=43)
  val = synth_csr(256, 600, 12, seed=44)

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    torch.manual_seed(37)
    model = DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.0, sparse=False)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
    kw = dict(val_dataset=RecommendationDataset(val, val), eval_freq=2) if with_val else {}
    rec.train(RecommendationDataset(csr), batch_size=128, lr=1e-3, weight_decay=1e-5, num_epochs=4,
              negative_sampling=True, **kw)
    assert (getattr(rec, "_graph_stepper", None) is not None) == graph
    return np.concatenate(rec.loss_history), torch.random.get_rng_state()
  l0, s0 = run(False)
  l1, s1 = run(True)
  assert len(l0) == 32 and np.array_equal(l0, l1), np.abs(l0 - l1).max()
  assert torch.equal(s0, s1)            # and the global RNG ends where the eager run leaves it


@pytest.mark.parametrize("seed", list(range(120)))
def test_graph_vs_eager_random_schedules(seed, monkeypatch):
  """Random epoch lengths (ragged or not, odd / even numbers of groups, shorter than one group),
  group sizes, step marks and hooks: the graph path must equal the eager path bit for bit."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  rng = np.random.RandomState(1000 + seed)
  B = int(rng.choice([32, 64, 100]))
  G = int(rng.choice([1, 2, 3, 4, 8]))
  n_steps = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12, 16]))
  n = n_steps * B + (int(rng.randint(1, B)) if rng.rand() < 0.5 else 0)
  epochs = int(rng.choice([2, 3, 4]))
  hooks = bool(rng.rand() < 0.5)
  n_marks = int(rng.choice([0, 0, 1, 2]))
  total = epochs * (n_steps + (1 if n % B else 0))
  marks = sorted(set(int(x) for x in rng.randint(1, max(2, total), size=n_marks)))
  sparse = bool(rng.rand() < 0.4)
  noise = float(rng.choice([0.0, 0.3]))
  act = str(rng.choice(["tanh", "relu", "sigmoid", "selu"]))
  loss = str(rng.choice(["mse", "mse", "logistic", "logloss"]))
  tied = bool(rng.rand() < 0.25)
  sampling = bool(rng.rand() < 0.8)
  milestones = [2] if rng.rand() < 0.5 else None
  with_val = bool(rng.rand() < 0.3)
  csr = synth_csr(n, 400, 9, seed=500 + seed, ratings=bool(rng.rand() < 0.5) and loss == "mse")
  val = synth_csr(150, 400, 9, seed=900 + seed)
  orders = [np.random.RandomState(70 + e).permutation(n).astype(np.int64) for e in range(epochs + 1)]
  monkeypatch.setenv("RK_GRAPH_GROUP", str(G))

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    torch.manual_seed(41 + seed)
    model = DynamicAutoencoder([24], activation_type=act, noise_prob=noise, sparse=sparse,
                               is_constrained=tied)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
    if hooks:
      rec.user_order_hook = lambda epoch, n_: orders[epoch] if n_ == n else None
    seen = []
    rec.step_marks = {m: (lambda m=m: seen.append(m) or False) for m in marks}
    kw = {}
    if with_val:       # validation loss + Recall@5 on a larger batch between the epochs (workspace growth)
      from recoder_amd.metrics import Recall
      kw = dict(val_dataset=RecommendationDataset(val, val), eval_freq=1, metrics=[Recall(5)],
                eval_num_recommendations=5, eval_batch_size=150)
    rec.train(RecommendationDataset(csr), batch_size=B, lr=1e-3, weight_decay=0.0 if sparse else 1e-5,
              num_epochs=epochs, negative_sampling=sampling, lr_milestones=milestones, **kw)
    assert seen == [m for m in marks if m < total]      # (called BEFORE step m is enqueued)
    return (np.concatenate(rec.loss_history),
            {k: v.detach().cpu().clone() for k, v in model.named_parameters()},
            getattr(rec, "_graph_stepper", None) is not None)
  l0, p0, g0 = run(False)
  l1, p1, g1 = run(True)
  desc = dict(B=B, G=G, n=n, epochs=epochs, hooks=hooks, marks=marks, sparse=sparse, noise=noise, graph=g1,
              act=act, loss=loss, tied=tied, sampling=sampling, milestones=milestones, with_val=with_val)
  assert not g0 and g1 == (n >= B), desc            # (the graph path was really taken)
  assert len(l0) == len(l1) == total, desc
  assert np.array_equal(l0, l1), (desc, np.abs(l0 - l1).max(), int(np.argmax(l0 != l1)))
  for k in p0:
    assert torch.equal(p0[k], p1[k]), (desc, k)


@pytest.mark.parametrize("seed", list(range(12)))
def test_recommend_strips_equal_dense_topk_random_shapes(seed, monkeypatch):
  """Strip-wise streaming decode + per-strip top-k + merge (Recoder.recommend) against the full
  score matrix + torch.topk, for random catalogue sizes, strip widths (incl. a last strip shorter
  than k), k and batch sizes; both model families."""
  from recoder_amd.data import RecommendationDataset, UsersInteractions
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  rng = np.random.RandomState(2000 + seed)
  n_items = int(rng.choice([90, 257, 1000, 4097]))
  n_users = int(rng.choice([40, 130]))
  k = int(rng.choice([1, 5, 20, 64]))
  k = min(k, n_items // 2)
  strip = int(rng.choice([k, k + 1, 64, 100, 333, 5000]))
  B = int(rng.choice([1, 7, 33]))
  kind = "mf" if seed % 4 == 3 else "ae"
  csr = synth_csr(n_users, n_items, 8, seed=600 + seed, ratings=bool(rng.rand() < 0.5))
  torch.manual_seed(50 + seed)
  if kind == "ae":
    model = DynamicAutoencoder([int(rng.choice([8, 32]))], activation_type="tanh", sparse=False)
  else:
    model = MatrixFactorization(16, activation_type="none", sparse=False)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
  rec.train(RecommendationDataset(csr), batch_size=32, lr=1e-2, weight_decay=0.0, num_epochs=1,
            negative_sampling=True)
  monkeypatch.setattr(type(rec), "eval_strip_items", strip, raising=False)
  users = rng.permutation(n_users)[:B]
  ui = UsersInteractions(users=users, interactions_matrix=csr[users])
  got = rec.recommend_array(ui, k)
  want = rec._recommend_dense(ui, k)
  assert got.shape == (B, k)
  if not np.array_equal(got, want):
    # equal scores may be ordered differently by torch.topk: compare the scores instead
    out, _ = rec.predict(ui)
    out = out.cpu().numpy()
    gs = np.take_along_axis(out, got, axis=1)
    ws = np.take_along_axis(out, np.asarray(want), axis=1)
    assert np.allclose(gs, ws, rtol=1e-6, atol=0), dict(n_items=n_items, k=k, strip=strip, B=B, kind=kind)
  seen = csr[users].toarray() > 0
  assert not np.take_along_axis(seen, got, axis=1).any()           # nothing already seen is recommended


@pytest.mark.parametrize("seed", list(range(10)))
def test_recommend_fused_filter_equals_the_strips(seed, monkeypatch):
  """Recoder.recommend with the top-k filter in the decode's epilogue (a strided sample bounds every
  row's k-th best score; only what reaches the bound leaves the kernel; rk_topk_pairs picks the k best
  by (score, lower id)) against the strip-by-strip path: the SAME ids, ties included (duplicated
  decoder rows), seen items never returned; small candidate lists force the overflow fallback."""
  from recoder_amd.data import RecommendationDataset, UsersInteractions
  from recoder_amd.engine import FusedEngine
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  rng = np.random.RandomState(3100 + seed)
  n_items = int(rng.choice([700, 2049, 6000]))
  n_users = 150
  k = int(rng.choice([1, 5, 20]))
  strip = int(rng.choice([128, 500, 1024]))
  B = int(rng.choice([3, 64, 130]))
  kind = "mf" if seed % 5 == 4 else "ae"
  act = "tanh" if seed % 2 == 0 else "relu"          # (unbounded: the Z image takes its scale from rk_amax)
  csr = synth_csr(n_users, n_items, 12, seed=700 + seed, ratings=bool(rng.rand() < 0.5))
  torch.manual_seed(60 + seed)
  if kind == "ae":
    model = DynamicAutoencoder([int(rng.choice([8, 36, 64]))], activation_type=act, sparse=False)
  else:
    model = MatrixFactorization(16, activation_type="none", sparse=False)
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
  rec.train(RecommendationDataset(csr), batch_size=50, lr=1e-2, weight_decay=0.0, num_epochs=1,
            negative_sampling=True)
  if seed % 3 == 0:
    # exact ties: a run of items shares one decoder row and bias
    W, b = rec._engine()._decoder_params()
    with torch.no_grad():
      W[100:140] = W[100:101]
      b[100:140] = b[100]
  cap = int(rng.choice([64, 256, 4096])) if k <= 5 else 4096
  monkeypatch.setattr(FusedEngine, "EVAL_CAND_CAP", cap)
  monkeypatch.setattr(FusedEngine, "EVAL_SAMPLE_MIN", int(rng.choice([64, 200, 512])))
  monkeypatch.setattr(type(rec), "eval_strip_items", strip, raising=False)
  users = rng.permutation(n_users)[:B]
  ui = UsersInteractions(users=users, interactions_matrix=csr[users])
  rec.eval_fused_batches = 0
  got = rec.recommend_array(ui, k)
  fused = rec.eval_fused_batches
  monkeypatch.setenv("RK_EVAL_FUSED", "0")
  want = rec.recommend_array(ui, k)
  assert rec.eval_fused_batches == fused
  assert np.array_equal(got, want), dict(n_items=n_items, k=k, B=B, kind=kind, cap=cap, fused=fused)
  from recoder_amd import _lib
  if _lib.load().rk_gemm_plain_bf16():
    assert fused == 0                                  # (plain-bf16 images: the fp16-pair filter must not run)
  elif cap == 4096:
    assert fused == 1                                  # (the filter really ran)
  seen = csr[users].toarray() > 0
  assert not np.take_along_axis(seen, got, axis=1).any()


def test_topk_tie_rule_and_strip_merge():
  """rk_topk_masked: exact ties resolve to the LOWER item id; only POSITIVE stored interactions are
  masked (model.py:537); the strip-wise top-k + merge equals the one-pass top-k."""
  import scipy.sparse as sp
  from recoder_amd import _lib
  from recoder_amd._lib import check
  from recoder_amd.device import Block, DeviceCSR, current_stream
  lib = _lib.load()
  dev = torch.device("cuda")
  B, n, k = 37, 3000, 25
  rng = np.random.RandomState(3)
  # heavily quantised scores: many exact ties
  sc = (rng.randint(0, 40, size=(B, n)) / 8.0).astype(np.float32)
  rows = np.repeat(np.arange(B), 30)
  cols = rng.randint(0, n, size=len(rows))
  vals = rng.choice([-2.0, 1.0, 3.0], size=len(rows)).astype(np.float32)    # negatives: NOT masked
  m = sp.coo_matrix((vals, (rows, cols)), shape=(B, n)).tocsr()
  m.sum_duplicates()
  m.eliminate_zeros()
  dcsr = DeviceCSR(m)
  blk = Block(B, int(m.nnz), n, dev, negative_sampling=False, need_bits_cr=False)
  blk.collate(dcsr, torch.arange(B, dtype=torch.int64, device=dev), negative_sampling=False)
  dense = np.asarray(m.todense())
  masked = sc.copy()
  masked[dense > 0] = -np.inf
  want = np.argsort(-masked, axis=1, kind="stable")[:, :k]       # ties: lower index first
  ld = 3008
  sd = torch.zeros(B, ld, device=dev)
  sd[:, :n] = torch.from_numpy(sc).to(dev)
  idx = torch.empty(B, k, dtype=torch.int64, device=dev)
  val = torch.empty(B, k, dtype=torch.float32, device=dev)
  check(lib.rk_topk_masked(sd.data_ptr(), B, n, ld, blk.ref, 0, k, 0, 1, idx.data_ptr(), val.data_ptr(), k,
                           current_stream()), "rk_topk_masked")
  assert np.array_equal(idx.cpu().numpy(), want)
  assert np.array_equal(val.cpu().numpy(), np.take_along_axis(masked, want, 1))
  # strips of 700 columns (the last one 200 wide), then the merge
  strip, ns = 700, 5
  cidx = torch.empty(B, ns * k, dtype=torch.int64, device=dev)
  cval = torch.empty(B, ns * k, dtype=torch.float32, device=dev)
  for s_ in range(ns):
    lo, hi = s_ * strip, min(n, (s_ + 1) * strip)
    part = sd[:, lo:hi].contiguous()
    check(lib.rk_topk_masked(part.data_ptr(), B, hi - lo, hi - lo, blk.ref, 0, k, lo, 1,
                                   cidx[:, s_ * k:].data_ptr(), cval[:, s_ * k:].data_ptr(), ns * k,
                                   current_stream()), "rk_topk_masked")
  pos = torch.empty(B, k, dtype=torch.int64, device=dev)
  check(lib.rk_topk_masked(cval.data_ptr(), B, ns * k, ns * k, None, 0, k, 0, 1, pos.data_ptr(), None, k,
                           current_stream()), "rk_topk_masked")
  assert np.array_equal(torch.gather(cidx, 1, pos).cpu().numpy(), want)


# --------------------------------------------------------------------------
# hidden nn.Linear stack (reference nn.py:242-249): the C-ABI entry points against torch fp64
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,K,wt,act,acc", [
  (500, 200, 200, 0, "tanh", 0), (500, 200, 200, 1, "tanh", 1),      # C3's layers (csrc/linear.hip)
  (37, 129, 77, 0, "sigmoid", 0), (37, 129, 77, 1, "relu", 1),        # nothing aligned, ragged tiles
  (1, 8, 4, 0, "none", 0), (64, 32, 600, 0, "selu", 0), (300, 50, 30, 1, "elu", 0),
  (33, 1100, 40, 0, "tanh", 0), (33, 40, 1100, 1, "tanh", 1),         # past the small kernel: LDS-tiled path
])
@pytest.mark.parametrize("pair", [0, 1])
def test_linear_layer_entry_points_match_torch(B, N, K, wt, act, acc, pair):
  from recoder_amd import _lib
  from recoder_amd._lib import ACT, check, ptr
  from recoder_amd.device import current_stream
  lib = _lib.load()
  lib.rk_linear_pair(pair)       # dX and dW as one launch (opt-in) or two: the same tiles either way
  g = torch.Generator(device="cpu").manual_seed(B * 7 + N * 3 + K)
  f = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
  X, W, b, dY0, dW0 = f(B, K), f(N, K) * 0.2, f(N) * 0.1, f(B, N), f(N, K)
  Wd = (W.t().contiguous() if wt else W).to(dev())          # wt: the layer is given as [K, N]
  Xd, bd = X.to(dev()), b.to(dev())
  Y = torch.empty(B, N, device=dev())
  st = current_stream()
  a = ACT[act]
  check(lib.rk_linear_fwd(ptr(Xd), ptr(Wd), ptr(bd), B, N, K, wt, a, ptr(Y), st), "rk_linear_fwd")
  pre = X.double() @ W.double().t() + b.double()
  fn = dict(tanh=torch.tanh, sigmoid=torch.sigmoid, relu=torch.relu, selu=torch.selu,
            elu=torch.nn.functional.elu, none=lambda x: x)[act]
  pre.requires_grad_(True)
  ref = fn(pre)
  scale = max(1.0, float(K) ** 0.5)
  assert torch.allclose(Y.cpu().double(), ref.detach(), rtol=1e-5, atol=2e-6 * scale)
  # backward: dY is overwritten with dY * act'(Y); dX, dW (optionally accumulated), db
  gpre, = torch.autograd.grad(ref, pre, dY0.double())
  dY = dY0.to(dev()).clone()
  dX = torch.empty(B, K, device=dev())
  dWd = ((dW0.t().contiguous() if wt else dW0).to(dev()).clone() if acc
         else torch.empty(K, N, device=dev()) if wt else torch.empty(N, K, device=dev()))
  db = torch.empty(N, device=dev())
  check(lib.rk_linear_bwd(ptr(dY), ptr(Y), ptr(Xd), ptr(Wd), B, N, K, wt, a, ptr(dX), ptr(dWd), acc, ptr(db),
                          st), "rk_linear_bwd")
  tol = dict(rtol=1e-4, atol=3e-6 * max(1.0, float(max(B, N)) ** 0.5))
  assert torch.allclose(dY.cpu().double(), gpre, rtol=1e-4, atol=1e-6)
  assert torch.allclose(dX.cpu().double(), gpre @ W.double(), **tol)
  want_dW = gpre.t() @ X.double() + (dW0.double() if acc else 0.0)
  got_dW = dWd.cpu().double()
  assert torch.allclose(got_dW.t() if wt else got_dW, want_dW, **tol)
  assert torch.allclose(db.cpu().double(), gpre.sum(0), **tol)
  # rk_linear_bwd_dact: the same call with act'(Xact) folded into dX's epilogue == the call above
  # followed by rk_act_grad, bit for bit (dW, db and dYpre unchanged)
  Xact = fn(f(B, K)).to(dev())
  dY2 = dY0.to(dev()).clone()
  dX2 = torch.empty(B, K, device=dev())
  dW2 = ((dW0.t().contiguous() if wt else dW0).to(dev()).clone() if acc
         else torch.empty(K, N, device=dev()) if wt else torch.empty(N, K, device=dev()))
  db2 = torch.empty(N, device=dev())
  check(lib.rk_linear_bwd_dact(ptr(dY2), ptr(Y), ptr(Xd), ptr(Wd), B, N, K, wt, a, ptr(dX2), ptr(dW2), acc,
                               ptr(db2), ptr(Xact), st), "rk_linear_bwd_dact")
  check(lib.rk_act_grad(ptr(dX), ptr(Xact), B * K, a, st), "rk_act_grad")
  torch.cuda.synchronize()
  # rk_linear_bwd_pre: dY already IS dYpre (what the call above left in dY2): dX * act'(Xact), dW and db in
  # one launch -- the same products bit for bit, db up to the order of its 8 row slices
  dX3 = torch.empty(B, K, device=dev())
  dW3 = ((dW0.t().contiguous() if wt else dW0).to(dev()).clone() if acc
         else torch.empty(K, N, device=dev()) if wt else torch.empty(N, K, device=dev()))
  db3 = torch.empty(N, device=dev())
  check(lib.rk_linear_bwd_pre(ptr(dY2), ptr(Xd), ptr(Wd), B, N, K, wt, a, ptr(dX3), ptr(dW3), acc, ptr(db3),
                              ptr(Xact), st), "rk_linear_bwd_pre")
  torch.cuda.synchronize()
  assert torch.equal(dX3, dX2) and torch.equal(dW3, dW2)
  assert torch.allclose(db3.cpu().double(), gpre.sum(0), **tol)
  lib.rk_linear_pair(1)          # (the default)
  assert torch.equal(dX2, dX) and torch.equal(dW2, dWd) and torch.equal(db2, db) and torch.equal(dY2, dY)


def test_hook_order_that_is_not_one_pass_takes_the_eager_path(monkeypatch):
  """A user_order_hook may return any list of users (here: a subset with a ragged tail).  The
  replayed graphs are laid out for one pass over the dataset, so such an epoch is sequenced eagerly
  with the same order -- same losses as with graph replay switched off, the hook asked once per
  epoch."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder
  csr = synth_csr(700, 400, 12, seed=77)
  sub = np.random.RandomState(3).permutation(700)[:300].astype(np.int64)

  def run(graph):
    monkeypatch.setenv("RK_GRAPH", "1" if graph else "0")
    torch.manual_seed(5)
    model = DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.0, sparse=False)
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss="mse")
    calls = []
    rec.user_order_hook = lambda epoch, n: calls.append(epoch) or sub
    rec.train(RecommendationDataset(csr), batch_size=128, lr=1e-3, weight_decay=1e-5, num_epochs=2,
              negative_sampling=True)
    assert calls == [1, 2]
    return np.concatenate(rec.loss_history)
  a, b = run(False), run(True)
  assert len(a) == 2 * 3 and np.array_equal(a, b)
