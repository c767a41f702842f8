"""Generate golden vectors from the REAL reference (build container only).

Imports amoussawi/recoder from /root/reference (read-only) behind four
in-memory shims, runs its own ``Recoder.train`` on tiny
seeded synthetic CSR matrices, and records -- per configuration -- the initial
state dict, every training ``Batch`` (users / items / indices / values), the
dropout keep-masks at the nnz positions, the per-step loss, parameter and
Adam-state snapshots, the validation loss and the top-k recommendations +
Recall/NDCG.  While doing so it asserts that ``oracle/recoder_oracle.py``
reproduces every recorded number bit-for-bit (that is the oracle's pin).

Only data (inputs and expected outputs) is written to ``tests/golden/*.npz``;
no reference source travels.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def import_reference():
  glog = types.ModuleType("glog")
  glog.info = lambda *a, **k: None
  sys.modules["glog"] = glog
  annoy = types.ModuleType("annoy")
  annoy.AnnoyIndex = object
  sys.modules["annoy"] = annoy
  import scipy.sparse._sputils as _spu
  import scipy.sparse.sputils as spu
  for n in ("issequence", "isintlike"):
    if not hasattr(spu, n):
      setattr(spu, n, getattr(_spu, n))
  np.int = int
  sys.path.insert(0, "/root/reference")
  import recoder  # noqa
  import tqdm
  import recoder.model as rmodel
  rmodel.tqdm = lambda it, **k: _Quiet(it)
  return recoder


class _Quiet:
  def __init__(self, it):
    self.it = it

  def set_postfix(self, *a, **k):
    pass

  def update(self, *a, **k):
    pass

  def close(self):
    pass


def _capture_dropout(layer, sink):
  """Record the true Bernoulli keep draw of an ``nn.Dropout`` without changing
  what the reference computes: save the RNG state before the layer runs, and
  afterwards replay ATen's ``empty_like(x).bernoulli_(1-p)`` from that state
  (checking it reproduces the layer's output bit-for-bit), then restore the
  post-call state."""
  st = {}

  def pre(mod, inp):
    if mod.training:
      st["rng"] = torch.get_rng_state()

  def post(mod, inp, out):
    if not mod.training:
      return
    after = torch.get_rng_state()
    torch.set_rng_state(st["rng"])
    keep = torch.empty_like(inp[0]).bernoulli_(1 - mod.p)
    assert torch.equal(inp[0].detach() * keep.clone().div_(1 - mod.p), out.detach())
    assert torch.equal(torch.get_rng_state(), after)
    torch.set_rng_state(after)
    sink.append(keep.to(torch.uint8))
  layer.register_forward_pre_hook(pre)
  layer.register_forward_hook(post)


def synth_csr(n_users, n_items, mean_deg, seed, ratings=False):
  rng = np.random.RandomState(seed)
  pop = 1.0 / np.arange(1, n_items + 1)
  pop /= pop.sum()
  rows, cols, vals = [], [], []
  for u in range(n_users):
    d = max(1, min(n_items // 2, int(rng.lognormal(np.log(mean_deg) - 0.5, 1.0))))
    it = np.unique(rng.choice(n_items, size=d, p=pop))
    rows += [u] * len(it)
    cols += list(it)
    if ratings:
      vals += list(rng.randint(1, 6, size=len(it)).astype(np.float32))
    else:
      vals += [1.0] * len(it)
  m = sp.coo_matrix((np.asarray(vals, dtype=np.float32), (rows, cols)),
                    shape=(n_users, n_items)).tocsr()
  m.sort_indices()
  return m


CONFIGS = {
  # name: dict(model kind, model kwargs, recoder kwargs, train kwargs, data kwargs)
  "ae_mse_dense": dict(
      kind="ae", model=dict(hidden_layers=[24], activation_type="tanh", noise_prob=0.5, sparse=False),
      loss="mse", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=3, negative_sampling=True,
                 lr_milestones=[3]),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=11), evaluate=True),
  "ae_mse_conf_sparse": dict(
      kind="ae", model=dict(hidden_layers=[24], activation_type="tanh", noise_prob=0.0, sparse=True),
      loss="mse", loss_params=dict(confidence=3),
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=12, ratings=True)),
  "ae2_logloss_dense": dict(
      kind="ae", model=dict(hidden_layers=[24, 16], activation_type="tanh", noise_prob=0.5,
                            dropout_prob=0.25, sparse=False),
      loss="logloss", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=13), evaluate=True),
  "ae2_constrained_bce": dict(
      kind="ae", model=dict(hidden_layers=[24, 16], activation_type="sigmoid", noise_prob=0.3,
                            is_constrained=True, sparse=False),
      loss="logistic", loss_params=None,
      train=dict(batch_size=32, lr=2e-3, weight_decay=1e-5, num_epochs=2, negative_sampling=True),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=14)),
  "ae_mse_sampling2": dict(
      kind="ae", model=dict(hidden_layers=[24], activation_type="relu", noise_prob=0.5, sparse=True),
      loss="mse", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=0.0, num_epochs=2, negative_sampling=True,
                 num_sampling_users=64),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=15)),
  "ae_mse_nosampling": dict(
      kind="ae", model=dict(hidden_layers=[24], activation_type="tanh", noise_prob=0.0, sparse=False),
      loss="mse", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=1, negative_sampling=False),
      data=dict(n_users=100, n_items=90, mean_deg=8, seed=16)),
  "mf_mse_sparse": dict(
      kind="mf", model=dict(embedding_size=16, activation_type="none", sparse=True),
      loss="mse", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=17)),
  "mf_bce_dense": dict(
      kind="mf", model=dict(embedding_size=16, activation_type="tanh", dropout_prob=0.3, sparse=False),
      loss="logistic", loss_params=None,
      train=dict(batch_size=32, lr=1e-3, weight_decay=2e-5, num_epochs=2, negative_sampling=True),
      data=dict(n_users=150, n_items=120, mean_deg=9, seed=18), evaluate=True),
}

SNAP_STEPS = (1, 2)   # parameter/optimizer snapshots after these many steps + final


def flat_state(prefix, st, out):
  for k, v in st.items():
    out["%s/%s" % (prefix, k)] = v.detach().cpu().numpy()


def adam_state_of(trainer, model):
  out = {}
  names = {id(p): n for n, p in model.named_parameters()}
  for opt in (trainer.optimizer, trainer.sparse_optimizer):
    if opt is None:
      continue
    for p, st in opt.state.items():
      if not st:
        continue
      n = names[id(p)]
      out[n + "/step"] = np.asarray(int(st["step"]))
      out[n + "/exp_avg"] = st["exp_avg"].detach().numpy().copy()
      out[n + "/exp_avg_sq"] = st["exp_avg_sq"].detach().numpy().copy()
  return out


def run_config(name, cfg, recoder_pkg):
  from recoder.model import Recoder
  from recoder.nn import DynamicAutoencoder, MatrixFactorization
  from recoder.data import RecommendationDataset
  from recoder.metrics import Recall, NDCG
  from oracle import recoder_oracle as orc

  csr = synth_csr(**cfg["data"])
  rng = np.random.RandomState(cfg["data"]["seed"] + 1000)
  # held-out target for validation/eval: a second, different matrix over the same users
  d2 = dict(cfg["data"]); d2["seed"] += 500
  csr_te = synth_csr(**d2)

  gold = {}
  gold["csr/indptr"] = csr.indptr.astype(np.int64)
  gold["csr/indices"] = csr.indices.astype(np.int32)
  gold["csr/data"] = csr.data.astype(np.float32)
  gold["csr/shape"] = np.asarray(csr.shape)
  gold["csr_te/indptr"] = csr_te.indptr.astype(np.int64)
  gold["csr_te/indices"] = csr_te.indices.astype(np.int32)
  gold["csr_te/data"] = csr_te.data.astype(np.float32)

  torch.manual_seed(1234)
  if cfg["kind"] == "ae":
    model = DynamicAutoencoder(**cfg["model"])
  else:
    model = MatrixFactorization(**cfg["model"])
  trainer = Recoder(model=model, use_cuda=False, optimizer_type="adam",
                    loss=cfg["loss"], loss_params=cfg["loss_params"])

  rec = dict(batches=[], noise=[], drop=[], losses=[], snaps={})
  state_holder = {}

  orig_init = model.init_model

  def init_model(num_items=None, num_users=None):
    orig_init(num_items, num_users)
    state_holder["init"] = {k: v.detach().clone() for k, v in model.named_parameters()}
    for attr, key in (("noise_layer", "noise"), ("dropout_layer", "drop")):
      layer = getattr(model, attr, None)
      if layer is not None:
        _capture_dropout(layer, rec[key])
  model.init_model = init_model

  orig_cl = trainer._Recoder__compute_loss

  def compute_loss(input, target):
    if model.training:
      k = len(rec["batches"])
      if k in SNAP_STEPS:
        rec["snaps"][k] = ({n: p.detach().clone() for n, p in model.named_parameters()},
                           adam_state_of(trainer, model))
      rec["batches"].append(input)
    loss = orig_cl(input, target)
    if model.training:
      rec["losses"].append(float(loss.item()))
    return loss
  trainer._Recoder__compute_loss = compute_loss

  train_ds = RecommendationDataset(csr)
  trainer.train(train_dataset=train_ds, **cfg["train"])
  final_state = {n: p.detach().clone() for n, p in model.named_parameters()}
  final_adam = adam_state_of(trainer, model)

  nsteps = len(rec["batches"])
  gold["nsteps"] = np.asarray(nsteps)
  flat_state("init", state_holder["init"], gold)
  flat_state("final", final_state, gold)
  for k, v in final_adam.items():
    gold["final_adam/" + k] = v
  for s, (st, ad) in rec["snaps"].items():
    flat_state("snap%d" % s, st, gold)
    for k, v in ad.items():
      gold["snap%d_adam/%s" % (s, k)] = v
  gold["losses"] = np.asarray(rec["losses"], dtype=np.float64)

  masks_noise, masks_drop = [], []
  for i, b in enumerate(rec["batches"]):
    gold["step%d/users" % i] = b.users.numpy()
    if b.items is not None:
      gold["step%d/items" % i] = b.items.numpy()
    gold["step%d/indices" % i] = b.indices.numpy()
    gold["step%d/values" % i] = b.values.numpy()
    gold["step%d/size" % i] = np.asarray(tuple(b.size))
    if rec["noise"]:
      idx = b.indices
      keep = rec["noise"][i][idx[0], idx[1]].numpy().astype(np.uint8)
      gold["step%d/noise_keep" % i] = keep
      masks_noise.append(keep)
    else:
      masks_noise.append(None)
    if rec["drop"]:
      keep = rec["drop"][i].numpy().astype(np.uint8)
      gold["step%d/drop_keep" % i] = keep
      masks_drop.append(keep)
    else:
      masks_drop.append(None)

  # ---- validation loss (model.py:439-452) with an independently collated target ----
  from recoder.data import RecommendationDataLoader
  val_ds = RecommendationDataset(csr, csr_te)
  torch.manual_seed(77)
  val_batches = []
  orig2 = trainer._Recoder__compute_loss

  def compute_loss_val(input, target):
    val_batches.append((input, target))
    return orig_cl(input, target)
  trainer._Recoder__compute_loss = compute_loss_val
  vdl = RecommendationDataLoader(val_ds, batch_size=cfg["train"]["batch_size"],
                                 negative_sampling=cfg["train"]["negative_sampling"],
                                 num_sampling_users=cfg["train"].get("num_sampling_users", 0))
  val_loss = trainer._validate(vdl)
  gold["val/loss"] = np.asarray(val_loss, dtype=np.float64)
  gold["val/nbatches"] = np.asarray(len(val_batches))
  for i, (bi, bt) in enumerate(val_batches):
    gold["val%d/users" % i] = bi.users.numpy()
    if bi.items is not None:
      gold["val%d/in_items" % i] = bi.items.numpy()
      gold["val%d/t_items" % i] = bt.items.numpy()

  # ---- evaluation (model.py:513-544, metrics.py) in fixed user order ----
  if cfg.get("evaluate"):
    from recoder.data import UsersInteractions
    K = 20
    users = np.arange(csr.shape[0])
    recs = []
    for off in range(0, len(users), 50):
      u = users[off:off + 50]
      ui = UsersInteractions(users=u, interactions_matrix=csr[u])
      recs += trainer.recommend(ui, K)
    recs = np.asarray(recs)
    gold["eval/topk"] = recs
    r20 = Recall(k=20); r5 = Recall(k=5); n20 = NDCG(k=20)
    vals = {"recall20": [], "recall5": [], "ndcg20": []}
    for i, u in enumerate(users):
      y = csr_te[u].nonzero()[1]
      vals["recall20"].append(r20.evaluate(recs[i], y))
      vals["recall5"].append(r5.evaluate(recs[i], y))
      vals["ndcg20"].append(n20.evaluate(recs[i], y))
    for k, v in vals.items():
      gold["eval/" + k] = np.asarray(np.mean(v))
    # full prediction for the first 8 users
    ui = UsersInteractions(users=users[:8], interactions_matrix=csr[users[:8]])
    out, _ = trainer.predict(ui)
    gold["eval/scores8"] = out.detach().numpy()

  # ---------------------------------------------------------------------
  # PIN THE ORACLE: replay with oracle/recoder_oracle.py, must be bit-exact
  # ---------------------------------------------------------------------
  torch.manual_seed(1234)
  if cfg["kind"] == "ae":
    st0 = orc.init_ae_state(csr.shape[1], cfg["model"]["hidden_layers"],
                            cfg["model"].get("is_constrained", False))
  else:
    st0 = orc.init_mf_state(csr.shape[1], csr.shape[0], cfg["model"]["embedding_size"])
  for k, v in state_holder["init"].items():
    assert torch.equal(st0[k], v), ("init mismatch", name, k)
  assert list(st0.keys()) == list(state_holder["init"].keys()), (list(st0.keys()), list(state_holder["init"].keys()))

  mk = cfg["model"]
  o = orc.OracleRecoder(cfg["kind"], st0,
                        hidden_layers=mk.get("hidden_layers"),
                        activation_type=mk.get("activation_type"),
                        is_constrained=mk.get("is_constrained", False),
                        noise_prob=mk.get("noise_prob", 0.0),
                        dropout_prob=mk.get("dropout_prob", 0.0),
                        sparse=mk.get("sparse", False),
                        loss=cfg["loss"], loss_params=cfg["loss_params"],
                        lr=cfg["train"]["lr"], weight_decay=cfg["train"]["weight_decay"])
  B = cfg["train"]["batch_size"]
  S = cfg["train"].get("num_sampling_users", 0) or B
  ns = cfg["train"]["negative_sampling"]
  steps_per_epoch = int(np.ceil(csr.shape[0] / B))
  milestones = cfg["train"].get("lr_milestones")
  i = 0
  while i < nsteps:
    epoch = i // steps_per_epoch + 1
    if milestones:
      o.set_lr(cfg["train"]["lr"] * (0.1 ** sum(1 for m in milestones if m <= epoch)))
    # the users of one sampling group = concat of the group's slices
    grp_users = []
    j = i
    while j < nsteps and len(grp_users) < S and (j // steps_per_epoch + 1) == epoch:
      grp_users += list(rec["batches"][j].users.numpy())
      j += 1
    batches = orc.collate(orc.extract_rows(csr, grp_users), grp_users, B, ns)
    assert len(batches) == j - i
    for b in batches:
      rb = rec["batches"][i]
      assert np.array_equal(b.indices, rb.indices.numpy()), (name, i)
      assert np.array_equal(b.values, rb.values.numpy())
      if ns:
        assert np.array_equal(b.items, rb.items.numpy())
      if i in SNAP_STEPS:
        for k, v in rec["snaps"][i][0].items():
          assert torch.equal(o.params[k].detach(), v), ("snap", name, i, k)
      loss = o.train_step(b, None, masks_noise[i], masks_drop[i])
      assert loss == rec["losses"][i], (name, i, loss, rec["losses"][i])
      i += 1
  for k, v in final_state.items():
    assert torch.equal(o.params[k].detach(), v), ("final", name, k)
  oad = o.adam_state()
  for k, (step, m, v) in oad.items():
    assert step == int(final_adam[k + "/step"])
    assert np.array_equal(m.numpy(), final_adam[k + "/exp_avg"]), (name, k)
    assert np.array_equal(v.numpy(), final_adam[k + "/exp_avg_sq"]), (name, k)
  if cfg.get("evaluate"):
    users = np.arange(csr.shape[0])
    recs = np.concatenate([o.recommend(csr[users[off:off + 50]], users[off:off + 50], 20)
                           for off in range(0, len(users), 50)])
    assert np.array_equal(recs, gold["eval/topk"]), name
    res = o.evaluate(csr, csr_te, 20, 50, [("recall", 20), ("recall", 5), ("ndcg", 20)])
    assert np.isclose(res[("recall", 20)], gold["eval/recall20"], rtol=1e-12)
    assert np.isclose(res[("recall", 5)], gold["eval/recall5"], rtol=1e-12)
    assert np.isclose(res[("ndcg", 20)], gold["eval/ndcg20"], rtol=1e-12)
  # validation loss replay
  o.training = False
  tot = 0.0
  for i, (bi, bt) in enumerate(val_batches):
    ob_in = orc.Batch(bi.users.numpy(), None if bi.items is None else bi.items.numpy(),
                      bi.indices.numpy(), bi.values.numpy(), tuple(bi.size))
    ob_t = orc.Batch(bt.users.numpy(), None if bt.items is None else bt.items.numpy(),
                     bt.indices.numpy(), bt.values.numpy(), tuple(bt.size))
    with torch.no_grad():
      tot += float(o.compute_loss(ob_in, ob_t).item())
  assert tot / len(val_batches) == val_loss, (name, tot / len(val_batches), val_loss)
  print("  [%s] %d steps, loss %.6f -> %.6f, val %.6f : oracle bit-exact"
        % (name, nsteps, rec["losses"][0], rec["losses"][-1], val_loss))
  return gold


def main():
  recoder_pkg = import_reference()
  only = sys.argv[1:] or list(CONFIGS)
  for name in only:
    gold = run_config(name, CONFIGS[name], recoder_pkg)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **gold)
    print("  wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
  main()
