"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (PyTorch-CPU eager + numpy/scipy, exactly the arithmetic
dependencies the reference itself uses: torch ATen ops, torch.optim.Adam /
SparseAdam, scipy CSR slicing, numpy.unique) of amoussawi/recoder's mini-batch
negative-sampling training path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``recoder_amd`` never does (it fails loudly without its HIP library).

Parity pin: this restatement is checked bit-for-bit against the *real*
reference imported in the build container (``tests/golden/make_golden.py``),
and the outputs of the real reference are committed as golden vectors under
``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).

The one thing the reference cannot give a GPU implementation is its RNG stream
(``nn.Dropout`` on the dense B x n_b tensor, ``RandomSampler``); so every
function here takes the user order and the dropout keep-masks as *inputs*
(captured from the reference when the golden vectors are generated).

Citations are ``file:line`` into the reference tree (v0.4.0).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

__all__ = [
    "Batch", "collate", "densify", "init_ae_state", "init_mf_state",
    "OracleRecoder", "recall", "ndcg", "average_precision",
]


# --------------------------------------------------------------------------
# data.py:170-251  -- Batch + BatchCollator.collate
# --------------------------------------------------------------------------
class Batch:
  """data.py:170-187 (numpy arrays instead of torch tensors)."""

  def __init__(self, users, items, indices, values, size):
    self.users = users      # int64 [rows]
    self.items = items      # int64 [n_b] or None
    self.indices = indices  # int64 [2, nnz]
    self.values = values    # float32 [nnz]
    self.size = size        # (rows, vector_dim)


def extract_rows(csr: sp.csr_matrix, users) -> sp.csr_matrix:
  """data.py:64-83 ``RecommendationDataset._extract`` (chunking is a scipy
  memory work-around and does not change the result)."""
  return csr[np.asarray(users).reshape(-1)]


def collate(csr_rows: sp.csr_matrix, users, batch_size: int,
            negative_sampling: bool):
  """data.py:203-251 ``BatchCollator.collate`` restated.

  ``csr_rows`` is the already row-gathered matrix (``dataset[users]``).
  """
  users = np.asarray(users, dtype=np.int64).reshape(-1)
  users_inds, items_inds = csr_rows.nonzero()                  # data.py:215
  if negative_sampling:
    batch_items, items_inds = np.unique(items_inds, return_inverse=True)  # :220
    vector_dim = len(batch_items)
    batch_items = batch_items.astype(np.int64)
  else:
    vector_dim = csr_rows.shape[1]                             # data.py:225
    batch_items = None
  slices = []
  cur = 0
  for off in range(0, csr_rows.shape[0], batch_size):          # data.py:231
    sl = csr_rows[off: off + batch_size]
    sl_users = users[off: off + batch_size]
    sl_rows = sl.nonzero()[0]
    nnz = sl.getnnz()
    sl_cols = items_inds[cur: cur + nnz]
    cur += nnz
    indices = np.stack([np.asarray(sl_rows, dtype=np.int64),
                        np.asarray(sl_cols, dtype=np.int64)])
    slices.append(Batch(users=sl_users, items=batch_items, indices=indices,
                        values=np.asarray(sl.data, dtype=np.float32),
                        size=(sl.shape[0], vector_dim)))
  return slices


def densify(batch: Batch) -> torch.Tensor:
  """model.py:457-458 COO -> dense (duplicates would sum; CSR has none)."""
  idx = torch.from_numpy(np.ascontiguousarray(batch.indices))
  val = torch.from_numpy(np.ascontiguousarray(batch.values))
  return torch.sparse_coo_tensor(idx, val, tuple(batch.size)).to_dense()


def dense_mask_from_nnz(batch: Batch, keep_nnz) -> torch.Tensor:
  """Scatter a per-nnz keep flag (row-major nnz order of ``batch.indices``)
  into a dense B x n_b 0/1 tensor; zero positions get keep=1 (irrelevant:
  dropout of a zero is zero)."""
  m = torch.ones(tuple(batch.size), dtype=torch.float32)
  idx = torch.from_numpy(np.ascontiguousarray(batch.indices))
  m[idx[0], idx[1]] = torch.from_numpy(np.asarray(keep_nnz, dtype=np.float32))
  return m


# --------------------------------------------------------------------------
# nn.py:145-226 / 314-330 -- parameter initialisation (consumes the global
# torch RNG in the same order as the reference's init_model)
# --------------------------------------------------------------------------
AE_EN_W = "en_embedding_layer.weight"
AE_EN_B = "_DynamicAutoencoder__en_linear_embedding_layer.bias"
AE_DE_W = "de_embedding_layer.weight"
AE_DE_B = "_DynamicAutoencoder__de_linear_embedding_layer.bias"


def _coding_layers(sizes):
  """nn.py:214-222."""
  out = []
  for ind in range(1, len(sizes)):
    lin = torch.nn.Linear(sizes[ind - 1], sizes[ind])
    torch.nn.init.xavier_uniform_(lin.weight)
    torch.nn.init.constant_(lin.bias, 0)
    out.append(lin)
  return out


def init_ae_state(num_items, hidden_layers, is_constrained=False):
  """Restates ``DynamicAutoencoder.init_model`` (nn.py:145-212): returns an
  OrderedDict of *parameters* in ``named_parameters()`` order."""
  h0 = hidden_layers[0]
  en = torch.nn.Embedding(num_items, h0)                       # nn.py:180
  en_b = torch.zeros(h0)                                       # nn.py:266,187
  enc = _coding_layers(hidden_layers)                          # nn.py:184
  torch.nn.init.xavier_uniform_(en.weight)                     # nn.py:186
  dec = _coding_layers(list(reversed(hidden_layers)))          # nn.py:190
  if not is_constrained:
    de = torch.nn.Embedding(num_items, h0)                     # nn.py:204
  else:
    de = en                                                    # nn.py:202
  torch.nn.init.xavier_uniform_(de.weight)                     # nn.py:211
  de_b = torch.zeros(num_items)                                # nn.py:212
  st = OrderedDict()
  st[AE_EN_W] = en.weight.detach().clone()
  st[AE_EN_B] = en_b
  for i, l in enumerate(enc):
    st["encoding_layers.%d.weight" % i] = l.weight.detach().clone()
    st["encoding_layers.%d.bias" % i] = l.bias.detach().clone()
  if not is_constrained:
    st[AE_DE_W] = de.weight.detach().clone()
  for i, l in enumerate(dec):
    if not is_constrained:
      st["decoding_layers.%d.weight" % i] = l.weight.detach().clone()
    st["decoding_layers.%d.bias" % i] = l.bias.detach().clone()
  st[AE_DE_B] = de_b
  return st


def init_mf_state(num_items, num_users, embedding_size):
  """Restates ``MatrixFactorization.init_model`` (nn.py:314-330)."""
  ue = torch.nn.Embedding(num_users, embedding_size)
  ie = torch.nn.Embedding(num_items, embedding_size)
  torch.nn.init.xavier_uniform_(ue.weight)
  torch.nn.init.xavier_uniform_(ie.weight)
  st = OrderedDict()
  st["bias"] = torch.zeros(num_items)
  st["user_embedding_layer.weight"] = ue.weight.detach().clone()
  st["item_embedding_layer.weight"] = ie.weight.detach().clone()
  return st


def _act(x, act):
  """nn.py:6-9."""
  if act == "none":
    return x
  return getattr(torch, act)(x)


def _dropout(x, keep, p):
  """ATen dropout: ``x * (bernoulli(1-p) / (1-p))``; ``keep`` is the injected
  0/1 Bernoulli draw."""
  noise = keep.to(torch.float32).div(1 - p)
  return x * noise


# --------------------------------------------------------------------------
# The trainer restatement (model.py:79-164, 383-404, 454-485, 487-544)
# --------------------------------------------------------------------------
class OracleRecoder:
  """Functional restatement of ``Recoder`` for DynamicAutoencoder ('ae') and
  MatrixFactorization ('mf') with adam / sparse-adam, fed explicit batches."""

  def __init__(self, kind, state, *, hidden_layers=None, activation_type=None,
               is_constrained=False, noise_prob=0.0, dropout_prob=0.0,
               sparse=False, loss="mse", loss_params=None,
               lr=1e-3, weight_decay=0.0, optimizer_type="adam"):
    assert kind in ("ae", "mf")
    self.kind = kind
    self.hidden_layers = hidden_layers
    self.act = activation_type if activation_type is not None else \
        ("tanh" if kind == "ae" else "none")
    self.is_constrained = is_constrained
    self.noise_prob = float(noise_prob)
    self.dropout_prob = float(dropout_prob)
    self.sparse = sparse
    self.loss = loss
    self.loss_params = loss_params or {}
    self.training = True
    self.params = OrderedDict(
        (k, torch.nn.Parameter(v.detach().clone().float()))
        for k, v in state.items())
    self._init_optimizer(lr, weight_decay, optimizer_type)

  # model.py:101-164
  def _init_optimizer(self, lr, weight_decay, optimizer_type):
    if self.kind == "ae":
      sparse_names = [AE_EN_W, AE_DE_W] if self.sparse else []
    else:
      sparse_names = (["user_embedding_layer.weight",
                       "item_embedding_layer.weight"] if self.sparse else [])
    groups, sgroups = [], []
    for name, p in self.params.items():
      wd = 0 if "bias" in name else weight_decay               # model.py:123
      g = {"params": p, "weight_decay": wd}
      (sgroups if name in sparse_names else groups).append(g)
    self.optimizer = None
    self.sparse_optimizer = None
    if optimizer_type == "adam":
      if groups:
        self.optimizer = torch.optim.Adam(groups, lr=lr)       # model.py:135
      if sgroups:
        self.sparse_optimizer = torch.optim.SparseAdam(sgroups, lr=lr)  # :138
    elif optimizer_type == "sgd":
      assert not sgroups
      self.optimizer = torch.optim.SGD(groups, lr=lr, momentum=0.9)
    else:
      raise ValueError(optimizer_type)

  def set_lr(self, lr):
    """MultiStepLR touches only the dense optimizer (model.py:329)."""
    for g in self.optimizer.param_groups:
      g["lr"] = lr

  # nn.py:269-280
  def _linear_embedding(self, weight, bias, idx, y, input_based):
    if idx is not None:
      w = F.embedding(idx, weight, sparse=self.sparse)
      b = bias if input_based else bias.index_select(0, idx)
    else:
      w, b = weight, bias
    return F.linear(y, w.t(), b) if input_based else F.linear(y, w, b)

  # nn.py:228-253
  def _ae_forward(self, x, input_items, target_items, noise_keep, drop_keep):
    z = self._ae_hidden(x, input_items, noise_keep, drop_keep)
    P = self.params
    de_w = P[AE_EN_W] if self.is_constrained else P[AE_DE_W]
    return self._linear_embedding(de_w, P[AE_DE_B], target_items, z, False)

  # nn.py:228-250: everything before the output embedding (what the decoder GEMM reads)
  def _ae_hidden(self, x, input_items, noise_keep, drop_keep):
    P = self.params
    nl = len(self.hidden_layers) - 1
    z = F.normalize(x, p=2, dim=1)                             # nn.py:235
    if self.noise_prob > 0.0 and self.training:
      z = _dropout(z, noise_keep, self.noise_prob)             # nn.py:237
    z = self._linear_embedding(P[AE_EN_W], P[AE_EN_B], input_items, z, True)
    z = _act(z, self.act)                                      # nn.py:240
    for i in range(nl):                                        # nn.py:242
      z = _act(F.linear(z, P["encoding_layers.%d.weight" % i],
                        P["encoding_layers.%d.bias" % i]), self.act)
    if self.dropout_prob > 0.0 and self.training:              # nn.py:245
      z = _dropout(z, drop_keep, self.dropout_prob)
    for i in range(nl):                                        # nn.py:248
      if self.is_constrained:                                  # nn.py:224-226
        w = P["encoding_layers.%d.weight" % (nl - 1 - i)].t()
      else:
        w = P["decoding_layers.%d.weight" % i]
      z = _act(F.linear(z, w, P["decoding_layers.%d.bias" % i]), self.act)
    return z

  # nn.py:344-362
  def _mf_forward(self, input_users, target_items, drop_keep):
    P = self.params
    u = F.embedding(input_users, P["user_embedding_layer.weight"],
                    sparse=self.sparse)
    u = _act(u, self.act)
    if self.dropout_prob > 0 and self.training:
      u = _dropout(u, drop_keep, self.dropout_prob)
    if target_items is None:
      iw, b = P["item_embedding_layer.weight"], P["bias"]
    else:
      iw = F.embedding(target_items, P["item_embedding_layer.weight"],
                       sparse=self.sparse)
      b = P["bias"].index_select(0, target_items)
    return F.linear(u, iw, b)

  # losses.py:43-47, 68-71; model.py:90-95 (all reduction='sum')
  def _loss(self, out, target):
    if self.loss == "mse":
      conf = self.loss_params.get("confidence", 0)
      w = 1 + conf * (target > 0).float()
      return (w * F.mse_loss(out, target, reduction="none")).sum()
    if self.loss == "logloss":
      return (-target * F.log_softmax(out, dim=1)).sum()
    if self.loss == "logistic":
      return F.binary_cross_entropy_with_logits(out, target, reduction="sum")
    raise ValueError(self.loss)

  def forward(self, batch: Batch, target: Batch = None, noise_keep=None,
              drop_keep=None):
    """model.py:454-485 up to the model call; returns (output, target_dense)."""
    x = densify(batch)
    in_items = None if batch.items is None else torch.from_numpy(batch.items)
    in_users = torch.from_numpy(np.asarray(batch.users, dtype=np.int64))
    if target is not None:
      t = densify(target)
      t_items = None if target.items is None else torch.from_numpy(target.items)
    else:
      t, t_items = x, in_items
    nk = None
    if noise_keep is not None:
      nk = dense_mask_from_nnz(batch, noise_keep)
    dk = None
    if drop_keep is not None:
      dk = torch.from_numpy(np.asarray(drop_keep, dtype=np.float32))
    if self.kind == "ae":
      out = self._ae_forward(x, in_items, t_items, nk, dk)
    else:
      out = self._mf_forward(in_users, t_items, dk)
    return out, t

  def decoder_input(self, batch: Batch, noise_keep=None, drop_keep=None):
    """The activations the output embedding multiplies (autoencoder; tests that size them)."""
    in_items = None if batch.items is None else torch.from_numpy(batch.items)
    nk = None if noise_keep is None else dense_mask_from_nnz(batch, noise_keep)
    dk = None if drop_keep is None else torch.from_numpy(np.asarray(drop_keep, dtype=np.float32))
    with torch.no_grad():
      return self._ae_hidden(densify(batch), in_items, nk, dk)

  def compute_loss(self, batch, target=None, noise_keep=None, drop_keep=None):
    out, t = self.forward(batch, target, noise_keep, drop_keep)
    norm = torch.FloatTensor([t.size(0)])                      # model.py:483
    return self._loss(out, t) / norm

  def train_step(self, batch, target=None, noise_keep=None, drop_keep=None):
    """model.py:383-404 one iteration. Returns the python float loss."""
    self.training = True
    if self.optimizer is not None:
      self.optimizer.zero_grad()
    if self.sparse_optimizer is not None:
      self.sparse_optimizer.zero_grad()
    loss = self.compute_loss(batch, target, noise_keep, drop_keep)
    loss.backward()
    if self.optimizer is not None:
      self.optimizer.step()
    if self.sparse_optimizer is not None:
      self.sparse_optimizer.step()
    return float(loss.item())

  def train_step_ddp(self, batches, noise_keeps=None):
    """One iteration of the reference trainer under conventional data parallelism (what wrapping its model in
    torch DistributedDataParallel does; the reference itself has no multi-device code): every rank runs
    model.py:454-485 on ITS OWN collated batch -- its own sampled item set --, the losses are averaged over the
    ranks (DDP averages the gradients), ONE optimizer step follows (model.py:397-402).  batches: one Batch per
    rank, equally sized.  Returns the averaged loss.  NOT the single-process semantics of a shared item set
    (data.py:216-223): the oracle of recoder_amd's opt-in RK_DP_ITEMSETS=local mode."""
    self.training = True
    if self.optimizer is not None:
      self.optimizer.zero_grad()
    if self.sparse_optimizer is not None:
      self.sparse_optimizer.zero_grad()
    total = 0.0
    n = len(batches)
    for r, b in enumerate(batches):
      loss = self.compute_loss(b, None, None if noise_keeps is None else noise_keeps[r], None) / n
      loss.backward()                    # (.grad accumulates over the ranks' batches)
      total += float(loss.item())
    if self.optimizer is not None:
      self.optimizer.step()
    if self.sparse_optimizer is not None:
      self.sparse_optimizer.step()
    return total

  def grads(self):
    out = {}
    for k, p in self.params.items():
      g = p.grad
      if g is None:
        continue
      out[k] = g.to_dense().clone() if g.is_sparse else g.clone()
    return out

  def state(self):
    return OrderedDict((k, p.detach().clone()) for k, p in self.params.items())

  def adam_state(self):
    """{param name: (step, exp_avg, exp_avg_sq)} for both optimizers."""
    out = {}
    for opt in (self.optimizer, self.sparse_optimizer):
      if opt is None:
        continue
      for name, p in self.params.items():
        st = opt.state.get(p)
        if st:
          out[name] = (int(st["step"]), st["exp_avg"].clone(),
                       st["exp_avg_sq"].clone())
    return out

  # model.py:487-511, 525-544
  @torch.no_grad()
  def predict(self, csr_rows, users):
    self.training = False
    b = collate(csr_rows, users, len(users), negative_sampling=False)[0]
    out, x = self.forward(b)
    return out, x

  @torch.no_grad()
  def recommend(self, csr_rows, users, k):
    out, x = self.predict(csr_rows, users)
    out[x > 0] = -float("inf")                                 # model.py:538
    _, top = torch.topk(out, k, dim=1, sorted=True)            # model.py:540
    return top.numpy()

  def evaluate(self, csr_in, csr_target, k, batch_size, metrics):
    """metrics.py:148-232 (single process) with an in-order user sweep."""
    res = {m: [] for m in metrics}
    n = csr_in.shape[0]
    for off in range(0, n, batch_size):
      users = np.arange(off, min(off + batch_size, n))
      rec = self.recommend(csr_in[users], users, k)
      for i, u in enumerate(users):
        y = csr_target[u].nonzero()[1]
        for m in metrics:
          name, kk = m
          res[m].append(METRICS[name](rec[i], y, kk))
    return {m: float(np.mean(v)) for m, v in res.items()}


# --------------------------------------------------------------------------
# metrics.py:9-45 (np.int -> int)
# --------------------------------------------------------------------------
def average_precision(x, y, k, normalize=True):
  x = np.asarray(x)[:k]
  hit = np.isin(x, y, assume_unique=True).astype(int)
  tp = hit.cumsum()
  prec = tp / (1 + np.arange(len(x)))
  norm = min(k, len(y)) if normalize else len(y)
  return float(np.multiply(prec, hit).sum() / norm)


def recall(x, y, k, normalize=True):
  x = np.asarray(x)[:k]
  hit = np.isin(x, y, assume_unique=True).astype(int)
  norm = min(k, len(y)) if normalize else len(y)
  return float(hit.sum() / norm)


def dcg(x, y, k):
  x = np.asarray(x)[:k]
  hit = np.isin(x, y, assume_unique=True).astype(int)
  return float((hit / np.log2(2 + np.arange(len(x)))).sum())


def ndcg(x, y, k):
  return dcg(x, y, k) / dcg(y, y, k)


METRICS = {"recall": recall, "ndcg": ndcg, "ap": average_precision}
