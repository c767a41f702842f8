"""Sparse-batch data API of the reference (recoder/data.py), backed by the
on-device collator.

Public names and semantics follow the reference: ``UsersInteractions``
(data.py:14-25), ``RecommendationDataset`` (data.py:28-83),
``RecommendationDataLoader`` (data.py:86-167), ``Batch`` (data.py:170-187) and
``BatchCollator`` (data.py:190-251).  The training fast path
(``Recoder.train``) never materialises ``Batch`` objects on the host: it keeps
the CSR in HBM (``RecommendationDataset.device_csr``) and collates with
``rk_collate``.  ``BatchCollator.collate`` exposes the same kernel through the
reference's host-side contract (list of ``Batch`` with COO tensors).
"""
import numpy as np
import scipy.sparse as sp
import torch

from .device import Block, DeviceCSR, require_gpu


class UsersInteractions:
  """Interactions of a set of users (data.py:14-25)."""

  def __init__(self, users, interactions_matrix):
    self.users = users
    self.interactions_matrix = interactions_matrix


def _is_index_like(index):
  return isinstance(index, (list, tuple, np.ndarray, int, np.integer))


class RecommendationDataset(torch.utils.data.Dataset):
  """User x item interactions (data.py:28-83) + lazily uploaded device copy."""

  def __init__(self, interactions_matrix, target_interactions_matrix=None):
    self.interactions_matrix = interactions_matrix
    self.target_interactions_matrix = target_interactions_matrix
    self.users = np.arange(self.interactions_matrix.shape[0])
    self.items = np.arange(self.interactions_matrix.shape[1])
    self._dev = None
    self._dev_target = None

  def __len__(self):
    return self.interactions_matrix.shape[0]

  def __getitem__(self, index):
    assert _is_index_like(index)
    users = np.array(index).reshape(-1,)
    rows = self.interactions_matrix[users]
    inp = UsersInteractions(users=users, interactions_matrix=rows)
    if self.target_interactions_matrix is None:
      return inp, None
    t_rows = self.target_interactions_matrix[users]
    return inp, UsersInteractions(users=users, interactions_matrix=t_rows)

  # ---- device residency (hot path) ----
  def device_csr(self):
    if self._dev is None:
      self._dev = DeviceCSR(self.interactions_matrix)
    return self._dev

  def device_target_csr(self):
    if self.target_interactions_matrix is None:
      return None
    if self._dev_target is None:
      self._dev_target = DeviceCSR(self.target_interactions_matrix)
    return self._dev_target


class Batch:
  """A sparse batch of users x items (data.py:170-187)."""

  def __init__(self, users, items, indices, values, size):
    self.users = users
    self.items = items
    self.indices = indices
    self.values = values
    self.size = size


class BatchCollator:
  """Collates ``UsersInteractions`` into ``Batch`` slices (data.py:190-251) on
  the GPU: the rows are uploaded, ``rk_collate`` builds the sorted-unique item
  set and the relabelled columns, and the result is handed back as the COO
  tensors the reference produces."""

  def __init__(self, batch_size, negative_sampling=False):
    self.batch_size = batch_size
    self.negative_sampling = negative_sampling

  def collate(self, users_interactions):
    dev = require_gpu()
    m = users_interactions.interactions_matrix
    S = m.shape[0]
    dcsr = DeviceCSR(m, dev)
    blk = Block(S, max(1, dcsr.nnz), m.shape[1], dev, negative_sampling=self.negative_sampling,
                need_bits_cr=False)
    rows = torch.arange(S, dtype=torch.int64, device=dev)
    blk.collate(dcsr, rows)
    h = blk.to_host()
    return batches_from_host_block(h, users_interactions.users, self.batch_size,
                                   self.negative_sampling, m.shape[1])


def batches_from_host_block(h, users, batch_size, negative_sampling, n_items):
  """Slice a downloaded block into the reference's list[Batch] (data.py:231-249)."""
  users_t = torch.as_tensor(np.asarray(users), dtype=torch.int64)
  if negative_sampling:
    items = torch.from_numpy(h["items"].astype(np.int64))
    vector_dim = int(h["n_b"])
  else:
    items = None
    vector_dim = int(n_items)
  indptr = h["indptr"].astype(np.int64)
  S = int(h["S"])
  slices = []
  for off in range(0, S, batch_size):
    end = min(S, off + batch_size)
    lo, hi = int(indptr[off]), int(indptr[end])
    counts = np.diff(indptr[off:end + 1])
    rows = np.repeat(np.arange(end - off, dtype=np.int64), counts)
    cols = h["cols"][lo:hi].astype(np.int64)
    indices = torch.from_numpy(np.stack([rows, cols]))
    values = torch.from_numpy(h["vals"][lo:hi].astype(np.float32))
    slices.append(Batch(users=users_t[off:end], items=items, indices=indices, values=values,
                        size=torch.Size([end - off, vector_dim])))
  return slices


def epoch_user_order(n):
  """The user permutation of one pass over the dataset, drawn exactly like the
  reference's loader does (data.py:124-136: torch ``DataLoader`` over
  ``BatchSampler(BatchSampler(RandomSampler))``): creating the DataLoader
  iterator draws a base seed from the global torch RNG, then ``RandomSampler``
  draws its own seed and permutes with a private generator."""
  torch.empty((), dtype=torch.int64).random_()            # DataLoader iterator _base_seed
  seed = int(torch.empty((), dtype=torch.int64).random_().item())
  g = torch.Generator()
  g.manual_seed(seed)
  return torch.randperm(n, generator=g).numpy().astype(np.int64)


class RecommendationDataLoader:
  """Iterates a ``RecommendationDataset`` in random sampling groups and yields
  ``(Batch, Batch | None)`` like the reference (data.py:86-167)."""

  def __init__(self, dataset, batch_size, negative_sampling=False, num_sampling_users=0,
               num_workers=0, collate_fn=None):
    self.dataset = dataset
    self.num_sampling_users = num_sampling_users
    self.num_workers = num_workers      # accepted for compatibility; collation is on the GPU
    self.batch_size = batch_size
    self.negative_sampling = negative_sampling
    if self.num_sampling_users == 0:
      self.num_sampling_users = batch_size
    assert self.num_sampling_users >= batch_size, \
        "num_sampling_users should be at least equal to the batch_size"
    self.batch_collator = BatchCollator(batch_size=self.batch_size,
                                        negative_sampling=self.negative_sampling)
    if collate_fn is None:
      self._collate_fn = self.batch_collator.collate
      self._use_default_data_generator = True
    else:
      self._collate_fn = collate_fn
      self._use_default_data_generator = False

  def _groups(self):
    order = epoch_user_order(len(self.dataset))
    S = self.num_sampling_users
    for off in range(0, len(order), S):
      idx = order[off:off + S]
      inp, tgt = self.dataset[idx]
      yield self._collate_fn(inp), (None if tgt is None else self._collate_fn(tgt))

  def __iter__(self):
    if not self._use_default_data_generator:
      return self._groups()
    return self._default_data_generator()

  def _default_data_generator(self):
    for inp, tgt in self._groups():
      for i in range(len(inp)):
        yield inp[i], (None if tgt is None else tgt[i])

  def __len__(self):
    return int(np.ceil(len(self.dataset) / self.batch_collator.batch_size))
