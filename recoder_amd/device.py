"""Device-resident data structures of the hot path.

DeviceCSR  -- the user x item interaction matrix in HBM (replaces the scipy
              matrix held by the reference's RecommendationDataset,
              data.py:41-48): int64 indptr, int32 indices, fp32 data.
Block      -- buffers of one collated sampling group (``rk_block_t``): what the
              reference's BatchCollator.collate (data.py:203-251) returns as a
              list of ``Batch`` sharing one item set, kept on the device.
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from ._lib import RkBlock, SCAN_CHUNK, check, ptr


def cdiv(a, b):
  return (a + b - 1) // b


def current_stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
  if not torch.cuda.is_available():
    raise _lib.RecoderHipError(
        "recoder_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
        "there is no CPU fallback for the training path")
  return torch.device("cuda", torch.cuda.current_device())


def canonical_csr(m):
  """Sorted indices, no duplicates, no explicit zeros, fp32 data (a copy).

  The reference drops explicit zeros in ``nonzero()`` (data.py:215) but counts
  them in ``getnnz()`` (data.py:238) -- a latent mismatch; here they are removed
  up front, which is what its dense arithmetic sees anyway."""
  m = sp.csr_matrix(m, copy=True)
  m.sum_duplicates()
  m.eliminate_zeros()
  m.sort_indices()
  if m.nnz and m.indices.max() >= 2 ** 31:
    raise ValueError("item ids must fit in int32")
  return m


class DeviceCSR:
  """The interaction matrix in HBM: int64 indptr, int32 indices, fp32 data (elided when every
  value is 1.0).  Built from a scipy matrix (canonicalised on the host), from arrays that are
  already canonical (``from_arrays``: nothing but the upload happens on the host -- .npz files,
  shards of a large matrix) or generated on the device (recoder_amd.synthetic)."""

  def __init__(self, matrix, device=None):
    device = device or require_gpu()
    m = canonical_csr(matrix)
    self._init_arrays(m.shape, torch.from_numpy(m.indptr.astype(np.int64)),
                      torch.from_numpy(m.indices.astype(np.int32)),
                      torch.from_numpy(m.data.astype(np.float32)), device)

  def _init_arrays(self, shape, indptr, indices, data, device):
    self.shape = (int(shape[0]), int(shape[1]))
    self.device = device
    self.indptr = indptr.to(device=device, dtype=torch.int64).contiguous()
    self.nnz = int(self.indptr[-1].item()) if self.indptr.numel() else 0
    self.degrees = (self.indptr[1:] - self.indptr[:-1]).cpu().numpy().astype(np.int64)
    self.indices = indices.to(device=device, dtype=torch.int32).contiguous()
    if data is not None:
      data = data.to(device=device, dtype=torch.float32).contiguous()
    self.implicit = bool(self.nnz == 0 or data is None or bool((data == 1.0).all().item()))
    # implicit-feedback matrices (all values 1.0) elide the value stream
    self.data = None if self.implicit else data
    if self.nnz == 0:
      self.indices = torch.zeros(1, dtype=torch.int32, device=device)

  @classmethod
  def from_arrays(cls, shape, indptr, indices, data=None, device=None, check=True):
    """CSR arrays (numpy or torch, host or device) that are ALREADY canonical: column indices
    strictly ascending inside every row, no explicit zeros.  ``check`` verifies that on the
    device (one pass of element-wise kernels) and raises ValueError otherwise."""
    device = device or require_gpu()
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    self = cls.__new__(cls)
    self._init_arrays(shape, as_t(indptr), as_t(indices), None if data is None else as_t(data), device)
    if check:
      # (also for an empty matrix: the collation kernels index indices / data through indptr)
      ip = self.indptr
      if ip.numel() != self.shape[0] + 1 or int(ip[0].item()) != 0:
        raise ValueError("indptr must start at 0 and have n_rows + 1 entries")
      if ip.numel() > 1 and bool((ip[1:] < ip[:-1]).any().item()):
        raise ValueError("indptr must be non-decreasing")
      n_idx = int(as_t(indices).numel())
      if int(ip[-1].item()) > n_idx or (data is not None and int(as_t(data).numel()) < int(ip[-1].item())):
        raise ValueError("indptr[-1] exceeds len(indices) (or len(data))")
    if check and self.nnz:
      idx = self.indices[:self.nnz].to(torch.int64)
      if int(idx.min().item()) < 0 or int(idx.max().item()) >= self.shape[1]:
        raise ValueError("column index out of range")
      # ascending inside a row <=> idx[j] > idx[j-1] wherever j is not a row start
      is_start = torch.zeros(self.nnz + 1, dtype=torch.bool, device=device)
      is_start[self.indptr.clamp(max=self.nnz)] = True
      bad = (idx[1:] <= idx[:-1]) & ~is_start[1:self.nnz]
      if bool(bad.any().item()):
        raise ValueError("column indices must be strictly ascending inside each row "
                         "(sorted, no duplicates); pass the matrix itself to DeviceCSR(...) instead")
      if self.data is not None and bool((self.data[:self.nnz] == 0).any().item()):
        raise ValueError("explicit zeros are not allowed")
    return self

  @classmethod
  def from_npz(cls, path, device=None):
    """A matrix saved with ``scipy.sparse.save_npz`` (what the reference's preprocessing scripts
    write, scripts/*/preprocess.py) straight into HBM; canonical CSR files skip the host-side
    copy + sort of ``DeviceCSR(matrix)``, anything else falls back to it."""
    with np.load(path, allow_pickle=False) as z:
      fmt = z["format"].item()
      fmt = fmt.decode() if isinstance(fmt, bytes) else str(fmt)
      if fmt == "csr":
        try:
          return cls.from_arrays(tuple(z["shape"]), z["indptr"], z["indices"], z["data"], device)
        except ValueError:
          pass
    return cls(sp.load_npz(path), device)

  def row_slice(self, lo, hi):
    """Rows [lo, hi) as a DeviceCSR of its own (device-side copy; data-parallel shards)."""
    lo, hi = int(lo), int(hi)
    a, b = int(self.indptr[lo].item()), int(self.indptr[hi].item())
    out = DeviceCSR.__new__(DeviceCSR)
    out._init_arrays((hi - lo, self.shape[1]), self.indptr[lo:hi + 1] - a, self.indices[a:max(b, a + 1)]
                     if b > a else torch.zeros(1, dtype=torch.int32, device=self.device),
                     None if self.data is None else self.data[a:b], self.device)
    return out

  @property
  def n_items(self):
    return self.shape[1]


class Block:
  """Capacity-sized device buffers for one sampling group + the ctypes view."""

  def __init__(self, S_cap, nnz_cap, n_items, device=None, negative_sampling=True,
               need_bits_cr=True, n_cap=None):
    device = device or require_gpu()
    S_cap = max(1, int(S_cap))
    nnz_cap = max(1, int(nnz_cap))
    self.device = device
    self.S_cap, self.nnz_cap, self.n_items = S_cap, nnz_cap, int(n_items)
    self.negative_sampling = bool(negative_sampling)
    # n_cap override: a data-parallel union item set can exceed one rank's nnz bound
    self.n_cap = min(self.n_items, n_cap or nnz_cap) if negative_sampling else self.n_items
    self.n_cap = max(1, self.n_cap)
    self.ld_cap = cdiv(self.n_cap, 32) * 32
    self.ldw_rc = cdiv(self.n_cap, 32)
    self.ldw_cr = cdiv(S_cap, 32)
    self.n_chunks = cdiv(self.n_items, SCAN_CHUNK)
    i32 = dict(dtype=torch.int32, device=device)
    self.counts = torch.zeros(72, **i32)    # rk_block_t.counts: 4 sizes + spare + 64 amax slots
    self.indptr = torch.zeros(S_cap + 1, **i32)
    self.cols = torch.zeros(nnz_cap, **i32)
    self.vals = torch.zeros(nnz_cap, dtype=torch.float32, device=device)
    self.svals = torch.zeros(nnz_cap, dtype=torch.float32, device=device)
    self.items = torch.zeros(self.n_cap, **i32)
    self.pos = torch.full((self.n_items,), -1, **i32)
    self.mark = torch.zeros(self.n_items, **i32)
    self.bits_rc = torch.zeros(S_cap * self.ldw_rc, **i32)
    # the transposed bitmap is only needed by the encoder backward (training)
    self.bits_cr = torch.zeros(self.n_cap * self.ldw_cr, **i32) if need_bits_cr else None
    self.scan_tmp = torch.zeros(2 * (self.n_chunks + 1), **i32)      # (64-bit slots: rk_block_t.scan_tmp)
    self.pref_rc = torch.zeros(S_cap * self.ldw_rc, **i32)
    self.gcols = torch.zeros(nnz_cap, **i32)
    self.stamp = 0
    self.users = None      # int64 device tensor of the rows of the last collate
    self.S = 0
    self.c = RkBlock(
        S_cap=S_cap, nnz_cap=nnz_cap, n_cap=self.n_cap, n_items=self.n_items,
        ldw_rc=self.ldw_rc, ldw_cr=self.ldw_cr, n_chunks=self.n_chunks, implicit=0,
        counts=ptr(self.counts), indptr=ptr(self.indptr), cols=ptr(self.cols),
        vals=ptr(self.vals), svals=ptr(self.svals), items=ptr(self.items), pos=ptr(self.pos),
        mark=ptr(self.mark), bits_rc=ptr(self.bits_rc), bits_cr=ptr(self.bits_cr),
        scan_tmp=ptr(self.scan_tmp), pref_rc=ptr(self.pref_rc), gcols=ptr(self.gcols))
    self.ref = ctypes.byref(self.c)

  def collate(self, dcsr, users_dev, negative_sampling=None, phase=0):
    """users_dev: int64 device tensor of user (row) ids of the group.
    phase 0: whole collation; 1: rows + marking; 2: the rest (data parallel)."""
    ns = self.negative_sampling if negative_sampling is None else bool(negative_sampling)
    S = int(users_dev.numel())
    assert users_dev.dtype == torch.int64 and users_dev.is_cuda
    assert dcsr.n_items == self.n_items
    if phase != 2:
      self.stamp += 1
      if self.stamp >= 2 ** 31 - 1:
        self.mark.zero_()
        self.stamp = 1
    self.users, self.S = users_dev, S
    self.c.implicit = 1 if dcsr.data is None else 0
    lib = _lib.load()
    check(lib.rk_collate(ptr(dcsr.indptr), ptr(dcsr.indices), ptr(dcsr.data), ptr(users_dev), S,
                         1 if ns else 0, self.stamp, phase, self.ref, current_stream()),
          "rk_collate")
    return self

  def set_items(self, items_dev_i32, n, S):
    """Describe a target item set without interactions (predict path):
    counts = (n, 0, round_up(n,32), S)."""
    n = int(n)
    assert n <= self.n_cap
    self.items[:n].copy_(items_dev_i32[:n])
    self.counts[:4].copy_(torch.tensor([n, 0, cdiv(n, 32) * 32, S], dtype=torch.int32), non_blocking=False)
    self.S = S

  def host_n_b(self):
    """n_b on the host.  After a CollatePrefetcher.submit this waits only for the
    (long finished) async copy; otherwise it synchronises on the counts."""
    ev = getattr(self, "counts_event", None)
    if ev is not None:
      ev.synchronize()
      return int(self.counts_pinned[0])
    return int(self.counts[0].item())

  # ---- host views (synchronising; tests / API compatibility only) ----
  def counts_host(self):
    c = self.counts.cpu().numpy()
    self._raise_if_truncated(int(c[5]) if c[5] else -int(c[6]))
    return int(c[0]), int(c[1]), int(c[2]), int(c[3])

  def _raise_if_truncated(self, n_all):
    if n_all:
      from ._lib import RecoderHipError
      if n_all < 0:
        raise RecoderHipError("a collated block held %d stored interactions but was sized for %d "
                              "(rk_collate dropped the surplus in bounds; everything computed from it "
                              "is wrong)" % (-n_all, self.nnz_cap))
      raise RecoderHipError("a collated block held %d distinct items but was sized for %d (rk_collate "
                            "truncated it in bounds; everything computed from it is wrong)"
                            % (n_all, self.n_cap))

  def check(self):
    """Raise if the last collation into this block overflowed its item capacity (counts[5]: the
    device clamps in bounds and leaves the true count there).  One 4-byte read-back: call it where
    the stream is drained anyway (the end of an epoch)."""
    f = self.counts[5:7].cpu().numpy()
    self._raise_if_truncated(int(f[0]) if f[0] else -int(f[1]))

  def to_host(self):
    n_b, nnz, ld, S = self.counts_host()
    return dict(n_b=n_b, nnz=nnz, ld=ld, S=S,
                indptr=self.indptr[:S + 1].cpu().numpy(),
                cols=self.cols[:nnz].cpu().numpy(),
                vals=self.vals[:nnz].cpu().numpy(),
                items=self.items[:n_b].cpu().numpy().astype(np.int64),
                pos=self.pos.cpu().numpy())


class CollatePrefetcher:
  """Double-buffered collation on a side HIP stream: the blocks of the next
  `group` steps are collated while the current ones train (the reference gets the
  same overlap from its DataLoader worker processes, data.py:135-136).

  Each of the two slots holds `group` blocks and ONE pair of events: a
  cross-stream dependency costs 10-20 us of latency on the stream it lands on
  (tools/sync_cost2.py), so the hand-over is paid once per `group` steps instead
  of once per step."""

  def __init__(self, make_block, dcsr, device=None, collate_fn=None, group=1):
    self.device = device or require_gpu()
    self.dcsr = dcsr
    self.group = int(group)
    self.blocks = [[make_block() for _ in range(self.group)] for _ in range(2)]
    self.stream = torch.cuda.Stream(device=self.device)
    self.ready = [torch.cuda.Event(), torch.cuda.Event()]
    self.free = [torch.cuda.Event(), torch.cuda.Event()]
    self._used = [False, False]
    self._count = [0, 0]
    self.collate_fn = collate_fn      # e.g. DataParallel.collate (two-phase, union item set)

  def reset(self):
    """Forget the hand-over state (a consumer abandoned its slots): the next
    submits order themselves after everything enqueued on the current stream."""
    self._used = [False, False]

  def submit(self, slot, users_list):
    """Enqueue the collation of up to `group` user batches into the blocks of
    `slot` on the side stream."""
    assert 0 < len(users_list) <= self.group
    main = torch.cuda.current_stream()
    with torch.cuda.stream(self.stream):
      if self._used[slot]:
        self.stream.wait_event(self.free[slot])      # previous consumer of these buffers is done
      else:
        self.stream.wait_stream(main)                # first use: order after setup work
      for blk, users_dev in zip(self.blocks[slot], users_list):
        if self.collate_fn is not None:
          self.collate_fn(blk, self.dcsr, users_dev)
        else:
          blk.collate(self.dcsr, users_dev)
        # the device-resident counts also go to pinned host memory: consumers that need
        # n_b on the host (RCCL message sizes) read it later without stalling a stream
        if getattr(blk, "counts_pinned", None) is None:
          blk.counts_pinned = torch.zeros(4, dtype=torch.int32).pin_memory()
          blk.counts_event = torch.cuda.Event()
        blk.counts_pinned.copy_(blk.counts[:4], non_blocking=True)
        blk.counts_event.record(self.stream)
      self.ready[slot].record(self.stream)
    self._count[slot] = len(users_list)
    return self.blocks[slot][:len(users_list)]

  def acquire(self, slot):
    """Make the current stream wait for the blocks of `slot`; returns them."""
    torch.cuda.current_stream().wait_event(self.ready[slot])
    return self.blocks[slot][:self._count[slot]]

  def release(self, slot):
    """Call after the last kernel reading the blocks of `slot` was enqueued."""
    self.free[slot].record(torch.cuda.current_stream())
    self._used[slot] = True
