"""Data parallelism over users (one process per GPU, torch.distributed: the
"nccl" backend is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no multi-device code; the exact
formulation comes from its own "one shared item set, several row blocks"
mechanism (num_sampling_users = k * batch_size, data.py:216-223,231-249):

  * every rank takes B_local users of its own shard per step;
  * the item set is the UNION over all ranks' users: all-reduce(MAX) of the
    per-item stamp array between the two phases of rk_collate (n_items int32s);
  * loss and gradients are normalised by the global row count N * B_local;
  * the compact gradient rows [n_b, h] (same n_b, same row order on every
    rank), the gathered-bias gradient [n_b], the small dense gradients and the
    scalar loss are all-reduced (SUM); every replica then applies the identical
    fused Adam -- bit-for-bit the mathematics of a single process running
    batch_size = N * B_local.

Item parallelism (class ItemParallel, RK_PARALLEL=items) is the second formulation, kept as
an opt-in: the ITEM dimension is sharded instead
of the users (item i lives on rank i % N: embedding rows, their Adam moments and
the matching columns of the interaction matrix).  Every rank processes all
N * B users of the global batch against its own items; the only exchange is an
all-reduce (SUM) of two [N*B, h] matrices per step (partial encoder sums, partial
dLoss/dZ) -- 2 x 3.2 MB at N = 8, h = 200 instead of 2 x 16 MB of gradient rows --
and the Adam sweep shrinks to the owned 1/N of the tables.  Same mathematics as
the single-process run with batch_size = N * B.

Nothing here touches the HIP library, so the same code runs under gloo on CPU
tensors in tests/test_parallel.py.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
  """Contiguous, balanced row range [lo, hi) of rank's shard."""
  base, rem = divmod(n, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def union_marks(mark, group=None):
  """mark: int32 [n_items] generation stamps (same stamp on every rank for the
  same step) -> after MAX all-reduce an item carries the stamp iff any rank
  touched it."""
  dist.all_reduce(mark, op=dist.ReduceOp.MAX, group=group)
  return mark


def allreduce_sum(views, group=None, small_threshold=65536):
  """SUM all-reduce of a list of tensors/views in place.  Large ones (the
  gradient row blocks) go as they are -- one collective each, sized for the
  per-link xGMI bandwidth; the small ones are coalesced into one flat bucket."""
  small = [v for v in views if v.numel() <= small_threshold]
  large = [v for v in views if v.numel() > small_threshold]
  for v in large:
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
  if small:
    flat = torch.cat([v.reshape(-1) for v in small])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for v in small:
      n = v.numel()
      v.copy_(flat[off:off + n].view_as(v))
      off += n


def sync_owned_rows(tensors, n_rows, group=None):
  """Each rank owns the contiguous row range shard_range(n_rows, rank, world) of every
  tensor in ``tensors`` (first dimension n_rows): after the call every replica holds the
  owners' rows.  MatrixFactorization under data parallelism: a user's embedding row (and
  its optimizer moments) only receives gradients on the rank that holds the user."""
  world = dist.get_world_size(group)
  for r in range(world):
    lo, hi = shard_range(n_rows, r, world)
    if hi > lo:
      for t in tensors:
        dist.broadcast(t[lo:hi], src=r if group is None else dist.get_global_rank(group, r),
                       group=group)


def _make_rccl(group, device, rank, world):
  """Our own RCCL communicator (recoder_amd/rccl.py: collectives enqueued IN ORDER on the stream
  they are issued on -- no cross-stream event pairs) with a self-check; None (= use
  torch.distributed) for RK_COMM=torch, non-nccl backends or any bootstrap failure.  A
  collective: every rank calls it."""
  if dist.get_backend(group) != "nccl" or os.environ.get("RK_COMM", "rccl") == "torch":
    return None
  try:
    from .rccl import RcclComm
    comm = RcclComm(group, device)
    probe = torch.full((8,), float(rank + 1), dtype=torch.float32, device=device)
    comm.all_reduce(probe)
    want = world * (world + 1) / 2.0
    if not bool((probe == want).all().item()):
      raise RuntimeError("self-check all-reduce returned %r, expected %r" % (probe[0].item(), want))
    return comm
  except Exception as e:      # noqa: BLE001 -- any bootstrap problem: torch.distributed
    import warnings
    warnings.warn("direct RCCL communicator unavailable (%s); using torch.distributed" % e)
    return None


class DataParallel:
  """Users sharded over the ranks (north_star's partitioning): glue between a FusedEngine and
  the collectives.  Device tensors on the nccl backend go through two communicators of our own
  (recoder_amd/rccl.py) -- one for the gradient bucket on the step's stream, one for the item-stamp
  MAX on the collation side stream -- each enqueued in order on the stream it is issued on;
  anything else (gloo in the CPU tests, RK_COMM=torch) through torch.distributed.  rank / world /
  the collectives can be injected (tests: several virtual ranks on one GPU)."""

  def __init__(self, group=None, rank=None, world=None, allreduce_fn=None, allreduce_max_fn=None):
    self.group = group
    self._sum_fn, self._max_fn = allreduce_fn, allreduce_max_fn
    self.virtual = allreduce_fn is not None
    if not self.virtual:
      assert dist.is_initialized()
    self.rank = dist.get_rank(group) if rank is None else rank
    self.world = dist.get_world_size(group) if world is None else world
    self.user_offset = 0          # first global user id of this rank's shard
    self._grad_comm = self._mark_comm = None

  def prepare(self, device):
    """Create the two direct communicators now (a collective: every rank calls it)."""
    if not self.virtual and device.type == "cuda":
      self._grad_comm = _make_rccl(self.group, device, self.rank, self.world)
      if self._grad_comm is not None:
        self._mark_comm = _make_rccl(self.group, device, self.rank, self.world)
    return self

  @property
  def direct(self):
    return self._grad_comm is not None

  def union_marks(self, mark):
    """MAX all-reduce of the item stamps, in order on the current stream."""
    if self._max_fn is not None:
      return self._max_fn(mark)
    if self._mark_comm is not None and mark.is_cuda:
      from .rccl import ncclMax
      return self._mark_comm.all_reduce(mark, op=ncclMax)
    return union_marks(mark, self.group)

  def union_marks_many(self, marks):
    """The MAX all-reduces of several blocks' stamp arrays, in order on the current stream (one
    RCCL group on the direct communicator: a replayed group's look-ahead blocks)."""
    if self._max_fn is None and self._mark_comm is not None and all(m.is_cuda for m in marks):
      from .rccl import ncclMax
      self._mark_comm.all_reduce_many(marks, op=ncclMax)
      return
    for m in marks:
      self.union_marks(m)

  def collate(self, blk, dcsr, users_dev):
    """Two-phase collation with the union item set.  ``users_dev`` are rows of this
    rank's shard; the block carries their GLOBAL ids (MatrixFactorization looks its
    user rows up by them, the dropout RNG is keyed on them)."""
    blk.collate(dcsr, users_dev, phase=1)
    self.union_marks(blk.mark)
    blk.collate(dcsr, users_dev, phase=2)
    if self.user_offset:
      blk.users = users_dev + self.user_offset

  def attach(self, engine):
    """Install the gradient exchange on a FusedEngine: the engine calls ``reduce(views)`` with
    the step's stream current once the backward pass is enqueued; the SUM all-reduces of all
    views go out as ONE in-order RCCL group before the (identical) Adam of every replica."""
    engine.world_size = self.world
    engine.allreduce = self
    return engine

  def n_b(self, blk):
    return blk.host_n_b()

  # The step's exchange in two parts on ONE communication stream (one communicator: its collectives
  # stay in order): the decoder-side gradients leave right after forward + dW and travel while the
  # step's stream runs dZ -> encoder backward; the encoder side follows; the Adam sweep waits for
  # both.  Injected / torch.distributed collectives (tests, gloo) run in line, in the same order.
  @property
  def overlapped(self):
    # (one rank has nothing to hide the two cross-stream edges behind: 0.228 vs 0.212 ms per step)
    return (self._sum_fn is None and self._grad_comm is not None and
            (self.world > 1 or os.environ.get("RK_DP_OVERLAP") == "1") and
            os.environ.get("RK_DP_OVERLAP", "1") != "0")

  def reduce_async(self, views, main_stream):
    """SUM all-reduce of `views` behind everything enqueued on main_stream so far, on the
    communication stream; returns at once.  join_async(main_stream) makes main_stream wait for it."""
    views = [v for v in views if v.numel() > 0]
    if not self.overlapped or not all(v.is_cuda for v in views):
      self.reduce(views)
      return
    if getattr(self, "_cstream", None) is None:
      self._cstream = torch.cuda.Stream(device=views[0].device)
      self._ev_go, self._ev_done = torch.cuda.Event(), torch.cuda.Event()
    self._ev_go.record(main_stream)
    self._cstream.wait_event(self._ev_go)
    self._grad_comm.all_reduce_many(views, stream=self._cstream)
    self._async_pending = True

  def join_async(self, main_stream):
    if getattr(self, "_async_pending", False):
      self._ev_done.record(self._cstream)
      main_stream.wait_event(self._ev_done)
      self._async_pending = False

  def reduce(self, views):
    views = [v for v in views if v.numel() > 0]
    if self._sum_fn is not None:
      for v in views:
        self._sum_fn(v)
    elif self._grad_comm is not None and all(v.is_cuda for v in views):
      self._grad_comm.all_reduce_many(views)
    else:
      allreduce_sum(views, self.group)


class ItemParallel:
  """Ownership, data sharding and the exchanges of item-parallel training."""

  def __init__(self, group=None, rank=None, world=None, allreduce_fn=None, allgather_fn=None):
    """rank / world / the two collectives default to torch.distributed's; tests inject
    an in-process fake to run several virtual ranks on one GPU."""
    self.group = group
    self.rank = dist.get_rank(group) if rank is None else rank
    self.world = dist.get_world_size(group) if world is None else world
    self._allreduce = allreduce_fn
    self._allgather = allgather_fn
    self.user_norm_dev = None
    self.user_tsum_dev = None
    self._rccl = None
    self._rccl_tried = False

  def _direct(self, t):
    """Our own RCCL communicator for device tensors on the nccl backend (see _make_rccl)."""
    if not self._rccl_tried:
      self._rccl_tried = True
      if t.is_cuda:
        self._rccl = _make_rccl(self.group, t.device, self.rank, self.world)
    return self._rccl if t.is_cuda else None

  def prepare(self, device):
    """Create the direct communicator now (a collective: every rank calls it)."""
    if self._allreduce is None and dist.is_initialized():
      self._direct(torch.zeros(1, dtype=torch.float32, device=device))
    return self

  def owns(self, item_ids):
    return (np.asarray(item_ids) % self.world) == self.rank

  def shard_csr(self, csr):
    """Every row, only the owned columns (global column ids, same shape)."""
    import scipy.sparse as sp
    csr = csr.tocsr()
    keep = self.owns(csr.indices)
    rows = np.repeat(np.arange(csr.shape[0]), np.diff(csr.indptr))
    counts = np.bincount(rows[keep], minlength=csr.shape[0])
    indptr = np.zeros(csr.shape[0] + 1, dtype=csr.indptr.dtype)
    np.cumsum(counts, out=indptr[1:])
    return sp.csr_matrix((csr.data[keep], csr.indices[keep], indptr), shape=csr.shape)

  # ---- the same three for a matrix that is resident in HBM (data.DeviceDataset, or a host dataset
  # after dataset.device_csr()): torch ops on the device arrays, no host pass over the matrix
  @staticmethod
  def _row_ids(dcsr):
    n = dcsr.shape[0]
    return torch.repeat_interleave(torch.arange(n, device=dcsr.device), dcsr.indptr[1:] - dcsr.indptr[:-1])

  def shard_device_csr(self, dcsr):
    """shard_csr on the device: every row, only the owned columns (global ids, same shape)."""
    from .device import DeviceCSR
    n, nnz = dcsr.shape[0], dcsr.nnz
    idx = dcsr.indices[:nnz]
    keep = (idx % self.world) == self.rank
    counts = torch.bincount(self._row_ids(dcsr)[keep], minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dcsr.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    data = None if dcsr.data is None else dcsr.data[:nnz][keep]
    return DeviceCSR.from_arrays(dcsr.shape, indptr, idx[keep], data, dcsr.device, check=False)

  @staticmethod
  def _row_sums_dev(dcsr, square):
    deg = (dcsr.indptr[1:] - dcsr.indptr[:-1]).to(torch.float64)
    if dcsr.data is None:                      # implicit feedback: every stored value is 1.0
      return deg
    v = dcsr.data[:dcsr.nnz].to(torch.float64)
    out = torch.zeros(dcsr.shape[0], dtype=torch.float64, device=dcsr.device)
    return out.index_add_(0, ItemParallel._row_ids(dcsr), v * v if square else v)

  @staticmethod
  def user_norms_dev(dcsr):
    """user_norms on the device (fp64 accumulation, fp32 result)."""
    return ItemParallel._row_sums_dev(dcsr, True).sqrt().to(torch.float32)

  @staticmethod
  def user_target_sums_dev(dcsr):
    return ItemParallel._row_sums_dev(dcsr, False).to(torch.float32)

  @staticmethod
  def user_norms(csr):
    """L2 norm of every user's WHOLE row (F.normalize's denominator, nn.py:235), fp32."""
    csr = csr.tocsr()
    sq = np.asarray(csr.multiply(csr).sum(axis=1), dtype=np.float64).reshape(-1)
    return np.sqrt(sq).astype(np.float32)

  @staticmethod
  def user_target_sums(csr):
    """Sum of every user's interaction values (the multinomial loss's sum_t t per row), fp32."""
    return np.asarray(csr.tocsr().sum(axis=1), dtype=np.float64).reshape(-1).astype(np.float32)

  def allgather(self, t):
    """[t of rank 0, ..., t of rank N-1] (same shape on every rank)."""
    if self._allgather is not None:
      return self._allgather(t)
    parts = [torch.empty_like(t) for _ in range(self.world)]
    dist.all_gather(parts, t.contiguous(), group=self.group)
    return parts

  def allreduce_sum(self, t):
    """In-place SUM over the ranks, ordered on the current stream."""
    if self._allreduce is not None:
      return self._allreduce(t)
    comm = self._direct(t)
    if comm is not None:
      return comm.all_reduce(t)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
    return t

  def broadcast(self, t, src=0):
    """Rank src's tensor on every rank (no-op with injected collectives: the virtual
    ranks of a test share their inputs by construction)."""
    if self._allreduce is None:
      dist.broadcast(t, src=src if self.group is None else dist.get_global_rank(self.group, src),
                     group=self.group)
    return t

  def sync_owned(self, tensors, n_rows):
    """Every replica gets rows r, r + N, ... of each tensor from their owner r."""
    per = (n_rows + self.world - 1) // self.world
    for t in tensors:
      mine = t[self.rank:n_rows:self.world]
      buf = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
      buf[:mine.shape[0]].copy_(mine)
      if self._allgather is not None:
        parts = self._allgather(buf)
      else:
        parts = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(parts, buf, group=self.group)
      for r in range(self.world):
        dst = t[r:n_rows:self.world]
        dst.copy_(parts[r][:dst.shape[0]])
