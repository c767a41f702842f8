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

Nothing here touches the HIP library, so the same code runs under gloo on CPU
tensors in tests/test_parallel.py.
"""
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


class DataParallel:
  """Glue between a FusedEngine and torch.distributed."""

  def __init__(self, group=None):
    assert dist.is_initialized()
    self.group = group
    self.rank = dist.get_rank(group)
    self.world = dist.get_world_size(group)
    self.user_offset = 0          # first global user id of this rank's shard

  def collate(self, blk, dcsr, users_dev):
    """Two-phase collation with the union item set.  ``users_dev`` are rows of this
    rank's shard; the block carries their GLOBAL ids (MatrixFactorization looks its
    user rows up by them, the dropout RNG is keyed on them)."""
    blk.collate(dcsr, users_dev, phase=1)
    union_marks(blk.mark, self.group)
    blk.collate(dcsr, users_dev, phase=2)
    if self.user_offset:
      blk.users = users_dev + self.user_offset

  def attach(self, engine):
    """Install the gradient exchange on a FusedEngine.  The engine calls
    ``reduce_async(views)`` as soon as a group of gradients is complete (decoder
    side right after dW, encoder side after the encoder backward) with the
    producing stream current, and ``wait(handles)`` before the matching Adam, so
    the RCCL transfers overlap the rest of the backward pass."""
    engine.world_size = self.world
    engine.allreduce = self
    return engine

  def n_b(self, blk):
    return blk.host_n_b()

  def reduce_async(self, views, small_threshold=65536, coalesce=True):
    if not coalesce:
      small_threshold = -1
    small = [v for v in views if v.numel() <= small_threshold]
    large = [v for v in views if v.numel() > small_threshold]
    handles = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
               for v in large]
    flat = None
    if small:
      flat = torch.cat([v.reshape(-1) for v in small])
      handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
    return handles, small, flat

  def wait(self, pending):
    handles, small, flat = pending
    for h in handles:
      h.wait()                 # the current stream waits for the collective
    if flat is not None:
      off = 0
      for v in small:
        n = v.numel()
        v.copy_(flat[off:off + n].view_as(v))
        off += n
