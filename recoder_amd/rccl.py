"""The data-parallel step's collectives: ctypes binding of librecoder_hip.so's rk_comm_* / rk_allreduce_* exports
(include/recoder_hip.h, csrc/comm.hip) -- RCCL enqueued IN ORDER on the step's stream.

torch.distributed runs every collective on ProcessGroupNCCL's own stream and joins it
to the caller's stream with an event pair on each side; on this stack a cross-stream
dependency costs 10-20 us of latency (tools/sync_cost2.py), i.e. several tens of us per
all-reduce of a step that takes 100-370 us.  A communicator of our own lets the
collective be enqueued in order on the step's stream -- and captured with the step.

The communicator is bootstrapped through the already initialised torch.distributed
group (rank 0's 128-byte id is broadcast with it); the C side binds the librccl.so that torch
itself loaded (the `librccl` argument of rk_comm_unique_id / rk_comm_init).  If anything fails, callers fall back to torch.distributed.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check

NCCL_UNIQUE_ID_BYTES = 128
# (values of include/recoder_hip.h RK_COMM_*; the names are what recoder_amd/parallel.py imports)
ncclSum, ncclMax = 0, 1
F32, I32 = 0, 1

def _load():
  return _lib.load()


def _librccl():
  """The librccl.so this process holds (torch's) -- the C side binds that one."""
  path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
  return path.encode() if os.path.exists(path) else None


def _h(stream):
  s = stream if stream is not None else torch.cuda.current_stream()
  return ctypes.c_void_p(s.cuda_stream)


class RcclComm:
  """One RCCL communicator spanning a torch.distributed group (default: the world)."""

  def __init__(self, group=None, device=None):
    lib = _load()
    self.rank = dist.get_rank(group)
    self.world = dist.get_world_size(group)
    device = device or torch.device("cuda", torch.cuda.current_device())
    uid = (ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)()
    if self.rank == 0:
      check(lib.rk_comm_unique_id(uid, _librccl()), "rk_comm_unique_id")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
    src = 0 if group is None else dist.get_global_rank(group, 0)
    dist.broadcast(t, src=src, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    ctypes.memmove(uid, raw, NCCL_UNIQUE_ID_BYTES)
    torch.cuda.synchronize(device)
    self.comm = ctypes.c_void_p(lib.rk_comm_init(uid, self.world, self.rank, _librccl()))
    if not self.comm:
      raise RuntimeError("rk_comm_init failed: %s" % lib.rk_last_error().decode())

  @staticmethod
  def _dt(t):
    if t.dtype == torch.float32:
      return F32
    if t.dtype == torch.int32:
      return I32
    raise TypeError("unsupported dtype %s" % t.dtype)

  def all_reduce(self, t, op=ncclSum, stream=None):
    """In place on the given (default: current) stream, ordered with the kernels around it."""
    self.all_reduce_many([t], op, stream)
    return t

  def all_reduce_many(self, tensors, op=ncclSum, stream=None):
    """The all-reduces of several tensors as ONE RCCL group (one fused launch), in place, in order
    on the given (default: current) stream."""
    tensors = [t for t in tensors if t.numel() > 0]
    if not tensors:
      return
    by_dt = {}
    for t in tensors:
      assert t.is_cuda and t.is_contiguous()
      by_dt.setdefault(self._dt(t), []).append(t)
    for dt, ts in by_dt.items():
      bufs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
      cnts = (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])
      check(_load().rk_allreduce_bucket(self.comm, bufs, cnts, len(ts), dt, op, _h(stream)), "rk_allreduce_bucket")

  def reduce_scatter(self, send, recv, op=ncclSum, stream=None):
    """recv[i] = sum over ranks of send[rank * recv.numel() + i]: send holds world equal shards."""
    assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
    assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype and op == ncclSum
    check(_load().rk_reduce_scatter(self.comm, send.data_ptr(), recv.data_ptr(), recv.numel(), self._dt(send),
                                    _h(stream)), "rk_reduce_scatter")
    return recv

  def all_gather(self, send, recv, stream=None):
    """recv[r * send.numel() + i] = rank r's send[i]."""
    assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
    assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
    check(_load().rk_all_gather(self.comm, send.data_ptr(), recv.data_ptr(), send.numel(), self._dt(send),
                                _h(stream)), "rk_all_gather")
    return recv

  def reduce_scatter_all_gather(self, t, scratch, stream=None):
    """SUM all-reduce of t (numel a multiple of world) in place as a reduce-scatter into this rank's shard
    + an all-gather of the shards: the same bytes as ncclAllReduce, but as two collectives whose direct
    (one-shot) algorithms drive all xGMI links of a fully connected node at once."""
    n = t.numel() // self.world
    mine = scratch[:n]
    self.reduce_scatter(t, mine, stream=stream)
    self.all_gather(mine, t, stream=stream)
    return t

  def exchange(self, sends, recvs, stream=None):
    """One grouped launch of point-to-point transfers: sends[q] goes to rank q, recvs[q] arrives from
    rank q (None / empty: nothing to / from that rank -- both sides must agree); the variable-count
    all-to-all of the owned-row exchange."""
    W = self.world
    dt = None
    sp, sc = (ctypes.c_void_p * W)(), (ctypes.c_int64 * W)()
    rp, rc = (ctypes.c_void_p * W)(), (ctypes.c_int64 * W)()
    for q in range(W):
      for t, ptrs, cnts in ((sends[q], sp, sc), (recvs[q], rp, rc)):
        if t is not None and t.numel() > 0:
          assert t.is_cuda and t.is_contiguous()
          assert dt is None or dt == self._dt(t), "one element type per exchange"
          dt = self._dt(t)
          ptrs[q], cnts[q] = t.data_ptr(), t.numel()
    if dt is None:
      return
    check(_load().rk_exchange(self.comm, W, sp, sc, rp, rc, dt, _h(stream)), "rk_exchange")

  def destroy(self):
    if self.comm:
      _load().rk_comm_destroy(self.comm)
      self.comm = ctypes.c_void_p()
