"""Thin RCCL binding for the exchanges of the training step.

torch.distributed runs every collective on ProcessGroupNCCL's own stream and joins it
to the caller's stream with an event pair on each side; on this stack a cross-stream
dependency costs 10-20 us of latency (tools/sync_cost2.py), i.e. several tens of us per
all-reduce of a step that takes 160-370 us.  A communicator of our own lets the
collective be enqueued IN ORDER on the step's stream: ncclAllReduce(..., stream).

The communicator is bootstrapped through the already initialised torch.distributed
group (rank 0's ncclUniqueId is broadcast with it) and uses the librccl.so that torch
itself loaded.  If anything fails, callers fall back to torch.distributed.
"""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
ncclSum, ncclMax = 0, 2
ncclInt32, ncclFloat32 = 2, 7


class _UniqueId(ctypes.Structure):
  _fields_ = [("internal", ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]   # opaque bytes (NULs inside)


_lib = None


def _load():
  global _lib
  if _lib is None:
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId,
                                     ctypes.c_int]
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclReduceScatter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p]
    lib.ncclRecv.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    _lib = lib
  return _lib


def _check(rc, what):
  if rc != 0:
    raise RuntimeError("%s failed: %s" % (what, _load().ncclGetErrorString(rc).decode()))


class RcclComm:
  """One RCCL communicator spanning a torch.distributed group (default: the world)."""

  def __init__(self, group=None, device=None):
    lib = _load()
    self.rank = dist.get_rank(group)
    self.world = dist.get_world_size(group)
    device = device or torch.device("cuda", torch.cuda.current_device())
    uid = _UniqueId()
    if self.rank == 0:
      _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    t = torch.frombuffer(bytearray(ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)),
                         dtype=torch.uint8).to(device)
    src = 0 if group is None else dist.get_global_rank(group, 0)
    dist.broadcast(t, src=src, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    ctypes.memmove(ctypes.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
    self.comm = ctypes.c_void_p()
    torch.cuda.synchronize(device)
    _check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank),
           "ncclCommInitRank")

  def all_reduce(self, t, op=ncclSum, stream=None):
    """In place on the given (default: current) stream, ordered with the kernels around it."""
    assert t.is_cuda and t.is_contiguous()
    if t.dtype == torch.float32:
      dt = ncclFloat32
    elif t.dtype == torch.int32:
      dt = ncclInt32
    else:
      raise TypeError("unsupported dtype %s" % t.dtype)
    s = stream if stream is not None else torch.cuda.current_stream()
    _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, op, self.comm,
                                 ctypes.c_void_p(s.cuda_stream)), "ncclAllReduce")
    return t

  def all_reduce_many(self, tensors, op=ncclSum, stream=None):
    """The all-reduces of several tensors as ONE RCCL group (one fused launch), in place, in order
    on the given (default: current) stream."""
    tensors = [t for t in tensors if t.numel() > 0]
    if not tensors:
      return
    if len(tensors) == 1:
      self.all_reduce(tensors[0], op, stream)
      return
    lib = _load()
    _check(lib.ncclGroupStart(), "ncclGroupStart")
    try:
      for t in tensors:
        self.all_reduce(t, op, stream)
    finally:
      _check(lib.ncclGroupEnd(), "ncclGroupEnd")

  @staticmethod
  def _dt(t):
    if t.dtype == torch.float32:
      return ncclFloat32
    if t.dtype == torch.int32:
      return ncclInt32
    raise TypeError("unsupported dtype %s" % t.dtype)

  def reduce_scatter(self, send, recv, op=ncclSum, stream=None):
    """recv[i] = sum over ranks of send[rank * recv.numel() + i]: send holds world equal shards."""
    assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
    assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype
    s = stream if stream is not None else torch.cuda.current_stream()
    _check(_load().ncclReduceScatter(send.data_ptr(), recv.data_ptr(), recv.numel(), self._dt(send), op,
                                     self.comm, ctypes.c_void_p(s.cuda_stream)), "ncclReduceScatter")
    return recv

  def all_gather(self, send, recv, stream=None):
    """recv[r * send.numel() + i] = rank r's send[i]."""
    assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
    assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
    s = stream if stream is not None else torch.cuda.current_stream()
    _check(_load().ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self._dt(send), self.comm,
                                 ctypes.c_void_p(s.cuda_stream)), "ncclAllGather")
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
    lib = _load()
    s = stream if stream is not None else torch.cuda.current_stream()
    st = ctypes.c_void_p(s.cuda_stream)
    _check(lib.ncclGroupStart(), "ncclGroupStart")
    try:
      for q in range(self.world):
        t = sends[q]
        if t is not None and t.numel() > 0:
          assert t.is_cuda and t.is_contiguous()
          _check(lib.ncclSend(t.data_ptr(), t.numel(), self._dt(t), q, self.comm, st), "ncclSend")
        r = recvs[q]
        if r is not None and r.numel() > 0:
          assert r.is_cuda and r.is_contiguous()
          _check(lib.ncclRecv(r.data_ptr(), r.numel(), self._dt(r), q, self.comm, st), "ncclRecv")
    finally:
      _check(lib.ncclGroupEnd(), "ncclGroupEnd")

  def destroy(self):
    if self.comm:
      _load().ncclCommDestroy(self.comm)
      self.comm = ctypes.c_void_p()
