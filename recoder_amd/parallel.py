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

  def __init__(self, group=None, rank=None, world=None, allreduce_fn=None, allreduce_max_fn=None,
               allgather_fn=None):
    self.group = group
    self._sum_fn, self._max_fn, self._gather_fn = allreduce_fn, allreduce_max_fn, allgather_fn
    self.virtual = allreduce_fn is not None
    if not self.virtual:
      assert dist.is_initialized()
    self.rank = dist.get_rank(group) if rank is None else rank
    self.world = dist.get_world_size(group) if world is None else world
    self.user_offset = 0          # first global user id of this rank's shard
    self._grad_comm = self._mark_comm = None
    self.exchange_mode = "allreduce"     # how a large gradient bucket is summed: ncclAllReduce | rsag
    self.calibration = None
    self.owner_bounds = None             # owned-row Adam: item-id boundaries of the ranks' row ranges
    self.zero = None                     # sharded dense Adam: setup_zero
    # per-rank item sets (RK_DP_ITEMSETS=local, opt-in): every rank samples its negatives from ITS OWN users'
    # items -- the semantics of the reference under conventional DDP, not of its shared item set -- and the
    # gradients travel laid out by item id (model.Recoder._setup_local_sets)
    self.local_sets = False

  def prepare(self, device):
    """Create the two direct communicators now (a collective: every rank calls it)."""
    if not self.virtual and device.type == "cuda":
      self._grad_comm = _make_rccl(self.group, device, self.rank, self.world)
      if self._grad_comm is not None:
        self._mark_comm = _make_rccl(self.group, device, self.rank, self.world)
        self._pick_exchange(device)
    return self

  # ---- how the large gradient buckets travel --------------------------------------------------
  # RK_DP_EXCHANGE = allreduce | rsag | auto (default).  A ring all-reduce over point-to-point xGMI is
  # bound by ONE link direction; reduce-scatter + all-gather of the same bucket (the same bytes) can run
  # direct, one-shot algorithms over all 7 links of a fully connected node.  Which one this
  # RCCL build runs faster at the step's bucket size is a property of the machine: `auto` times both
  # once, at communicator creation, on an 8 MB bucket and keeps the faster (the MAX over the ranks
  # decides, so every rank takes the same path).
  def _pick_exchange(self, device):
    mode = os.environ.get("RK_DP_EXCHANGE", "auto")
    if mode in ("allreduce", "rsag"):
      self.exchange_mode = mode
      return
    if self.world == 1:
      return
    want = "allreduce"
    ok = 1
    try:
      self.calibration = self.microbench(8 << 20, device, iters=5)
      if self.calibration["rsag_us"] < 0.95 * self.calibration["allreduce_us"]:
        want = "rsag"
    except Exception as e:            # noqa: BLE001 -- never fail a training run on the calibration
      import warnings
      warnings.warn("exchange calibration failed (%s): ncclAllReduce" % e)
      ok = 0
    # every rank must issue the SAME collectives: a calibration that failed on one rank only (or left the
    # ranks with different answers) would pair ncclAllReduce with reduce-scatter / all-gather and hang the
    # first exchange -- agree over torch.distributed (not the communicator that may just have failed):
    # rsag only if every rank measured it, succeeded and chose it
    self.exchange_mode = self._agree_mode(want if ok else "allreduce", device)

  def _agree_mode(self, want, device):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
      return want
    backend = str(dist.get_backend(self.group))
    t = torch.tensor([1 if want == "rsag" else 0], dtype=torch.int32,
                     device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
    return "rsag" if int(t.item()) == 1 else "allreduce"

  def microbench(self, nbytes, device, iters=10):
    """Time ncclAllReduce against reduce-scatter + all-gather on a bucket of `nbytes` (a collective:
    every rank calls it): {bytes, allreduce_us, rsag_us, *_busbw_GBs} with the slowest rank's times and
    the bus bandwidth 2 (N - 1) / N . bytes / t as nccl-tests define it."""
    comm = self._grad_comm
    n = max(self.world * 64, (nbytes // 4) // (self.world * 64) * (self.world * 64))
    buf = torch.ones(n, dtype=torch.float32, device=device)
    scratch = torch.empty(n // self.world, dtype=torch.float32, device=device)
    st = torch.cuda.current_stream()
    res = {}
    for name, fn in (("allreduce", lambda: comm.all_reduce(buf)),
                     ("rsag", lambda: comm.reduce_scatter_all_gather(buf, scratch))):
      for _ in range(2):
        buf.fill_(1.0)
        fn()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      st.synchronize()
      e0.record(st)
      for _ in range(iters):
        fn()
      e1.record(st)
      e1.synchronize()
      res[name] = e0.elapsed_time(e1) * 1e3 / iters
    t = torch.tensor([res["allreduce"], res["rsag"]], dtype=torch.float32, device=device)
    from .rccl import ncclMax
    comm.all_reduce(t, op=ncclMax)
    ar, rs = (float(x) for x in t.cpu())
    bus = 2.0 * (self.world - 1) / self.world * n * 4
    return dict(bytes=n * 4, allreduce_us=ar, rsag_us=rs, allreduce_busbw_GBs=bus / ar / 1e3 if ar > 0 else None,
                rsag_busbw_GBs=bus / rs / 1e3 if rs > 0 else None, world=self.world, picked=self.exchange_mode)

  def _sum_many(self, views, stream=None):
    """SUM over the ranks of every view, in place, as one in-order RCCL group; large views go as
    reduce-scatter + all-gather when that is this machine's faster exchange."""
    comm = self._grad_comm
    if self.exchange_mode != "rsag":
      comm.all_reduce_many(views, stream=stream)
      return
    small = []
    for v in views:
      n = v.numel()
      if n >= 65536 and n % self.world == 0 and v.dtype == torch.float32:
        key = (n // self.world, v.device)
        sc = getattr(self, "_rs_scratch", None)
        if sc is None or sc.numel() < n // self.world or sc.device != v.device:
          self._rs_scratch = sc = torch.empty(n // self.world, dtype=torch.float32, device=v.device)
        comm.reduce_scatter_all_gather(v, sc, stream=stream)
      else:
        small.append(v)
    if small:
      comm.all_reduce_many(small, stream=stream)

  def round_rows(self, n_rows, cap_rows):
    """Rows of a gradient bucket that actually travel: the live count rounded up so that rows * h is a
    multiple of the world size (reduce-scatter shards), never past the buffer's capacity."""
    g = self.world * 8
    return min(cap_rows, -(-n_rows // g) * g)

  @property
  def direct(self):
    return self._grad_comm is not None

  # With more than one rank EVERY collective of this object goes through ONE communicator on ONE
  # communication stream, tied to the issuing stream by an event on each side: two communicators with
  # collectives in flight at once and no cross-rank order between them -- the stamp MAX of the look-ahead
  # collation on the side stream next to a step's gradient exchange, inside one replayed graph -- is the
  # pattern RCCL documents as deadlock-prone.  One stream + one communicator = one total
  # order, the same on every rank (the enqueue / capture order is).  With one rank there is nothing to
  # order and the collectives stay on the issuing stream (no cross-stream edges).
  def _on_comm_stream(self, device, fn):
    cur = torch.cuda.current_stream(device)
    if getattr(self, "_cstream", None) is None:
      self._cstream = torch.cuda.Stream(device=device)
      self._ev_go, self._ev_done = torch.cuda.Event(), torch.cuda.Event()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(cur)
    self._cstream.wait_event(ea)
    fn(self._cstream)
    eb.record(self._cstream)
    cur.wait_event(eb)

  def union_marks(self, mark):
    """MAX all-reduce of the item stamps, ordered behind the current stream's work."""
    if self._max_fn is not None:
      return self._max_fn(mark)
    if self._grad_comm is not None and mark.is_cuda:
      from .rccl import ncclMax
      if self.world > 1:
        self._on_comm_stream(mark.device, lambda cs: self._grad_comm.all_reduce(mark, op=ncclMax, stream=cs))
        return mark
      return self._mark_comm.all_reduce(mark, op=ncclMax)
    return union_marks(mark, self.group)

  def union_marks_many(self, marks):
    """The MAX all-reduces of several blocks' stamp arrays as one RCCL group (a replayed group's
    look-ahead blocks), ordered behind the current stream's work."""
    if self._max_fn is None and self._grad_comm is not None and all(m.is_cuda for m in marks):
      from .rccl import ncclMax
      if self.world > 1:
        self._on_comm_stream(marks[0].device,
                             lambda cs: self._grad_comm.all_reduce_many(marks, op=ncclMax, stream=cs))
        return
      self._mark_comm.all_reduce_many(marks, op=ncclMax)
      return
    for m in marks:
      self.union_marks(m)

  def collate(self, blk, dcsr, users_dev):
    """Two-phase collation with the union item set.  ``users_dev`` are rows of this
    rank's shard; the block carries their GLOBAL ids (MatrixFactorization looks its
    user rows up by them, the dropout RNG is keyed on them)."""
    if self.local_sets:
      blk.collate(dcsr, users_dev)         # (this rank's own item set: no exchange of the stamps)
    else:
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
    self._sum_many(views, stream=self._cstream)
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
      if self.world > 1:
        self._on_comm_stream(views[0].device, lambda cs: self._sum_many(views, stream=cs))
      else:
        self._sum_many(views)
    else:
      allreduce_sum(views, self.group)

  # ---- owned-row Adam (SparseAdam tables) -----------------------------------------------------
  # With replicated weights every rank applies SparseAdam to ALL rows of the union item set: at C5 (8
  # ranks, ~330 k union rows x 512 x 2 tables x 28 B = 9.5 GB) that is ~1.7 ms of HBM time per rank and
  # step, eight times over.  Instead every item belongs to ONE rank -- contiguous item-id ranges
  # [owner_bounds[r], owner_bounds[r + 1]), balanced by the items' expected presence in a batch; the
  # compact rows of a block are sorted by item id, so a rank's rows are ONE contiguous slice of every
  # compact gradient array -- and a step exchanges
  #     partial gradient rows of range r  -> rank r        (variable-count all-to-all: the reduce-scatter)
  #     SparseAdam on the owned rows only (moments of a row live on its owner)
  #     updated parameter rows of range r -> every rank    (the all-gather)
  # the bytes of ONE all-reduce, and 1/N of the Adam sweep.  Exactly the replicated update: the sum of
  # the N partial rows is taken in rank order inside the Adam job (g_parts = N).
  def set_owner_bounds(self, bounds):
    self.owner_bounds = np.asarray(bounds, dtype=np.int64)
    assert len(self.owner_bounds) == self.world + 1

  @staticmethod
  def balanced_bounds(item_freq, n_users, rows_per_step, world):
    """Item-id boundaries that give every rank the same EXPECTED number of union rows per step:
    item i is in the union of a step's rows with probability 1 - (1 - f_i / n_users) ** rows_per_step."""
    f = np.asarray(item_freq, dtype=np.float64) / max(1, n_users)
    p = 1.0 - np.power(np.clip(1.0 - f, 0.0, 1.0), rows_per_step)
    c = np.concatenate([[0.0], np.cumsum(p)])
    tot = c[-1] if c[-1] > 0 else 1.0
    b = [int(np.searchsorted(c, tot * r / world, side="left")) for r in range(world + 1)]
    b[0], b[-1] = 0, len(f)
    return np.maximum.accumulate(np.asarray(b, dtype=np.int64))

  @staticmethod
  def owned_rows_estimate(item_freq, n_users, rows_per_step, world, h, tables, hbm_bps=5.0e12, host_us=150.0):
    """What owned-row SparseAdam would save and cost per step (us), from the expected union item set of a
    global batch: saved = (1 - 1/N) of the replicated sweep (28 B per element of the union rows); cost = the
    host-sequenced step (measured at one forced rank: ~150 us of launches enqueued one by one + the read of
    the row offsets) + gathering the owned rows and scattering every received row into the tables (16 B
    per element).  RK_DP_OWNED=auto takes the owned rows when saved >= 2 x cost."""
    f = np.asarray(item_freq, dtype=np.float64) / max(1, n_users)
    union = float((1.0 - np.power(np.clip(1.0 - f, 0.0, 1.0), rows_per_step)).sum())
    elems = tables * union * h
    return dict(union_rows=union, saved_us=elems * 28.0 * (1.0 - 1.0 / world) / hbm_bps * 1e6,
                cost_us=host_us + elems * 16.0 / hbm_bps * 1e6)

  def owned_offsets(self, items, n_b):
    """Compact-row offsets [world + 1] of the ranks' segments in the block's sorted item list (host)."""
    b = torch.as_tensor(self.owner_bounds[1:-1], dtype=items.dtype, device=items.device)
    mid = torch.searchsorted(items[:n_b].contiguous(), b).cpu().tolist() if self.world > 1 else []
    return [0] + [int(x) for x in mid] + [int(n_b)]

  def exchange_rows(self, G, offs, h):
    """G: this rank's partial compact gradient rows [n_b, h] (flat).  Returns R [world, cnt, h] (flat):
    R[q] = rank q's partial rows of THIS rank's segment (cnt = its row count)."""
    lo, hi = offs[self.rank], offs[self.rank + 1]
    cnt = hi - lo
    R = torch.empty(self.world * max(cnt, 1) * h, dtype=G.dtype, device=G.device)
    if self._sum_fn is None and self._grad_comm is not None and G.is_cuda:
      sends = [G[offs[q] * h:offs[q + 1] * h] for q in range(self.world)]
      recvs = [R[q * cnt * h:(q + 1) * cnt * h] for q in range(self.world)]
      if self.world > 1:
        self._on_comm_stream(G.device, lambda cs: self._grad_comm.exchange(sends, recvs, stream=cs))
      else:
        self._grad_comm.exchange(sends, recvs)
      return R, cnt
    parts = self._gather_all(G[:offs[-1] * h].contiguous())
    for q in range(self.world):
      R[q * cnt * h:(q + 1) * cnt * h].copy_(parts[q][lo * h:hi * h])
    return R, cnt

  def publish_rows(self, S, offs, h):
    """S: the updated parameter rows of this rank's segment [cnt, h].  Returns T [n_b, h]: every rank's
    rows in compact order."""
    n_b = offs[-1]
    T = torch.empty(n_b * h, dtype=S.dtype, device=S.device)
    if self._sum_fn is None and self._grad_comm is not None and S.is_cuda:
      S = S.contiguous().view(-1)
      sends = [S for _ in range(self.world)]
      recvs = [T[offs[q] * h:offs[q + 1] * h] for q in range(self.world)]
      if self.world > 1:
        self._on_comm_stream(S.device, lambda cs: self._grad_comm.exchange(sends, recvs, stream=cs))
      else:
        self._grad_comm.exchange(sends, recvs)
      return T
    cap = max(offs[q + 1] - offs[q] for q in range(self.world))
    pad = torch.zeros(max(cap, 1) * h, dtype=S.dtype, device=S.device)
    pad[:S.numel()].copy_(S.reshape(-1))
    parts = self._gather_all(pad)
    for q in range(self.world):
      n = (offs[q + 1] - offs[q]) * h
      T[offs[q] * h:offs[q] * h + n].copy_(parts[q][:n])
    return T

  def _gather_all(self, t):
    """[t of rank 0, ..., t of rank N - 1] (equal shapes): injected (virtual ranks) or torch.distributed."""
    if self._gather_fn is not None:
      return self._gather_fn(t)
    parts = [torch.empty_like(t) for _ in range(self.world)]
    dist.all_gather(parts, t, group=self.group)
    return parts

  # ---- sharded dense Adam (ZeRO-1) ------------------------------------------------------------------
  # optim.Adam with sparse=False sweeps ALL rows of a table every step (weight decay, decaying moments:
  # reference model.py:135,398-399), 38 us of HBM time at C2, 76 at C3 -- replicated, every rank repeats it.
  # A dense Adam step treats every row independently, so rank r can own the rows [r sh, (r + 1) sh) of every
  # dense table, sh = ceil(n_items / N): the compact gradient rows are laid out by item id
  # (rk_rows_to_dense), REDUCE-SCATTERED over the equal row ranges (fixed counts: capturable), the owner
  # sweeps its 1/N of the rows -- the only place their moments are kept up to date -- and the updated rows
  # are ALL-GATHERED.  The bytes of today's reduce-scatter + all-gather of the gradient rows at capacity,
  # 1/N of the sweep.
  def setup_zero(self, n_items):
    sh = -(-int(n_items) // self.world)
    self.zero = dict(n_items=int(n_items), sh=sh, rows_pad=sh * self.world,
                     lo=min(int(n_items), self.rank * sh), hi=min(int(n_items), (self.rank + 1) * sh))
    return self.zero

  def zero_bounds(self):
    z = self.zero
    return [min(z["n_items"], r * z["sh"]) for r in range(self.world + 1)]

  def zero_reduce_scatter(self, D, shard, stream=None):
    """shard <- SUM over the ranks of D[rank's row range] (D: [rows_pad * h] laid out by item id)."""
    if self._sum_fn is None and self._grad_comm is not None and D.is_cuda:
      self._grad_comm.reduce_scatter(D, shard, stream=stream)
      return
    parts = self._gather_all(D)            # (virtual ranks / gloo: the same sum in rank order)
    n = shard.numel()
    acc = parts[0][self.rank * n:(self.rank + 1) * n].clone()
    for q in range(1, self.world):
      acc += parts[q][self.rank * n:(self.rank + 1) * n]
    shard.copy_(acc)

  def zero_all_gather(self, tables, h, stream=None):
    """Every rank's freshly updated row range of each [n_items, h] table -> every replica, in place."""
    b = self.zero_bounds()
    if self._sum_fn is None and self._grad_comm is not None and tables[0].is_cuda:
      for t in tables:
        flat = t.view(-1)
        mine = flat[b[self.rank] * h:b[self.rank + 1] * h]
        sends = [mine if q != self.rank else None for q in range(self.world)]
        recvs = [flat[b[q] * h:b[q + 1] * h] if q != self.rank else None for q in range(self.world)]
        if self.world > 1:
          self._grad_comm.exchange(sends, recvs, stream=stream)
      return
    sh = self.zero["sh"]
    for t in tables:
      pad = torch.zeros(sh * h, dtype=t.dtype, device=t.device)
      mine = t.view(-1)[b[self.rank] * h:b[self.rank + 1] * h]
      pad[:mine.numel()].copy_(mine)
      parts = self._gather_all(pad)
      for q in range(self.world):
        n = (b[q + 1] - b[q]) * h
        if q != self.rank and n:
          t.view(-1)[b[q] * h:b[q] * h + n].copy_(parts[q][:n])

  def zero_exchange(self, stage, D, shard, small, main_stream, overlap=True):
    """One half of a sharded-Adam step's exchange: stage(stream) lays the compact gradient rows out by item
    id in D, D is reduce-scattered into `shard`, the small gradients `small` are all-reduced -- on the
    communication stream behind everything enqueued on main_stream so far when the exchange may overlap the
    step (join_async makes main_stream wait for it), else in line.  shard None (per-rank item sets without
    sharding): D itself is all-reduced in place, with the small ones."""
    small = [v for v in small if v.numel() > 0]
    if shard is None:
      direct = self._sum_fn is None and self._grad_comm is not None and D.is_cuda
      if direct and overlap and self.overlapped:
        if getattr(self, "_cstream", None) is None:
          self._cstream = torch.cuda.Stream(device=D.device)
          self._ev_go, self._ev_done = torch.cuda.Event(), torch.cuda.Event()
        self._ev_go.record(main_stream)
        self._cstream.wait_event(self._ev_go)
        stage(self._cstream)
        self._sum_many([D] + small, stream=self._cstream)
        self._async_pending = True
        return
      stage(main_stream)
      self.reduce([D] + small)
      return
    direct = self._sum_fn is None and self._grad_comm is not None and D.is_cuda
    if direct and overlap and self.overlapped:
      if getattr(self, "_cstream", None) is None:
        self._cstream = torch.cuda.Stream(device=D.device)
        self._ev_go, self._ev_done = torch.cuda.Event(), torch.cuda.Event()
      self._ev_go.record(main_stream)
      self._cstream.wait_event(self._ev_go)
      stage(self._cstream)
      self.zero_reduce_scatter(D, shard, stream=self._cstream)
      if small:
        self._grad_comm.all_reduce_many(small, stream=self._cstream)
      self._async_pending = True
      return
    stage(main_stream)
    if direct and self.world > 1:
      def go(cs):
        self.zero_reduce_scatter(D, shard, stream=cs)
        if small:
          self._grad_comm.all_reduce_many(small, stream=cs)
      self._on_comm_stream(D.device, go)
    elif direct:
      self.zero_reduce_scatter(D, shard)
      if small:
        self._grad_comm.all_reduce_many(small)
    else:
      self.zero_reduce_scatter(D, shard)
      self.reduce(small)

  def zero_publish(self, tables, h, extra_max=None):
    """The updated row ranges of `tables` to every replica (+ the MAX over the ranks of `extra_max`, the
    decoder table's |W| bound: a rank only saw the rows it wrote), ordered behind the current stream's work
    and in front of whatever it enqueues next."""
    direct = self._sum_fn is None and self._grad_comm is not None and tables[0].is_cuda
    if direct and self.world > 1:
      def go(cs):
        self.zero_all_gather(tables, h, stream=cs)
        if extra_max is not None:
          from .rccl import ncclMax
          self._grad_comm.all_reduce(extra_max, op=ncclMax, stream=cs)
      self._on_comm_stream(tables[0].device, go)
      return
    self.zero_all_gather(tables, h)
    if extra_max is not None and self.world > 1:
      self.union_marks(extra_max)

  def sync_owned_moments(self, tensors, bounds=None):
    """Every replica gets the rows [owner_bounds[r], owner_bounds[r + 1]) of each tensor (the Adam
    moments of the owned-row update) from their owner: before a checkpoint / at the end of train()."""
    bounds = self.owner_bounds if bounds is None else bounds
    if bounds is None or self.world == 1:
      return
    for r in range(self.world):
      lo, hi = int(bounds[r]), int(bounds[r + 1])
      if hi <= lo:
        continue
      for t in tensors:
        seg = t[lo:hi]
        if self._gather_fn is not None:
          seg.copy_(self._gather_fn(seg.contiguous())[r])
        else:
          dist.broadcast(seg, src=r if self.group is None else dist.get_global_rank(self.group, r),
                         group=self.group)


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
