"""HIP-graph replay of the hot loop (reference model.py:383-418).

One optimisation step of the common autoencoder case is 6 collation launches + 7 training launches;
enqueued one by one the host spends 50-120 us per step on them -- as much as the GPU needs to run
them.  ``GraphStepper`` captures a GROUP of G steps as one HIP graph,

    main stream :  step(block[v][0]) ... step(block[v][G-1])   (the last one publishes cursor + G)
    side stream :  collate(block[1-v][0..G-1])  for the NEXT group      (forked / joined by events)

and replays it with one ``hipGraphLaunch`` per G steps.  A replayed launch cannot take new
arguments, so everything that changes from step to step is derived ON THE DEVICE from a cursor
(``rk_ae_step_t.cursor``, csrc/common.h ``rk_cur_t``): which users are collated (an offset into the
epoch's user order, resident in HBM), the collation stamp, the dropout RNG step, Adam's bias
corrections and step size (a per-epoch table of constants, ``rk_adam_consts``) and the slot of the
per-step loss.  The captured kernels are the same C-ABI entry points the eager path launches
(``rk_collate_at``, ``rk_ae_train_step``); steps that do not fill a group (the tail of an epoch,
the ragged last batch, groups bench.py brackets with timing events) run eagerly through them.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import PAR_B_DE, PAR_B_EN, PAR_W_DE, PAR_W_EN, RkAeStep, check, ptr

_PAR_NAMES = {PAR_W_EN: "en_embedding_layer.weight",
              PAR_B_EN: "_DynamicAutoencoder__en_linear_embedding_layer.bias",
              PAR_W_DE: "de_embedding_layer.weight",
              PAR_B_DE: "_DynamicAutoencoder__de_linear_embedding_layer.bias"}


class GraphStepper:
  MULTI_MAX = 8          # RK_COLLATE_MULTI (include/recoder_hip.h)

  def __init__(self, engine, dcsr, make_block, B, negative_sampling, group, n_users, device):
    self.lib = _lib.load()
    self.eng, self.dcsr, self.B, self.ns, self.G = engine, dcsr, int(B), bool(negative_sampling), int(group)
    self.device = device
    self.multi = True    # batched collation launches
    # data parallel over users (parallel.DataParallel attached to the engine): every block's item
    # set is the union over the ranks -- the MAX all-reduce of the blocks' stamp arrays sits between
    # the two collation phases, captured with them -- and the step's two gradient exchanges are
    # captured between its phases (engine._c_train_step); all of fixed size
    self.dp = getattr(engine, "allreduce", None)
    assert self.dp is None or (self.multi and self.G <= self.MULTI_MAX and not getattr(engine, "owned_rows", False))
    self.blocks = [[make_block() for _ in range(self.G)] for _ in range(2)]
    self.tail_blk = make_block()            # ragged last batch: eager, host-provided arguments
    engine.ensure_capacity(self.B, self.blocks[0][0].n_cap)
    # two cursors: the group on slot v reads cursor[v]; its last Adam launch writes cursor[1 - v]
    # (= its own + the steps it ran) for the group that follows on the other slot
    self.cursors = torch.zeros(2, 2, dtype=torch.int64, device=device)
    self.steps_cap = -(-n_users // self.B)
    # the epoch's user order, padded so that the look-ahead collation of the group after the last
    # one reads valid user ids (its blocks are never trained on)
    self.order = torch.zeros((self.steps_cap + 2 * self.G) * self.B, dtype=torch.int64, device=device)
    # (data parallel: `order` holds rows of this rank's shard; the dropout RNG is keyed on GLOBAL ids)
    self.order_global = self.order if self.dp is None else torch.zeros_like(self.order)
    self.loss_buf = torch.zeros(self.steps_cap + self.G, dtype=torch.float32, device=device)
    # per-epoch table of Adam constants: one 8-float entry per (step, parameter slot).  The one-call
    # step has its four parameters (slots = RK_PAR_*); the entry-by-entry sequencing one per state
    self.c_step = engine.c_step_eligible()
    self.slots = None if self.c_step else {name: i for i, name in enumerate(engine.states)}
    self.tab_stride = 4 if self.c_step else max(1, len(engine.states))
    self.table = torch.zeros((self.steps_cap + self.G) * self.tab_stride * 8, dtype=torch.float32, device=device)
    self.table_host = torch.zeros((self.steps_cap + self.G) * self.tab_stride * 8, dtype=torch.float32).pin_memory()
    # stream capture needs a stream of its own (not the default stream torch work runs on)
    self.main = torch.cuda.Stream(device=device)
    self.side = torch.cuda.Stream(device=device)
    self.ev_fork, self.ev_join = self.lib.rk_event_create(0), self.lib.rk_event_create(0)
    # the group right behind a cut (epoch start, step mark) has no look-ahead blocks: its G
    # collations are independent chains of small launches, run side by side on streams of their own
    self.pre = [torch.cuda.Stream(device=device) for _ in range(self.G - 1)]
    self.ev_pre = [self.lib.rk_event_create(0) for _ in range(self.G - 1)]
    self.st = [[RkAeStep() for _ in range(self.G)] for _ in range(2)]
    self._gen = None                       # engine.alloc_gen the graphs were captured under
    self.recaptures = 0
    self.regen = True                      # (tests switch it off to show what it protects from)
    self.exec = [None, None]
    self._la_slot = 1                      # slot holding the newest look-ahead blocks
    self.exec_first = [None, None]         # the group right behind a cut: its collation + its steps, per slot
    self.exec_tail = [{}, {}]              # per slot: {n: the last n < G steps in front of a cut, no look-ahead}
    self.capture_tails = True              # (bench.py --no-tails / tools/probes/ab_s20.sh: the A/B of this)
    self.exec_timed = {}                   # (slot, first global index) -> a group captured WITH timing events
    self.warmed = False
    self.global_step = 0                   # steps this stepper's cursor has seen
    self.epoch_base = 0
    # lazy dense Adam (engine.lazy_tables): a step's sweep skips the rows that neither carry a gradient nor are
    # read by the NEXT step -- whose block this stepper has collated by then; run() leaves every row up to date
    self.lazy = list(engine.lazy_tables()) if self.G >= 2 else []
    self.ev_join_early = self.lib.rk_event_create(0)
    # need lists of the lazy sweeps (rk_lazy_need_lists): per block the rows that it or the block behind it holds, built
    # right behind their collation; a step whose two blocks were collated by one such call hands its sweep the list
    # Where most rows are skipped: a list saves the waves of the skipped rows (C3, 41 140 items of which a step's two
    # blocks hold 13 k: the sweep 47.9 -> 44 us, the step 0.191 -> 0.188 ms), and costs two small launches per group on
    # the side stream -- at C2 (20 108 items, 60 % of them swept) the sweep gains nothing inside the step and the step
    # loses 1.2 us.  Rule: a block's expected item set (from the matrix's item frequencies) covers less than 30 % of
    # the catalogue (C2: 39 %, C3: 20 %); RK_ADAM_LAZY=16,list / 16,scan force it
    mode = getattr(engine, "lazy_lists", "auto")
    self.need_lists = bool(self.lazy) and self.multi and self.G <= self.MULTI_MAX and mode != "scan"
    if self.need_lists and mode == "auto":
      n_items = int(self.blocks[0][0].c.n_items)
      freq = torch.zeros(n_items, dtype=torch.float64, device=device)
      for lo in range(0, int(dcsr.indices.numel()), 1 << 26):          # (in pieces: bincount wants int64 ids)
        freq += torch.bincount(dcsr.indices[lo:lo + (1 << 26)].long(), minlength=n_items)[:n_items]
      freq /= max(1, int(n_users))
      rows = self.B * (1 if self.dp is None else self.dp.world)
      self.need_cover = float((1.0 - (1.0 - freq).clamp(0.0, 1.0).pow(rows)).sum()) / max(1, n_items)
      self.need_lists = self.need_cover < 0.3             # (+ RK_NEED_LIST_MAX_ITEMS, below)
    # id(block) -> (list [n_items] int32, count [1] int32), allocated HERE: an allocation inside a stream capture
    # would put its zero fill into the graph, i.e. clear the list again at every replay
    self._need = {}
    if self.need_lists and int(self.blocks[0][0].c.n_items) <= 64 * 2048:
      for blk in self.blocks[0] + self.blocks[1]:
        self._need[id(blk)] = (torch.zeros(int(blk.c.n_items), dtype=torch.int32, device=device),
                               torch.zeros(1, dtype=torch.int32, device=device))
    self._stamps_at = None                 # global step at which every stamp says "up to date"
    self.lazy_flushes = 0

  def _drop_graphs(self):
    for e in self.exec:
      if e:
        self.lib.rk_graph_destroy(e)
    self.exec = [None, None]
    for e in self.exec_first + [x for d in self.exec_tail for x in d.values()]:
      if e:
        self.lib.rk_graph_destroy(e)
    self.exec_first = [None, None]
    self.exec_tail = [{}, {}]
    for e in self.exec_timed.values():
      self.lib.rk_graph_destroy(e)
    self.exec_timed = {}
    self.warmed = False

  def close(self):
    self._drop_graphs()
    for e in [self.ev_fork, self.ev_join, self.ev_join_early] + self.ev_pre:
      self.lib.rk_event_destroy(e)

  # ----------------------------------------------------------------- pieces
  def _h(self, stream):
    return ctypes.c_void_p(stream.cuda_stream)

  def _cur(self, slot):
    return self.cursors.data_ptr() + 16 * slot

  @staticmethod
  def _mark_collated(blk):
    """`blk` is (about to be) collated again: its need list and the one of the block in front of it are stale."""
    blk._gen = getattr(blk, "_gen", 0) + 1
    blk._need_for = None

  def _need_build(self, blks, stream):
    """The need lists of consecutive collated blocks (blks[i] followed by blks[i + 1]) on `stream`, behind their
    collation; blks[i]._need_for names the successor (and its collation count) the list was built for."""
    n = len(blks)
    n_items = int(blks[0].c.n_items)
    if not self.need_lists or n < 2 or n_items > 64 * 2048:       # (RK_NEED_LIST_MAX_ITEMS: larger catalogues scan)
      return
    arr = (ctypes.POINTER(_lib.RkBlock) * n)(*[ctypes.pointer(blk.c) for blk in blks])
    lists, counts = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    for i, blk in enumerate(blks[:-1]):
      buf = self._need.get(id(blk))
      if buf is None:                       # (a block that is not one of the stepper's: no list)
        return
      lists[i], counts[i] = buf[0].data_ptr(), buf[1].data_ptr()
    check(self.lib.rk_lazy_need_lists(arr, n, lists, counts, self._h(stream)), "rk_lazy_need_lists")
    for i, blk in enumerate(blks[:-1]):
      blk._need_for = (blks[i + 1], getattr(blks[i + 1], "_gen", 0))

  def _need_of(self, blk, nxt):
    """(list, count) pointers for the step of `blk` followed by `nxt`, or None (no list for this pair: the sweep scans)."""
    tag = getattr(blk, "_need_for", None) if self.need_lists and nxt is not None else None
    if tag is None or tag[0] is not nxt or tag[1] != getattr(nxt, "_gen", 0):
      return None
    buf = self._need[id(blk)]
    return ptr(buf[0]), ptr(buf[1])

  def _collate(self, blk, off, stream, slot):
    d = self.dcsr
    self._mark_collated(blk)
    blk.c.implicit = 1 if d.data is None else 0
    blk.S = self.B
    check(self.lib.rk_collate_at(ptr(d.indptr), ptr(d.indices), ptr(d.data), ptr(self.order), self.B,
                                 1 if self.ns else 0, self._cur(slot), off, blk.ref, self._h(stream)),
          "rk_collate_at")

  def _collate_many(self, blks, off0, stream, slot):
    """rk_collate_at for several blocks (cursor offsets off0, off0 + 1, ...) in ONE set of launches."""
    d = self.dcsr
    n = len(blks)
    cache = self.__dict__.setdefault("_blk_arrays", {})
    key = tuple(id(blk) for blk in blks)
    arr = cache.get(key)
    if arr is None:                       # (built once per block list: this sits on the restart path)
      arr = (ctypes.POINTER(_lib.RkBlock) * n)()
      for g, blk in enumerate(blks):
        blk.c.implicit = 1 if d.data is None else 0
        blk.S = self.B
        arr[g] = ctypes.pointer(blk.c)
      cache[key] = arr
    for blk in blks:
      self._mark_collated(blk)
    args = (ptr(d.indptr), ptr(d.indices), ptr(d.data), ptr(self.order), self.B, 1 if self.ns else 0,
            self._cur(slot), off0, arr, n)
    if self.dp is None or getattr(self.dp, "local_sets", False):     # (per-rank item sets: no stamp exchange)
      check(self.lib.rk_collate_at_multi(*args, 0, self._h(stream)), "rk_collate_at_multi")
    else:
      check(self.lib.rk_collate_at_multi(*args, 1, self._h(stream)), "rk_collate_at_multi")
      with torch.cuda.stream(stream):
        self.dp.union_marks_many([blk.mark for blk in blks])
      check(self.lib.rk_collate_at_multi(*args, 2, self._h(stream)), "rk_collate_at_multi")
    self._need_build(blks, stream)

  def _step(self, slot, g, index=None, advance=None, next_blk=None):
    """Enqueue the training step of block [slot][g] (cursor offset g) on the main stream; index
    != None: an eager step that bench.py's time plan may bracket; advance: the group's last step
    publishes the other slot's cursor; next_blk: the block of the step that follows (lazy Adam; None:
    this step's sweep leaves every row up to date)."""
    replay = dict(st=self.st[slot][g], cursor=self._cur(slot), off=g, table=ptr(self.table),
                  users=ptr(self.order_global), timed=index is not None, index=index, dw_stream=self.side,
                  next=None if advance is None else (self._cur(1 - slot), advance))
    if self.lazy:
      replay["lazy"] = dict(names=self.lazy, pos_next=None if next_blk is None else ptr(next_blk.pos),
                            need=self._need_of(self.blocks[slot][g], next_blk))
    if self.c_step:
      self.eng._c_train_step(self.blocks[slot][g], 0, self.B, None, self.loss_buf,
                             None if self.dp is None else self.B * self.dp.world, self.main, replay=replay)
    else:
      # entry-by-entry sequencing (hidden stacks, dropout, MatrixFactorization) under the replay
      # context: every state's Adam constants have a slot of their own in the table
      replay.update(tab_stride=self.tab_stride, slots=self.slots, users_t=self.order)
      self.eng.train_step(self.blocks[slot][g], 0, self.B, out=self.loss_buf, replay=replay,
                          global_rows=None if self.dp is None else self.B * self.dp.world)

  def _group(self, slot, n_steps=None, first_index=None, lookahead=True):
    """One group on slot `slot`: its steps on the main stream, the collation of the NEXT group's
    blocks on the side stream (lookahead=False: the caller knows that a cut follows and nothing
    would read them), the cursor advance.  Called inside a capture or eagerly."""
    lib, G = self.lib, self.G
    n_steps = G if n_steps is None else n_steps
    check(lib.rk_event_record(self.ev_fork, self._h(self.main)), "rk_event_record")
    check(lib.rk_stream_wait_event(self._h(self.side), self.ev_fork), "rk_stream_wait_event")
    # ONE side stream carries both kinds of side work, interleaved: the dW kernel of step g (the
    # engine enqueues it there behind the step's decode, engine.dw_branch) and then the collation of
    # block g of the next group -- dW(g) is needed by the Adam sweep of step g, the collation only
    # at the end of the group, and a third concurrent branch ends up behind one of the others on
    # the same hardware queue.  The order of the branches inside a capture decides which one the
    # graph keeps on the launching stream's queue: with step 0 captured first the training chain
    # stays on one hardware queue from launch to launch (11 us between groups; with the collation
    # first it moved to another queue every launch, 28-31 us).
    multi = G <= self.MULTI_MAX and self.multi
    for g in range(max(n_steps, G)):
      if g < n_steps:
        nxt = None
        if self.lazy and g + 1 < n_steps:
          nxt = self.blocks[slot][g + 1]
        elif self.lazy and lookahead and g > 0:
          # the group's last step: its Adam sweep reads the item map of the NEXT group's first block, collated on
          # the side stream since step 0 -- joined here instead of behind the step (long done by now)
          check(lib.rk_event_record(self.ev_join_early, self._h(self.side)), "rk_event_record")
          check(lib.rk_stream_wait_event(self._h(self.main), self.ev_join_early), "rk_stream_wait_event")
          nxt = self.blocks[1 - slot][0]
        self._step(slot, g, None if first_index is None else first_index + g,
                   advance=n_steps if g == n_steps - 1 else None, next_blk=nxt)
      if lookahead and multi:
        # the G look-ahead blocks in ONE set of launches, behind the dW kernel of step 0 (which the
        # Adam sweep of step 0 waits for; nothing needs the blocks before the end of the group)
        if g == 0:
          # (released by the event the step records behind its decode: the collation's workgroups
          # beside the fused decode + dZ launch doubled it, 26 -> 47 us)
          ev = getattr(self.eng, "_dw_objs", None)
          if self.c_step and ev is not None and self.eng._ws_dw_live:
            check(lib.rk_stream_wait_event(self._h(self.side), ev[1]), "rk_stream_wait_event")
          self._collate_many(self.blocks[1 - slot], n_steps, self.side, slot)
          if self.need_lists and n_steps >= 2:
            # (... and the list of this group's LAST step, which is followed by the first of those blocks: the early
            # join in front of that step covers it)
            self._need_build([self.blocks[slot][n_steps - 1], self.blocks[1 - slot][0]], self.side)
      elif g < G and lookahead:
        self._collate(self.blocks[1 - slot][g], n_steps + g, self.side, slot)
    check(lib.rk_event_record(self.ev_join, self._h(self.side)), "rk_event_record")
    check(lib.rk_stream_wait_event(self._h(self.main), self.ev_join), "rk_stream_wait_event")

  def _pre_collate(self, n0, slot):
    """Collate blocks[slot][0..n0) side by side: block 0 on the main stream, the others on streams
    of their own, joined back.  Called inside a capture or eagerly."""
    lib = self.lib
    if n0 <= self.MULTI_MAX and self.multi:
      # one set of launches on the main stream for all of them (4 parallel branches of a graph took
      # ~140 us to get going; three launches of 4x the workgroups take one collation's time)
      self._collate_many(self.blocks[slot][:n0], 0, self.main, slot)
      return
    if n0 > 1:
      check(lib.rk_event_record(self.ev_fork, self._h(self.main)), "rk_event_record")
    self._collate(self.blocks[slot][0], 0, self.main, slot)
    for g in range(1, n0):
      check(lib.rk_stream_wait_event(self._h(self.pre[g - 1]), self.ev_fork), "rk_stream_wait_event")
      self._collate(self.blocks[slot][g], g, self.pre[g - 1], slot)
      check(lib.rk_event_record(self.ev_pre[g - 1], self._h(self.pre[g - 1])), "rk_event_record")
    for g in range(1, n0):
      check(lib.rk_stream_wait_event(self._h(self.main), self.ev_pre[g - 1]), "rk_stream_wait_event")

  def _capture(self, enqueue):
    check(self.lib.rk_graph_begin(self._h(self.main)), "rk_graph_begin")
    try:
      enqueue()
    finally:
      ex = self.lib.rk_graph_end(self._h(self.main))
    if not ex:
      raise _lib.RecoderHipError("graph capture failed: %s" % self.lib.rk_last_error().decode())
    return ex

  # ------------------------------------------------------------------ epoch
  def begin_epoch(self, order_np, global_step):
    """Upload the epoch's user order and Adam constants; point the cursor at its first step.
    Returns the number of whole-batch steps."""
    assert torch.cuda.current_stream() == self.main, "run the epoch under torch.cuda.stream(stepper.main)"
    n = len(order_np)
    n_full = n // self.B
    assert n_full <= self.steps_cap
    # the look-ahead collation of the group BEHIND the epoch's last one reads past the order: pad
    # it with the order's own first users (B consecutive entries stay DISTINCT users, so a padding
    # block never holds more interactions / items than the blocks are sized for -- B copies of one
    # heavy user did, and overran them); those blocks are never trained on
    pad = self.order.numel() - n
    order_np = np.ascontiguousarray(order_np, dtype=np.int64)
    self.order.copy_(torch.from_numpy(np.concatenate([order_np, np.resize(order_np, pad)])),
                     non_blocking=False)
    if self.dp is not None:
      torch.add(self.order, int(self.dp.user_offset), out=self.order_global)
    # Adam constants of every step of the epoch (exactly what rk_adam_multi derives itself)
    th = self.table_host
    S = self.eng.states
    for k, name in self._slot_items():
      s = S[name]
      lr, b1, b2, eps = self.eng._adam_args(s)
      wd = 0.0 if s.sparse else float(s.wd)
      # entry (step i of the epoch, slot k) at float offset (i * tab_stride + k) * 8
      check(self.lib.rk_adam_consts(lr, b1, b2, eps, wd, s.step + 1, n_full + self.G, self.tab_stride * 8,
                                    th.data_ptr() + k * 8 * 4), "rk_adam_consts")
    self.table.copy_(th, non_blocking=False)
    self.global_step = int(global_step)
    self.epoch_base = int(global_step)
    self._cursor_at = None
    self._collated = None                # slot whose blocks hold the steps at the cursor
    return n_full

  def run(self, n_steps, eager_plan=None):
    """Run `n_steps` whole-batch steps starting at the cursor.  eager_plan(global index) -> bool:
    groups containing a step for which it is True are enqueued eagerly (timing events)."""
    lib, G = self.lib, self.G
    if n_steps <= 0:
      return
    if self.warmed and self._gen != getattr(self.eng, "alloc_gen", 0) and self.regen:
      # the engine re-allocated its workspaces since the capture (an evaluation with a larger batch
      # or item strip between two epochs, a loaded optimizer state): the graphs hold dead
      # addresses -- drop them; the next group runs eagerly and they are captured again
      torch.cuda.current_stream().synchronize()
      self._drop_graphs()
      self.recaptures += 1
    if self.lazy and self._stamps_at != self.global_step:
      # (first run, or steps ran outside this stepper -- the ragged batch of an epoch -- since the last one)
      self.eng.lazy_mark_current(self.lazy, self.global_step)
      self._stamps_at = self.global_step
    need_pre = self._collated is None
    if need_pre:
      # nothing of the first group is in flight yet: its blocks are collated in front of its steps
      # (one graph with them, or eagerly).  It restarts on the slot that does NOT hold the stale
      # look-ahead blocks: those were collated for the same step numbers -- i.e. with the same
      # stamps (rk_cur_stamp) -- but possibly other users (the padding behind an epoch's last
      # step), and a block collated twice in a row with one stamp keeps the first set's items.
      slot = 1 - self._la_slot
      self._set_cursor(slot)
    else:
      slot = self._collated
    done = 0
    while done < n_steps:
      left = n_steps - done
      idx0 = self.global_step
      if left >= G:
        eager = eager_plan is not None and any(eager_plan(idx0 + g) for g in range(G))
        # the very first group always runs eagerly: every kernel has then been launched (its code
        # object loaded) before it is captured -- graphs captured cold replayed ~8x slower on the
        # host -- and the graphs are captured right behind it (capturing enqueues nothing), i.e.
        # in the warm-up of a benchmark, never inside its timed region
        timed = self.exec_timed.get((slot, idx0, left > G)) if (eager and self.warmed and
                                                                not need_pre) else None
        if timed is not None:
          la = left > G                    # a bracketed group, captured with its events (prepare_timed)
          check(lib.rk_graph_launch(timed, self._h(self.main)), "rk_graph_launch")
        elif eager or not self.warmed:
          if need_pre:
            self._pre_collate(G, slot)
          la = left > G                    # (run() ends with this group: a cut or the epoch's end
          self._group(slot, G, first_index=idx0, lookahead=la)     # follows, nothing reads them)
          self._warm_capture()
        else:
          la = True
          if need_pre and self.multi and G <= self.MULTI_MAX:
            # behind a cut: the group's blocks are collated by ONE eagerly enqueued set of launches --
            # the GPU starts on them a few us later, and the ~50 us it takes a graph launch to reach
            # the GPU hide behind them -- then the ordinary group graph
            self._pre_collate(G, slot)
            check(lib.rk_graph_launch(self.exec[slot], self._h(self.main)), "rk_graph_launch")
          else:
            check(lib.rk_graph_launch(self.exec_first[slot] if need_pre else self.exec[slot],
                                      self._h(self.main)), "rk_graph_launch")
        k = G
      else:
        eager = eager_plan is not None and any(eager_plan(idx0 + g) for g in range(left))
        tail = self.exec_tail[slot].get(left) if (self.warmed and not need_pre and not eager) else None
        la = False
        if tail is not None:
          check(lib.rk_graph_launch(tail, self._h(self.main)), "rk_graph_launch")
        else:
          if need_pre:
            self._pre_collate(left, slot)
          self._group(slot, left, first_index=idx0, lookahead=False)   # tail: fewer than G steps, eager
          self._warm_capture()
          if self._tails_ok() and not eager:
            self._capture_tail(slot, left)     # (the next tail of this length on this slot replays)
        k = left
      need_pre = False
      self._advance_host(k)
      done += k
      slot = 1 - slot
    if self.lazy:
      # a last group WITH look-ahead left the rows stale that its look-ahead block does not read: whatever follows
      # run() (a mark, validation, a checkpoint, the ragged batch, the next epoch's constants table) sees the
      # tables the dense sweeps would have left.  (Without look-ahead the last step's own sweep did this.)
      if la:
        slots = ({"en_embedding_layer.weight": PAR_W_EN, "de_embedding_layer.weight": PAR_W_DE} if self.c_step
                 else self.slots)
        self.eng.lazy_flush(self.lazy, slots, ptr(self.table), self.tab_stride, self.global_step, self.epoch_base,
                            self._h(self.main))
        self.lazy_flushes += 1
      self._stamps_at = self.global_step
    # the look-ahead blocks of the next group -- if the last group collated them (an eagerly
    # enqueued last group does not: every caller cuts behind run(), see model._run_epoch_graph)
    self._collated = slot if la else None
    if la:
      self._la_slot = slot               # (remembered across cuts / epochs: see the restart above)

  def prepare_timed(self, first_index, lookahead=False):
    """Capture, for both slots, the group of G steps starting at global step `first_index` WITH the
    timing events the engine's time plan asks for (event-record nodes: csrc/step.hip timer_record),
    with or without the look-ahead collation (without: run() ends with that group).  bench.py calls
    this in front of its timed region; run() then replays it instead of enqueueing the bracketed
    group eagerly."""
    if not self.warmed or not self.lib.rk_graph_timing_supported():
      return False
    from ._lib import ENTRY
    self.eng.reserve_timing_events(2 * self.G * 2 * (max(ENTRY.values()) + 1))   # (created outside the capture)
    for v in (0, 1):
      key = (v, int(first_index), bool(lookahead))
      if key not in self.exec_timed:
        self.exec_timed[key] = self._capture(
            lambda v=v: self._group(v, self.G, first_index=int(first_index), lookahead=bool(lookahead)))
    return True

  def _warm_capture(self):
    """Capture the graphs once every kernel has been launched eagerly (the first steps run)."""
    if self.warmed:
      return
    self.warmed = True
    self._gen = getattr(self.eng, "alloc_gen", 0)
    G = self.G
    for v in (0, 1):
      if self.exec[v] is None:
        self.exec[v] = self._capture(lambda v=v: self._group(v))
    for v in (0, 1):
      if self.exec_first[v] is None:
        self.exec_first[v] = self._capture(lambda v=v: (self._pre_collate(G, v), self._group(v)))

  # the tails: n < G steps in front of a cut or the epoch's end, without a look-ahead collation.  Enqueued launch
  # by launch such a step costs the GPU ~17 us more than replayed (the driver's 20-step run = 8 + 8 + 4 steps:
  # DESIGN.md section 5); one-call step, single process only.  Captured on demand (all 2 (G - 1) of
  # them at warm-up were ~56 step captures that every re-allocation of the engine's workspaces threw away, and an
  # epoch uses one length): the first tail of a length runs launch by launch and is captured behind it;
  # prepare_tails captures the lengths a caller knows it will need (bench.py, in front of its clock)
  def _tails_ok(self):
    return self.warmed and self.c_step and self.dp is None and self.capture_tails

  def _capture_tail(self, slot, n):
    if n not in self.exec_tail[slot]:
      self.exec_tail[slot][n] = self._capture(lambda: self._group(slot, n, lookahead=False))

  def prepare_tails(self, lengths):
    if not self._tails_ok():
      return False
    for v in (0, 1):
      for n in lengths:
        if 0 < int(n) < self.G:
          self._capture_tail(v, int(n))
    return True

  def cut(self):
    """Forget the look-ahead blocks (a step mark / an eager ragged step follows).  The cursor of the
    slot the next run() restarts on is pointed at the next step HERE (a one-thread launch that knows
    nothing of the steps to come: it need not sit between a caller's mark and its first step)."""
    self._collated = None
    if self.warmed:
      self._set_cursor(1 - self._la_slot)

  def precollate(self):
    """Behind a cut(): collate the next group's blocks NOW instead of in front of its steps -- what the
    look-ahead collation of the previous group would have done had there been no cut.  bench.py calls
    it from its start mark, in front of the clock: the timed region then holds exactly one look-ahead
    collation per group (the one behind its last group included), like any stretch of steady state."""
    if not self.warmed or self._collated is not None or not (self.multi and self.G <= self.MULTI_MAX):
      return False
    slot = 1 - self._la_slot
    self._set_cursor(slot)
    self._pre_collate(self.G, slot)
    self._collated = slot
    return True

  def _set_cursor(self, slot):
    key = (slot, self.global_step, self.epoch_base)
    if getattr(self, "_cursor_at", None) == key:
      return
    check(self.lib.rk_cursor_set(self._cur(slot), self.global_step, self.epoch_base, self._h(self.main)),
          "rk_cursor_set")
    self._cursor_at = key

  def _slot_items(self):
    """(table slot, state name) of every parameter the replayed steps update."""
    if not self.c_step:
      return [(i, name) for name, i in self.slots.items()]
    tied = bool(self.eng.model.is_constrained)
    return [(k, name) for k, name in _PAR_NAMES.items() if not (k == PAR_W_DE and tied)]

  def _advance_host(self, k):
    self._cursor_at = None               # (the group's last Adam launch moved the cursors)
    self.global_step += k
    S = self.eng.states
    for _, name in self._slot_items():
      S[name].step += k
    self.eng.rng_step += k

  def losses(self, n):
    return self.loss_buf[:n]
