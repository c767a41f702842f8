"""Fused training / inference step on the HIP kernels.

``FusedEngine`` sequences the C-ABI calls that replace one iteration of the
reference's hot loop (model.py:383-404): ``__compute_loss`` (model.py:454-485:
densify -> model forward -> loss / B), ``loss.backward()`` and the
``Adam`` / ``SparseAdam`` steps -- without densifying the batch and without a
host synchronisation.  All arithmetic is in the HIP library; torch is used for
device memory and streams only.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ACT, LOSS_BCE, LOSS_MNLL, LOSS_MSE, LOSS_NONE, check, ptr
from .device import Block, cdiv, current_stream, require_gpu

ADAM_MULTI_MAX = 10           # RK_ADAM_MULTI_MAX (include/recoder_hip.h): jobs of one rk_adam_multi launch

LOSS_IDS = {"mse": LOSS_MSE, "logistic": LOSS_BCE, "logloss": LOSS_MNLL}


def _f32(x):
  return float(np.float32(x))


class ParamState:
  """One trainable tensor + its Adam moments (tensors live in the
  torch.optim state dict so checkpoints keep the reference layout)."""
  __slots__ = ("name", "p", "m", "v", "step", "wd", "group", "sparse", "st")

  def __init__(self, name, p, m, v, step, wd, group, sparse, st):
    self.name, self.p, self.m, self.v = name, p, m, v
    self.step, self.wd, self.group, self.sparse, self.st = step, wd, group, sparse, st


class TimedLib:
  """Pass-through to the C ABI that can bracket every call with HIP events on
  the launch stream (bench.py's roofline measurement); disabled by default."""

  def __init__(self, lib):
    self._lib = lib
    self.enabled = False
    self.records = {}       # entry point -> list of (start_event, end_event)
    self.streams = {}       # raw hipStream_t value -> torch stream (for event placement)

  def __getattr__(self, name):
    fn = getattr(self._lib, name)
    # (sizing accessors -- plain Python over rk_plan -- and host-only calls are not launches)
    if not name.startswith("rk_") or not hasattr(fn, "argtypes") or name in (
        "rk_plan", "rk_planes_layout", "rk_last_error", "rk_version",
        "rk_ae_step_uses_pg"):
      return fn

    def call(*args):
      if not self.enabled:
        return fn(*args)
      st = self.streams.get(getattr(args[-1], "value", None)) if args else None
      st = st or torch.cuda.current_stream()
      s = torch.cuda.Event(enable_timing=True)
      e = torch.cuda.Event(enable_timing=True)
      s.record(st)
      rc = fn(*args)
      e.record(st)
      self.records.setdefault(name, []).append((s, e))
      return rc
    return call

  def summary(self):
    """{entry point: (calls, mean ms)}; synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), float(np.mean([s.elapsed_time(e) for s, e in v])))
            for k, v in self.records.items()}

  def reset(self):
    self.records = {}


class FusedEngine:
  """Holds the workspaces for batches of up to ``B_cap`` rows x ``n_cap`` items."""

  def __init__(self, model, kind, loss="mse", loss_params=None, device=None):
    self.lib = TimedLib(_lib.load())
    self.device = device or require_gpu()
    self.model = model
    self.kind = kind                       # 'ae' | 'mf'
    if isinstance(loss, str):
      if loss not in LOSS_IDS:
        raise ValueError("Unknown loss function {}".format(loss))
      self.loss_id = LOSS_IDS[loss]
    else:
      raise ValueError("the fused engine needs a named loss ('mse', 'logistic', 'logloss')")
    self.confidence = float((loss_params or {}).get("confidence", 0))
    self.act = ACT[model.activation_type]
    self.B_cap = 0
    self.n_cap = 0
    self.seed = 0x5eed
    self.rng_step = 0
    self.states = {}                       # name -> ParamState
    self._jobs = []                        # rk_adam_job_t records of the step being assembled
    self.world_size = 1
    self.allreduce = None                  # callable(list of tensors) for data parallel
    self.use_c_step = True                 # one-FFI-call step driver (rk_ae_train_step)
    # bench.py's roofline measurement: time_plan(call index) -> C-ABI entry name (or None) whose
    # launch(es) the C step driver brackets with HIP events on the step's stream in that call
    self.time_plan = None
    self._time_samples = []                # (entry name, event0, event1)
    self._time_keep = []
    self._gb_lazy = None
    self._pending_loss = None
    self.item_parallel = None              # parallel.ItemParallel when the items are sharded
    self._cstep = None
    self._c_calls = 0
    # dW on a stream of its own next to the dZ -> encoder-backward chain; RK_DW_BRANCH=0: in line
    self.dw_branch = True
    self._dw_objs = None
    # operand ranges of the split-fp16 decoder contractions (include/recoder_hip.h rk_amax):
    # [0..63] max |Z| (filled per call when the activation is unbounded), [64..127] an upper bound
    # of |decoder table| (its maximum now; the Adam sweep keeps it running from there)
    self.ranges = torch.zeros(128, dtype=torch.int32, device=self.device)
    # lazy dense Adam of the embedding tables (include/recoder_hip.h rk_adam_job_t.lazy_stamp): the round-robin
    # period that bounds a row's lag; RK_ADAM_LAZY=0 switches it off (every row swept every step)
    # ("<period>,list" / "<period>,scan": force the sweeps' need lists on / off -- graph.GraphStepper decides by shape)
    _lz = os.environ.get("RK_ADAM_LAZY", "16").split(",")
    self.lazy_period = max(0, int(_lz[0]))
    self.lazy_lists = _lz[1] if len(_lz) > 1 and _lz[1] in ("list", "scan") else "auto"
    self._lazy_stamps = {}
    self.act_bounded = model.activation_type in ("tanh", "sigmoid")
    self._w_range_stale = True
    self._w_range_key = None
    if kind == "ae":
      self.h = list(model.hidden_layers)
      self.nl = len(self.h) - 1
      if self.h[0] % 4 != 0:
        raise ValueError("hidden_layers[0] must be a multiple of 4 (16-byte embedding rows)")
    else:
      self.h = [model.embedding_size]
      self.nl = 0
      if self.h[0] % 4 != 0:
        raise ValueError("embedding_size must be a multiple of 4 (16-byte embedding rows)")

  # ------------------------------------------------------------------ setup
  def ensure_capacity(self, B_cap, n_cap):
    if B_cap <= self.B_cap and n_cap <= self.n_cap:
      return
    B_cap, n_cap = max(B_cap, self.B_cap), max(n_cap, self.n_cap)
    # every workspace below is re-allocated: captured HIP graphs hold the old addresses
    # (graph.GraphStepper compares this counter and re-captures)
    self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
    dev = self.device
    f = dict(dtype=torch.float32, device=dev)
    h0 = self.h[0]
    ld_cap = cdiv(n_cap, 32) * 32
    self.B_cap, self.n_cap, self.ld_cap = B_cap, n_cap, ld_cap
    # (zeroed: the padding columns [n_b, ld) of its rows meet the zeros of the W^T plane image in the
    # dZ contraction and must be finite)
    # (rows in whole groups of 32: as a plane IMAGE -- csrc/pgemm.h, the same footprint -- dW reads it
    # along its rows in 32-row k-tiles)
    self.do_rows = cdiv(B_cap, 32) * 32
    self.dO = torch.zeros(self.do_rows * ld_cap, **f)
    # per-granule split scales of that image (rk_pg_decode_loss)
    self.do_scales = torch.ones(self.lib.rk_pg_scale_floats(B_cap, n_cap), **f)
    self.G_de = torch.zeros(n_cap * h0, **f)
    # rows of the compact gradient arrays at or past this mark are zero (rk_zero_tail_rows, data-parallel replay)
    self._grad_hwm = torch.zeros(1, dtype=torch.int32, device=dev)
    # MF: [B | the step's user rows as int32] left by the forward's gather for the SparseAdam job of
    # the user table (rk_gather_rows_amax)
    self._users32_buf = torch.zeros(B_cap + 1, dtype=torch.int32, device=dev) if self.kind == "mf" else None
    # the fused dW + encoder-backward launch writes G_en in row segments (long item columns)
    self.G_en = torch.zeros(n_cap * h0 * self.lib.rk_encode_bwd_segments(B_cap), **f)
    # small gradients in ONE buffer [gb_en (h0) | loss | pad | gb_de (n_cap)] so that a
    # data-parallel step reduces them with a single collective over [0, off + n_b)
    self.small_off = cdiv(h0 + 1, 4) * 4
    self.small = torch.zeros(self.small_off + n_cap, **f)
    self.gb_de = self.small[self.small_off:]
    self.row_tile = self.lib.rk_decode_row_tile()
    self.gb_part = torch.empty(cdiv(B_cap, self.row_tile) * ld_cap, **f)   # per-row-tile colsums of dO
    self.gb_en = self.small[:h0]
    # the one-call step gets the encoder-bias gradient as row-segment partial vectors
    self.gb_en_parts = torch.zeros(8 * h0, **f)
    self.loss_dp = self.small[h0:h0 + 1]
    # one workspace, used in turn by the dZ split-K slabs, then by dW (bf16-pipe kernel: Z^T
    # planes + K slabs, which stay live until the Adam sweep has read them)
    self.ws = torch.zeros(max(self.lib.rk_dz_workspace_bytes(B_cap, h0),
                              # (any batch below 1024 rows -- a ragged last one too -- may take the fused form)
                              self.lib.rk_dz_fused_workspace_bytes(min(B_cap, 1023), h0, n_cap),
                              self.lib.rk_fdec_workspace_bytes(B_cap, h0, n_cap),
                              self.lib.rk_dw_workspace_bytes(B_cap, h0, n_cap),
                              self.lib.rk_dw3_workspace_bytes(B_cap, h0, n_cap),
                              self.lib.rk_pg_dz_workspace_bytes(B_cap, h0),
                              self.lib.rk_pg_dw_workspace_bytes(B_cap, h0, n_cap)) // 4 + 64, **f)
    self.split16 = bool(self.lib.rk_gemm_split16())
    self._dw_slabs = None
    # dW as a branch of the one-call step (rk_ae_step_t.dw_stream): a workspace of its own
    self.ws_dw = (torch.zeros(max(self.lib.rk_dw3_workspace_bytes(B_cap, h0, n_cap),
                                  self.lib.rk_pg_dw_workspace_bytes(B_cap, h0, n_cap)) // 4 + 64, **f)
                  if self.split16 and self.dw_branch else None)
    # Z^T as bf16 planes for the dW kernel, written by the encoder forward of the one-call step
    # (zeroed once: the padding columns are never written)
    self.zt_planes = torch.zeros(self.lib.rk_dw3_planes_bytes(B_cap, h0) // 4 + 16, **f)
    # pre-split operand planes of the decoder contractions (csrc/planes.h); RK_PLANES=0: the
    # in-loop split of round 2
    self.planes = None
    if self.split16:
      from ._lib import RkPlanes
      self.planes_buf = torch.zeros(self.lib.rk_planes_bytes(B_cap, h0, n_cap) // 4 + 64, **f)
      self.planes = RkPlanes()
      check(self.lib.rk_planes_layout(ptr(self.planes_buf), B_cap, h0, n_cap, ctypes.byref(self.planes)),
            "rk_planes_layout")
    # multinomial NLL on the pipelined kernels: the statistics pass's pairs + the rows' target sums
    self.mnll_ws = (torch.zeros(self.lib.rk_pg_mnll_workspace_floats(B_cap, n_cap), **f)
                    if self.loss_id == LOSS_MNLL else None)
    self.planes_nowt = None
    if self.planes is not None:
      # the same images without the W^T one (csrc/pgemm.h reads the W image along its rows)
      from ._lib import RkPlanes
      self.planes_nowt = RkPlanes()
      ctypes.memmove(ctypes.byref(self.planes_nowt), ctypes.byref(self.planes), ctypes.sizeof(RkPlanes))
      self.planes_nowt.wt = None
    self.n_part = self.lib.rk_loss_partials(B_cap, n_cap)
    self.loss_part = torch.zeros(self.n_part, **f)
    self.loss_out = torch.zeros(1, **f)
    # activations: enc[i] = output of encoder layer i (post activation), i = 0..nl
    self.enc = [torch.empty(B_cap * self.h[i], **f) for i in range(self.nl + 1)]
    self.denc = [torch.empty(B_cap * self.h[i], **f) for i in range(self.nl + 1)]
    hb = self.h[-1]
    self.bott = torch.empty(B_cap * hb, **f)         # post-dropout bottleneck
    # dec[i] = output of decoder layer i (i = 0..nl-1); sizes reversed(h)[i+1]
    rh = list(reversed(self.h))
    self.dec = [torch.empty(B_cap * rh[i + 1], **f) for i in range(self.nl)]
    self.ddec = [torch.empty(B_cap * rh[i + 1], **f) for i in range(self.nl)]
    self.dbott = torch.empty(B_cap * hb, **f)
    # gradients of the hidden Linear stack
    if self.kind == "ae":
      m = self.model
      self.g_enc_w = [torch.empty_like(l.weight) for l in m.encoding_layers]
      self.g_enc_b = [torch.empty_like(l.bias) for l in m.encoding_layers]
      self.g_dec_w = [None if m.is_constrained else torch.empty_like(l.weight)
                      for l in m.decoding_layers]
      self.g_dec_b = [torch.empty_like(l.bias) for l in m.decoding_layers]
    else:
      self.pos_u = torch.full((self.model.num_users,), -1, dtype=torch.int32, device=dev)

  # ------------------------------------------------------------- optimiser
  def bind_optimizers(self, optimizer, sparse_optimizer):
    """Create / adopt the Adam moment tensors inside the torch.optim state
    (so ``optimizer.state_dict()`` has the reference layout, model.py:210)."""
    self.states = {}
    self._w_range_stale = True
    self.alloc_gen = getattr(self, "alloc_gen", 0) + 1     # (the moment tensors may be new ones)
    names = {id(p): n for n, p in self.model.named_parameters()}
    for opt, is_sparse in ((optimizer, False), (sparse_optimizer, True)):
      if opt is None:
        continue
      for group in opt.param_groups:
        for p in group["params"]:
          st = opt.state[p]
          if "exp_avg" not in st:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
          else:
            st["exp_avg"] = st["exp_avg"].to(p.device).contiguous()
            st["exp_avg_sq"] = st["exp_avg_sq"].to(p.device).contiguous()
          step = int(st["step"]) if not torch.is_tensor(st["step"]) else int(st["step"].item())
          self.states[names[id(p)]] = ParamState(
              names[id(p)], p, st["exp_avg"], st["exp_avg_sq"], step,
              0.0 if is_sparse else float(group.get("weight_decay", 0.0)), group, is_sparse, st)

  def sync_optimizer_steps(self):
    """Write the host-side step counters back into the torch.optim state."""
    for s in self.states.values():
      s.st["step"] = torch.tensor(float(s.step))

  def _adam_args(self, s):
    g = s.group
    b1, b2 = g["betas"]
    return float(g["lr"]), float(b1), float(b2), float(g["eps"])

  # ---- lazy dense Adam (graph.GraphStepper drives it: it knows the NEXT step's block) ----
  def lazy_tables(self):
    """Names of the dense-Adam embedding tables whose sweeps can skip rows (DynamicAutoencoder, single process):
    optim.Adam on a dense embedding gradient (model.py:135,398-399) touches every row every step; rows outside
    the step's item set that the next step does not read are caught up later by replaying their missed
    steps bit for bit (csrc/optim.hip table_sweep_lazy)."""
    if (self.lazy_period < 1 or self.kind != "ae" or self.item_parallel is not None or
        self.h[0] % 4 != 0):
      return []
    if self.allreduce is not None:
      # users-DP: the replicated update only (every rank sweeps the same rows of identical tables; the blocks' item
      # maps are the union sets, equal on all ranks) -- not the sharded / per-rank-item-set / owned-row variants
      dp = self.allreduce
      if (getattr(self, "zero_adam", False) or getattr(dp, "local_sets", False) or getattr(self, "owned_rows", False)):
        return []
    names = ["en_embedding_layer.weight"] + ([] if self.model.is_constrained else ["de_embedding_layer.weight"])
    if not all(n in self.states and not self.states[n].sparse for n in names):
      return []
    if any(self.states[n].p.numel() >= (1 << 32) for n in names):     # (the lean table sweeps index with 32 bits)
      return []
    return names

  def lazy_stamp(self, name):
    """[n_rows] int32: global index of the first step not yet applied to a row of table `name`."""
    t = self._lazy_stamps.get(name)
    n = self.states[name].p.shape[0]
    if t is None or t.numel() != n:
      t = torch.zeros(n, dtype=torch.int32, device=self.device)
      self._lazy_stamps[name] = t
    return t

  def lazy_mark_current(self, names, step):
    """Every row of the tables is up to date in front of global step `step` (the dense sweeps ran)."""
    for n in names:
      self.lazy_stamp(n).fill_(int(step))

  def lazy_flush(self, names, slots, table, tab_stride, next_step, epoch_base, stream):
    """Replay every row's missed steps up to (excluding) global step `next_step`: rk_adam_lazy_flush."""
    from ._lib import RkAdamJob
    arr = (RkAdamJob * len(names))()
    sl = (ctypes.c_int32 * len(names))()
    W_dec = self._decoder_params()[0]
    for i, n in enumerate(names):
      st = self.states[n]
      a = arr[i].par
      a.p, a.m, a.v = ptr(st.p), ptr(st.m), ptr(st.v)
      arr[i].n_rows, arr[i].h = st.p.shape[0], st.p.shape[1]
      arr[i].lazy_stamp, arr[i].lazy_period = ptr(self.lazy_stamp(n)), 1
      if st.p is W_dec:
        arr[i].amax_out = self.ranges.data_ptr() + 64 * 4
      sl[i] = slots[n]
    check(self.lib.rk_adam_lazy_flush(arr, len(names), table, tab_stride, sl, int(next_step), int(epoch_base), stream),
          "rk_adam_lazy_flush")

  # The updates of one step are collected as rk_adam_job_t records and issued through
  # rk_adam_multi, ten per launch (a hidden-stack model has ~11 parameter tensors, i.e. ~11 launches and FFI calls
  # otherwise).  Index arrays that are int64 (MF user rows) go through rk_adam_rows directly.
  def _job(self, s, n_rows, h, g, pos=None, rows=None, n_dev=None, n_cap=0, parts=None):
    """parts = (pointer, g_parts, g_stride, gstride_dev, gparts_dev): the gradient is the sum, in
    order, of that many partial arrays (rk_adam_job_t) instead of the single array `g`."""
    from ._lib import RkAdamJob
    rp = getattr(self, "_replay", None)
    if rp is None:
      s.step += 1
    lr, b1, b2, eps = self._adam_args(s)
    j = RkAdamJob()
    a = j.par
    a.p, a.m, a.v = ptr(s.p), ptr(s.m), ptr(s.v)
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, b1, b2, eps, float(s.wd)
    # (replay: `step` carries the parameter's slot + 1 in the per-epoch constants table, rk_replay_t)
    a.step = s.step if rp is None else rp["slots"][s.name] + 1
    a.sparse = 1 if rows is not None else 0
    j.n_rows, j.h, j.g, j.g_parts = n_rows, h, ptr(g), 1
    j.pos, j.rows, j.n_dev, j.n_cap = ptr(pos), ptr(rows), ptr(n_dev), n_cap
    if parts is not None:
      j.g, j.g_parts, j.g_stride, j.gstride_dev, j.gparts_dev = parts
    if s.p is self._decoder_params()[0]:       # the table the decoder GEMMs read: keep its bound
      j.amax_out = self.ranges.data_ptr() + 64 * 4
    lz = rp.get("lazy") if rp is not None else None
    if lz is not None and pos is not None and rows is None and s.name in lz["names"]:
      j.lazy_stamp, j.lazy_pos_next, j.lazy_period = ptr(self.lazy_stamp(s.name)), lz["pos_next"], self.lazy_period
      if lz["pos_next"] is not None and lz.get("need"):
        j.lazy_need_list, j.lazy_need_count = lz["need"]
    self._jobs.append(j)

  def _flush_jobs(self, stream):
    from ._lib import RkAdamJob
    jobs, self._jobs = self._jobs, []
    for i in range(0, len(jobs), ADAM_MULTI_MAX):
      chunk = jobs[i:i + ADAM_MULTI_MAX]
      arr = (RkAdamJob * len(chunk))(*chunk)
      pend = self._pending_loss if i + ADAM_MULTI_MAX >= len(jobs) else None
      if pend is not None:
        # the deferred reduction of the step's loss partials rides on the sweep (as in rk_ae_train_step)
        n_part, denom, out = pend[:3]
        src = pend[3] if len(pend) > 3 else self.loss_part
        self._pending_loss = None
        check(self.lib.rk_adam_multi(arr, len(chunk), ptr(src), n_part, denom, ptr(out), stream),
              "rk_adam_multi")
      else:
        check(self.lib.rk_adam_multi(arr, len(chunk), None, 0, 1.0, None, stream), "rk_adam_multi")

  def _adam_table(self, s, pos, G, h, n_rows, stream, parts=None):
    self._job(s, n_rows, h, G, pos=pos, parts=parts)

  def _adam_rows(self, s, idx32, idx64, n_dev, n_cap, G, h, stream, parts=None):
    if idx32 is not None and n_dev is not None:
      self._job(s, 0, h, G, rows=idx32, n_dev=n_dev, n_cap=n_cap, parts=parts)
      return
    assert parts is None
    rp = getattr(self, "_replay", None)
    if rp is None:
      s.step += 1
    lr, b1, b2, eps = self._adam_args(s)
    check(self.lib.rk_adam_rows(ptr(s.p), ptr(s.m), ptr(s.v), h, ptr(idx32), ptr(idx64), ptr(n_dev),
                                n_cap, ptr(G), lr, b1, b2, eps,
                                s.step if rp is None else rp["slots"][s.name] + 1, stream), "rk_adam_rows")

  def _adam_dense(self, s, g, stream):
    self._job(s, 1, s.p.numel(), g)

  # --------------------------------------------------------------- forward
  def _ae_forward(self, blk, row_off, B, keep_noise, keep_drop, train, stream):
    m, lib = self.model, self.lib
    p_noise = float(m.noise_prob) if train else 0.0
    ip = self.item_parallel if train else None
    if ip is not None:
      # item parallel: partial sums over this rank's items -> all-reduce -> bias + activation
      h0 = self.h[0]
      check(lib.rk_ae_encode_fwd_partial(blk.ref, row_off, B, ptr(m.en_embedding_layer.weight), h0,
                                         ptr(keep_noise), p_noise, self.seed, self.rng_step,
                                         ptr(blk.users), ptr(ip.user_norm_dev), ptr(self.enc[0]),
                                         stream), "rk_ae_encode_fwd_partial")
      ip.allreduce_sum(self.enc[0][:B * h0])
      check(lib.rk_bias_act(ptr(self.enc[0]), ptr(m.en_bias), B, h0, self.act, stream), "rk_bias_act")
    elif train and getattr(self, "_split_w_with_fwd", False):
      # the W_de[items] half of the decode's operand split rides on this launch (as in the one-call
      # step); _loss then only cuts Z
      self._check_weight_range()
      W_de, _ = self._decoder_params()
      check(lib.rk_ae_encode_fwd_split_w(blk.ref, row_off, B, ptr(m.en_embedding_layer.weight),
                                         ptr(m.en_bias), self.h[0], ptr(keep_noise), p_noise, self.seed,
                                         self.rng_step, ptr(blk.users), self.act, ptr(self.enc[0]),
                                         ptr(W_de), ptr(self.ranges),
                                         ctypes.byref(self.planes_nowt if getattr(self, "_split_nowt", False)
                                                      else self.planes), stream),
            "rk_ae_encode_fwd")
      self._w_split_of = blk
    else:
      check(lib.rk_ae_encode_fwd(blk.ref, row_off, B, ptr(m.en_embedding_layer.weight),
                                 ptr(m.en_bias), self.h[0], ptr(keep_noise), p_noise, self.seed,
                                 self.rng_step, ptr(blk.users), self.act, ptr(self.enc[0]), stream),
            "rk_ae_encode_fwd")
    for i, layer in enumerate(m.encoding_layers):
      check(lib.rk_linear_fwd(ptr(self.enc[i]), ptr(layer.weight), ptr(layer.bias), B, self.h[i + 1],
                              self.h[i], 0, self.act, ptr(self.enc[i + 1]), stream), "rk_linear_fwd")
    z = self.enc[self.nl]
    self.drop_active = bool(train and m.dropout_prob > 0.0)
    if self.drop_active:
      n = B * self.h[-1]
      self.bott[:n].copy_(z[:n])
      check(lib.rk_dropout(ptr(self.bott), ptr(keep_drop), n, self.h[-1], float(m.dropout_prob),
                           self.seed ^ 0xd0d0, self.rng_step, stream), "rk_dropout")
      z = self.bott
    self.dec_in = z
    rh = list(reversed(self.h))
    for i, layer in enumerate(m.decoding_layers):
      if m.is_constrained:
        w, wt = m.encoding_layers[self.nl - 1 - i].weight, 1
      else:
        w, wt = layer.weight, 0
      check(lib.rk_linear_fwd(ptr(z), ptr(w), ptr(layer.bias), B, rh[i + 1], rh[i], wt, self.act,
                              ptr(self.dec[i]), stream), "rk_linear_fwd")
      z = self.dec[i]
    return z

  def _mf_forward(self, users, B, keep_drop, train, stream):
    m, lib = self.model, self.lib
    d = self.h[0]
    self.drop_active = bool(train and m.dropout_prob > 0)
    self._amax_of = None
    self._users32 = None
    want_amax = not self.act_bounded and not self.drop_active and self.split16
    # SparseAdam on the user table: the gather leaves the step's users as an int32 index array (+
    # count) so that the update is a job of the step's rk_adam_multi launch, not an rk_adam_rows launch
    su = self.states.get("user_embedding_layer.weight") if train else None
    want_rows = su is not None and su.sparse and self.allreduce is None and self.item_parallel is None
    if (want_amax or want_rows) and B > 0:
      # unbounded activation: the split contractions need max |z| -- the gather publishes it
      # (64 slots, one per workgroup) instead of an rk_amax launch behind it
      if want_rows and (getattr(self, "_users32_buf", None) is None or self._users32_buf.numel() < B + 1):
        want_rows = False        # (ensure_capacity has not run for this batch size yet: the plain update)
      check(lib.rk_gather_rows_amax(ptr(m.user_embedding_layer.weight), ptr(users), B, d, self.act,
                                    ptr(self.enc[0]), ptr(self.ranges) if want_amax else None,
                                    ptr(self._users32_buf) if want_rows else None, stream),
            "rk_gather_rows_amax")
      if want_amax:
        self._amax_of = (self.enc[0].data_ptr(), B * d)
      if want_rows:
        self._users32 = self._users32_buf
    else:
      if B > 0:
        check(lib.rk_gather_rows_amax(ptr(m.user_embedding_layer.weight), ptr(users), B, d, self.act,
                                      ptr(self.enc[0]), None, None, stream), "rk_gather_rows_amax")
    z = self.enc[0]
    if self.drop_active:
      n = B * d
      self.bott[:n].copy_(z[:n])
      check(lib.rk_dropout(ptr(self.bott), ptr(keep_drop), n, d, float(m.dropout_prob),
                           self.seed ^ 0xd0d0, self.rng_step, stream), "rk_dropout")
      z = self.bott
    return z

  def _decoder_params(self):
    m = self.model
    if self.kind == "ae":
      return m.de_embedding_layer.weight, m.de_bias
    return m.item_embedding_layer.weight, m.bias

  def refresh_weight_range(self):
    """(Re)compute the bound of |decoder table| from the table itself: after the parameters were
    set from outside (construction, load_state_dict); training keeps it current on the device."""
    W, _ = self._decoder_params()
    m = torch.linalg.vector_norm(W.detach(), float("inf")).reshape(1).to(torch.float32)
    self.ranges[64:].zero_()
    self.ranges[64:65].copy_(m.view(torch.int32))
    self._w_range_stale = False
    self._w_range_key = (W.data_ptr(), W._version)

  def _check_weight_range(self):
    if not self._w_range_stale:
      W, _ = self._decoder_params()
      if (W.data_ptr(), W._version) == self._w_range_key:   # in-place writes from torch bump it
        return
    self.refresh_weight_range()

  def _ranges(self, z, n, stream):
    """Device pointer of the operand ranges for a decode of `z` (n elements)."""
    self._check_weight_range()
    if self.act_bounded:
      return ptr(self.ranges)         # |Z| <= 1 (tanh / sigmoid): the static scale of Z is exact
    if getattr(self, "_amax_of", None) == (z.data_ptr(), n):
      return ptr(self.ranges)         # (rk_gather_rows_amax of this forward left it there)
    check(self.lib.rk_amax(ptr(z), n, ptr(self.ranges), stream), "rk_amax")
    return ptr(self.ranges)

  def _fdec_entry_ok(self, B, n_cap, row_off=0, own_block=True):
    """Steps sequenced entry by entry on the register-resident fused decode (rk_fdec_loss_dz: decode + loss + dZ
    partials, dLoss/dLogits as a plane image) -- round 5.  MatrixFactorization: dW from the image, its column sums
    and the slab reduce in ONE launch behind it (rk_pg_dw_dz_reduce; users-DP: its dense form).  Autoencoders with
    hidden stacks / bottleneck dropout (single process, untied, MSE / BCE, the block is its own target):
    rk_fdec_dz_reduce, the stack's backward, then dW || column sums || encoder backward (rk_pg_dw_encode_bwd)."""
    lib = self.lib
    if not (self.planes is not None and self.split16 and self.ws_dw is not None and self.item_parallel is None and
            not lib.rk_gemm_plain_bf16() and bool(lib.rk_mf_fdec_ok(B, self.h[0], n_cap, self.loss_id))):
      return False
    if self.kind == "mf":
      return True
    return (self.allreduce is None and not bool(self.model.is_constrained) and own_block and
            bool(lib.rk_dw_encode_bwd_fused_ok(row_off, B)))

  def _pg_entry_ok(self, B, n_cap):
    """Entry-by-entry steps: the three contractions on the pipelined pair-plane kernels (csrc/pgemm.h)
    -- everything outside the fused decode + dZ launch's domain: the multinomial loss, h > 256,
    >= 1024 rows."""
    lib = self.lib
    # (the multinomial loss as TWO decode passes pays where a pass is flop-bound, not at B = 500: C3,
    # n_b = 8.4 k -- 25.6 + 5.2 + 27.5 us for statistics / merge / loss passes against 19.7 + 18.9 us for
    # the decode that writes the logits + rk_mnll_finish; RK_PG_MNLL=1 forces it)
    if self.loss_id == LOSS_MNLL and B < 1024:
      return False
    return (self.planes is not None and self.split16 and bool(lib.rk_pg_enabled()) and
            not lib.rk_gemm_plain_bf16() and self.item_parallel is None and
            not lib.rk_decode_dz_fused_ok(B, self.h[0], n_cap, self.loss_id))

  def _loss(self, z, B, tgt, row_off, denom_rows, stream, out=None, ip=None, defer=False, fuse_dz=False,
            zt_ws=None, pg_ok=False, fdec_ok=False, own_block=True):
    """decode + loss; leaves dLoss/dLogits in self.dO. Returns device scalar.  ip: the
    block holds an item shard (parallel.ItemParallel) -- only the multinomial loss needs to
    know: its softmax statistics are combined over the ranks."""
    lib = self.lib
    W, b = self._decoder_params()
    inv_B = _f32(np.float32(1.0) / np.float32(denom_rows))
    out = self.loss_out if out is None else out
    self._dz_in_ws = False
    self._dz_on_planes = False
    self._dz_pg = False
    self._dz_fdec = False
    # zt_ws: the workspace the step's dW launch will use -- the split launch then writes Z^T as that
    # kernel's fp16 pair planes at its head (one launch less inside rk_decode_bwd_dw2)
    self._zt_ready = None
    if zt_ws is not None and not self.lib.rk_split_zt_ok():
      zt_ws = None
    # (the encoder forward of this step already cut W_de[items of this block]: rk_ae_encode_fwd_split_w)
    w_done = getattr(self, "_w_split_of", None) is tgt and tgt is not None
    self._w_split_of = None
    if fuse_dz and fdec_ok and ip is None and self._fdec_entry_ok(B, tgt.n_cap, row_off, own_block):
      h0 = self.h[0]
      rg = self._ranges(z, B * h0, stream)
      check(lib.rk_split_wz(None if w_done else ptr(W), ptr(z), B, h0, tgt.ref, rg,
                               ctypes.byref(self.planes_nowt), None, stream), "rk_split_wz")
      check(lib.rk_fdec_loss_dz(ctypes.byref(self.planes), B, tgt.ref, row_off, ptr(b), self.loss_id,
                                self.confidence, inv_B, ptr(self.dO), self.do_rows, ptr(self.do_scales),
                                ptr(self.loss_part), ptr(self.ws), stream), "rk_fdec_loss_dz")
      self._dz_pg = True          # (dO is a plane image: dW reads it -- granule 32 x 64, self._dz_fdec)
      self._dz_fdec = True
    elif fuse_dz and ip is None and self.planes is not None and self.split16 and self.ws_dw is not None and \
        self.item_parallel is None and lib.rk_decode_dz_fused_ok(B, self.h[0], tgt.n_cap, self.loss_id):
      # training steps sequenced entry by entry (hidden stacks, bottleneck dropout, MatrixFactorization):
      # the operands are split once (two launches) and the decode launch leaves the dZ partials of its
      # column tiles in self.ws (rk_decode_loss_dz_planes) -- the dZ launch, its pass over dO and the
      # in-loop operand splits go; rk_decode_dz_reduce follows where rk_decode_bwd_dz stood, dW works
      # in its own workspace in between
      h0 = self.h[0]
      rg = self._ranges(z, B * h0, stream)
      check(lib.rk_split_wz(None if w_done else ptr(W), ptr(z), B, h0, tgt.ref, rg, ctypes.byref(self.planes),
                               ptr(zt_ws), stream), "rk_split_wz")
      self._zt_ready = None if zt_ws is None else zt_ws.data_ptr()
      check(lib.rk_decode_loss_dz_planes(ctypes.byref(self.planes), B, tgt.ref, row_off, ptr(b), self.loss_id,
                                         self.confidence, inv_B, ptr(self.dO), ptr(self.loss_part),
                                         ptr(self.gb_part), ptr(self.ws), stream), "rk_decode_loss_dz_planes")
      self._dz_in_ws = True
    elif fuse_dz and pg_ok and ip is None and self._pg_entry_ok(B, tgt.n_cap):
      # outside the fused launch's domain: decode + loss, dZ and dW on the pipelined pair-plane kernels --
      # ONE split launch (W image unless the encoder forward cut it, Z image; no W^T image, no Z^T planes),
      # dLoss/dLogits as a plane image; the multinomial loss as a statistics pass + the decode / loss pass
      # (no logits matrix, no rk_mnll_finish)
      h0 = self.h[0]
      rg = self._ranges(z, B * h0, stream)
      check(lib.rk_split_wz(None if w_done else ptr(W), ptr(z), B, h0, tgt.ref, rg,
                               ctypes.byref(self.planes_nowt), None, stream), "rk_split_wz")
      if self.loss_id == LOSS_MNLL:
        check(lib.rk_pg_decode_mnll(ctypes.byref(self.planes), B, tgt.ref, row_off, ptr(b), inv_B,
                                    ptr(self.mnll_ws), ptr(self.dO), self.do_rows, ptr(self.do_scales), None,
                                    ptr(self.loss_part), ptr(self.gb_part), stream), "rk_pg_decode_mnll")
      else:
        check(lib.rk_pg_decode_loss(ctypes.byref(self.planes), B, tgt.ref, row_off, ptr(b), self.loss_id,
                                    self.confidence, inv_B, ptr(self.dO), self.do_rows, ptr(self.do_scales),
                                    None, ptr(self.loss_part), ptr(self.gb_part), stream), "rk_pg_decode_loss")
      self._dz_pg = True
    elif fuse_dz and ip is None and self.planes is not None and self.split16 and self.item_parallel is None:
      # outside the fused launch's domain (multinomial loss, h > 256, >= 1024 rows): still the plane
      # kernels -- ONE split launch, then the copy -> LDS -> MFMA decode; the dZ product follows on the
      # W^T image (rk_decode_bwd_dz_planes) where rk_decode_bwd_dz would split W_de in its k-loop again
      h0 = self.h[0]
      rg = self._ranges(z, B * h0, stream)
      check(lib.rk_split_wz(None if w_done else ptr(W), ptr(z), B, h0, tgt.ref, rg, ctypes.byref(self.planes),
                               ptr(zt_ws), stream), "rk_split_wz")
      self._zt_ready = None if zt_ws is None else zt_ws.data_ptr()
      check(lib.rk_decode_loss_planes(ctypes.byref(self.planes), B, tgt.ref, row_off, ptr(b), self.loss_id,
                                      self.confidence, inv_B, ptr(self.dO), 0, ptr(self.loss_part),
                                      ptr(self.gb_part), stream), "rk_decode_loss_planes")
      self._dz_on_planes = True
    else:
      check(lib.rk_decode_loss(ptr(z), B, self.h[0], tgt.ref, row_off, ptr(W), ptr(b), self.loss_id,
                               self.confidence, inv_B, ptr(self.dO), 0, ptr(self.loss_part),
                               ptr(self.gb_part), self._ranges(z, B * self.h[0], stream), stream),
            "rk_decode_loss")
    if self.loss_id == LOSS_MNLL and ip is not None:
      # per-row {max, sum exp} of the local logits -> all ranks' pairs -> global log-sum-exp
      stats = torch.empty(B, 2, dtype=torch.float32, device=self.device)
      check(lib.rk_mnll_row_stats(ptr(self.dO), B, tgt.ref, ptr(stats), stream), "rk_mnll_row_stats")
      st = torch.stack(ip.allgather(stats))                      # [N, B, 2]
      gmax = st[..., 0].max(dim=0).values
      glog = torch.log((st[..., 1] * torch.exp(st[..., 0] - gmax)).sum(dim=0))
      tsum = ip.user_tsum_dev[tgt.users[row_off:row_off + B]].contiguous()
      gmax, glog = gmax.contiguous(), glog.contiguous()
      check(lib.rk_mnll_finish(ptr(self.dO), B, tgt.ref, row_off, inv_B, ptr(gmax), ptr(glog),
                               ptr(tsum), ptr(self.loss_part), stream), "rk_mnll_finish")
      n_part = B
    elif self.loss_id == LOSS_MNLL and self._dz_pg:
      n_part = self.lib.rk_loss_partials(B, tgt.n_cap)     # (one partial per tile, as mse / logistic)
    elif self.loss_id == LOSS_MNLL:
      check(lib.rk_mnll_finish(ptr(self.dO), B, tgt.ref, row_off, inv_B, None, None, None, ptr(self.loss_part),
                               stream), "rk_mnll_finish")
      n_part = B
    else:
      n_part = self.lib.rk_loss_partials(B, tgt.n_cap)     # all slots (unused ones hold 0)
    if defer:        # (train_step: summed by the step's Adam launch, see _flush_jobs)
      self._pending_loss = (n_part, float(denom_rows), out)
      return out
    check(lib.rk_loss_reduce(ptr(self.loss_part), n_part, float(denom_rows), ptr(out), stream),
          "rk_loss_reduce")
    return out

  # ------------------------------------------------------------------ steps
  def compute_loss(self, blk, row_off, B, tgt=None, out=None):
    """Evaluation-mode loss (model.py:439-452 ``_validate`` body)."""
    self.ensure_capacity(B, max(blk.n_cap, tgt.n_cap if tgt is not None else 0))
    stream = current_stream()
    if self.kind == "ae":
      z = self._ae_forward(blk, row_off, B, None, None, False, stream)
    else:
      z = self._mf_forward(blk.users[row_off:row_off + B], B, None, False, stream)
    return self._loss(z, B, tgt if tgt is not None else blk, row_off, B, stream, out)

  def train_step(self, blk, row_off, B, keep_noise=None, keep_drop=None, out=None,
                 global_rows=None, tgt=None, replay=None):
    """One optimisation step on rows [row_off, row_off+B) of the collated
    block (model.py:383-404).  ``global_rows`` = rows summed over all ranks
    (data parallel); the loss/gradients are normalised by it.

    Everything is enqueued in order on the caller's stream (cross-stream events
    cost 10-20 us of dependency latency each, more than the overlap they bought),
    the RCCL all-reduces of a data-parallel step included."""
    # tgt: a separately collated TARGET block for the same rows (datasets with a target
    # matrix, model.py:464-472): decode, loss, dW and the decoder-side updates run over ITS item
    # set, the encoder side over the input block's
    tb = blk if tgt is None else tgt
    self.ensure_capacity(B, max(blk.n_cap, tb.n_cap))
    lib, m = self.lib, self.model
    main_s = torch.cuda.current_stream()
    ip = self.item_parallel
    if tgt is not None:
      # (never reached through Recoder.train: it routes these to the generic engine / replicated
      # training -- model.Recoder._pick_engine_for, _setup_data_parallel)
      if self.kind == "ae" and bool(m.is_constrained):
        raise RuntimeError("FusedEngine: tied weights with a separate target matrix belong to the generic engine")
      if ip is not None or self.allreduce is not None:
        raise RuntimeError("FusedEngine: a separate target matrix has no sharded formulation")
    if self.c_step_eligible() and tgt is None and not (ip is not None and self.loss_id == LOSS_MNLL):
      return self._c_train_step(blk, row_off, B, keep_noise, out, global_rows, main_s, replay=replay)
    if replay is not None:
      # graph replay of the per-entry sequencing (graph.GraphStepper): what changes per step comes
      # from the device-resident cursor (include/recoder_hip.h rk_replay_t); the host counters are
      # advanced by the stepper, `out` is the base of the epoch's loss buffer
      from ._lib import RkReplay
      ctx = RkReplay()
      ctx.cursor, ctx.off, ctx.B = replay["cursor"], replay["off"], B
      ctx.users_base, ctx.adam_table, ctx.tab_stride = replay["users"], replay["table"], replay["tab_stride"]
      ctx.cursor_next, ctx.advance = replay["next"] if replay.get("next") is not None else (None, 0)
      raw_lib = _lib.load()
      raw_lib.rk_replay_set(ctypes.byref(ctx))
      self._replay = replay
      try:
        return self._entry_train_step(blk, row_off, B, keep_noise, keep_drop, out, global_rows, tgt, main_s)
      finally:
        self._replay = None
        raw_lib.rk_replay_set(None)
    return self._entry_train_step(blk, row_off, B, keep_noise, keep_drop, out, global_rows, tgt, main_s)

  def _entry_train_step(self, blk, row_off, B, keep_noise, keep_drop, out, global_rows, tgt, main_s):
    """The step sequenced entry by entry from here (hidden stacks, bottleneck dropout,
    MatrixFactorization, separate target blocks, the multi-GPU variants)."""
    lib, m = self.lib, self.model
    ip = self.item_parallel
    tb = blk if tgt is None else tgt
    self._gb_lazy = None
    self._gb_en_segs = 0
    self._pg_step = False
    stream = ctypes.c_void_p(main_s.cuda_stream)
    if getattr(self, "_replay", None) is None:
      self.rng_step += 1
    h0 = self.h[0]
    rows = B if global_rows is None else global_rows
    if self.kind == "ae":
      # (the plane kernels will decode this block: its W_de[items] split can ride on the encoder forward)
      self._w_split_of = None
      self._split_w_with_fwd = (tgt is None and ip is None and self.planes is not None and self.split16 and
                                self.item_parallel is None and not bool(m.is_constrained))
      self._split_nowt = bool(self._split_w_with_fwd and self.allreduce is None and
                              (self._pg_entry_ok(B, blk.n_cap) or
                               self._fdec_entry_ok(B, blk.n_cap, row_off, tgt is None)))
      z = self._ae_forward(blk, row_off, B, keep_noise, keep_drop, True, stream)
      self._split_w_with_fwd = False
    else:
      rp = getattr(self, "_replay", None)
      # (replay: the C entry points take the step's users from the cursor; the pointer is a placeholder)
      users = blk.users[row_off:row_off + B] if rp is None else rp["users_t"]
      z = self._mf_forward(users, B, keep_drop, True, stream)
    # single process: nothing has to exist as an array of its own for an exchange, so the small
    # reductions of the step ride on its Adam launch as they do in rk_ae_train_step -- the loss
    # partials, the decode epilogue's row-tile column sums (decoder bias gradient) and the K slabs
    # of the bf16-pipe dW (three launches less)
    tied = self.kind == "ae" and bool(m.is_constrained)
    lazy = ip is None and self.allreduce is None
    # (dW will work in ws_dw with its K slabs kept for the Adam sweep: see keep_slabs below)
    zt_ws = self.ws_dw if (lazy and not tied and self.split16 and self.ws_dw is not None and
                           True) else None
    # data parallel + graph replay: the rank's share of the loss goes to a scalar of its own, travels with
    # the gradients, and the step's Adam launch files the sum under the step's slot of `out` (the epoch's
    # loss buffer) and publishes the next cursor -- as the one-call step does (_c_train_step)
    dp_replay = self.allreduce is not None and getattr(self, "_replay", None) is not None and ip is None
    loss = self._loss(z, B, tb, row_off, rows, stream, self.loss_dp if dp_replay else out, ip=ip, defer=lazy,
                      fuse_dz=True, zt_ws=zt_ws,
                      pg_ok=lazy and not tied and self.ws_dw is not None,
                      # (users-DP too: dW then leaves ONE dense array for the exchange -- rk_pg_dw_dz_reduce dense)
                      fdec_ok=ip is None and not tied and self.ws_dw is not None, own_block=tb is blk)
    self._loss_target = loss

    # ---- dW = dO^T . z  (+ decoder bias gradient) ----
    keep_slabs = lazy and not tied and self.split16 and self.ws_dw is not None
    # replayed steps: dW (it needs dO and z only) as a branch on the stepper's side stream next to
    # dZ -> hidden stacks -> encoder backward, joined in front of the Adam sweeps -- as the one-call
    # step does (rk_ae_step_t.dw_stream); its K slabs live in a workspace of their own
    rp = getattr(self, "_replay", None)
    # (off by default: with a dozen more launches on the chain the two cross-queue edges cost more
    # than the overlap buys -- C3 0.307 vs 0.300 ms, C4 0.150 vs 0.147 ms per step; RK_DW_BRANCH_ENTRY=1)
    dw_side = rp.get("dw_stream") if (rp is not None and keep_slabs and self.dw_branch and
                                      False) else None
    dw_stream = stream
    if dw_side is not None:
      if getattr(self, "_dw_ev", None) is None:
        self._dw_ev = (torch.cuda.Event(), torch.cuda.Event())
      self._dw_ev[0].record(main_s)
      dw_side.wait_event(self._dw_ev[0])
      dw_stream = ctypes.c_void_p(dw_side.cuda_stream)
    # dW (it needs dO and z only) can wait for the encoder backward and share its launch
    # (rk_decode_bwd_dw2_encode_bwd: one launch less on the chain) when both work on ONE block, the K
    # slabs stay in the dW workspace for the Adam sweep and no side stream is in play
    defer_dw = (keep_slabs and dw_side is None and self.kind == "ae" and tb is blk and
                True and
                bool(lib.rk_dw_encode_bwd_fused_ok(row_off, B)))
    self._dw_deferred = None
    self._dw_colsum = False
    if defer_dw:
      if self.loss_id == LOSS_MNLL and not self._dz_pg:
        # (dO comes from rk_mnll_finish: its column sums -- the decoder bias gradient -- are taken by
        # extra workgroups of the dW || encoder-backward launch)
        self._dw_colsum = True
      elif getattr(self, "_dz_fdec", False):
        pass                     # (the fused decode has no column sums to give: the deferred launch takes them from the image)
      elif lazy:
        self._gb_lazy = (cdiv(B, self.row_tile), tb)
      else:
        check(lib.rk_colsum(ptr(self.gb_part), cdiv(B, self.row_tile), tb.n_cap, 0, ptr(tb.counts),
                            ptr(self.gb_de), stream), "rk_colsum")
      self._dw_deferred = z
    elif self.loss_id == LOSS_MNLL and not self._dz_pg:
      # dO was produced by rk_mnll_finish: column sums need a pass over dO
      self._dw(z, B, tb, self.gb_de, dw_stream, keep_slabs)
    elif getattr(self, "_dz_fdec", False):
      # the fused decode leaves no column sums: the dW launch takes the bias gradient from the image's columns
      # -- and, MatrixFactorization (nothing before the Adam sweep reads dZ), the slab reduce as a third range
      zact = None if self.drop_active else self.enc[0]
      check(lib.rk_pg_dw_dz_reduce(ptr(self.dO), ptr(self.do_scales), 32, 64, B, ctypes.byref(self.planes), tb.ref,
                                   ptr(self.ws_dw if keep_slabs else self.G_de), ptr(self.gb_de), ptr(self.ws),
                                   ptr(zact), self.act, ptr(self.dbott), 0 if keep_slabs else 1, stream),
            "rk_pg_dw_dz_reduce")
      # (single process: the K slabs stay in ws_dw for the Adam sweep to add up; users-DP: one dense G_de)
      self._dw_slabs = (tb, B) if keep_slabs else None
      self._ws_dw_live = bool(keep_slabs)
      self._pg_step = bool(keep_slabs)
      self._dz_done = True
    else:
      # the loss epilogue already reduced dO per row tile: sum those few rows
      if lazy:
        self._gb_lazy = (cdiv(B, self.row_tile), tb)
      else:
        check(lib.rk_colsum(ptr(self.gb_part), cdiv(B, self.row_tile), tb.n_cap, 0, ptr(tb.counts),
                            ptr(self.gb_de), stream), "rk_colsum")
      # MatrixFactorization: nothing between the decode and the Adam sweep reads dZ (the user rows'
      # gradient), so the reduce of the decode launch's dZ partials rides on the dW launch
      red = None
      if (self.kind != "ae" and getattr(self, "_dz_in_ws", False) and keep_slabs and dw_side is None and
          ip is None and self.lib.rk_dw_pairs()):
        red = (self.ws, None if self.drop_active else self.enc[0], self.dbott)
      self._dw(z, B, tb, None, dw_stream, keep_slabs, red=red)
      if red is not None:
        self._dz_in_ws = False
        self._dz_done = True
    if dw_side is not None:
      self._dw_ev[1].record(dw_side)
    # (replayed: a captured collective has a fixed size -- the exchange covers the block's capacity, rows past
    # n_b are never read by the update -- and nothing of the step is read on the host)
    n_b_host = None if self.allreduce is None else (blk.n_cap if dp_replay else self.allreduce.n_b(blk))

    # ---- dZ = dO . W_de[T] and everything upstream of it ----
    W_de, _ = self._decoder_params()
    simple = (self.kind == "ae" and self.nl == 0 and not self.drop_active)
    # gradient w.r.t. the decoder's input: straight into the encoder side's buffer unless the
    # bottleneck dropout has to be undone on the way (MF: dbott is the gathered user rows' gradient)
    bott_grad = self.denc[self.nl] if (self.kind == "ae" and not self.drop_active) else self.dbott
    dz = bott_grad
    if self.kind == "ae" and self.nl > 0:
      dz = self.ddec[self.nl - 1]
    # act' folded into the split-K reduce (MF without dropout: the gathered rows ARE the decoder's input)
    fuse_act = (simple or (self.kind != "ae" and not self.drop_active)) and ip is None
    # hidden stacks without bottleneck dropout: every act' of the backward is multiplied in by the
    # PRODUCER of the gradient it applies to -- the dZ reduce for the decoder's last hidden output, each
    # layer's dX epilogue for its input activation -- so a layer's backward is ONE launch (dX, dW and
    # the bias gradient's column sums: rk_linear_bwd_pre) instead of an act' pass + the products
    stack_pre = (self.kind == "ae" and self.nl > 0 and not self.drop_active and ip is None and
                 True)
    zact = self.enc[0] if fuse_act else (self.dec[self.nl - 1] if stack_pre else None)
    if getattr(self, "_dz_done", False):
      self._dz_done = False            # (summed by the dW launch: rk_decode_bwd_dw2_dz_reduce)
    elif getattr(self, "_dz_in_ws", False):
      check(lib.rk_decode_dz_reduce(ptr(self.ws), B, h0, tb.ref, ptr(zact),
                                    self.act, ptr(dz), stream), "rk_decode_dz_reduce")
      self._dz_in_ws = False
    elif getattr(self, "_dz_fdec", False):
      check(lib.rk_fdec_dz_reduce(ptr(self.ws), B, h0, tb.ref, ptr(zact), self.act, ptr(dz), stream),
            "rk_fdec_dz_reduce")
    elif getattr(self, "_dz_pg", False):
      check(lib.rk_pg_dz(ptr(self.dO), ptr(self.do_scales), 64, 32, B, ctypes.byref(self.planes), tb.ref,
                         ptr(zact), self.act, ptr(dz), ptr(self.ws), stream), "rk_pg_dz")
    elif getattr(self, "_dz_on_planes", False):
      check(lib.rk_decode_bwd_dz_planes(ptr(self.dO), B, ctypes.byref(self.planes), tb.ref,
                                        ptr(zact), self.act, ptr(dz),
                                        ptr(self.ws), stream), "rk_decode_bwd_dz_planes")
      self._dz_on_planes = False
    else:
      check(lib.rk_decode_bwd_dz(ptr(self.dO), B, h0, tb.ref, ptr(W_de),
                                 ptr(zact), self.act, ptr(dz),
                                 ptr(self.ws), ptr(self.ranges), stream), "rk_decode_bwd_dz")
    if ip is not None:
      # item parallel: dLoss/d(decoder input) summed over the ranks' item shards; everything
      # upstream (hidden stacks, user rows) is replicated and sees identical inputs
      ip.allreduce_sum(dz[:B * h0])

    if self.kind == "ae":
      rh = list(reversed(self.h))
      # decoder Linear stack, last to first
      for i in range(self.nl - 1, -1, -1):
        layer = m.decoding_layers[i]
        x = self.dec[i - 1] if i > 0 else self.dec_in
        dx = self.ddec[i - 1] if i > 0 else bott_grad
        if m.is_constrained:
          j = self.nl - 1 - i
          w, wt, gw, acc = m.encoding_layers[j].weight, 1, self.g_enc_w[j], 0
        else:
          w, wt, gw, acc = layer.weight, 0, self.g_dec_w[i], 0
        if stack_pre:
          # (ddec[i] arrives with act'(dec[i]) in it; dX leaves with act'(x): x is the layer's input)
          check(lib.rk_linear_bwd_pre(ptr(self.ddec[i]), ptr(x), ptr(w), B, rh[i + 1], rh[i], wt, self.act,
                                      ptr(dx), ptr(gw), acc, ptr(self.g_dec_b[i]), ptr(x), stream),
                "rk_linear_bwd_pre")
          continue
        check(lib.rk_linear_bwd(ptr(self.ddec[i]), ptr(self.dec[i]), ptr(x), ptr(w), B, rh[i + 1],
                                rh[i], wt, self.act, ptr(dx), ptr(gw), acc, ptr(self.g_dec_b[i]),
                                stream), "rk_linear_bwd")
      if self.drop_active:
        # gradient w.r.t. the bottleneck activation: undo dropout, into denc[nl]
        n = B * self.h[-1]
        check(lib.rk_dropout(ptr(self.dbott), ptr(keep_drop), n, self.h[-1],
                             float(m.dropout_prob), self.seed ^ 0xd0d0, self.rng_step, stream),
              "rk_dropout")
        self.denc[self.nl][:n].copy_(self.dbott[:n])
      # encoder Linear stack, last to first
      for i in range(self.nl - 1, -1, -1):
        layer = m.encoding_layers[i]
        # (the stack's first layer: its dX is the embedding layer's gradient -- act'(enc[0]) is
        # folded into that product's epilogue instead of an rk_act_grad launch behind it)
        if stack_pre:
          check(lib.rk_linear_bwd_pre(ptr(self.denc[i + 1]), ptr(self.enc[i]), ptr(layer.weight), B,
                                      self.h[i + 1], self.h[i], 0, self.act, ptr(self.denc[i]),
                                      ptr(self.g_enc_w[i]), 1 if m.is_constrained else 0,
                                      ptr(self.g_enc_b[i]), ptr(self.enc[i]), stream), "rk_linear_bwd_pre")
          if i == 0:
            fuse_act = True          # (act'(enc[0]) went into that dX)
          continue
        last = i == 0 and not fuse_act
        check(lib.rk_linear_bwd_dact(ptr(self.denc[i + 1]), ptr(self.enc[i + 1]), ptr(self.enc[i]),
                                     ptr(layer.weight), B, self.h[i + 1], self.h[i], 0, self.act,
                                     ptr(self.denc[i]), ptr(self.g_enc_w[i]),
                                     1 if m.is_constrained else 0, ptr(self.g_enc_b[i]),
                                     ptr(self.enc[0]) if last else None, stream),
              "rk_linear_bwd")
        if last:
          fuse_act = True
      if not fuse_act:
        check(lib.rk_act_grad(ptr(self.denc[0]), ptr(self.enc[0]), B * h0, self.act, stream),
              "rk_act_grad")
      G_en = self.G_de if tied else self.G_en      # tied: accumulates on top of dW's rows
      if getattr(self, "_dw_deferred", None) is not None:
        zz, self._dw_deferred = self._dw_deferred, None
        zt = ptr(self.ws_dw) if getattr(self, "_zt_ready", None) == self.ws_dw.data_ptr() else None
        if getattr(self, "_dz_fdec", False):
          # (the fused decode's image: granule 32 x 64; its columns' sums = the decoder bias gradient ride along)
          check(lib.rk_pg_dw_encode_bwd(ptr(self.dO), ptr(self.do_scales), 32, 64, B, ctypes.byref(self.planes),
                                        blk.ref, ptr(self.ws_dw), row_off, ptr(self.denc[0]), ptr(G_en),
                                        ptr(self.gb_en), ptr(self.gb_de), stream), "rk_pg_dw_encode_bwd")
          self._pg_step = True
        elif getattr(self, "_dz_pg", False):
          check(lib.rk_pg_dw_encode_bwd(ptr(self.dO), ptr(self.do_scales), 64, 32, B, ctypes.byref(self.planes),
                                        blk.ref, ptr(self.ws_dw), row_off, ptr(self.denc[0]), ptr(G_en),
                                        ptr(self.gb_en), None, stream), "rk_pg_dw_encode_bwd")
          self._pg_step = True
        elif self._dw_colsum:
          check(lib.rk_decode_bwd_dw2_encode_bwd_colsum(ptr(self.dO), ptr(zz), B, h0, blk.ref, ptr(self.ws_dw),
                                                        zt, ptr(self.ranges), row_off, ptr(self.denc[0]),
                                                        ptr(G_en), ptr(self.gb_en), ptr(self.gb_de), stream),
                "rk_decode_bwd_dw2_encode_bwd_colsum")
        else:
          check(lib.rk_decode_bwd_dw2_encode_bwd(ptr(self.dO), ptr(zz), B, h0, blk.ref, ptr(self.ws_dw), zt,
                                                 ptr(self.ranges), row_off, ptr(self.denc[0]), ptr(G_en),
                                                 ptr(self.gb_en), stream), "rk_decode_bwd_dw2_encode_bwd")
        self._dw_slabs = (blk, B)
        self._ws_dw_live = True
      else:
        check(lib.rk_ae_encode_bwd(blk.ref, row_off, B, ptr(self.denc[0]), h0, ptr(G_en),
                                   1 if tied else 0, ptr(self.gb_en), stream), "rk_ae_encode_bwd")
    else:
      # MF: gradient of the gathered user rows = dU * act'(U) (after dropout)
      n = B * h0
      if self.drop_active:
        check(lib.rk_dropout(ptr(self.dbott), ptr(keep_drop), n, h0, float(m.dropout_prob),
                             self.seed ^ 0xd0d0, self.rng_step, stream), "rk_dropout")
      if not fuse_act:
        check(lib.rk_act_grad(ptr(self.dbott), ptr(self.enc[0]), n, self.act, stream), "rk_act_grad")

    if self.allreduce is not None:
      # data parallel over users: every gradient of the step (live rows of both tables, gathered
      # bias, dense layers, loss) is SUM all-reduced as one in-order RCCL group on this stream
      if dp_replay:
        tails = [(self.G_de, h0), (self.gb_de, 1)] + ([(self.G_en, h0)] if self.kind == "ae" and not tied else [])
        self._zero_grad_tails(tb, tails, stream)
      if getattr(self, "owned_rows", False):
        self._owned_exchange(blk, n_b_host)
      else:
        self.allreduce.reduce(self.grad_views(n_b_host, "all"))
      if dp_replay:
        self._pending_loss = (1, 1.0, out, self.loss_dp)
        loss = out
    if dw_side is not None:
      main_s.wait_event(self._dw_ev[1])
    self._apply_updates(blk, row_off, B, stream, "all", tgt=tb)
    if getattr(self, "_own", None) is not None:
      self._owned_publish(blk)
    return loss

  # ---- owned-row Adam under users-DP (parallel.DataParallel, "owned-row Adam") ----
  def _sparse_tables(self):
    """(state name, compact gradient rows) of every SparseAdam embedding table of the step."""
    S = self.states
    if self.kind == "ae":
      if self.model.is_constrained:
        cand = [("en_embedding_layer.weight", self.G_de)]
      else:
        cand = [("en_embedding_layer.weight", self.G_en), ("de_embedding_layer.weight", self.G_de)]
    else:
      cand = [("item_embedding_layer.weight", self.G_de)]
    return [(n, g) for n, g in cand if S[n].sparse]

  def _owned_exchange(self, blk, n_b):
    """The SparseAdam tables' partial gradient rows go to the ranks that OWN them (variable-count
    all-to-all); everything else is all-reduced as ever."""
    dp, h0 = self.allreduce, self.h[0]
    offs = dp.owned_offsets(blk.items, n_b)
    tabs = {}
    own_g = set()
    for name, G in self._sparse_tables():
      R, cnt = dp.exchange_rows(G[:n_b * h0], offs, h0)
      tabs[name] = R
      own_g.add(G.data_ptr())
    lo = offs[dp.rank]
    self._own = dict(offs=offs, tabs=tabs, lo=lo, cnt=offs[dp.rank + 1] - lo, n_b=n_b,
                     cnt_dev=torch.tensor([offs[dp.rank + 1] - lo], dtype=torch.int32, device=self.device))
    rest = [v for v in self.grad_views(n_b, "all") if not (v.data_ptr() in own_g and v.numel() == n_b * h0)]
    dp.reduce(rest)

  def _owned_publish(self, blk):
    """Every rank's freshly updated rows of the SparseAdam tables -> every replica (the all-gather)."""
    own, self._own = self._own, None
    dp, h0 = self.allreduce, self.h[0]
    n_b, lo, cnt = own["n_b"], own["lo"], own["cnt"]
    items = blk.items[:n_b].long()
    for name in own["tabs"]:
      p = self.states[name].p.data
      S = p.index_select(0, items[lo:lo + cnt]) if cnt else p.new_zeros((0, h0))
      T = dp.publish_rows(S, own["offs"], h0)
      p.index_copy_(0, items, T.view(n_b, h0))
    # the decoder table's running |W| bound (split scales of the contractions): a rank only sees the
    # rows it wrote -- MAX over the ranks of the 64 slots (non-negative fp32 bit patterns order as ints)
    dp.union_marks(self.ranges[64:])

  def _dw(self, z, B, blk, gb_de, stream, keep_slabs=False, red=None):
    """G_de = dO^T . z (+ gb_de = colsum(dO) if asked): the bf16-pipe kernel (csrc/dw3.hip) unless
    RK_GEMM_PREC=f32 keeps the contractions on the fp32 MFMA.  keep_slabs: leave the K slabs
    unsummed in the dW workspace of its own (the dZ product that follows reuses `ws`) for the Adam
    sweep to add up."""
    h0 = self.h[0]
    self._dw_slabs = None
    self._ws_dw_live = False
    if self.split16:
      # fp16 pairs (rk_decode_bwd_dw2: three products, the scale of z from self.ranges -- the bound
      # rk_amax left there for this z when the activation is unbounded) unless RK_DW_PREC=bf16x3
      G, ws = (None, self.ws_dw) if keep_slabs else (self.G_de, self.ws)
      if getattr(self, "_dz_in_ws", False):
        ws = self.ws_dw                  # (self.ws holds the decode launch's dZ partials until the reduce)
      if getattr(self, "_dz_pg", False):
        assert keep_slabs and gb_de is None and red is None
        check(self.lib.rk_pg_dw(ptr(self.dO), ptr(self.do_scales), 64, 32, B, ctypes.byref(self.planes), blk.ref,
                                ptr(ws), None, stream), "rk_pg_dw")
        self._pg_step = True
      elif self.lib.rk_dw_pairs():
        # (Z^T pair planes already at the head of this workspace: rk_split_wz of this step's decode)
        zt = ptr(ws) if getattr(self, "_zt_ready", None) == ws.data_ptr() else None
        if red is not None:
          # red = (the decode launch's dZ partials, Zact or None, dZ): summed by extra workgroups here
          assert G is None and gb_de is None
          check(self.lib.rk_decode_bwd_dw2_dz_reduce(ptr(self.dO), ptr(z), B, h0, blk.ref, ptr(ws), zt,
                                                     ptr(self.ranges), ptr(red[0]), ptr(red[1]), self.act,
                                                     ptr(red[2]), stream), "rk_decode_bwd_dw2_dz_reduce")
        else:
          check(self.lib.rk_decode_bwd_dw2(ptr(self.dO), ptr(z), B, h0, blk.ref, ptr(G), ptr(gb_de), ptr(ws),
                                           zt, ptr(self.ranges), stream), "rk_decode_bwd_dw2")
      else:
        check(self.lib.rk_decode_bwd_dw3(ptr(self.dO), ptr(z), B, h0, blk.ref, ptr(G), ptr(gb_de), ptr(ws),
                                         None, stream), "rk_decode_bwd_dw3")
      if keep_slabs:
        self._dw_slabs = (blk, B)
        self._ws_dw_live = True
    else:
      check(self.lib.rk_decode_bwd_dw(ptr(self.dO), ptr(z), B, h0, blk.ref, ptr(self.G_de), ptr(gb_de),
                                      stream), "rk_decode_bwd_dw")

  def c_step_eligible(self):
    """The one-call step (rk_ae_train_step) covers DynamicAutoencoder([h]) without bottleneck
    dropout; everything else runs the per-entry sequencing."""
    m = self.model
    if getattr(self, "owned_rows", False):
      return False          # (owned-row Adam under users-DP: sequenced from Python, _owned_exchange)
    return self.use_c_step and self.kind == "ae" and self.nl == 0 and not (m.dropout_prob > 0.0)

  def _c_train_step(self, blk, row_off, B, keep_noise, out, global_rows, main_s, replay=None):
    """The same step through rk_ae_train_step: one FFI call, the kernels
    sequenced in C on the caller's stream.

    replay (graph.GraphStepper): dict(st=RkAeStep of its own, cursor=device int64[2], off=position
    in the replayed group, table=Adam constants table) -- what changes per step is then derived on
    the device (rk_ae_step_t.cursor); `out` is the BASE of the epoch's loss buffer and the host
    counters (Adam steps, rng step) are advanced by the caller."""
    from ._lib import (ENTRY, PAR_B_DE, PAR_B_EN, PAR_W_DE, PAR_W_EN, STEP_ALL, STEP_DZ_ENC,
                       STEP_FWD_DW, STEP_UPDATE, RkAeStep)
    raw = _lib.load()
    m, S = self.model, self.states
    st = self._cstep if replay is None else replay["st"]
    if st is None:
      st = RkAeStep()
      self._cstep = st
    if replay is None:
      self.rng_step += 1
    self._gb_lazy = None
    rows = B if global_rows is None else global_rows
    st.blk = ctypes.pointer(blk.c)
    st.row_off, st.B, st.h, st.act = row_off, B, self.h[0], self.act
    st.loss_kind, st.tied = self.loss_id, 1 if m.is_constrained else 0
    st.confidence = self.confidence
    st.inv_B = _f32(np.float32(1.0) / np.float32(rows))
    st.denom = float(rows)
    st.noise_p = float(m.noise_prob)
    st.seed, st.rng_step = self.seed, self.rng_step
    st.keep, st.users = ptr(keep_noise), ptr(blk.users)
    names = {PAR_W_EN: "en_embedding_layer.weight",
             PAR_B_EN: "_DynamicAutoencoder__en_linear_embedding_layer.bias",
             PAR_W_DE: "de_embedding_layer.weight",
             PAR_B_DE: "_DynamicAutoencoder__de_linear_embedding_layer.bias"}
    for k, name in names.items():
      if k == PAR_W_DE and m.is_constrained:
        continue
      s = S[name]
      if replay is None:
        s.step += 1
      lr, b1, b2, eps = self._adam_args(s)
      a = st.par[k]
      a.p, a.m, a.v = ptr(s.p), ptr(s.m), ptr(s.v)
      a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, b1, b2, eps, float(s.wd)
      a.step, a.sparse = max(1, s.step), 1 if s.sparse else 0
    out = self.loss_out if out is None else out
    dp = self.allreduce
    loss_dst = self.loss_dp if dp is not None else out
    st.Z0, st.dZ0, st.dO = ptr(self.enc[0]), ptr(self.denc[0]), ptr(self.dO)
    st.G_de, st.G_en, st.gb_de = ptr(self.G_de), ptr(self.G_en), ptr(self.gb_de)
    st.gb_part, st.ws = ptr(self.gb_part), ptr(self.ws)
    st.zt_planes = ptr(self.zt_planes)
    st.planes = (ctypes.addressof(self.planes)
                 if self.planes is not None and self.item_parallel is None else None)
    st.do_scales, st.do_rows = ptr(self.do_scales), self.do_rows
    self._check_weight_range()
    st.ranges = ptr(self.ranges)
    plain = dp is None and not m.is_constrained and self.loss_id != LOSS_MNLL
    # the fused fp32 dW || encoder-backward launch writes row-segment partials; the bf16-pipe dW
    # (whole single-GPU steps) leaves its K slabs in the workspace and runs the plain encoder backward
    dw3 = plain and self.split16 and self.item_parallel is None
    segmented = plain and not dw3
    self._dw_slabs = (blk, B) if dw3 else None
    st.gb_en = ptr(self.gb_en_parts if segmented else self.gb_en)
    self._gb_en_segs = self.lib.rk_encode_bwd_segments(B) if segmented else 0
    self.n_cap_last = blk.n_cap
    st.loss_part, st.loss_out = ptr(self.loss_part), ptr(loss_dst)
    st.stream = main_s.cuda_stream
    st.cursor, st.cursor_off, st.adam_table, st.cursor_next, st.cursor_advance = None, 0, None, None, 0
    st.ws_dw = st.dw_stream = st.dw_fork = st.dw_join = None
    st.zero_lo = st.zero_hi = 0
    st.zero_g_en = st.zero_g_de = st.zero_gb_de = None
    st.lazy_stamp_en = st.lazy_stamp_de = st.lazy_pos_next = st.lazy_need_list = st.lazy_need_count = None
    st.lazy_period = 0
    lz = replay.get("lazy") if replay is not None else None
    if lz is not None and self.item_parallel is None:
      # (the stepper knows the next step's block: rows without a gradient that it does not read are caught up later)
      st.lazy_stamp_en = ptr(self.lazy_stamp("en_embedding_layer.weight"))
      if not m.is_constrained:
        st.lazy_stamp_de = ptr(self.lazy_stamp("de_embedding_layer.weight"))
      st.lazy_pos_next, st.lazy_period = lz["pos_next"], self.lazy_period
      st.lazy_need_list, st.lazy_need_count = lz.get("need") or (None, None)
    if dp is not None and self.ws_dw is not None and not m.is_constrained:
      st.ws_dw = ptr(self.ws_dw)        # (phased steps: dW's own workspace lets the decode launch keep its dZ slabs)
    self._ws_dw_live = False
    if dw3 and self.ws_dw is not None:
      if self._dw_objs is None:
        raw0 = _lib.load()
        self._dw_objs = (torch.cuda.Stream(device=self.device), raw0.rk_event_create(0), raw0.rk_event_create(0))
      # (the graph stepper lends its side stream: dW and its collation share one branch)
      dws = replay.get("dw_stream") if replay is not None else None
      st.ws_dw, st.dw_stream = ptr(self.ws_dw), (dws or self._dw_objs[0]).cuda_stream
      st.dw_fork, st.dw_join = self._dw_objs[1], self._dw_objs[2]
      self._ws_dw_live = True
    if replay is not None:
      st.cursor, st.cursor_off, st.adam_table = replay["cursor"], replay["off"], replay["table"]
      if replay.get("next") is not None:         # last step of a group: publish the next cursor
        st.cursor_next, st.cursor_advance = replay["next"]
      st.users = replay["users"]             # base of the epoch's user order (offset on the device)
    self._c_calls += 1
    self._pg_step = False
    name = None
    if self.time_plan is not None and (replay is None or replay.get("timed")):
      name = self.time_plan(replay["index"] if replay is not None else self._c_calls)
      if name == "eager":                  # (an eagerly enqueued step of a bracketed group: no events)
        name = None
    st.time_all = None
    if name == "all":
      # every launch group of this step gets its own pair of timing events
      from ._lib import ENTRY_ALL
      n_ent = max(ENTRY.values()) + 1
      evs = (ctypes.c_void_p * (2 * n_ent))()
      for ename, eid in ENTRY.items():
        evs[2 * eid], evs[2 * eid + 1] = self._new_timing_event(raw), self._new_timing_event(raw)
        self._time_samples.append((ename, evs[2 * eid], evs[2 * eid + 1]))
      self._time_keep.append(evs)          # (the array must outlive the call)
      st.time_entry, st.time_all = ENTRY_ALL, evs
    elif name is not None:
      e0, e1 = self._new_timing_event(raw), self._new_timing_event(raw)
      self._time_samples.append((name, e0, e1))
      st.time_entry, st.time_ev0, st.time_ev1 = ENTRY[name], e0, e1
    else:
      st.time_entry = 0
    ip = self.item_parallel
    if ip is not None:
      # item parallel (parallel.ItemParallel): the block holds every user of the global
      # batch restricted to this rank's items; two [B, h] all-reduces per step.  `out` gets
      # this rank's share of the loss (summed over the ranks once per epoch).
      from ._lib import STEP_IP_ENC, STEP_IP_MID, STEP_IP_TAIL
      h0 = self.h[0]
      st.user_norm = ptr(ip.user_norm_dev)
      st.own_rank, st.own_world = ip.rank, ip.world
      st.phase = STEP_IP_ENC
      check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
      ip.allreduce_sum(self.enc[0][:B * h0])
      st.phase = STEP_IP_MID
      check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
      ip.allreduce_sum(self.denc[0][:B * h0])
      st.phase = STEP_IP_TAIL
      check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
      if self.loss_id != LOSS_MNLL:
        self._gb_lazy = (cdiv(B, self.row_tile), blk)
    elif dp is None:
      st.phase = STEP_ALL
      flags = int(raw.rk_ae_step_uses_pg(ctypes.byref(st)))
      mode = flags & 15
      self._step_flags = flags     # (bit 4: the bias gradient as output column h of the dW tiles -- recoder_hip.h)
      self._pg_step = bool(mode)
      self._step_mode = mode       # (0: decode16 / dw3 kernels, 1: csrc/pgemm.h, 3: csrc/fdecode.hip + pgemm's dW; bench.py names the kernels by it)
      check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
      if self.loss_id != LOSS_MNLL and mode != 3:
        self._gb_lazy = (cdiv(B, self.row_tile), blk)    # (mode 3: gb_de itself, from the dO image)
      elif flags & 16:
        self._gb_lazy = ("slabs", blk)                   # (... or one slab per K slab of dW in gb_part)
    else:
      # data parallel over users: forward + whole backward locally, then the live gradient rows
      # of both tables, the gathered-bias gradient, the encoder bias gradient and the loss go out
      # as ONE in-order RCCL group on this stream, then the identical Adam on every replica
      h0 = self.h[0]
      tied = bool(m.is_constrained)
      # replayed (graph.GraphStepper): a captured collective has a fixed size -- the exchange covers
      # the blocks' whole capacity (rows past n_b are never read by the update) instead of the live
      # rows, and nothing of the step is read on the host
      n_b = dp.n_b(blk) if replay is None else blk.n_cap
      if replay is not None:
        self._zero_grad_tails(blk, [(self.G_de, self.h[0]), (self.G_en, self.h[0]), (self.gb_de, 1)],
                              ctypes.c_void_p(main_s.cuda_stream))
      # (the gradient rows that travel: the live count rounded up to a granule the reduce-scatter can
      # shard -- rows past n_b are zeros or stale rows nobody reads)
      n_x = dp.round_rows(n_b, blk.n_cap) if hasattr(dp, "round_rows") else n_b
      G_enc = self.G_de if tied else self.G_en
      zero = getattr(dp, "zero", None) if getattr(self, "zero_adam", False) else None
      local = bool(getattr(dp, "local_sets", False))
      if zero is not None or local:
        # The gradients travel laid out by ITEM ID (rk_rows_to_dense) and the update reads them row by row:
        #  * sharded dense Adam (parallel.DataParallel "ZeRO-1"): the dense layout is reduce-scattered -- this
        #    rank's row range comes back --, the update covers that range only (rk_ae_step_t.zero_lo), the
        #    updated rows are all-gathered;
        #  * per-rank item sets (dp.local_sets): the ranks' compact columns differ, so the dense layout is the
        #    only common one -- all-reduced in place (or reduce-scattered, with the sharded update on top); the
        #    gathered decoder-bias gradient travels as a dense vector too (rk_ae_step_t.zero_gb_de).
        n_items = blk.n_items
        if zero is not None:
          lo, hi, sh, rp_ = zero["lo"], zero["hi"], zero["sh"], zero["rows_pad"]
          D, S_en, S_de = self._zero_buffers(rp_, sh, h0)
          D_en = D_de = D
          sh_en, sh_de = S_en, S_de
        else:
          lo, hi, rp_ = 0, n_items, n_items
          D_en, D_de = self._dense_grad_buffers(n_items, h0)
          S_en, S_de = D_en, D_de
          sh_en = sh_de = None               # (all-reduced in place)
        gb_dense = self._gb_dense_buffer(n_items) if local else None

        def stage_of(G, D_, with_gb=False):
          def go(s_):
            hs = ctypes.c_void_p(s_.cuda_stream)
            check(raw.rk_rows_to_dense(ptr(G), ptr(blk.pos), n_items, rp_, h0, ptr(D_), hs), "rk_rows_to_dense")
            if with_gb and gb_dense is not None:
              check(raw.rk_rows_to_dense(ptr(self.gb_de), ptr(blk.pos), n_items, n_items, 1, ptr(gb_dense), hs),
                    "rk_rows_to_dense")
          return go
        loss_v, gb_en_v = self.small[h0:h0 + 1], self.small[:h0]
        dec_small = [loss_v, gb_dense] if local else [self.small[h0:self.small_off + n_b]]
        if tied:
          st.phase = STEP_FWD_DW | STEP_DZ_ENC
          check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
          dp.zero_exchange(stage_of(G_enc, D_en, True), D_en, sh_en, [gb_en_v] + dec_small, main_s, overlap=False)
        else:
          st.phase = STEP_FWD_DW
          check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
          dp.zero_exchange(stage_of(self.G_de, D_de, True), D_de, sh_de, dec_small, main_s)
          st.phase = STEP_DZ_ENC
          check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
          dp.zero_exchange(stage_of(G_enc, D_en), D_en, sh_en, [gb_en_v], main_s)
          dp.join_async(main_s)
        st.zero_lo, st.zero_hi = lo, hi
        st.zero_g_en, st.zero_g_de = ptr(S_en), (None if tied else ptr(S_de))
        st.zero_gb_de = ptr(gb_dense) if local else None
      elif tied:
        # tied weights: the encoder backward accumulates onto dW's rows -- nothing may leave before it
        st.phase = STEP_FWD_DW | STEP_DZ_ENC
        check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
        dp.reduce([G_enc[:n_x * h0], self.small[:self.small_off + n_b]])
      else:
        # the decoder-side gradients (dW rows, loss, gathered-bias gradient) travel on the
        # communication stream while this stream runs dZ -> encoder backward; the encoder side
        # (gradient rows, encoder bias) follows behind them; the Adam sweep waits for both
        st.phase = STEP_FWD_DW
        check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
        dp.reduce_async([self.G_de[:n_x * h0], self.small[h0:self.small_off + n_b]], main_s)
        st.phase = STEP_DZ_ENC
        check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
        dp.reduce_async([G_enc[:n_x * h0], self.small[:h0]], main_s)
        dp.join_async(main_s)
      if replay is None:
        out.copy_(self.loss_dp)
      else:
        # the Adam launch files the exchanged loss under the step's slot of the epoch's buffer
        st.loss_part, st.loss_out = ptr(self.loss_dp), ptr(out)
      st.phase = STEP_UPDATE
      check(raw.rk_ae_train_step(ctypes.byref(st)), "rk_ae_train_step")
      if zero is not None:
        names_ = ["en_embedding_layer.weight"] + ([] if tied else ["de_embedding_layer.weight"])
        dp.zero_publish([S[n_].p.data for n_ in names_], h0, extra_max=self.ranges[64:])
    self._loss_target = out
    return out

  def _zero_grad_tails(self, blk, arrays, stream):
    """Rows [n_b, n_cap) of the compact gradient arrays <- 0 (rk_zero_tail_rows): in front of a replayed
    data-parallel exchange over the block's capacity, whose in-place sum would otherwise multiply whatever those
    rows hold by the world size step after step.  arrays: [(tensor, width)]."""
    n = len(arrays)
    X = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in arrays])
    H = (ctypes.c_int32 * n)(*[int(w) for _, w in arrays])
    # (the arrays are allocated zeroed: ensure_capacity resets the high-water mark with them)
    check(self.lib.rk_zero_tail_rows(X, H, n, ptr(blk.counts), min(blk.n_cap, self.n_cap), ptr(self._grad_hwm), stream),
          "rk_zero_tail_rows")

  def _dense_grad_buffers(self, n_items, h0):
    """(per-rank item sets without sharding) the two tables' gradients laid out by item id, all-reduced in place"""
    z = getattr(self, "_dense_bufs", None)
    if z is None or z[0].numel() != n_items * h0:
      f = dict(dtype=torch.float32, device=self.device)
      z = (torch.zeros(n_items * h0, **f), torch.zeros(n_items * h0, **f))
      self._dense_bufs = z
      self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
    return z

  def _gb_dense_buffer(self, n_items):
    z = getattr(self, "_gb_dense", None)
    if z is None or z.numel() != n_items:
      z = torch.zeros(n_items, dtype=torch.float32, device=self.device)
      self._gb_dense = z
      self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
    return z

  def _zero_buffers(self, rows_pad, sh, h0):
    """(D [rows_pad, h]: the gradient rows laid out by item id -- one staging buffer, the two halves of a
    step's exchange use it one after the other on one stream --, the two reduce-scattered shards [sh, h])."""
    z = getattr(self, "_zero_bufs", None)
    if z is None or z[0].numel() != rows_pad * h0:
      f = dict(dtype=torch.float32, device=self.device)
      z = (torch.zeros(rows_pad * h0, **f), torch.zeros(sh * h0, **f), torch.zeros(sh * h0, **f))
      self._zero_bufs = z
      self.alloc_gen = getattr(self, "alloc_gen", 0) + 1
    return z

  def _new_timing_event(self, raw):
    """A timing event -- from the pool graph.GraphStepper.prepare_timed filled BEFORE its capture
    began when there is one (an event created while a stream capture is active cannot be recorded
    as an event-record node: hipEventRecordWithFlags returns invalid argument)."""
    pool = getattr(self, "_event_pool", None)
    if pool:
      return pool.pop()
    return raw.rk_event_create(1)

  def reserve_timing_events(self, n):
    raw = _lib.load()
    pool = self.__dict__.setdefault("_event_pool", [])
    while len(pool) < n:
      pool.append(raw.rk_event_create(1))

  def decoder_bias_grad(self, n_b):
    """gb_de[:n_b] of the last training step (tests).  The one-call step consumes the
    decode epilogue's row-tile partials directly and never materialises gb_de."""
    if self._gb_lazy is None:
      return self.gb_de[:n_b].clone()
    tiles, blk = self._gb_lazy
    if tiles == "slabs":
      ns = int(blk.counts[4].item())
      return sum(self.gb_part[k * blk.n_cap:k * blk.n_cap + n_b] for k in range(ns))
    ld = blk.counts_host()[2]
    return self.gb_part[:tiles * ld].view(tiles, ld)[:, :n_b].sum(0)

  def decoder_row_grad(self, n_b):
    """G_de[:n_b] of the last training step (tests): the one-call step leaves it as K slabs in the
    workspace (summed by rk_adam_multi while it reads the gradient)."""
    h0 = self.h[0]
    if self._dw_slabs is None:
      return self.G_de[:n_b * h0].view(n_b, h0).clone()
    blk, B = self._dw_slabs
    ws = self.ws_dw if getattr(self, "_ws_dw_live", False) else self.ws
    if getattr(self, "_pg_step", False):           # rk_pg_dw: the slabs start at the workspace's head
      ns, off = int(blk.counts[4].item()), 0
      stride = blk.n_cap * h0
      return sum(ws[off + k * stride:off + k * stride + n_b * h0].view(n_b, h0) for k in range(ns))
    ns = int(blk.counts[4].item())
    off = (self.lib.rk_dw3_slabs(ptr(ws), B, h0) - ws.data_ptr()) // 4
    stride = blk.n_cap * h0
    return sum(ws[off + k * stride:off + k * stride + n_b * h0].view(n_b, h0) for k in range(ns))

  def encoder_bias_grad(self):
    """gb_en of the last training step (tests): the one-call step leaves it as row-segment
    partial vectors for rk_adam_multi."""
    n, h0 = getattr(self, "_gb_en_segs", 0), self.h[0]
    if not n:
      return self.gb_en.clone()
    return self.gb_en_parts[:n * h0].view(n, h0).sum(0)

  def encoder_row_grad(self, n_b):
    """G_en[:n_b] of the last training step (tests): the one-call step leaves it as
    row-segment partial arrays for rk_adam_multi when the batch has more than 512 rows."""
    n, h0 = getattr(self, "_gb_en_segs", 0), self.h[0]
    if n <= 1:
      return self.G_en[:n_b * h0].view(n_b, h0).clone()
    stride = self.n_cap_last * h0
    return sum(self.G_en[k * stride:k * stride + n_b * h0].view(n_b, h0) for k in range(n))

  def event_pair_overhead_ms(self, n=64):
    """Elapsed time of a timing-event pair with nothing between the two records."""
    raw = _lib.load()
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pairs = [(raw.rk_event_create(1), raw.rk_event_create(1)) for _ in range(n)]
    torch.cuda.synchronize()
    for e0, e1 in pairs:
      hip.hipEventRecord(ctypes.c_void_p(e0), st)
      hip.hipEventRecord(ctypes.c_void_p(e1), st)
    ms = sorted(raw.rk_event_elapsed_ms(e0, e1) for e0, e1 in pairs)
    for e0, e1 in pairs:
      raw.rk_event_destroy(e0)
      raw.rk_event_destroy(e1)
    return float(ms[len(ms) // 2])

  def timed_samples_ms(self, clear=True):
    """{entry name: [ms per bracketed call]} of the samples taken so far (synchronises)."""
    raw = _lib.load()
    out = {}
    for name, e0, e1 in self._time_samples:
      ms = raw.rk_event_elapsed_ms(e0, e1)
      if ms >= 0:                          # (an entry this step variant never launches: unrecorded)
        out.setdefault(name, []).append(ms)
    if clear:
      for _, e0, e1 in self._time_samples:
        raw.rk_event_destroy(e0)
        raw.rk_event_destroy(e1)
      self._time_samples = []
      self._time_keep = []
    return out

  # ------------------------------------------------------- data parallelism
  def grad_views(self, n_b, part="all"):
    """Views of everything a data-parallel step must SUM over the ranks: the
    live n_b gradient rows, the gathered-bias gradient, the dense gradients and
    the (already 1/(N*B)-scaled) loss.  'decoder' = what the dW chain produces,
    'encoder' = the rest.  MF user-row gradients are rank-private (each user
    lives on one rank) and are not reduced."""
    h0 = self.h[0]
    tied = self.kind == "ae" and bool(self.model.is_constrained)
    dec = [self.gb_de[:n_b], self._loss_target]
    if not tied:
      dec.append(self.G_de[:n_b * h0])
    enc = []
    if self.kind == "ae":
      m = self.model
      enc.append(self.G_de[:n_b * h0] if tied else self.G_en[:n_b * h0])
      enc.append(self.gb_en)
      enc += self.g_enc_w + self.g_enc_b + [g for g in self.g_dec_w if g is not None] + self.g_dec_b
    return dec if part == "decoder" else enc if part == "encoder" else dec + enc

  # ---------------------------------------------------------------- updates
  def _apply_updates(self, blk, row_off, B, stream, part="all", tgt=None):
    """part: 'decoder' = the decoder / item table and its gathered bias (their
    gradients come from the dW chain), 'encoder' = everything else, 'all'.  tgt: the block whose
    item set the decoder-side gradient rows are indexed by (default: blk)."""
    m, S = self.model, self.states
    h0 = self.h[0]
    n_items = blk.n_items
    tb = blk if tgt is None else tgt
    dec = part in ("all", "decoder")
    enc = part in ("all", "encoder")

    own = getattr(self, "_own", None)

    def table(name, G, b=None, parts=None):
      b = blk if b is None else b
      s = S[name]
      if s.sparse and own is not None and name in own["tabs"]:
        # this rank's item range only: rows items[lo : lo + cnt], the gradient = the sum, in rank order,
        # of the world partial arrays the exchange left in R
        R = own["tabs"][name]
        cnt = own["cnt"]
        self._adam_rows(s, b.items[own["lo"]:], None, own["cnt_dev"], max(cnt, 1), R, h0, stream,
                        parts=(ptr(R), self.allreduce.world, max(cnt, 1) * h0, None, None))
      elif s.sparse:
        self._adam_rows(s, b.items, None, b.counts, b.n_cap, G, h0, stream, parts=parts)
      else:
        self._adam_table(s, b.pos, G, h0, n_items, stream, parts=parts)

    # gradients the step left as partial arrays (train_step, single process)
    dw_parts = gb_parts = None
    if dec and self._dw_slabs is not None and self._dw_slabs[0] is tb:
      ws = self.ws_dw if self._ws_dw_live else self.ws
      if getattr(self, "_pg_step", False):
        dw_parts = (ptr(ws), self.lib.rk_pg_dw_splits(self._dw_slabs[1], h0, tb.n_cap), tb.n_cap * h0, None,
                    tb.counts.data_ptr() + 4 * 4)
      else:
        dw_parts = (self.lib.rk_dw3_slabs(ptr(ws), self._dw_slabs[1], h0), self.lib.rk_dw3_max_splits(),
                    tb.n_cap * h0, None, tb.counts.data_ptr() + 4 * 4)
    if dec and self._gb_lazy is not None and self._gb_lazy[1] is tb:
      gb_parts = (ptr(self.gb_part), self._gb_lazy[0], 0, tb.counts.data_ptr() + 2 * 4, None)

    if self.kind == "ae":
      en_w = "en_embedding_layer.weight"
      if m.is_constrained:
        if enc:
          table(en_w, self.G_de)          # tied table: G_de holds dW + encoder rows
      else:
        if enc:
          table(en_w, self.G_en)
        if dec:
          table("de_embedding_layer.weight", self.G_de, tb, dw_parts)
      if enc:
        self._adam_dense(S["_DynamicAutoencoder__en_linear_embedding_layer.bias"], self.gb_en, stream)
        for i in range(self.nl):
          self._adam_dense(S["encoding_layers.%d.weight" % i], self.g_enc_w[i], stream)
          self._adam_dense(S["encoding_layers.%d.bias" % i], self.g_enc_b[i], stream)
          if not m.is_constrained:
            self._adam_dense(S["decoding_layers.%d.weight" % i], self.g_dec_w[i], stream)
          self._adam_dense(S["decoding_layers.%d.bias" % i], self.g_dec_b[i], stream)
      if dec:
        # decoder bias: a dense [n_items] gradient (index_select backward), wd = 0
        self._adam_table(S["_DynamicAutoencoder__de_linear_embedding_layer.bias"], tb.pos,
                         self.gb_de, 1, n_items, stream, parts=gb_parts)
    else:
      lib = self.lib
      if enc:
        rp = getattr(self, "_replay", None)
        users = blk.users[row_off:row_off + B] if rp is None else rp["users_t"]
        su = S["user_embedding_layer.weight"]
        if su.sparse and getattr(self, "_users32", None) is not None:
          # (the forward's gather left the users as int32 rows + count: a job of the one Adam launch)
          u32 = self._users32
          self._adam_rows(su, u32[1:], None, u32[:1], B, self.dbott, h0, stream)
        elif su.sparse:
          self._adam_rows(su, None, users, None, B, self.dbott, h0, stream)
        else:
          check(lib.rk_scatter_pos(ptr(self.pos_u), ptr(users), B, 0, stream), "rk_scatter_pos")
          self._adam_table(su, self.pos_u, self.dbott, h0, m.num_users, stream)
          if dec:
            table("item_embedding_layer.weight", self.G_de, tb, dw_parts)
            self._adam_table(S["bias"], tb.pos, self.gb_de, 1, n_items, stream, parts=gb_parts)
            dec = False
          self._flush_jobs(stream)       # before the user-row map is cleared again
          check(lib.rk_scatter_pos(ptr(self.pos_u), ptr(users), B, 1, stream), "rk_scatter_pos")
      if dec:
        table("item_embedding_layer.weight", self.G_de, tb, dw_parts)
        self._adam_table(S["bias"], tb.pos, self.gb_de, 1, n_items, stream, parts=gb_parts)
    self._flush_jobs(stream)

  # ------------------------------------------------------------- inference
  def encode_eval(self, blk, row_off, B):
    """Evaluation-mode encoder output of rows of `blk` (the input of the decoder GEMM)."""
    self.ensure_capacity(B, blk.n_cap)
    stream = current_stream()
    if self.kind == "ae":
      return self._ae_forward(blk, row_off, B, None, None, False, stream)
    return self._mf_forward(blk.users[row_off:row_off + B], B, None, False, stream)

  def decode_scores(self, z, B, items_blk, out, ld_out):
    """Logits of `z` against the item set of `items_blk` (any subset / strip of the catalogue)."""
    self.ensure_capacity(B, items_blk.n_cap)
    W, b = self._decoder_params()
    stream = current_stream()
    check(self.lib.rk_decode_loss(ptr(z), B, self.h[0], items_blk.ref, 0, ptr(W), ptr(b), LOSS_NONE,
                                  0.0, 1.0, ptr(out), ld_out, None, None,
                                  self._ranges(z, B * self.h[0], stream), stream), "rk_decode_loss")
    return out

  # ---- Recoder.recommend without a score matrix (include/recoder_hip.h "The fused form") ----
  EVAL_CAND_CAP = 4096          # candidate pairs per row (a power of two <= rk_topk_pairs_max_cap)
  EVAL_SAMPLE_MIN = 16384       # least number of sampled items

  def _eval_images(self, n_items, stride):
    """Plane images of the decoder table (all items) and of its strided sample, split once per
    state of the weights (the Adam step counts / torch's version counter change when they do)."""
    W, _ = self._decoder_params()
    h = self.h[0]
    key = (W.data_ptr(), W._version, n_items, stride, tuple(s.step for s in self.states.values()),
           bool(self.lib.rk_gemm_plain_bf16()))
    ev = getattr(self, "_eval_img", None)
    if ev is not None and ev["key"] == key:
      return ev
    self._check_weight_range()
    KT = -(-h // 32)
    m = -(-n_items // stride)
    stream = current_stream()
    ev = dict(key=key, m=m,
              scales=torch.zeros(64, dtype=torch.float32, device=self.device),
              w=torch.empty(n_items * KT * 32, dtype=torch.float32, device=self.device),
              ws=torch.empty(m * KT * 32, dtype=torch.float32, device=self.device),
              n_dev=torch.tensor([n_items], dtype=torch.int32, device=self.device))
    amax_w = self.ranges.data_ptr() + 64 * 4
    check(self.lib.rk_split_image(ptr(W), n_items, h, h, amax_w, 128.0, ptr(ev["w"]), ptr(ev["scales"]), 1,
                                  stream), "rk_split_image")
    check(self.lib.rk_split_image(ptr(W), m, h, h * stride, amax_w, 128.0, ptr(ev["ws"]), ptr(ev["scales"]), 1,
                                  stream), "rk_split_image")
    sblk = Block(1, 1, n_items, self.device, negative_sampling=True, need_bits_cr=False, n_cap=m)
    sblk.set_items(torch.arange(0, n_items, stride, dtype=torch.int32, device=self.device), m, 1)
    ev["sblk"] = sblk
    self._eval_img = ev
    return ev

  def recommend_fused(self, blk, B, k, n_items):
    """Top-k item ids [B, k] (int64, device) of the users collated in `blk` (the whole catalogue as
    columns), or None when a candidate list overflowed / a row has fewer than k unseen items (the
    caller then decodes strip by strip)."""
    from ._lib import RkPlanes
    lib, h, cap = self.lib, self.h[0], self.EVAL_CAND_CAP
    if not (self.split16 and k <= cap // 8 and h % 4 == 0):
      return None
    if lib.rk_gemm_plain_bf16():
      # RK_GEMM_PREC=bf16: rk_split_image writes PLAIN bf16 images, the fused filter launch multiplies
      # fp16 pairs (it would read bf16 bits as fp16 and return wrong ids with status 0): strip path
      return None
    # sample size: the expected number of survivors per row is n_items * k / m -- a quarter of the list
    m_target = max(self.EVAL_SAMPLE_MIN, -(-4 * n_items * k // cap))
    stride = max(1, n_items // m_target)
    ev = self._eval_images(n_items, stride)
    m = ev["m"]
    if m < k:
      return None
    stream = current_stream()
    z = self.encode_eval(blk, 0, B)
    KT = -(-h // 32)
    ws = self.__dict__.setdefault("_eval_buf", {})
    if ws.get("B", 0) < B or ws.get("m") != m or ws.get("k") != k:
      f = dict(dtype=torch.float32, device=self.device)
      ws.update(B=B, m=m, k=k, zimg=torch.empty(B * KT * 32, **f),
                scores=torch.empty(B * (-(-m // 32) * 32), **f),
                s_idx=torch.empty(B, k, dtype=torch.int64, device=self.device), s_val=torch.empty(B, k, **f),
                thr=torch.empty(B, **f), c_val=torch.empty(B * cap, **f),
                c_idx=torch.empty(B * cap, dtype=torch.int32, device=self.device),
                c_cnt=torch.zeros(B + 1, dtype=torch.int32, device=self.device),
                out=torch.empty(B, k, dtype=torch.int64, device=self.device))
    # Z image (scale slot 0): the bound of |z| for unbounded activations, else the static scale
    amax_z = None
    if not self.act_bounded:
      check(lib.rk_amax(ptr(z), B * h, ptr(self.ranges), stream), "rk_amax")
      amax_z = ptr(self.ranges)
    check(lib.rk_split_image(ptr(z), B, h, h, amax_z, 32.0, ptr(ws["zimg"]), ptr(ev["scales"]), 0, stream),
          "rk_split_image")
    _, b = self._decoder_params()
    # 1. the strided sample: scores -> masked top k -> the k-th best is the row's bound
    pl = RkPlanes()
    pl.scales, pl.z, pl.w, pl.wt = ptr(ev["scales"]), ptr(ws["zimg"]), ptr(ev["ws"]), None
    pl.h, pl.B_cap, pl.n_cap, pl.n_ld = h, B, m, -(-m // 32) * 32
    ld = -(-m // 32) * 32
    sblk = ev["sblk"]
    sblk.c.S_cap = max(sblk.c.S_cap, B)        # (an item set without rows: any B)
    check(lib.rk_decode_loss_planes(ctypes.byref(pl), B, sblk.ref, 0, ptr(b), LOSS_NONE, 0.0, 1.0,
                                    ptr(ws["scores"]), ld, None, None, stream), "rk_decode_loss_planes")
    stride_ = ev["key"][3]
    check(lib.rk_topk_masked(ptr(ws["scores"]), B, m, ld, blk.ref, 0, k, 0, stride_,
                                     ptr(ws["s_idx"]), ptr(ws["s_val"]), k, stream), "rk_topk_masked")
    ws["thr"][:B].copy_(ws["s_val"][:B, k - 1])
    # 2. the whole catalogue with the filter in the decode's epilogue
    ws["c_cnt"].zero_()
    status = ws["c_cnt"][-1:]                  # (behind the rows' counters)
    check(lib.rk_decode_filter_planes(ptr(ws["zimg"]), ptr(ev["w"]), ptr(ev["scales"]), h, B, n_items, 0,
                                      ptr(b), blk.ref, 0, ptr(ws["thr"]), ptr(ws["c_val"]), ptr(ws["c_idx"]),
                                      ptr(ws["c_cnt"]), cap, ptr(ev["n_dev"]), stream),
          "rk_decode_filter_planes")
    # 3. the k best candidates of every row
    check(lib.rk_topk_pairs(ptr(ws["c_val"]), ptr(ws["c_idx"]), ptr(ws["c_cnt"]), B, cap, k, ptr(ws["out"]),
                            k, ptr(status), stream), "rk_topk_pairs")
    return ws["out"][:B], status

  def predict_scores(self, blk, row_off, B, out, ld_out, tgt_items_blk):
    """Logits for rows of ``blk`` against the item set of ``tgt_items_blk``
    (model.py:487-511 with input_items=None: the whole catalogue)."""
    self.ensure_capacity(B, max(blk.n_cap, tgt_items_blk.n_cap))
    stream = current_stream()
    if self.kind == "ae":
      z = self._ae_forward(blk, row_off, B, None, None, False, stream)
    else:
      z = self._mf_forward(blk.users[row_off:row_off + B], B, None, False, stream)
    W, b = self._decoder_params()
    check(self.lib.rk_decode_loss(ptr(z), B, self.h[0], tgt_items_blk.ref, 0, ptr(W), ptr(b),
                                  LOSS_NONE, 0.0, 1.0, ptr(out), ld_out, None, None,
                                  self._ranges(z, B * self.h[0], stream), stream), "rk_decode_loss")
    return out


# ----------------------------------------------------------------------------
# dense-input forward of the nn modules (API compatibility: nn.py:228-253,
# 344-362); builds a block from the dense tensor and runs the same kernels.
# ----------------------------------------------------------------------------
class _DenseCSR:
  def __init__(self, x):
    B, n = x.shape
    nzmask = x != 0
    counts = nzmask.sum(dim=1)
    self.indptr = torch.zeros(B + 1, dtype=torch.int64, device=x.device)
    self.indptr[1:] = torch.cumsum(counts, 0)
    idx = nzmask.nonzero(as_tuple=False)
    self.indices = idx[:, 1].to(torch.int32).contiguous()
    self.data = x[nzmask].to(torch.float32).contiguous()
    self.nnz = int(self.indices.numel())
    if self.nnz == 0:
      self.indices = torch.zeros(1, dtype=torch.int32, device=x.device)
      self.data = torch.zeros(1, dtype=torch.float32, device=x.device)
    self.shape = (B, n)
    self.n_items = n


def _items_block(items_i32, n, n_items, B, device):
  blk = Block(B, 1, n_items, device, negative_sampling=False, need_bits_cr=False) \
      if items_i32 is None else Block(B, max(1, n), max(n_items, n), device, negative_sampling=True,
                                      need_bits_cr=False)
  if items_i32 is None:
    ar = torch.arange(n_items, dtype=torch.int32, device=device)
    blk.set_items(ar, n_items, B)
  else:
    blk.set_items(items_i32, n, B)
  return blk


def _model_engine(model, kind):
  eng = getattr(model, "_rk_engine", None)
  if eng is None:
    eng = FusedEngine(model, kind, "mse", None)
    object.__setattr__(model, "_rk_engine", eng)
  return eng


@torch.no_grad()
def ae_dense_forward(model, x, input_items=None, target_items=None):
  dev = require_gpu()
  x = x.to(dev, torch.float32).contiguous()
  B, n_in = x.shape
  dcsr = _DenseCSR(x)
  blk = Block(B, max(1, dcsr.nnz), n_in, dev, negative_sampling=False, need_bits_cr=False)
  users = torch.arange(B, dtype=torch.int64, device=dev)
  blk.collate(dcsr, users, negative_sampling=False)
  if input_items is not None:
    blk.items[:n_in].copy_(input_items.to(dev).to(torch.int32))
    blk.c.gcols = None        # the per-entry global ids of the collation no longer apply
  n_items = model.num_items
  if target_items is not None:
    t = target_items.to(dev).to(torch.int32).contiguous()
    tblk = _items_block(t, int(t.numel()), n_items, B, dev)
    n_t = int(t.numel())
  else:
    tblk = _items_block(None, n_items, n_items, B, dev)
    n_t = n_items
  eng = _model_engine(model, "ae")
  train = model.training
  eng.ensure_capacity(B, max(blk.n_cap, tblk.n_cap))
  stream = current_stream()
  eng.rng_step += 1
  z = eng._ae_forward(blk, 0, B, None, None, train, stream)
  W, b = eng._decoder_params()
  out = torch.empty(B, n_t, dtype=torch.float32, device=dev)
  eng._w_range_stale = True          # (a bare module: its weights may have been set from anywhere)
  check(eng.lib.rk_decode_loss(ptr(z), B, eng.h[0], tblk.ref, 0, ptr(W), ptr(b), LOSS_NONE, 0.0, 1.0,
                               ptr(out), n_t, None, None, eng._ranges(z, B * eng.h[0], stream), stream),
        "rk_decode_loss")
  return out


@torch.no_grad()
def mf_dense_forward(model, input_users, target_items=None):
  dev = require_gpu()
  users = input_users.to(dev).to(torch.int64).contiguous()
  B = int(users.numel())
  n_items = model.num_items
  if target_items is not None:
    t = target_items.to(dev).to(torch.int32).contiguous()
    tblk = _items_block(t, int(t.numel()), n_items, B, dev)
    n_t = int(t.numel())
  else:
    tblk = _items_block(None, n_items, n_items, B, dev)
    n_t = n_items
  eng = _model_engine(model, "mf")
  eng.ensure_capacity(B, tblk.n_cap)
  stream = current_stream()
  eng.rng_step += 1
  z = eng._mf_forward(users, B, None, model.training, stream)
  W, b = eng._decoder_params()
  out = torch.empty(B, n_t, dtype=torch.float32, device=dev)
  eng._w_range_stale = True          # (a bare module: its weights may have been set from anywhere)
  check(eng.lib.rk_decode_loss(ptr(z), B, eng.h[0], tblk.ref, 0, ptr(W), ptr(b), LOSS_NONE, 0.0, 1.0,
                               ptr(out), n_t, None, None, eng._ranges(z, B * eng.h[0], stream), stream),
        "rk_decode_loss")
  return out
