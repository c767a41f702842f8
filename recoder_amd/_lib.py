"""ctypes binding of librecoder_hip.so (the C ABI in include/recoder_hip.h).

This is the binding a maintainer of the reference would add:
plain pointers and sizes, no torch types cross the boundary.  There is NO
fallback: if the library is missing or a symbol is absent, import of the hot
path fails loudly.
"""
import ctypes
import os

# PyTorch-ROCm must load its HIP runtime BEFORE this library is dlopen'ed: both link
# libamdhip64 and the process must end up with ONE runtime instance (otherwise our
# launches see "no ROCm-capable device" on streams torch created).
import torch  # noqa: F401  (side effect: loads the HIP runtime torch will use)

from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librecoder_hip.so")

ACT = {"none": 0, "tanh": 1, "sigmoid": 2, "relu": 3, "selu": 4, "elu": 5}
LOSS_MSE, LOSS_BCE, LOSS_MNLL, LOSS_NONE = 0, 1, 2, 3
SCAN_CHUNK = 2048


class RkBlock(Structure):
  """mirror of rk_block_t"""
  _fields_ = [
    ("S_cap", c_int32), ("nnz_cap", c_int32), ("n_cap", c_int32), ("n_items", c_int32),
    ("ldw_rc", c_int32), ("ldw_cr", c_int32), ("n_chunks", c_int32), ("implicit", c_int32),
    ("counts", c_void_p), ("indptr", c_void_p), ("cols", c_void_p), ("vals", c_void_p),
    ("svals", c_void_p), ("items", c_void_p), ("pos", c_void_p), ("mark", c_void_p),
    ("bits_rc", c_void_p), ("bits_cr", c_void_p), ("scan_tmp", c_void_p),
    ("pref_rc", c_void_p), ("gcols", c_void_p),
  ]


class RkAdamParam(Structure):
  """mirror of rk_adam_param_t"""
  _fields_ = [("p", c_void_p), ("m", c_void_p), ("v", c_void_p),
              ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
              ("weight_decay", c_double), ("step", c_int32), ("sparse", c_int32)]


class RkPlan(Structure):
  """mirror of rk_plan_t"""
  _fields_ = [
    ("B", c_int32), ("h", c_int32), ("n_cap", c_int32), ("loss_kind", c_int32), ("row_off", c_int32),
    ("gemm_split16", c_int32), ("gemm_plain_bf16", c_int32), ("dw_pairs", c_int32), ("split_zt_ok", c_int32),
    ("pg_enabled", c_int32), ("graph_timing_supported", c_int32), ("decode_row_tile", c_int32),
    ("dw3_max_splits", c_int32), ("topk_max_k", c_int32), ("topk_pairs_max_cap", c_int32),
    ("loss_partials", c_int32), ("dw_splits", c_int32), ("pg_dw_splits", c_int32),
    ("dw3_rows_pad", c_int32), ("dw3_cols_pad", c_int32), ("pg_granule_rows", c_int32), ("pg_granule_cols", c_int32),
    ("planes_bytes", c_int64), ("dz_workspace_bytes", c_int64), ("dz_fused_workspace_bytes", c_int64),
    ("dw_workspace_bytes", c_int64), ("dw3_workspace_bytes", c_int64), ("dw3_planes_bytes", c_int64),
    ("fdec_workspace_bytes", c_int64), ("pg_dz_workspace_bytes", c_int64), ("pg_dw_workspace_bytes", c_int64),
    ("pg_scale_floats", c_int64), ("pg_mnll_workspace_floats", c_int64),
    ("decode_dz_fused_ok", c_int32), ("fdec_ok", c_int32), ("dw_encode_bwd_fused_ok", c_int32),
    ("encode_bwd_segments", c_int32), ("dw3_slabs_offset_bytes", c_int64),
    ("mf_fdec_ok", c_int32),
  ]


PAR_W_EN, PAR_B_EN, PAR_W_DE, PAR_B_DE = 0, 1, 2, 3
ENTRY_ALL = -1
ENTRY = {"rk_ae_encode_fwd": 1, "rk_decode_loss": 2, "rk_decode_bwd_dz": 3, "rk_decode_bwd_dw": 4,
         "rk_ae_encode_bwd": 5, "rk_adam_multi": 6}


class RkAeStep(Structure):
  """mirror of rk_ae_step_t"""
  _fields_ = [
    ("blk", POINTER(RkBlock)),
    ("row_off", c_int32), ("B", c_int32), ("h", c_int32), ("act", c_int32), ("loss_kind", c_int32),
    ("tied", c_int32),
    ("confidence", c_float), ("inv_B", c_float), ("denom", c_float), ("noise_p", c_float),
    ("seed", c_uint64), ("rng_step", c_uint64),
    ("keep", c_void_p), ("users", c_void_p),
    ("par", RkAdamParam * 4),
    ("Z0", c_void_p), ("dZ0", c_void_p), ("dO", c_void_p), ("G_de", c_void_p), ("G_en", c_void_p),
    ("gb_de", c_void_p), ("gb_part", c_void_p), ("gb_en", c_void_p), ("ws", c_void_p),
    ("loss_part", c_void_p), ("loss_out", c_void_p),
    ("stream", c_void_p),
    ("time_entry", c_int32), ("phase", c_int32),
    ("time_ev0", c_void_p), ("time_ev1", c_void_p),
    ("user_norm", c_void_p), ("own_rank", c_int32), ("own_world", c_int32),
    ("zt_planes", c_void_p),
    ("ranges", c_void_p),
    ("cursor", c_void_p), ("cursor_off", c_int32), ("cursor_advance", c_int32), ("adam_table", c_void_p),
    ("cursor_next", c_void_p),
    ("time_all", POINTER(c_void_p)),
    ("ws_dw", c_void_p), ("dw_stream", c_void_p), ("dw_fork", c_void_p), ("dw_join", c_void_p),
    ("planes", c_void_p),
    ("do_scales", c_void_p), ("do_rows", c_int32),
    ("zero_lo", c_int32), ("zero_hi", c_int32), ("zero_g_en", c_void_p), ("zero_g_de", c_void_p),
    ("zero_gb_de", c_void_p),
    ("lazy_stamp_en", c_void_p), ("lazy_stamp_de", c_void_p), ("lazy_pos_next", c_void_p), ("lazy_period", c_int32),
    ("lazy_need_list", c_void_p), ("lazy_need_count", c_void_p),
  ]


class RkReplay(Structure):
  """mirror of rk_replay_t"""
  _fields_ = [("cursor", c_void_p), ("off", c_int32), ("B", c_int32), ("users_base", c_void_p),
              ("adam_table", c_void_p), ("tab_stride", c_int32), ("advance", c_int32),
              ("cursor_next", c_void_p)]


class RkPlanes(Structure):
  """mirror of rk_planes_t"""
  _fields_ = [("scales", c_void_p), ("z", c_void_p), ("w", c_void_p), ("wt", c_void_p),
              ("h", c_int32), ("B_cap", c_int32), ("n_cap", c_int32), ("n_ld", c_int32)]


class RkAdamJob(Structure):
  """mirror of rk_adam_job_t"""
  _fields_ = [
    ("par", RkAdamParam),
    ("n_rows", c_int32), ("h", c_int32),
    ("pos", c_void_p), ("rows", c_void_p), ("n_dev", c_void_p),
    ("n_cap", c_int32), ("g_parts", c_int32), ("g_stride", c_int32),
    ("gstride_dev", c_void_p), ("g", c_void_p),
    ("row0", c_int32), ("row_step", c_int32),
    ("amax_out", c_void_p), ("gparts_dev", c_void_p),
    ("lazy_pos_next", c_void_p), ("lazy_stamp", c_void_p), ("lazy_period", c_int32),
    ("lazy_need_list", c_void_p), ("lazy_need_count", c_void_p),
  ]


STEP_FWD_DW, STEP_DZ_ENC, STEP_UPDATE, STEP_ALL = 1, 2, 4, 7
STEP_IP_ENC, STEP_IP_MID, STEP_IP_TAIL = 8, 16, 32


_P = c_void_p
_BLK = POINTER(RkBlock)

# name -> (restype, argtypes); every symbol include/recoder_hip.h declares
SIGNATURES = {
  "rk_plan": (c_int32, [POINTER(RkPlan)]),
  "rk_probe_buffer": (c_int32, [c_int32, _P]),
  "rk_tune": (c_int32, [c_int32, c_int32]),
  "rk_version": (c_int32, []),
  "rk_last_error": (c_char_p, []),
  "rk_collate": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _BLK, _P]),
  "rk_ae_encode_fwd_partial": (c_int32, [_BLK, c_int32, c_int32, _P, c_int32, _P, c_float, c_uint64,
                                         c_uint64, _P, _P, _P, _P]),
  "rk_bias_act": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P]),
  "rk_rows_to_dense": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P]),
  "rk_zero_tail_rows": (c_int32, [POINTER(c_void_p), POINTER(c_int32), c_int32, _P, c_int32, _P, _P]),
  "rk_densify": (c_int32, [_BLK, c_int32, c_int32, c_int32, _P, c_int32, _P]),
  "rk_ae_encode_fwd": (c_int32, [_BLK, c_int32, c_int32, _P, _P, c_int32, _P, c_float, c_uint64,
                                 c_uint64, _P, c_int32, _P, _P]),
  "rk_ae_encode_fwd_split_w": (c_int32, [_BLK, c_int32, c_int32, _P, _P, c_int32, _P, c_float, c_uint64,
                                         c_uint64, _P, c_int32, _P, _P, _P, POINTER(RkPlanes), _P]),
  "rk_ae_encode_bwd": (c_int32, [_BLK, c_int32, c_int32, _P, c_int32, _P, c_int32, _P, _P]),
  "rk_decode_loss": (c_int32, [_P, c_int32, c_int32, _BLK, c_int32, _P, _P, c_int32, c_float,
                               c_float, _P, c_int32, _P, _P, _P, _P]),
  "rk_amax": (c_int32, [_P, c_int64, _P, _P]),
  "rk_planes_layout": (c_int32, [_P, c_int32, c_int32, c_int32, POINTER(RkPlanes)]),
  "rk_split_wz": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, POINTER(RkPlanes), _P, _P]),
  "rk_decode_loss_planes": (c_int32, [POINTER(RkPlanes), c_int32, _BLK, c_int32, _P, c_int32, c_float,
                                      c_float, _P, c_int32, _P, _P, _P]),
  "rk_decode_bwd_dz_planes": (c_int32, [_P, c_int32, POINTER(RkPlanes), _BLK, _P, c_int32, _P, _P, _P]),
  "rk_mnll_row_stats": (c_int32, [_P, c_int32, _BLK, _P, _P]),
  "rk_mnll_finish": (c_int32, [_P, c_int32, _BLK, c_int32, c_float, _P, _P, _P, _P, _P]),
  "rk_loss_reduce": (c_int32, [_P, c_int32, c_float, _P, _P]),
  "rk_decode_bwd_dz": (c_int32, [_P, c_int32, c_int32, _BLK, _P, _P, c_int32, _P, _P, _P, _P]),
  "rk_decode_bwd_dw": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P]),
  "rk_decode_bwd_dw3": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P, _P, _P]),
  "rk_decode_bwd_dw2": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P, _P, _P, _P]),
  "rk_decode_bwd_dw2_encode_bwd": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P, c_int32, _P, _P, _P, _P]),
  "rk_decode_bwd_dw2_dz_reduce": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P, _P, _P, c_int32, _P, _P]),
  "rk_decode_bwd_dw2_encode_bwd_colsum": (c_int32, [_P, _P, c_int32, c_int32, _BLK, _P, _P, _P, c_int32, _P, _P,
                                                    _P, _P, _P]),
  "rk_decode_loss_dz_planes": (c_int32, [_P, c_int32, _BLK, c_int32, _P, c_int32, c_float, c_float, _P, _P, _P,
                                         _P, _P]),
  "rk_decode_dz_reduce": (c_int32, [_P, c_int32, c_int32, _BLK, _P, c_int32, _P, _P]),
  "rk_fdec_loss_dz": (c_int32, [POINTER(RkPlanes), c_int32, _BLK, c_int32, _P, c_int32, c_float, c_float, _P, c_int32,
                                _P, _P, _P, _P]),
  "rk_fdec_dz_reduce": (c_int32, [_P, c_int32, c_int32, _BLK, _P, c_int32, _P, _P]),
  "rk_pg_dw_dz_reduce": (c_int32, [_P, _P, c_int32, c_int32, c_int32, POINTER(RkPlanes), _BLK, _P, _P, _P, _P, c_int32,
                                   _P, c_int32, _P]),
  "rk_ae_step_uses_pg": (c_int32, [_P]),
  "rk_pg_decode_loss": (c_int32, [POINTER(RkPlanes), c_int32, _BLK, c_int32, _P, c_int32, c_float, c_float, _P,
                                  c_int32, _P, _P, _P, _P, _P]),
  "rk_pg_decode_mnll": (c_int32, [POINTER(RkPlanes), c_int32, _BLK, c_int32, _P, c_float, _P, _P, c_int32, _P, _P, _P,
                                  _P, _P]),
  "rk_pg_dz": (c_int32, [_P, _P, c_int32, c_int32, c_int32, POINTER(RkPlanes), _BLK, _P, c_int32, _P, _P, _P]),
  "rk_pg_dw": (c_int32, [_P, _P, c_int32, c_int32, c_int32, POINTER(RkPlanes), _BLK, _P, _P, _P]),
  "rk_pg_dw_encode_bwd": (c_int32, [_P, _P, c_int32, c_int32, c_int32, POINTER(RkPlanes), _BLK, _P, c_int32, _P, _P, _P,
                                    _P, _P]),
  "rk_linear_fwd": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
  "rk_linear_bwd": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                              c_int32, _P, _P]),
  "rk_linear_bwd_pre": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P,
                                  _P]),
  "rk_linear_bwd_dact": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                   c_int32, _P, _P, _P]),
  "rk_act_grad": (c_int32, [_P, _P, c_int64, c_int32, _P]),
  "rk_dropout": (c_int32, [_P, _P, c_int64, c_int32, c_float, c_uint64, c_uint64, _P]),
  "rk_colsum": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P, _P]),
  "rk_gather_rows_amax": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P]),
  "rk_adam_rows": (c_int32, [_P, _P, _P, c_int32, _P, _P, _P, c_int32, _P, c_double, c_double,
                             c_double, c_double, c_int32, _P]),
  "rk_adam_multi": (c_int32, [POINTER(RkAdamJob), c_int32, _P, c_int32, c_float, _P, _P]),
  "rk_adam_lazy_flush": (c_int32, [POINTER(RkAdamJob), c_int32, _P, c_int32, POINTER(c_int32), c_int64, c_int64, _P]),
  "rk_scatter_pos": (c_int32, [_P, _P, c_int32, c_int32, _P]),
  "rk_comm_unique_id": (c_int32, [_P, c_char_p]),
  "rk_comm_init": (c_void_p, [_P, c_int32, c_int32, c_char_p]),
  "rk_comm_destroy": (None, [_P]),
  "rk_allreduce_bucket": (c_int32, [_P, POINTER(c_void_p), POINTER(c_int64), c_int32, c_int32, c_int32, _P]),
  "rk_reduce_scatter": (c_int32, [_P, _P, _P, c_int64, c_int32, _P]),
  "rk_all_gather": (c_int32, [_P, _P, _P, c_int64, c_int32, _P]),
  "rk_exchange": (c_int32, [_P, c_int32, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), POINTER(c_int64),
                            c_int32, _P]),
  "rk_event_create": (c_void_p, [c_int32]),
  "rk_event_destroy": (None, [c_void_p]),
  "rk_event_elapsed_ms": (c_float, [c_void_p, c_void_p]),
  "rk_ae_train_step": (c_int32, [POINTER(RkAeStep)]),
  "rk_collate_at": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P, c_int32, _BLK, _P]),
  "rk_lazy_need_lists": (c_int32, [POINTER(_BLK), c_int32, POINTER(c_void_p), POINTER(c_void_p), _P]),
  "rk_collate_at_multi": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P, c_int32, POINTER(_BLK), c_int32,
                                          c_int32, _P]),
  "rk_cursor_set": (c_int32, [_P, c_int64, c_int64, _P]),
  "rk_adam_consts": (c_int32, [c_double, c_double, c_double, c_double, c_double, c_int32, c_int32,
                               c_int32, _P]),
  "rk_replay_set": (None, [POINTER(RkReplay)]),
  "rk_graph_begin": (c_int32, [_P]),
  "rk_graph_end": (c_void_p, [_P]),
  "rk_graph_launch": (c_int32, [_P, _P]),
  "rk_graph_destroy": (None, [_P]),
  "rk_event_record": (c_int32, [_P, _P]),
  "rk_stream_wait_event": (c_int32, [_P, _P]),
  "rk_split_image": (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_float, _P, _P, c_int32, _P]),
  "rk_topk_masked": (c_int32, [_P, c_int32, c_int32, c_int32, _BLK, c_int32, c_int32, c_int32, c_int32,
                                       _P, _P, c_int32, _P]),
  "rk_decode_filter_planes": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _BLK, c_int32, _P,
                                        _P, _P, _P, c_int32, _P, _P]),
  "rk_topk_pairs": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P, c_int32, _P, _P]),
}

_lib = None


class RecoderHipError(RuntimeError):
  pass


def load():
  """Load the shared library (once) and bind every declared symbol."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RecoderHipError(
        "librecoder_hip.so not found at %s -- build it with `python -m recoder_amd.build` "
        "(there is no CPU fallback for the training path)" % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
    fn.restype = res
    fn.argtypes = args
  _install_plan_accessors(lib)
  # the probe header's two entry points under the names tools/ and tests/ were written against
  lib.rk_gemm_probe = lambda buf: lib.rk_probe_buffer(0, buf)
  lib.rk_dw3_probe = lambda buf: lib.rk_probe_buffer(1, buf)
  lib.rk_enc_probe = lambda buf: lib.rk_probe_buffer(2, buf)
  lib.rk_planes_probe = lambda buf: lib.rk_probe_buffer(3, buf)
  def tune(knob):
    def set_(value):
      _plans.clear()                      # (the plan's predicates follow the knobs)
      return lib.rk_tune(knob, value)
    return set_
  lib.rk_linear_pair, lib.rk_planes_tile = tune(0), tune(1)
  _lib = lib
  return lib


_plans = {}


def plan(B=0, h=0, n_cap=0, loss_kind=0, row_off=0):
  """rk_plan of one shape (cached: the plan is a pure function of its inputs and the process environment)."""
  key = (int(B), int(h), int(n_cap), int(loss_kind), int(row_off))
  p = _plans.get(key)
  if p is None:
    p = RkPlan()
    p.B, p.h, p.n_cap, p.loss_kind, p.row_off = key
    check(load().rk_plan(ctypes.byref(p)), "rk_plan")
    _plans[key] = p
  return p


def _install_plan_accessors(lib):
  """lib.rk_<field>(...) for the fields of rk_plan_t: the one-value accessors the engine, the tests and the
  tools were written against (the library itself exports rk_plan only)."""
  def field(name, args):
    def get(*a):
      return getattr(plan(**dict(zip(args, a))), name)
    return get
  for name, args in (
      ("dz_workspace_bytes", ("B", "h")), ("loss_partials", ("B", "n_cap")), ("decode_row_tile", ()),
      ("planes_bytes", ("B", "h", "n_cap")), ("split_zt_ok", ()), ("dw_workspace_bytes", ("B", "h", "n_cap")),
      ("dw3_workspace_bytes", ("B", "h", "n_cap")), ("dw3_max_splits", ()), ("dw_pairs", ()),
      ("dw_encode_bwd_fused_ok", ("row_off", "B")), ("decode_dz_fused_ok", ("B", "h", "n_cap", "loss_kind")),
      ("dz_fused_workspace_bytes", ("B", "h", "n_cap")), ("fdec_ok", ("B", "h", "n_cap", "loss_kind")),
      ("mf_fdec_ok", ("B", "h", "n_cap", "loss_kind")),
      ("fdec_workspace_bytes", ("B", "h", "n_cap")), ("pg_enabled", ()), ("pg_scale_floats", ("B", "n_cap")),
      ("pg_mnll_workspace_floats", ("B", "n_cap")), ("pg_dz_workspace_bytes", ("B", "h")),
      ("pg_dw_splits", ("B", "h", "n_cap")), ("pg_dw_workspace_bytes", ("B", "h", "n_cap")),
      ("dw3_planes_bytes", ("B", "h")), ("dw3_rows_pad", ("B",)), ("dw3_cols_pad", ("h",)), ("gemm_split16", ()),
      ("gemm_plain_bf16", ()), ("dw_splits", ("B",)), ("graph_timing_supported", ()), ("topk_max_k", ()),
      ("topk_pairs_max_cap", ()), ("encode_bwd_segments", ("B",))):
    setattr(lib, "rk_" + name, field(name, args))
  # (the K slabs of rk_decode_bwd_dw3 / dw2 inside their workspace: a pointer, as the export of rounds 2-5 returned it)
  lib.rk_dw3_slabs = lambda ws, B, h: (ws or 0) + plan(B=B, h=h).dw3_slabs_offset_bytes

  def granule(B, n_cap, gr, gc):
    p = plan(B=B, n_cap=n_cap)
    gr._obj.value, gc._obj.value = p.pg_granule_rows, p.pg_granule_cols
  lib.rk_pg_decode_granule = granule


def check(rc, what=""):
  if rc != 0:
    msg = load().rk_last_error()
    raise RecoderHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
  """device (or host) pointer of a torch tensor / None"""
  return None if t is None else t.data_ptr()
