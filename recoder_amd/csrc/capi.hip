// Error plumbing + version of the C ABI (include/recoder_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void rk_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *rk_last_error(void) { return g_err; }
extern "C" int rk_version(void) { return 100; }

// the sizing plan of one step shape (include/recoder_hip.h rk_plan_t): the helpers of csrc/internal.h
extern "C" int rk_plan(rk_plan_t *p) {
  RK_REQUIRE(p != nullptr && p->B >= 0 && p->h >= 0 && p->n_cap >= 0, "rk_plan: B, h, n_cap >= 0");
  // (an input left at 0 -- a caller interested in one field only -- is sized as 1: no helper divides by it)
  const int32_t B = p->B > 0 ? p->B : 1, h = p->h > 0 ? p->h : 1, n = p->n_cap > 0 ? p->n_cap : 1, loss = p->loss_kind;
  p->gemm_split16 = rk_gemm_split16();
  p->gemm_plain_bf16 = rk_gemm_plain_bf16();
  p->dw_pairs = rk_dw_pairs();
  p->split_zt_ok = rk_split_zt_ok();
  p->pg_enabled = rk_pg_enabled();
  p->graph_timing_supported = rk_graph_timing_supported();
  p->decode_row_tile = rk_decode_row_tile();
  p->dw3_max_splits = rk_dw3_max_splits();
  p->topk_max_k = rk_topk_max_k();
  p->topk_pairs_max_cap = rk_topk_pairs_max_cap();
  p->loss_partials = rk_loss_partials(B, n);
  p->dw_splits = rk_dw_splits(B);
  p->pg_dw_splits = rk_pg_dw_splits(B, h, n);
  p->dw3_rows_pad = rk_dw3_rows_pad(B);
  p->dw3_cols_pad = rk_dw3_cols_pad(h);
  rk_pg_decode_granule(B, n, &p->pg_granule_rows, &p->pg_granule_cols);
  p->planes_bytes = rk_planes_bytes(B, h, n);
  p->dz_workspace_bytes = rk_dz_workspace_bytes(B, h);
  p->dz_fused_workspace_bytes = rk_dz_fused_workspace_bytes(B, h, n);
  p->dw_workspace_bytes = rk_dw_workspace_bytes(B, h, n);
  p->dw3_workspace_bytes = rk_dw3_workspace_bytes(B, h, n);
  p->dw3_planes_bytes = rk_dw3_planes_bytes(B, h);
  p->fdec_workspace_bytes = rk_fdec_workspace_bytes(B, h, n);
  p->pg_dz_workspace_bytes = rk_pg_dz_workspace_bytes(B, h);
  p->pg_dw_workspace_bytes = rk_pg_dw_workspace_bytes(B, h, n);
  p->pg_scale_floats = rk_pg_scale_floats(B, n);
  p->pg_mnll_workspace_floats = rk_pg_mnll_workspace_floats(B, n);
  p->decode_dz_fused_ok = rk_decode_dz_fused_ok(B, h, n, loss);
  p->fdec_ok = rk_fdec_ok(B, h, n, loss);
  p->dw_encode_bwd_fused_ok = rk_dw_encode_bwd_fused_ok(p->row_off, B);
  p->encode_bwd_segments = rk_encode_bwd_segments(B);
  p->dw3_slabs_offset_bytes = (const char *)rk_dw3_slabs(nullptr, B, h) - (const char *)nullptr;
  p->mf_fdec_ok = (rk_tune_get(RK_TUNE_MF_FDEC) != 0 && B < 1024 && rk_fdec_ok(B, h, n, loss)) ? 1 : 0;
  return 0;
}

// replay context of the per-entry sequencing (include/recoder_hip.h rk_replay_t)
static thread_local rk_replay_t g_replay;
static thread_local bool g_replay_on = false;
extern "C" void rk_replay_set(const rk_replay_t *ctx) {
  g_replay_on = ctx != nullptr && ctx->cursor != nullptr;
  if (g_replay_on) g_replay = *ctx;
}
const rk_replay_t *rk_replay_get(void) { return g_replay_on ? &g_replay : nullptr; }

// include/recoder_hip_probe.h: the tuning probes and switches of tools/ and tests/ behind two entry points
extern "C" int rk_probe_buffer(int32_t which, unsigned long long *buffer) {
  switch (which) {
    case 0: rk_gemm_probe(buffer); return 0;
    case 1: rk_dw3_probe(buffer); return 0;
    case 2: rk_enc_probe(buffer); return 0;
    case 3: rk_planes_probe(buffer); return 0;
  }
  rk_set_error("rk_probe_buffer: unknown probe %d", which);
  return -1;
}
// defaults of the knobs (include/recoder_hip_probe.h RK_TUNE_*)
static int g_tune[RK_TUNE_COUNT] = {1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 1};
int rk_tune_get(int knob) { return g_tune[knob]; }
extern "C" int rk_tune(int32_t knob, int32_t value) {
  if (knob < 0 || knob >= RK_TUNE_COUNT) {
    rk_set_error("rk_tune: unknown knob %d", knob);
    return -1;
  }
  g_tune[knob] = value;
  return 0;
}
