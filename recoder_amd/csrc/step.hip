// rk_ae_train_step: one C-ABI call = one optimisation step of the
// DynamicAutoencoder hot path (reference model.py:383-404 for the common
// hidden_layers=[h] case): encode -> decode + loss -> {dW chain || dZ chain} ->
// Adam / SparseAdam.  It only sequences the kernels of this library on the two
// caller-provided HIP streams; keeping the sequencing in C removes ~35
// Python->ctypes transitions per step (the host enqueue time was approaching the
// GPU time of the step).
#include <stdlib.h>

#include "common.h"
#include "planes.h"

extern "C" void *rk_event_create(int32_t timing) {
  // timing == 0: ordering-only events between streams of ONE device -- no timing, and no
  // system-scope fence (host / peer visibility is not needed for them)
  hipEvent_t e = nullptr;
  const hipError_t rc = timing ? hipEventCreate(&e)
                               : hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  if (rc != hipSuccess) {
    rk_set_error("hipEventCreate failed");
    return nullptr;
  }
  return (void *)e;
}

extern "C" void rk_event_destroy(void *e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}

extern "C" float rk_event_elapsed_ms(void *e0, void *e1) {
  float ms = -1.f;
  if (hipEventSynchronize((hipEvent_t)e1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) {
    (void)hipGetLastError();      // (an entry this step never launched: do not leave the error behind)
    return -1.f;
  }
  return ms;
}

extern "C" int rk_event_record(void *event, void *stream) {
  if (hipEventRecord((hipEvent_t)event, (hipStream_t)stream) != hipSuccess) {
    rk_set_error("hipEventRecord failed");
    return -1;
  }
  return 0;
}

extern "C" int rk_stream_wait_event(void *stream, void *event) {
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) != hipSuccess) {
    rk_set_error("hipStreamWaitEvent failed");
    return -1;
  }
  return 0;
}

extern "C" int rk_graph_begin(void *stream) {
  // relaxed mode: other threads (torch's allocator, a second virtual rank) may call into HIP
  const hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    rk_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e));
    return -1;
  }
  return 0;
}

extern "C" void *rk_graph_end(void *stream) {
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    rk_set_error("hipStreamEndCapture: %s", hipGetErrorString(e));
    return nullptr;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    rk_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return nullptr;
  }
  // move the executable graph to the device now, not inside its first (timed) launch
  (void)hipGraphUpload(exec, (hipStream_t)stream);
  (void)hipGetLastError();
  return (void *)exec;
}

// Can a timing event be recorded as an event-record NODE of a stream capture
// (hipEventRecordWithFlags(..., hipEventRecordExternal))?  Works on the ROCm 7.2 runtime
// (tools/probes/graph_event_probe.hip), returns "invalid argument" on the 7.0 runtime PyTorch
// bundles: probed once on a scratch stream, so that callers can fall back to eager brackets.
static int timing_route();
extern "C" int32_t rk_graph_timing_supported(void) { return timing_route() != 0 ? 1 : 0; }

// 1: hipEventRecordExternal inside a capture; 2: explicit event-record nodes (capture_record_node,
// below); 0: neither -- probed once
static int external_flag_ok() {
  static const int ok = [] {
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    hipGraph_t g = nullptr;
    int good = 0;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e) == hipSuccess &&
        hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
      good = hipEventRecordWithFlags(e, s, hipEventRecordExternal) == hipSuccess;
      (void)hipStreamEndCapture(s, &g);
    }
    if (g) (void)hipGraphDestroy(g);
    if (e) (void)hipEventDestroy(e);
    if (s) (void)hipStreamDestroy(s);
    (void)hipGetLastError();
    return good;
  }();
  return ok;
}

extern "C" float rk_graph_event_node_probe(void);
static int timing_route() {
  if (external_flag_ok()) return 1;
  if (rk_tune_get(RK_TUNE_GRAPH_EVENT_NODES) != 1) return 0;
  static const int route = [] {
    // (opt-in, rk_tune(RK_TUNE_GRAPH_EVENT_NODES, 1): measured on the 7.0 runtime the bracketed group costs the same
    // replayed with event nodes as enqueued eagerly -- 0.1385-0.1403 vs 0.1367-0.1385 ms per step of a
    // 20-step run -- and its intervals read 1.5-5 us longer than rocprofv3's kernel durations, where
    // the eager brackets agree with them)
    return rk_graph_event_node_probe() > 0.f ? 2 : 0;
  }();
  return route;
}

// Add an event-record node for `e` at the current point of the capture on `s` through the explicit
// graph API (capture info -> hipGraphAddEventRecordNode on the capturing graph -> the node becomes
// the stream's capture dependency): the route that does not need hipEventRecordExternal.
static hipError_t capture_record_node(hipEvent_t e, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t *deps = nullptr;
  size_t n_deps = 0;
  hipError_t rc = hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n_deps);
  if (rc != hipSuccess) return rc;
  if (st != hipStreamCaptureStatusActive || g == nullptr) return hipErrorInvalidValue;
  hipGraphNode_t node = nullptr;
  rc = hipGraphAddEventRecordNode(&node, g, deps, n_deps, e);
  if (rc != hipSuccess) return rc;
  return hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
}

// Probe of that route in THIS process (its HIP runtime): two event-record nodes around a 64 MB
// memset inside a captured graph, replayed twice; returns the interval in ms read from the events
// (> 0: usable), or -(step that failed) - 0.001 * hip error code.
extern "C" float rk_graph_event_node_probe(void) {
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  void *buf = nullptr;
  float out = 0.f;
  hipError_t rc = hipSuccess;
  int step = 0;
#define PSTEP(call) do { ++step; rc = (call); if (rc != hipSuccess) goto done; } while (0)
  PSTEP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  PSTEP(hipEventCreate(&e0));
  PSTEP(hipEventCreate(&e1));
  PSTEP(hipMalloc(&buf, 64 << 20));
  PSTEP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  PSTEP(hipMemsetAsync(buf, 0, 4096, s));
  PSTEP(capture_record_node(e0, s));
  PSTEP(hipMemsetAsync(buf, 1, 64 << 20, s));
  PSTEP(capture_record_node(e1, s));
  PSTEP(hipMemsetAsync(buf, 2, 4096, s));
  PSTEP(hipStreamEndCapture(s, &g));
  PSTEP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  PSTEP(hipGraphLaunch(ex, s));
  PSTEP(hipGraphLaunch(ex, s));
  PSTEP(hipStreamSynchronize(s));
  PSTEP(hipEventElapsedTime(&out, e0, e1));
#undef PSTEP
done:
  if (rc != hipSuccess) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive) {
      hipGraph_t junk = nullptr;
      (void)hipStreamEndCapture(s, &junk);
      if (junk) (void)hipGraphDestroy(junk);
    }
    out = -(float)step - 0.001f * (float)rc;
  }
  if (ex) (void)hipGraphExecDestroy(ex);
  if (g) (void)hipGraphDestroy(g);
  if (buf) (void)hipFree(buf);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamDestroy(s);
  (void)hipGetLastError();
  return out;
}

extern "C" int rk_graph_launch(void *graph_exec, void *stream) {
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    rk_set_error("hipGraphLaunch: %s", hipGetErrorString(e));
    return -1;
  }
  return 0;
}

extern "C" void rk_graph_destroy(void *graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

namespace {

// A timing event recorded while its stream is being CAPTURED becomes an event-record NODE of the
// graph (hipEventRecordExternal): every replay re-records it, and hipEventElapsedTime between two
// of them reads the interval inside the replayed graph (tools/probes/graph_event_probe.hip) -- so a
// bracketed group of steps can be a graph replay like every other one.
inline void timer_record(hipEvent_t e, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t rc;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive)
    // (the HIP 7.0 runtime PyTorch bundles refuses the external flag but runs event-record nodes
    // added through the graph API: rk_graph_event_node_probe)
    rc = timing_route() == 2 ? capture_record_node(e, s) : hipEventRecordWithFlags(e, s, hipEventRecordExternal);
  else
    rc = hipEventRecord(e, s);
  (void)rc;
}

struct Timer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t s;
  Timer(const rk_ae_step_t *a, int id, void *stream) : s((hipStream_t)stream) {
    if (a->time_entry == id && a->time_ev0 && a->time_ev1) {
      e0 = (hipEvent_t)a->time_ev0; e1 = (hipEvent_t)a->time_ev1;
    } else if (a->time_entry == RK_ENTRY_ALL && a->time_all) {
      e0 = (hipEvent_t)a->time_all[2 * id]; e1 = (hipEvent_t)a->time_all[2 * id + 1];
    }
    if (e0) timer_record(e0, s);
  }
  ~Timer() {
    if (e1) timer_record(e1, s);
  }
};

// |act(x)| <= 1: the static split scale of Z is always in range
inline bool act_bounded(int act) { return act == RK_ACT_TANH || act == RK_ACT_SIGMOID; }

// one rk_adam_multi job for a [n_rows, h] table (dense Adam through pos, or SparseAdam
// on the block's item rows) or a flat tensor (pos == items == null)
rk_adam_job_t table_job(const rk_adam_param_t &par, const rk_block_t *blk, int n_rows, int h,
                        const float *G, bool table) {
  rk_adam_job_t j = {};
  j.par = par;
  j.n_rows = n_rows; j.h = h; j.g = G; j.g_parts = 1;
  if (table && par.sparse) {
    j.rows = blk->items; j.n_dev = blk->counts; j.n_cap = blk->n_cap;
  } else {
    j.par.sparse = 0;
    if (table) j.pos = blk->pos;
  }
  return j;
}

// dW = dO^T . Z: the bf16-pipe kernel (dw3.hip) when the split contractions are on and the step
// has a workspace, else the fp32-MFMA tiles.  G_de == NULL (dw3 only): the K slabs stay in the
// workspace for rk_adam_multi.
int dw_call(const rk_ae_step_t *a, float *G_de, float *gb_de, bool have_planes,
            float *ws = nullptr, void *stream = nullptr) {
  ws = ws ? ws : a->ws;
  stream = stream ? stream : a->stream;
  if (rk_gemm_split16() && ws) {
    // fp16 pairs (three products) by default; the Z^T planes come from the encoder forward when it
    // could write them (bounded activation: static scale), else they are made from Z with the
    // bound rk_amax left in ranges
    if (rk_dw_pairs())
      return rk_decode_bwd_dw2(a->dO, a->Z0, a->B, a->h, a->blk, G_de, gb_de, ws,
                               have_planes ? a->zt_planes : nullptr, a->ranges, stream);
    return rk_decode_bwd_dw3(a->dO, a->Z0, a->B, a->h, a->blk, G_de, gb_de, ws,
                             have_planes ? a->zt_planes : nullptr, stream);
  }
  return rk_decode_bwd_dw(a->dO, a->Z0, a->B, a->h, a->blk, G_de, gb_de, stream);
}

}  // namespace

#define RK_TRY(call)          \
  do {                        \
    int rc__ = (call);        \
    if (rc__ != 0) return rc__; \
  } while (0)

// Item-parallel step (rk_ae_step_t.own_world ranks, item i owned by rank i % own_world).
// The block holds the rows of ALL users of the global batch restricted to this rank's
// items; three segments with an all-reduce (SUM) of a [B, h] matrix between them:
//   IP_ENC : Z0 = partial encoder sums over the local items (whole-row norms)
//   -- all-reduce Z0 --
//   IP_MID : Z0 = act(Z0 + b_en) ; decode + loss over the local items ;
//            dZ0 = (dO . W_de) * act'(Z0)   (this rank's share of the pre-activation gradient)
//   -- all-reduce dZ0 --
//   IP_TAIL: dW || encoder backward ; Adam on the OWNED rows (+ b_en, which is
//            replicated: every rank computes the identical update) ; local loss partial
static int step_item_parallel(const rk_ae_step_t *a, int phase) {
  const rk_block_t *blk = a->blk;
  hipStream_t sm = (hipStream_t)a->stream;
  const int B = a->B, h = a->h, n_items = blk->n_items;
  const int row_tiles = rk_cdiv(B, rk_decode_row_tile());
  const bool mnll = a->loss_kind == RK_LOSS_MNLL;
  const float *W_de = a->tied ? a->par[RK_PAR_W_EN].p : a->par[RK_PAR_W_DE].p;
  float *G_en = a->tied ? a->G_de : a->G_en;
  const int n_part = mnll ? B : rk_loss_partials(B, blk->n_cap);
  const bool planes = false;     // (the partial encoder sums are not Z yet)
  RK_REQUIRE(a->own_world >= 1 && a->own_rank >= 0 && a->own_rank < a->own_world, "bad ownership");
  // the softmax of the multinomial loss spans every target item of a row, i.e. all ranks
  RK_REQUIRE(!mnll, "item-parallel training supports the mse / logistic losses");
  if (phase & RK_STEP_IP_ENC) {
    Timer t(a, RK_ENTRY_ENCODE_FWD, sm);
    RK_TRY(rk_ae_encode_fwd_partial(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, h, a->keep, a->noise_p,
                                    a->seed, a->rng_step, a->users, a->user_norm, a->Z0, sm));
  }
  if (phase & RK_STEP_IP_MID) {
    RK_TRY(rk_bias_act(a->Z0, a->par[RK_PAR_B_EN].p, B, h, a->act, sm));
    {
      Timer t(a, RK_ENTRY_DECODE_LOSS, sm);
      if (a->ranges && !act_bounded(a->act)) RK_TRY(rk_amax(a->Z0, (int64_t)B * h, a->ranges, sm));
      RK_TRY(rk_decode_loss(a->Z0, B, h, blk, a->row_off, W_de, a->par[RK_PAR_B_DE].p, a->loss_kind,
                            a->confidence, a->inv_B, a->dO, 0, a->loss_part, a->gb_part, a->ranges, sm));
      if (mnll) RK_TRY(rk_mnll_finish(a->dO, B, blk, a->row_off, a->inv_B, nullptr, nullptr, nullptr, a->loss_part, sm));
    }
    // act'(Z0) is applied to this rank's PARTIAL dZ0 here, inside the split-K reduce: Z0 is the
    // same on every rank, so sum_ranks(dZ0_r) * act'(Z0) = sum_ranks(dZ0_r * act'(Z0)) and the
    // element-wise launch after the all-reduce disappears
    Timer t(a, RK_ENTRY_DECODE_BWD_DZ, sm);
    RK_TRY(rk_decode_bwd_dz(a->dO, B, h, blk, W_de, a->Z0, a->act, a->dZ0, a->ws, a->ranges, sm));
  }
  const int dw_slabs = (a->tied || mnll || a->ws == nullptr) ? 1 : rk_dw_splits(B);
  if (phase & RK_STEP_IP_TAIL) {
    if (a->tied || mnll) {
      RK_TRY(dw_call(a, a->G_de, mnll ? a->gb_de : nullptr, planes));
      RK_TRY(rk_ae_encode_bwd(blk, a->row_off, B, a->dZ0, h, G_en, a->tied ? 1 : 0, a->gb_en, sm));
    } else {
      // large global batches: dW comes out as K slabs; rk_adam_multi sums them in slab order
      // while it reads the gradient (g_parts), so the separate summing launch is skipped
      Timer t(a, RK_ENTRY_DECODE_BWD_DW, sm);
      RK_TRY(rk_decode_bwd_dw_encode_bwd(a->dO, a->Z0, B, h, blk, dw_slabs > 1 ? nullptr : a->G_de,
                                         a->row_off, a->dZ0, G_en, a->gb_en, a->ws, sm));
    }
    rk_adam_job_t jobs[4];
    int n = 0;
    jobs[n++] = table_job(a->par[RK_PAR_W_EN], blk, n_items, h, G_en, true);
    if (a->tied && a->ranges) jobs[0].amax_out = a->ranges + 64;     // the decoder reads this table
    if (!a->tied) {
      jobs[n] = table_job(a->par[RK_PAR_W_DE], blk, n_items, h, a->G_de, true);
      if (a->ranges) jobs[n].amax_out = a->ranges + 64;
      if (dw_slabs > 1) { jobs[n].g = a->ws; jobs[n].g_parts = dw_slabs; jobs[n].g_stride = blk->n_cap * h; }
      ++n;
    }
    jobs[n] = table_job(a->par[RK_PAR_B_DE], blk, n_items, 1, a->gb_de, true);
    jobs[n].par.sparse = 0; jobs[n].rows = nullptr; jobs[n].n_dev = nullptr; jobs[n].pos = blk->pos;
    if (!mnll) {
      jobs[n].g = a->gb_part; jobs[n].g_parts = row_tiles; jobs[n].gstride_dev = blk->counts + 2;
    }
    ++n;
    if (!(a->tied || mnll)) {        // the fused launch wrote G_en in row segments
      jobs[0].g_parts = rk_encode_bwd_segments(B); jobs[0].g_stride = blk->n_cap * h;
    }
    for (int j = 0; j < n; ++j)
      if (!jobs[j].par.sparse) { jobs[j].row0 = a->own_rank; jobs[j].row_step = a->own_world; }
    jobs[n] = table_job(a->par[RK_PAR_B_EN], blk, 1, h, a->gb_en, false);
    if (!(a->tied || mnll)) { jobs[n].g_parts = rk_encode_bwd_segments(B); jobs[n].g_stride = h; }
    ++n;
    Timer t(a, RK_ENTRY_ADAM_MULTI, sm);
    RK_TRY(rk_adam_multi(jobs, n, a->loss_part, n_part, a->denom, a->loss_out, sm));
  }
  return 0;
}

// Does this (whole, single-process) step run its contractions on the pipelined pair-plane kernels
// (csrc/pgemm.h)?  Untied MSE / BCE steps with operand planes and a scale table, outside the fused
// decode + dZ launch's domain (h > 256 or >= 1024 rows).
// Returns 0: neither; 1: all three contractions on csrc/pgemm.h; 3: the register-resident fused decode
// (csrc/fdecode.hip) with its image, dW on rk_pg_dw  (4, round 5's streaming form of that decode, is gone)
static bool step_dw_ones(const rk_ae_step_t *a, const int pg_mode);
// the merged dW || encoder-backward launch of csrc/pgemm.hip covers this row window (rk_dw_encode_bwd_fused_ok also asks for
// the fp16-pair dW of dw3.hip, which plain bf16 operands -- RK_GEMM_PREC=bf16 -- do not have: the pgemm tiles do not care)
static bool pg_merged_ok(int row_off, int B) {
  if (rk_dw_encode_bwd_fused_ok(row_off, B)) return true;
  return rk_gemm_plain_bf16() && rk_tune_get(RK_TUNE_DW_ENC_FUSED) == 1 && (((row_off + B + 31) >> 5) - (row_off >> 5) <= 64);
}
static int step_pg_mode(const rk_ae_step_t *a) {
  const int phase = a->phase == 0 ? RK_STEP_ALL : a->phase;
  if (a->tied || a->loss_kind == RK_LOSS_MNLL) return 0;
  if (!rk_gemm_split16() || a->ws == nullptr || a->planes == nullptr) return 0;
  if (a->do_scales == nullptr || !rk_pg_enabled()) return 0;
  if (rk_gemm_plain_bf16()) {
    // RK_GEMM_PREC=bf16 on the current family (round 6): whole single-process steps in the fused decode's domain whose
    // decoder bias gradient comes out of the dW tiles (step_dw_ones: the image's column-sum range reads fp16 pairs) --
    // fdec_kernel<.., PLAIN> + dw_encbwd_kernel<64, 128, .., PLAIN>; everything else keeps the round-2/3 plain kernels
    if (phase == RK_STEP_ALL && a->B < 1024 && a->ws_dw != nullptr && rk_fdec_ok(a->B, a->h, a->blk->n_cap, a->loss_kind) &&
        pg_merged_ok(a->row_off, a->B) && rkp::kp_of(a->h) <= 256 && step_dw_ones(a, 3)) return 3;
    return 0;
  }
  if (phase != RK_STEP_ALL) {
    // the PHASED (data-parallel) step: the register-resident fused decode in FWD_DW, dW as one dense array +
    // the bias gradient from its image (rk_pg_dw_dense), the slab reduce + encoder backward in DZ_ENC.  The
    // same answer in all three calls of a step (it depends on nothing a phase changes).
    if ((phase & RK_STEP_IP_ALL) == 0 && a->ws_dw != nullptr && rk_fdec_ok(a->B, a->h, a->blk->n_cap, a->loss_kind) &&
        rk_decode_dz_fused_ok(a->B, a->h, a->blk->n_cap, a->loss_kind))
      return 3;
    return 0;
  }
  if (rk_decode_dz_fused_ok(a->B, a->h, a->blk->n_cap, a->loss_kind) == 0) return 1;
  // 3: the register-resident fused decode (csrc/fdecode.hip) + rk_pg_dw || encoder backward || image
  // column sums -- when the dW / encoder-backward launch can be the fused one (a->ws_dw, the row window)
  if (a->ws_dw != nullptr && rk_fdec_ok(a->B, a->h, a->blk->n_cap, a->loss_kind) &&
      rk_dw_encode_bwd_fused_ok(a->row_off, a->B)) return 3;
  // (a third form -- the LDS-staged fused decode of decode16.hip writing the image as well -- was measured
  // and dropped: at C2 the image pass cost that launch 6 us and rk_pg_dw gained 0.8: 0.1222 vs 0.1188 ms)
  return 0;
}
// the decoder bias gradient as output column h of the dW tiles (a ones column in the Z image's padding, written
// by this call's encoder forward; csrc/internal.h): whole steps on the fused decode with the merged dW || encoder
// backward launch, h % 32 != 0, and room for one bias slab per K slab in gb_part (the fused decode leaves it unused)
static bool step_dw_ones(const rk_ae_step_t *a, const int pg_mode) {
  const int phase = a->phase == 0 ? RK_STEP_ALL : a->phase;
  if (phase != RK_STEP_ALL || pg_mode != 3 || !act_bounded(a->act) || a->gb_part == nullptr) return false;
  const rk_block_t *blk = a->blk;
  const int B = a->B, h = a->h;
  const bool dz_fused = rk_decode_dz_fused_ok(B, h, blk->n_cap, a->loss_kind) != 0 || pg_mode == 3;
  if (!dz_fused || !pg_merged_ok(a->row_off, B)) return false;      // (the merged dW || encoder backward launch)
  const int row_tiles = rk_cdiv(B, rk_decode_row_tile());
  return rk_tune_get(RK_TUNE_DW_ONES) != 0 && rk_pg_dw_ones_ok(B, h, blk->n_cap) != 0 &&
         (int64_t)row_tiles * (rk_cdiv(blk->n_cap, 32) * 32) >= (int64_t)rk_pg_dw_splits(B, h, blk->n_cap) * blk->n_cap;
}
extern "C" int32_t rk_ae_step_uses_pg(const rk_ae_step_t *a) {
  if (!(a && a->blk)) return 0;
  const int mode = step_pg_mode(a);
  return mode | (step_dw_ones(a, mode) ? 16 : 0);
}

// The whole step is a serial chain on ONE stream:
//   encode_fwd ; decode+loss ; dW ; dZ split-K ; reduce ; encode_bwd (+gb_en) ; update
// (an earlier version ran the dW chain on a second stream: each cross-stream event
// costs 10-20 us of dependency latency on this stack -- tools/sync_cost2.py -- which
// ate the overlap; the small kernels it needed are folded into the big ones instead).
extern "C" int rk_ae_train_step(const rk_ae_step_t *a) {
  RK_REQUIRE(a && a->blk, "null step / block");
  RK_REQUIRE(a->phase >= 0 && a->phase <= 63, "phase is a mask of RK_STEP_*");
  RK_REQUIRE(a->time_entry >= RK_ENTRY_ALL && a->time_entry < RK_ENTRY_COUNT, "time_entry");
  if (a->phase & RK_STEP_IP_ALL) {
    RK_REQUIRE((a->phase & RK_STEP_ALL) == 0, "item-parallel and data-parallel phases do not mix");
    return step_item_parallel(a, a->phase);
  }
  const int phase = a->phase == 0 ? RK_STEP_ALL : a->phase;
  const bool whole = phase == RK_STEP_ALL;
  const rk_block_t *blk = a->blk;
  hipStream_t sm = (hipStream_t)a->stream;
  const int B = a->B, h = a->h, n_items = blk->n_items;
  const int row_tiles = rk_cdiv(B, rk_decode_row_tile());
  const bool mnll = a->loss_kind == RK_LOSS_MNLL;
  const float *W_de = a->tied ? a->par[RK_PAR_W_EN].p : a->par[RK_PAR_W_DE].p;
  float *G_en = a->tied ? a->G_de : a->G_en;
  const int n_part = mnll ? B : rk_loss_partials(B, blk->n_cap);   // unused slots hold 0
  // whole untied MSE / BCE steps on the 16-bit pipe: dW as its own bf16-pipe launch (slabs summed
  // by the Adam sweep) + the plain encoder backward; otherwise the fused fp32 dW || encoder launch
  const bool dw3 = whole && !(a->tied || mnll) && rk_gemm_split16() && a->ws != nullptr;
  // the encoder forward writes the Z^T planes of the bf16-pipe dW kernel along with Z
  const bool planes = rk_gemm_split16() && a->ws != nullptr && a->zt_planes != nullptr &&
                      (phase & RK_STEP_FWD_DW) != 0 && (!rk_dw_pairs() || act_bounded(a->act));
  // dW on a stream of its own next to the dZ -> encoder-backward chain (rk_ae_step_t.dw_stream)
  const bool dw_branch = dw3 && a->dw_stream != nullptr;
  RK_REQUIRE(!dw_branch || (a->ws_dw && a->dw_fork && a->dw_join), "dw_stream needs ws_dw, dw_fork, dw_join");

  // pre-split operand planes (decode16.hip): W_de[items] is split by extra workgroups of the
  // encoder-forward launch, Z by that kernel's epilogue; decode and dZ copy the images into LDS
  const bool pl = a->planes != nullptr && rk_gemm_split16() != 0;
  // dZ fused into the decode launch (decode16.hip DZT): the phases that hold both halves and leave the
  // dZ workspace alone in between -- the untied MSE / BCE step whose dW has a workspace of its own
  // Phased (data-parallel) steps: the slabs must survive from the FWD_DW call to the DZ_ENC call, so
  // the in-line dW of the first takes the workspace of its own (ws_dw) -- the same predicate in both calls.
  const bool both = (phase & RK_STEP_FWD_DW) && (phase & RK_STEP_DZ_ENC);
  const int pg_mode = step_pg_mode(a);
  const bool dz_fused = pl && !a->tied && !mnll && a->ws != nullptr && (both ? whole : a->ws_dw != nullptr) &&
                        (phase & (RK_STEP_FWD_DW | RK_STEP_DZ_ENC)) != 0 &&
                        (rk_decode_dz_fused_ok(B, h, blk->n_cap, a->loss_kind) != 0 || pg_mode == 3);
  // dW and the encoder backward as ONE launch on the chain instead of a side-stream branch
  // (in the small-shape domain of the fused decode only: at C5's sizes -- dW 100+ us -- the side-stream
  // branch next to dZ -> encoder backward is worth more than its two edges: 0.75 vs 0.83 ms per step)
  const bool dw_enc_fused = dw3 && dz_fused && (rk_dw_encode_bwd_fused_ok(a->row_off, B) != 0 || (pg_mode == 3 && pg_merged_ok(a->row_off, B)));
  // the three contractions on the pipelined pair-plane kernels (csrc/pgemm.h; include/recoder_hip.h
  // rk_ae_step_t.do_scales): whole untied MSE / BCE steps outside the fused decode's domain
  const bool pg = pg_mode == 1, fdec = pg_mode == 3;
  const float *dw_slabs_pg = dw_branch ? a->ws_dw : a->ws;
  const bool dw_ones = dw_enc_fused && step_dw_ones(a, pg_mode);
  if (phase & RK_STEP_FWD_DW) {
    {
      Timer t(a, RK_ENTRY_ENCODE_FWD, sm);
      if (pl) {
        rk_enc_split_t es = {};
        es.sw = rk_split_w_args(W_de, blk, a->ranges, a->planes);
        if (pg || fdec) es.sw.wtp = nullptr;  // (dZ reads the W image along its rows: no W^T image)
        es.n_split = rk_cdiv(blk->n_cap, 32);
        es.zimg = act_bounded(a->act) ? (char *)a->planes->z : nullptr;
        es.z_kt = rkp::kp_of(h) / 32;
        es.z_ones = dw_ones ? 1 : 0;
        RK_REQUIRE(a->planes->h == h && B <= a->planes->B_cap && blk->n_cap <= a->planes->n_cap,
                   "planes were laid out for another shape");
        RK_TRY(rk_ae_encode_fwd_at(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, a->par[RK_PAR_B_EN].p, h,
                                   a->keep, a->noise_p, a->seed, a->cursor, a->cursor_off, a->users,
                                   a->act, a->Z0, (planes && !pg && !fdec) ? a->zt_planes : nullptr, sm, &es,
                                   a->rng_step));
      } else if (a->cursor)
        RK_TRY(rk_ae_encode_fwd_at(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, a->par[RK_PAR_B_EN].p, h,
                                   a->keep, a->noise_p, a->seed, a->cursor, a->cursor_off, a->users,
                                   a->act, a->Z0, planes ? a->zt_planes : nullptr, sm, nullptr, 0));
      else if (planes)
        RK_TRY(rk_ae_encode_fwd_planes(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, a->par[RK_PAR_B_EN].p,
                                       h, a->keep, a->noise_p, a->seed, a->rng_step, a->users, a->act,
                                       a->Z0, a->zt_planes, sm));
      else
        RK_TRY(rk_ae_encode_fwd(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, a->par[RK_PAR_B_EN].p, h,
                                a->keep, a->noise_p, a->seed, a->rng_step, a->users, a->act, a->Z0, sm));
    }
    {
      Timer t(a, RK_ENTRY_DECODE_LOSS, sm);
      if (a->ranges && !act_bounded(a->act)) RK_TRY(rk_amax(a->Z0, (int64_t)B * h, a->ranges, sm));
      if (pl) {
        // (unbounded activations: the split scale of Z needs its maximum first)
        if (!act_bounded(a->act)) RK_TRY(rk_split_z(a->Z0, B, h, a->ranges, a->planes, sm));
        if (dz_fused && fdec)
          RK_TRY(rk_fdec_loss_dz(a->planes, B, blk, a->row_off, a->par[RK_PAR_B_DE].p, a->loss_kind, a->confidence,
                                 a->inv_B, a->dO, a->do_rows, a->do_scales, a->loss_part, a->ws, sm));
        else if (dz_fused)
          RK_TRY(rk_decode_loss_dz_planes(a->planes, B, blk, a->row_off, a->par[RK_PAR_B_DE].p, a->loss_kind,
                                          a->confidence, a->inv_B, a->dO, a->loss_part, a->gb_part, a->ws, sm));
        else if (pg)
          RK_TRY(rk_pg_decode_loss(a->planes, B, blk, a->row_off, a->par[RK_PAR_B_DE].p, a->loss_kind,
                                   a->confidence, a->inv_B, a->dO, a->do_rows, a->do_scales, nullptr,
                                   a->loss_part, a->gb_part, sm));
        else
          RK_TRY(rk_decode_loss_planes(a->planes, B, blk, a->row_off, a->par[RK_PAR_B_DE].p, a->loss_kind,
                                       a->confidence, a->inv_B, a->dO, 0, a->loss_part, a->gb_part, sm));
      } else {
        RK_TRY(rk_decode_loss(a->Z0, B, h, blk, a->row_off, W_de, a->par[RK_PAR_B_DE].p, a->loss_kind,
                              a->confidence, a->inv_B, a->dO, 0, a->loss_part, a->gb_part, a->ranges, sm));
      }
      if (mnll) RK_TRY(rk_mnll_finish(a->dO, B, blk, a->row_off, a->inv_B, nullptr, nullptr, nullptr, a->loss_part, sm));
    }
    // dO and the Z^T planes are ready: the dW branch starts here -- and so does whatever else the
    // caller queues behind this event (graph.GraphStepper: the look-ahead collation, which must not
    // run next to the decode: its workgroups do not fit into the LDS the fused decode leaves)
    if (dw_branch) RK_TRY(rk_event_record(a->dw_fork, sm));
    // dW: on its own (tied weights: the encoder backward accumulates onto its rows;
    // MNLL: + column sums of dO; data parallel: G_de must travel early), otherwise
    // fused with the encoder backward below
    if (a->tied || mnll || !whole) {
      Timer t(a, RK_ENTRY_DECODE_BWD_DW, sm);
      // (the fused decode's dZ slabs sit in a->ws until the DZ_ENC call: dW works in ws_dw then)
      if (fdec)    // (phased, on the fused decode's image: G_de dense, gb_de from the image's columns)
        RK_TRY(rk_pg_dw_dense(a->dO, a->do_scales, 32, 64, B, a->planes, blk, a->G_de, a->gb_de, sm));
      else
        RK_TRY(dw_call(a, a->G_de, mnll ? a->gb_de : nullptr, planes, dz_fused ? a->ws_dw : nullptr));
    }
    if (!whole) {
      // the data-parallel exchange needs gb_de and the loss scalar as arrays of their own
      if (!mnll && !fdec) RK_TRY(rk_colsum(a->gb_part, row_tiles, blk->n_cap, 0, blk->counts, a->gb_de, sm));
      RK_TRY(rk_loss_reduce(a->loss_part, n_part, a->denom, a->loss_out, sm));
    }
  }
  if (phase & RK_STEP_DZ_ENC) {
    {
      Timer t(a, RK_ENTRY_DECODE_BWD_DZ, sm);
      if (dz_fused && fdec)   // (the fused decode left one slab per group of column tiles in the workspace)
        RK_TRY(rk_fdec_dz_reduce(a->ws, B, h, blk, a->Z0, a->act, a->dZ0, sm));
      else if (dz_fused)   // (the decode launch left the column tiles' partials in the workspace)
        RK_TRY(rk_decode_dz_reduce(a->ws, B, h, blk, a->Z0, a->act, a->dZ0, sm));
      else if (pg)
        RK_TRY(rk_pg_dz(a->dO, a->do_scales, 64, 32, B, a->planes, blk, a->Z0, a->act, a->dZ0, a->ws, sm));
      else if (pl)
        RK_TRY(rk_decode_bwd_dz_planes(a->dO, B, a->planes, blk, a->Z0, a->act, a->dZ0, a->ws, sm));
      else
        RK_TRY(rk_decode_bwd_dz(a->dO, B, h, blk, W_de, a->Z0, a->act, a->dZ0, a->ws, a->ranges, sm));
    }
    if (a->tied || mnll || !whole) {
      Timer t(a, RK_ENTRY_ENCODE_BWD, sm);
      RK_TRY(rk_ae_encode_bwd(blk, a->row_off, B, a->dZ0, h, G_en, a->tied ? 1 : 0, a->gb_en, sm));
    } else if (dw_enc_fused) {
      // dW || encoder backward in ONE launch on the chain (dw3.hip dw_encbwd_kernel): no side stream
      Timer t(a, RK_ENTRY_DECODE_BWD_DW, sm);
      if (fdec && dw_ones)     // (the bias gradient as the dW tiles' output column h: no column-sum range)
        RK_TRY(rk_pg_dw_encode_bwd_ones(a->dO, a->do_scales, 32, 64, B, a->planes, blk, dw_branch ? a->ws_dw : a->ws,
                                        a->row_off, a->dZ0, G_en, a->gb_en, a->gb_part, sm));
      else if (fdec)
        RK_TRY(rk_pg_dw_encode_bwd(a->dO, a->do_scales, 32, 64, B, a->planes, blk, dw_branch ? a->ws_dw : a->ws,
                                   a->row_off, a->dZ0, G_en, a->gb_en, a->gb_de, sm));
      else
      RK_TRY(rk_decode_bwd_dw2_encode_bwd(a->dO, a->Z0, B, h, blk, dw_branch ? a->ws_dw : a->ws,
                                          planes ? a->zt_planes : nullptr, a->ranges, a->row_off, a->dZ0,
                                          G_en, a->gb_en, sm));
    } else if (dw_branch) {
      {
        Timer t(a, RK_ENTRY_ENCODE_BWD, sm);
        RK_TRY(rk_ae_encode_bwd(blk, a->row_off, B, a->dZ0, h, G_en, 0, a->gb_en, sm));
      }
      // (enqueued behind the chain: see rk_ae_step_t.dw_stream)
      RK_TRY(rk_stream_wait_event(a->dw_stream, a->dw_fork));
      {
        Timer t(a, RK_ENTRY_DECODE_BWD_DW, a->dw_stream);
        if (pg) RK_TRY(rk_pg_dw(a->dO, a->do_scales, 64, 32, B, a->planes, blk, a->ws_dw, nullptr, a->dw_stream));
        else if (fdec) RK_TRY(rk_pg_dw(a->dO, a->do_scales, 32, 64, B, a->planes, blk, a->ws_dw, a->gb_de, a->dw_stream));
        else RK_TRY(dw_call(a, nullptr, nullptr, planes, a->ws_dw, a->dw_stream));
      }
      RK_TRY(rk_event_record(a->dw_join, a->dw_stream));
      RK_TRY(rk_stream_wait_event(sm, a->dw_join));
    } else if (dw3) {
      // the dZ slabs in the workspace are consumed: the bf16-pipe dW takes it over (Z^T planes +
      // its own K slabs, which rk_adam_multi sums while it reads the gradient)
      {
        Timer t(a, RK_ENTRY_DECODE_BWD_DW, sm);
        if (pg) RK_TRY(rk_pg_dw(a->dO, a->do_scales, 64, 32, B, a->planes, blk, a->ws, nullptr, sm));
        else if (fdec) RK_TRY(rk_pg_dw(a->dO, a->do_scales, 32, 64, B, a->planes, blk, a->ws, a->gb_de, sm));
        else RK_TRY(dw_call(a, nullptr, nullptr, planes));
      }
      Timer t(a, RK_ENTRY_ENCODE_BWD, sm);
      RK_TRY(rk_ae_encode_bwd(blk, a->row_off, B, a->dZ0, h, G_en, 0, a->gb_en, sm));
    } else {
      Timer t(a, RK_ENTRY_DECODE_BWD_DW, sm);
      RK_TRY(rk_decode_bwd_dw_encode_bwd(a->dO, a->Z0, B, h, blk, a->G_de, a->row_off, a->dZ0, G_en,
                                         a->gb_en, a->ws, sm));
    }
  }
  if (phase & RK_STEP_UPDATE) {
    rk_adam_job_t jobs[4];
    int32_t slots[4];
    int n = 0;
    slots[n] = RK_PAR_W_EN;
    jobs[n++] = table_job(a->par[RK_PAR_W_EN], blk, n_items, h, G_en, true);
    if (whole && !(a->tied || mnll) && !dw3) {   // the fused launch wrote G_en in row segments
      jobs[0].g_parts = rk_encode_bwd_segments(B); jobs[0].g_stride = blk->n_cap * h;
    }
    if (a->tied && a->ranges) jobs[0].amax_out = a->ranges + 64;     // the decoder reads this table
    if (!a->tied) {
      jobs[n] = table_job(a->par[RK_PAR_W_DE], blk, n_items, h, a->G_de, true);
      if (a->ranges) jobs[n].amax_out = a->ranges + 64;
      if ((pg || fdec) && whole) {
        jobs[n].g = dw_slabs_pg; jobs[n].g_parts = rk_pg_dw_splits(B, h, blk->n_cap);
        jobs[n].g_stride = blk->n_cap * h; jobs[n].gparts_dev = blk->counts + 4;
      } else if (dw3) {
        jobs[n].g = rk_dw3_slabs(dw_branch ? a->ws_dw : a->ws, B, h); jobs[n].g_parts = rk_dw3_max_splits();
        jobs[n].g_stride = blk->n_cap * h; jobs[n].gparts_dev = blk->counts + 4;
      }
      slots[n] = RK_PAR_W_DE;
      ++n;
    }
    // sharded dense Adam (rk_ae_step_t.zero_lo): the table jobs cover this rank's rows only and read the
    // reduce-scattered dense gradient shard -- row r of the table at zero_g + (r - zero_lo) * h
    if (a->zero_hi > 0) {
      RK_REQUIRE(!whole && a->zero_lo >= 0 && a->zero_lo <= a->zero_hi && a->zero_hi <= n_items && a->zero_g_en,
                 "zero_lo / zero_hi: a row range of the tables, phased steps, with the gradient shards");
      for (int k = 0; k < n; ++k) {
        if (jobs[k].par.sparse) continue;
        const float *shard = slots[k] == RK_PAR_W_DE ? a->zero_g_de : a->zero_g_en;
        RK_REQUIRE(shard != nullptr, "zero_g_de");
        jobs[k].pos = nullptr; jobs[k].g_parts = 1; jobs[k].gparts_dev = nullptr;
        jobs[k].g = shard - (int64_t)a->zero_lo * h;
        jobs[k].row0 = a->zero_lo; jobs[k].row_step = 1; jobs[k].n_rows = a->zero_hi;
      }
    }
    // lazy dense Adam of the embedding tables (rk_adam_job_t.lazy_stamp): rows without a gradient that the next
    // step does not read are caught up later
    if (a->lazy_stamp_en) {
      // (whole steps, or the UPDATE call of a phased data-parallel step with the replicated update: every rank sweeps
      // the same rows of identical tables)
      RK_REQUIRE((whole || phase == RK_STEP_UPDATE) && a->cursor != nullptr && a->zero_hi == 0 && a->zero_gb_de == nullptr &&
                 a->lazy_period >= 1 && (a->tied || a->lazy_stamp_de),
                 "lazy Adam: replayed steps with the replicated update, both stamp arrays, lazy_period >= 1");
      for (int k = 0; k < n; ++k) {
        if (jobs[k].par.sparse) continue;
        jobs[k].lazy_stamp = slots[k] == RK_PAR_W_DE ? a->lazy_stamp_de : a->lazy_stamp_en;
        jobs[k].lazy_pos_next = a->lazy_pos_next;
        jobs[k].lazy_period = a->lazy_period;
        jobs[k].lazy_need_list = a->lazy_pos_next ? a->lazy_need_list : nullptr;
        jobs[k].lazy_need_count = a->lazy_pos_next ? a->lazy_need_count : nullptr;
      }
    }
    slots[n] = RK_PAR_B_DE;
    jobs[n] = table_job(a->par[RK_PAR_B_DE], blk, n_items, 1, a->gb_de, true);
    jobs[n].par.sparse = 0; jobs[n].rows = nullptr; jobs[n].n_dev = nullptr; jobs[n].pos = blk->pos;
    if (a->zero_gb_de) {           // (per-rank item sets: the summed gradient arrives laid out by item id)
      RK_REQUIRE(!whole, "zero_gb_de: phased steps");
      jobs[n].g = a->zero_gb_de; jobs[n].pos = nullptr;
    } else
    if (whole && !mnll && !fdec) { // straight from the decode epilogue's row-tile partials
      jobs[n].g = a->gb_part; jobs[n].g_parts = row_tiles; jobs[n].gstride_dev = blk->counts + 2;
    } else if (dw_ones) {          // one slab per K slab of the dW launch (its output column h: rk_pg_dw_encode_bwd_ones)
      jobs[n].g = a->gb_part; jobs[n].g_parts = rk_pg_dw_splits(B, h, blk->n_cap); jobs[n].g_stride = blk->n_cap;
      jobs[n].gparts_dev = blk->counts + 4;
    }                              // (fdec otherwise: gb_de itself, written by the dW launch's column-sum range)
    ++n;
    slots[n] = RK_PAR_B_EN;
    jobs[n] = table_job(a->par[RK_PAR_B_EN], blk, 1, h, a->gb_en, false);
    if (whole && !(a->tied || mnll) && !dw3) {
      jobs[n].g_parts = rk_encode_bwd_segments(B); jobs[n].g_stride = h;
    }
    ++n;
    RK_REQUIRE(a->cursor == nullptr || a->adam_table != nullptr,
               "graph replay needs the Adam constants table");
    // replayed PHASED steps (data parallel): the caller's exchange has left the global loss in a
    // scalar of its own; the update phase gets it as loss_part[0] (denom 1) and this launch's loss
    // block files it under the step's slot of loss_out and publishes the next cursor
    const bool with_loss = whole || a->cursor != nullptr;
    // the two bias jobs go FIRST in the grid (the launch dispatches its workgroups in order): a few
    // dozen workgroups with a long chain of dependent loads (pos -> 8 partial gradient rows), which
    // as the LAST ones dispatched were the tail of the whole sweep (43 vs 37 us in isolation)
    {
      rk_adam_job_t j2[4];
      int32_t s2[4];
      j2[0] = jobs[n - 2]; s2[0] = slots[n - 2];
      j2[1] = jobs[n - 1]; s2[1] = slots[n - 1];
      for (int k = 0; k < n - 2; ++k) { j2[2 + k] = jobs[k]; s2[2 + k] = slots[k]; }
      for (int k = 0; k < n; ++k) { jobs[k] = j2[k]; slots[k] = s2[k]; }
    }
    Timer t(a, RK_ENTRY_ADAM_MULTI, sm);
    RK_TRY(rk_adam_multi_at(jobs, n, with_loss ? a->loss_part : nullptr, whole ? n_part : 1,
                            whole ? a->denom : 1.0f, with_loss ? a->loss_out : nullptr, a->cursor,
                            a->cursor_off, a->adam_table,
                            RK_PAR_COUNT, slots, a->cursor_next, a->cursor_advance, sm));
  }
  return 0;
}
