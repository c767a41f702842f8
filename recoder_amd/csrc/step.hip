// rk_ae_train_step: one C-ABI call = one optimisation step of the
// DynamicAutoencoder hot path (reference model.py:383-404 for the common
// hidden_layers=[h] case): encode -> decode + loss -> {dW chain || dZ chain} ->
// Adam / SparseAdam.  It only sequences the kernels of this library on the two
// caller-provided HIP streams; keeping the sequencing in C removes ~35
// Python->ctypes transitions per step (the host enqueue time was approaching the
// GPU time of the step).
#include "common.h"

extern "C" void *rk_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
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
  if (hipEventElapsedTime(&ms, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) return -1.f;
  return ms;
}

namespace {

struct Timer {
  const rk_ae_step_t *a;
  int id;
  hipStream_t s;
  bool on;
  Timer(const rk_ae_step_t *a_, int id_, void *stream) : a(a_), id(id_), s((hipStream_t)stream) {
    on = (a->time_entry == id) && a->time_ev0 && a->time_ev1;
    if (on) (void)hipEventRecord((hipEvent_t)a->time_ev0, s);
  }
  ~Timer() {
    if (on) (void)hipEventRecord((hipEvent_t)a->time_ev1, s);
  }
};

int adam_param(const rk_ae_step_t *a, int k, int n_rows, int h, const int32_t *pos,
               const int32_t *items, const int32_t *counts, int n_cap, const float *G, bool table,
               void *stream) {
  const rk_adam_param_t &p = a->par[k];
  if (table && p.sparse)
    return rk_adam_rows(p.p, p.m, p.v, h, items, nullptr, counts, n_cap, G, p.lr, p.beta1, p.beta2,
                        p.eps, p.step, stream);
  if (table)
    return rk_adam_table(p.p, p.m, p.v, n_rows, h, pos, G, p.lr, p.beta1, p.beta2, p.eps,
                         p.weight_decay, p.step, stream);
  return rk_adam_dense(p.p, p.m, p.v, G, (int64_t)n_rows * h, p.lr, p.beta1, p.beta2, p.eps,
                       p.weight_decay, p.step, stream);
}

}  // namespace

#define RK_TRY(call)          \
  do {                        \
    int rc__ = (call);        \
    if (rc__ != 0) return rc__; \
  } while (0)

extern "C" int rk_ae_train_step(const rk_ae_step_t *a) {
  RK_REQUIRE(a && a->blk, "null step / block");
  RK_REQUIRE(a->stream_main != a->stream_aux, "the two streams must differ");
  RK_REQUIRE(a->ev_loss && a->ev_dz && a->ev_dw && a->ev_aux_done, "events missing");
  const rk_block_t *blk = a->blk;
  hipStream_t sm = (hipStream_t)a->stream_main, sa = (hipStream_t)a->stream_aux;
  const int B = a->B, h = a->h, n_items = blk->n_items;
  const int row_tiles = rk_cdiv(B, rk_decode_row_tile());
  const float *W_de = a->tied ? a->par[RK_PAR_W_EN].p : a->par[RK_PAR_W_DE].p;
  float *G_en = a->tied ? a->G_de : a->G_en;
  RK_REQUIRE(a->phase >= 0 && a->phase <= 2, "phase must be 0, 1 or 2");
  if (a->phase != 2) {
  // ---- forward: encoder SpMM, decoder GEMM + fused loss ----
  {
    Timer t(a, RK_ENTRY_ENCODE_FWD, sm);
    RK_TRY(rk_ae_encode_fwd(blk, a->row_off, B, a->par[RK_PAR_W_EN].p, a->par[RK_PAR_B_EN].p, h,
                            a->keep, a->noise_p, a->seed, a->rng_step, a->users, a->act, a->Z0, sm));
  }
  {
    Timer t(a, RK_ENTRY_DECODE_LOSS, sm);
    RK_TRY(rk_decode_loss(a->Z0, B, h, blk, a->row_off, W_de, a->par[RK_PAR_B_DE].p, a->loss_kind,
                          a->confidence, a->inv_B, a->dO, 0, a->loss_part, a->gb_part, sm));
  }
  int n_part = rk_loss_partials(B, blk->n_cap);   // all slots (unused ones hold 0)
  if (a->loss_kind == RK_LOSS_MNLL) {
    RK_TRY(rk_mnll_finish(a->dO, B, blk, a->row_off, a->inv_B, a->loss_part, sm));
    n_part = B;
  }
  if (hipEventRecord((hipEvent_t)a->ev_loss, sm) != hipSuccess) { rk_set_error("event record"); return -1; }
  if (hipStreamWaitEvent(sa, (hipEvent_t)a->ev_loss, 0) != hipSuccess) { rk_set_error("event wait"); return -1; }

  // ---- auxiliary stream: loss scalar, dW (+ decoder bias gradient) ----
  RK_TRY(rk_loss_reduce(a->loss_part, n_part, a->denom, a->loss_out, sa));
  if (a->loss_kind == RK_LOSS_MNLL) {
    Timer t(a, RK_ENTRY_DECODE_BWD_DW, sa);
    RK_TRY(rk_decode_bwd_dw(a->dO, a->Z0, B, h, blk, a->G_de, a->gb_de, sa));
  } else {
    RK_TRY(rk_colsum(a->gb_part, row_tiles, blk->n_cap, 0, blk->counts, a->gb_de, sa));
    Timer t(a, RK_ENTRY_DECODE_BWD_DW, sa);
    RK_TRY(rk_decode_bwd_dw(a->dO, a->Z0, B, h, blk, a->G_de, nullptr, sa));
  }
  (void)hipEventRecord((hipEvent_t)a->ev_dw, sa);

  // ---- main stream: dZ (x act') ----
  {
    Timer t(a, RK_ENTRY_DECODE_BWD_DZ, sm);
    RK_TRY(rk_decode_bwd_dz(a->dO, B, h, blk, W_de, a->Z0, a->act, a->dZ0, a->ws, sm));
  }
  (void)hipEventRecord((hipEvent_t)a->ev_dz, sm);

  // ---- main stream: encoder bias / row gradients ----
  RK_TRY(rk_colsum(a->dZ0, B, h, h, nullptr, a->gb_en, sm));
  if (a->tied) (void)hipStreamWaitEvent(sm, (hipEvent_t)a->ev_dw, 0);
  {
    Timer t(a, RK_ENTRY_ENCODE_BWD, sm);
    RK_TRY(rk_ae_encode_bwd(blk, a->row_off, B, a->dZ0, h, G_en, a->tied ? 1 : 0, sm));
  }
  }  // phase != 2
  if (a->phase == 1) return 0;

  // decoder-side Adam on the auxiliary stream; it writes W_de, so it follows dZ
  (void)hipStreamWaitEvent(sa, (hipEvent_t)a->ev_dz, 0);
  if (!a->tied) {
    Timer t(a, RK_ENTRY_ADAM_TABLE, sa);
    RK_TRY(adam_param(a, RK_PAR_W_DE, n_items, h, blk->pos, blk->items, blk->counts, blk->n_cap,
                      a->G_de, true, sa));
  }
  RK_TRY(rk_adam_table(a->par[RK_PAR_B_DE].p, a->par[RK_PAR_B_DE].m, a->par[RK_PAR_B_DE].v, n_items,
                       1, blk->pos, a->gb_de, a->par[RK_PAR_B_DE].lr, a->par[RK_PAR_B_DE].beta1,
                       a->par[RK_PAR_B_DE].beta2, a->par[RK_PAR_B_DE].eps,
                       a->par[RK_PAR_B_DE].weight_decay, a->par[RK_PAR_B_DE].step, sa));
  (void)hipEventRecord((hipEvent_t)a->ev_aux_done, sa);

  // ---- main stream: encoder-side Adam ----
  RK_TRY(adam_param(a, RK_PAR_W_EN, n_items, h, blk->pos, blk->items, blk->counts, blk->n_cap, G_en,
                    true, sm));
  RK_TRY(adam_param(a, RK_PAR_B_EN, 1, h, nullptr, nullptr, nullptr, 0, a->gb_en, false, sm));
  (void)hipStreamWaitEvent(sm, (hipEvent_t)a->ev_aux_done, 0);
  return 0;
}
