// Internal sizing helpers of librecoder_hip.so: not exported (the public form is rk_plan / rk_plan_t in
// include/recoder_hip.h, which calls every one of them).
#pragma once
#include <stdint.h>
struct rk_planes;
int rk_pg_dw_dense(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                   const struct rk_planes *pl, const rk_block_t *tgt, float *G_de, float *gb_de, void *stream);
// rk_pg_dw_encode_bwd for a Z image whose padding column h holds the constant 1 (h % 32 != 0; the encoder forward of
// the same rk_ae_train_step call wrote it: rk_enc_split_t.z_ones): no column-sum range -- the dW tiles' output
// column h IS the decoder bias gradient, one slab of it per K slab: gb_slabs[s * tgt->n_cap + c], s < counts[4]
int rk_pg_dw_encode_bwd_ones(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                             const struct rk_planes *pl, const rk_block_t *tgt, float *slabs, int32_t row_off,
                             const float *dZ0pre, float *G_en, float *gb_en, float *gb_slabs, void *stream);
int rk_pg_dw_ones_ok(int32_t B, int32_t h, int32_t n_cap);
int rk_splitk_reduce_tiles(const float *ws, int M, int N, const int32_t *Kdev, int max_splits, int tile_k,
                           const float *Zact, int act, float *out, void *stream_);
int rk_fdec_slabs(int B, int n_cap);
const float *rk_dw3_slabs(const void *workspace, int32_t B, int32_t h);   // K slabs of dw3 / dw2 inside their workspace
int32_t rk_encode_bwd_segments(int32_t B);                                // row segments of the fused fp32 dW || encoder backward
// gemm.hip: the fp32 tiles' dW + rk_ae_encode_bwd(accumulate = 0) in ONE launch (RK_GEMM_PREC=f32, item-parallel steps)
int rk_decode_bwd_dw_encode_bwd(const float *dO, const float *Z, int32_t B, int32_t h, const rk_block_t *blk, float *G_de,
                                int32_t row_off, const float *dZ0pre, float *G_en, float *gb_en, float *workspace,
                                void *stream);
extern "C" {
int64_t rk_dz_workspace_bytes(int32_t B, int32_t h);
int32_t rk_loss_partials(int32_t B, int32_t n_cap);
int32_t rk_decode_row_tile(void);
int64_t rk_dw_workspace_bytes(int32_t B, int32_t h, int32_t n_cap);
int32_t rk_dw_splits(int32_t B);
int64_t rk_dw3_workspace_bytes(int32_t B, int32_t h, int32_t n_cap);
int32_t rk_dw3_max_splits(void);
int32_t rk_dw_pairs(void);
int32_t rk_dw_encode_bwd_fused_ok(int32_t row_off, int32_t B);
int64_t rk_dw3_planes_bytes(int32_t B, int32_t h);
int32_t rk_dw3_rows_pad(int32_t B);
int32_t rk_dw3_cols_pad(int32_t h);
int32_t rk_gemm_split16(void);
int32_t rk_gemm_plain_bf16(void);
int64_t rk_planes_bytes(int32_t B_cap, int32_t h, int32_t n_cap);
int32_t rk_decode_dz_fused_ok(int32_t B, int32_t h, int32_t n_cap, int32_t loss_kind);
int64_t rk_dz_fused_workspace_bytes(int32_t B, int32_t h, int32_t n_cap);
int32_t rk_fdec_ok(int32_t B, int32_t h, int32_t n_cap, int32_t loss_kind);
int64_t rk_fdec_workspace_bytes(int32_t B, int32_t h, int32_t n_cap);
int32_t rk_pg_enabled(void);
void rk_pg_decode_granule(int32_t B, int32_t n_cap, int32_t *gr, int32_t *gc);
int64_t rk_pg_scale_floats(int32_t B_cap, int32_t n_cap);
int64_t rk_pg_mnll_workspace_floats(int32_t B, int32_t n_cap);
int64_t rk_pg_dz_workspace_bytes(int32_t B, int32_t h);
int32_t rk_pg_dw_splits(int32_t B, int32_t h, int32_t n_cap);
int64_t rk_pg_dw_workspace_bytes(int32_t B, int32_t h, int32_t n_cap);
int32_t rk_split_zt_ok(void);
int32_t rk_graph_timing_supported(void);
int32_t rk_topk_max_k(void);
int32_t rk_topk_pairs_max_cap(void);
int rk_ae_encode_fwd_planes(const rk_block_t *blk, int32_t row_off, int32_t B,
                            const float *W_en, const float *b_en, int32_t h,
                            const uint8_t *keep, float p, uint64_t seed,
                            uint64_t rng_step, const int64_t *users, int32_t act,
                            float *Z0, void *zt_planes, void *stream);
float rk_graph_event_node_probe(void);
void rk_gemm_probe(unsigned long long *buffer);
void rk_dw3_probe(unsigned long long *buffer);
void rk_enc_probe(unsigned long long *buffer);
void rk_planes_probe(unsigned long long *buffer);
int rk_split_w(const float *W_de, int32_t h, const rk_block_t *tgt, const int32_t *ranges,
               const rk_planes_t *pl, void *stream);
int rk_split_z(const float *Z, int32_t B, int32_t h, const int32_t *ranges, const rk_planes_t *pl,
               void *stream);
}
