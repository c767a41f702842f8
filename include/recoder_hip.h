/*
 * recoder_hip.h -- C ABI of librecoder_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the training hot path of amoussawi/recoder
 * (reference v0.4.0).  The reference has no FFI of its own: every entry point
 * below replaces a run of eager PyTorch ops at the cited reference file:line.
 * The binding a maintainer would add on the reference side is the ctypes
 * loader in recoder_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; rk_last_error() gives a
 *     thread-local message.
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr())
 *     unless the name ends in _host; nothing is retained past the call.
 *   - every launch goes on the caller's hipStream_t (passed as void*); no call
 *     synchronises the host; no call allocates.
 *   - "dev count": sizes that only exist on the device (n_b = |sampled item
 *     set|) are read by the kernels from rk_block_t.counts; grids are sized by
 *     the host-known capacities and surplus workgroups exit.
 *   - all matrices are row-major fp32; item/user embedding tables are
 *     [rows, h] exactly as torch.nn.Embedding.weight stores them.
 */
#ifndef RECODER_HIP_H
#define RECODER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* (the library is built with -fvisibility=hidden: what this header declares is what it exports) */
#pragma GCC visibility push(default)

/*
 * The sizing plan of one step shape: every workspace size, slab count, padding rule and "which fused form
 * applies" predicate of the entry points below, for (B rows, hidden size h, item capacity n_cap, loss) --
 * ONE call before the caller allocates (nothing here touches the device).  Fill the four inputs (row_off
 * only matters for dw_encode_bwd_fused_ok), call rk_plan, read the rest.  The comments of the entry points
 * name the field that sizes each buffer.
 */
typedef struct rk_plan {
  /* inputs */
  int32_t B, h, n_cap, loss_kind, row_off;
  /* process-wide configuration */
  int32_t gemm_split16;            /* decode / dZ / dW multiply fp16 hi+lo pairs (default) */
  int32_t gemm_plain_bf16;         /* RK_GEMM_PREC=bf16: plain bf16 operands (a separate data point) */
  int32_t dw_pairs;                /* dW on fp16 pairs (rk_decode_bwd_dw3) */
  int32_t split_zt_ok;             /* rk_split_wz's transposed Z planes (dw_workspace) apply */
  int32_t pg_enabled;              /* the pipelined pair-plane family (rk_pg_*) is on */
  int32_t graph_timing_supported;  /* time_ev0 / time_ev1 may be recorded inside a captured graph */
  int32_t decode_row_tile;         /* rows per decode tile: gb_part holds ceil(B / this) rows */
  int32_t dw3_max_splits, topk_max_k, topk_pairs_max_cap;
  /* sizes of this shape */
  int32_t loss_partials;           /* floats of loss_part */
  int32_t dw_splits;               /* slabs of rk_decode_bwd_dw */
  int32_t pg_dw_splits;            /* most slabs rk_pg_dw writes */
  int32_t dw3_rows_pad, dw3_cols_pad;
  int32_t pg_granule_rows, pg_granule_cols;   /* scale granule of rk_pg_decode_loss's image */
  int64_t planes_bytes, dz_workspace_bytes, dz_fused_workspace_bytes, dw_workspace_bytes,
          dw3_workspace_bytes, dw3_planes_bytes, fdec_workspace_bytes, pg_dz_workspace_bytes,
          pg_dw_workspace_bytes;
  int64_t pg_scale_floats, pg_mnll_workspace_floats;
  /* which fused forms cover this shape */
  int32_t decode_dz_fused_ok;      /* rk_decode_loss_dz_planes */
  int32_t fdec_ok;                 /* rk_fdec_loss_dz */
  int32_t dw_encode_bwd_fused_ok;  /* dW || encoder backward in one launch */
  int32_t encode_bwd_segments;     /* the fused fp32 dW || encoder-backward call writes G_en as this many partial
                                      arrays of n_cap * h floats (gb_en: of h floats), summed by rk_adam_multi */
  int64_t dw3_slabs_offset_bytes;  /* rk_decode_bwd_dw3 / dw2 with G_de == NULL leave their K slabs this far into the
                                      workspace (behind the Z^T planes) */
  int32_t mf_fdec_ok;              /* entry-by-entry sequenced steps (MatrixFactorization: rk_fdec_loss_dz +
                                      rk_pg_dw_dz_reduce; hidden stacks / dropout with MSE / BCE: rk_fdec_loss_dz +
                                      rk_fdec_dz_reduce + rk_pg_dw_encode_bwd): the fused decode covers this shape
                                      (the probe header's RK_TUNE_MF_FDEC value, 0 = off) */
} rk_plan_t;
int rk_plan(rk_plan_t *plan);

/* activation ids (reference nn.py:6-9 `activation(x, act)`) */
enum { RK_ACT_NONE = 0, RK_ACT_TANH = 1, RK_ACT_SIGMOID = 2, RK_ACT_RELU = 3,
       RK_ACT_SELU = 4, RK_ACT_ELU = 5 };
/* loss ids (reference model.py:87-99) */
enum { RK_LOSS_MSE = 0,      /* recoder/losses.py:43-47  (confidence weighted) */
       RK_LOSS_BCE = 1,      /* torch BCEWithLogitsLoss, model.py:91 */
       RK_LOSS_MNLL = 2,     /* recoder/losses.py:68-71 */
       RK_LOSS_NONE = 3 };   /* store logits only (predict, model.py:487-511) */

/*
 * One collated sampling group: S user rows restricted to the n_b item columns
 * that at least one of them touched (reference data.py:203-251 Batch list; all
 * slices of a group share `items`).  Plain struct of capacities + device
 * buffers, filled by rk_collate.
 */
typedef struct rk_block {
  int32_t S_cap;      /* max rows of a group */
  int32_t nnz_cap;    /* max stored interactions of a group */
  int32_t n_cap;      /* max |item set| = min(n_items, nnz_cap) */
  int32_t n_items;    /* catalogue size */
  int32_t ldw_rc;     /* words per row of bits_rc  (>= ceil(n_cap/32)) */
  int32_t ldw_cr;     /* words per column of bits_cr (>= ceil(S_cap/32)) */
  int32_t n_chunks;   /* ceil(n_items / RK_SCAN_CHUNK) */
  int32_t implicit;   /* != 0: every stored value is 1.0 (vals is not read by the
                         loss / backward kernels) -- implicit-feedback data */
  int32_t *counts;    /* [72] dev: n_b, nnz_b, ld (= round_up(n_b,32)), S; [4]: K slabs the last
                         rk_decode_bwd_dw3 wrote; [5] / [6]: overflow flags of the last rk_collate --
                         the true number of distinct items / stored interactions when it exceeded
                         n_cap / nnz_cap (the block is then truncated IN BOUNDS: the caller must
                         check and discard it), else 0; [8..71]: fp32 bit patterns of the running max
                         |dLoss/dLogit| the loss kernels saw since rk_collate zeroed them
                         (rk_decode_bwd_dz scales its fp16 split by it) */
  int32_t *indptr;    /* [S_cap+1] block CSR row pointers */
  int32_t *cols;      /* [nnz_cap] relabelled column (index into items) */
  float   *vals;      /* [nnz_cap] interaction values */
  float   *svals;     /* [nnz_cap] normalised * input-dropout-scaled values */
  int32_t *items;     /* [n_cap] sorted unique item ids (Batch.items) */
  int32_t *pos;       /* [n_items] item id -> compact column, -1 if absent */
  int32_t *mark;      /* [n_items] generation stamps */
  uint32_t *bits_rc;  /* [S_cap][ldw_rc] bit (r,c) set iff (r,c) stored */
  uint32_t *bits_cr;  /* [n_cap][ldw_cr] transposed bitmap; NULL = not built
                         (inference-only blocks) */
  int32_t *scan_tmp;  /* [2 * (n_chunks + 1)], 8-byte aligned: per-chunk counts (large catalogues) / 64-bit
                         {stamp, count} slots of the batched collation's look-back scan (<= 64 k items) */
  int32_t *pref_rc;   /* [S_cap][ldw_rc] exclusive prefix popcount of bits_rc per row:
                         entry index of (r,c) = indptr[r] + pref_rc[r][c>>5]
                         + popc(bits_rc[r][c>>5] & ((1<<(c&31))-1)) */
  int32_t *gcols;     /* [nnz_cap] global item id of every entry (= items[cols[j]]):
                         saves the encoder forward one dependent load per entry */
} rk_block_t;

#define RK_SCAN_CHUNK 2048

int rk_version(void);
const char *rk_last_error(void);
/* bytes of split-K workspace rk_decode_bwd_dz needs for (B, h) */

/*
 * rk_collate -- replaces RecommendationDataset.__getitem__/_extract
 * (data.py:50-83) + BatchCollator.collate (data.py:203-251): row-gather of
 * `users` from the device-resident CSR, sorted-unique item set
 * (np.unique(return_inverse)) and relabelled columns.  With
 * negative_sampling == 0 the item set is the whole catalogue (data.py:224-226).
 * stamp must differ between consecutive calls on the same blk->mark.
 * phase: 0 = everything; 1 = row pointers + item marking only; 2 = the rest.
 * Data-parallel training calls phase 1, all-reduces blk->mark with MAX over
 * the ranks (RCCL) so that every rank derives the same union item set, then
 * phase 2.
 */
int rk_collate(const int64_t *ds_indptr, const int32_t *ds_indices,
               const float *ds_data /* NULL => all 1.0 */,
               const int64_t *users, int32_t S, int32_t negative_sampling,
               int32_t stamp, int32_t phase, const rk_block_t *blk,
               void *stream);

/*
 * rk_densify -- rows [row_off, row_off+B) of a collated block as a dense
 * [B, ld] fp32 matrix with n live columns (model.py:457-458, the reference's
 * `torch.sparse.FloatTensor(indices, values, size).to_dense()`).  The fused path
 * never densifies; this feeds the generic torch-autograd path (user-defined
 * FactorizationModel subclasses, nn.Module losses, sgd/adagrad/rmsprop).
 */
int rk_densify(const rk_block_t *blk, int32_t row_off, int32_t B, int32_t n,
               float *out, int32_t ld, void *stream);

/*
 * rk_ae_encode_fwd -- DynamicAutoencoder.forward first layer (nn.py:235-240):
 * F.normalize(p=2,dim=1) -> input dropout -> LinearEmbedding(input_based)
 * (nn.py:269-278) -> activation, as one CSR x embedding SpMM over rows
 * [row_off, row_off+B) of the block.
 *   keep : per-nnz uint8 keep flags for the whole block (NULL: drawn from the
 *          counter RNG (seed, rng_step) when p > 0; all kept when p == 0)
 *   Z0   : [B, h] activated output
 */
int rk_ae_encode_fwd(const rk_block_t *blk, int32_t row_off, int32_t B,
                     const float *W_en, const float *b_en, int32_t h,
                     const uint8_t *keep, float p, uint64_t seed,
                     uint64_t rng_step, const int64_t *users, int32_t act,
                     float *Z0, void *stream);
/* the same launch with the W_de[tgt = blk items] half of the decode's operand split (rk_split_wz: pl->w,
 * pl->wt, the W scale from ranges[64..127]) as extra workgroups -- for steps sequenced entry by entry
 * whose split launch then only cuts Z (rk_split_wz with W_de == NULL) */
struct rk_planes;
int rk_ae_encode_fwd_split_w(const rk_block_t *blk, int32_t row_off, int32_t B, const float *W_en,
                             const float *b_en, int32_t h, const uint8_t *keep, float p, uint64_t seed,
                             uint64_t rng_step, const int64_t *users, int32_t act, float *Z0,
                             const float *W_de, const int32_t *ranges, const struct rk_planes *pl,
                             void *stream);

/*
 * rk_ae_encode_fwd_partial -- the raw partial sum of rk_ae_encode_fwd (no bias, no
 * activation) over the columns the block holds, normalised by user_norm[users[r]]
 * (the L2 norm of the user's WHOLE row; NULL: computed from the block as above).
 * Item-parallel training: every rank holds a column shard of the interactions,
 * the partial sums are all-reduced, then rk_bias_act finishes the layer.
 */
int rk_ae_encode_fwd_partial(const rk_block_t *blk, int32_t row_off, int32_t B,
                             const float *W_en, int32_t h, const uint8_t *keep,
                             float p, uint64_t seed, uint64_t rng_step,
                             const int64_t *users, const float *user_norm,
                             float *Zpart, void *stream);
/* X[r,c] = act(X[r,c] + bias[c]) in place (bias nullable) */
int rk_bias_act(float *X, const float *bias, int32_t rows, int32_t cols,
                int32_t act, void *stream);

/*
 * rk_ae_encode_bwd -- autograd of the above w.r.t. the gathered encoder rows
 * (model.py:397): G_en[c,:] (+)= sum_r svals[r,c] * dZ0pre[r,:]  (deterministic
 * ascending-row order via the transposed bitmap).  accumulate != 0 adds into
 * G_en (tied weights, nn.py:191-202).  gb_en (nullable): also the encoder-bias
 * gradient gb_en[h] = colsum(dZ0pre), computed by a few extra workgroups of the
 * same launch.
 */
int rk_ae_encode_bwd(const rk_block_t *blk, int32_t row_off, int32_t B,
                     const float *dZ0pre, int32_t h, float *G_en,
                     int32_t accumulate, float *gb_en, void *stream);

/*
 * rk_decode_loss -- LinearEmbedding(output) (nn.py:271-280) fused with the
 * loss (losses.py:43-47,68-71 / BCEWithLogits) and its gradient w.r.t. the
 * logits; the loss is divided by B as model.py:483-484.
 *   Z [B,h]; target block `tgt` rows [row_off,row_off+B) (training: the input
 *   block itself, model.py:473-476).
 *   MSE/BCE : dO[B,ld] <- dLoss/dLogits, loss partials -> loss_part
 *   MNLL    : dO <- logits (finish with rk_mnll_finish)
 *   NONE    : out[B, ld_out] <- logits (+bias), ld_out host-given
 *   loss_part : [rk_plan_t.loss_partials] floats, all-zero on entry
 *   gb_part   : nullable [ceil(B/row_tile)][ld] per-row-tile column sums of dO
 *               (MSE/BCE); colsum over those few rows = gradient of the
 *               gathered decoder bias (saves a second pass over dO)
 */
/* rows per decode tile: gb_part holds ceil(B / rk_plan_t.decode_row_tile) rows */
int rk_decode_loss(const float *Z, int32_t B, int32_t h, const rk_block_t *tgt,
                   int32_t row_off, const float *W_de, const float *b_de,
                   int32_t loss_kind, float confidence, float inv_B,
                   float *dO, int32_t ld_out, float *loss_part, float *gb_part,
                   const int32_t *ranges /* nullable, below */, void *stream);
/*
 * Operand ranges of the split-fp16 contractions (rk_decode_loss, rk_decode_bwd_dz): `ranges` is a
 * device array of 128 int32 = fp32 bit patterns, [0..63] maxima / upper bounds of |Z|, [64..127] of
 * |W_de| (the maximum over the slots of a half is used; an all-zero half or ranges == NULL means
 * the documented static range |Z| < 2048, |W_de| < 512).  The kernels pick power-of-two split
 * scales from them on the device, so no operand magnitude can overflow the fp16 pieces.
 * rk_amax writes max |x| into slots[0] and zeroes slots[1..63] (one small launch; the trainer
 * calls it on Z only when the activation is unbounded -- tanh / sigmoid need nothing);
 * rk_adam_job_t.amax_out keeps the running maximum of a parameter tensor inside the Adam sweep.
 */
int rk_amax(const float *x, int64_t n, int32_t *slots, void *stream);
/* MNLL second pass: row max / logsumexp over the logits in dO, loss, and
 * dO <- (softmax * sum_t - t) * inv_B  (losses.py:68-71 + autograd). */
/* Item-parallel form of the above (the softmax spans every rank's items):
 * rk_mnll_row_stats writes stats[r] = {max, sum exp(o - max)} over the block's shard of row r;
 * the caller combines them over the ranks and passes, per row, the global max, the log of the
 * global sum (relative to that max) and the target sum of the WHOLE row to rk_mnll_finish. */
int rk_mnll_row_stats(const float *logits, int32_t B, const rk_block_t *tgt, float *stats,
                      void *stream);
int rk_mnll_finish(float *dO, int32_t B, const rk_block_t *tgt, int32_t row_off,
                   float inv_B, const float *row_max /* nullable, all three: the block holds whole rows */,
                   const float *row_logsum, const float *row_tsum, float *loss_part, void *stream);
/* sum the (unscaled) loss partials in a fixed order (double) and divide by
 * denom = rows of the slice in fp32 (model.py:483-484) -> loss[0]; the consumed
 * partials are reset to 0 (rk_decode_loss requires loss_part zeroed on entry) */
int rk_loss_reduce(float *loss_part, int32_t n, float denom, float *loss,
                   void *stream);

/*
 * Decoder backward (autograd of F.linear(z, W_de[T], b_de[T]), nn.py:280):
 *   rk_decode_bwd_dz : dZ[B,h] = dO[B,n_t] . W_de[T]   (* act'(Zact) if Zact)
 *   rk_decode_bwd_dw : G_de[n_t,h] = dO^T . Z ;  gb_de[n_t] = colsum(dO)
 */
int rk_decode_bwd_dz(const float *dO, int32_t B, int32_t h,
                     const rk_block_t *tgt, const float *W_de,
                     const float *Zact /* nullable */, int32_t act,
                     float *dZ, float *workspace, const int32_t *ranges /* nullable */,
                     void *stream);
int rk_decode_bwd_dw(const float *dO, const float *Z, int32_t B, int32_t h,
                     const rk_block_t *tgt, float *G_de, float *gb_de,
                     void *stream);
/* (The fp32 tiles' dW + rk_ae_encode_bwd(accumulate = 0) on one block also exist as ONE launch -- the MFMA-bound dW
 * tiles and the latency-bound encoder-backward gathers share the GPU -- inside rk_ae_train_step (RK_GEMM_PREC=f32,
 * item-parallel steps); rounds 1-5 exported it as rk_decode_bwd_dw_encode_bwd.  For large B that call cuts the
 * contraction along K into `workspace` slabs, summed in fixed order.) */
/* (rk_plan_t.dw_splits: the number of K slabs that call cuts dW into for B rows, 1 = none.  With more
 * than one, G_de == NULL leaves them unsummed in `workspace` as arrays of n_cap*h floats -- G_de = their sum
 * in slab order: rk_adam_multi consumes them through g_parts / g_stride and the summing launch disappears.
 * rk_plan_t.dw_workspace_bytes sizes the workspace; NULL disables the split-K of large batches.) */
/*
 * rk_decode_bwd_dw3 -- the same contraction G_de[n_t,h] = dO^T . Z (autograd of F.linear,
 * nn.py:280) on the 16-bit matrix pipe at fp32 accuracy and with no operand range: every fp32
 * operand is cut into three bf16 pieces (x = hi + mid + lo exactly), six products per pair are
 * accumulated in fp32 (csrc/dw3.hip).  dO tiles reach LDS by DMA; Z is split once per call into
 * k-contiguous bf16 planes at the head of `workspace`.
 *   workspace : rk_plan_t.dw3_workspace_bytes bytes, 256-byte aligned
 *   G_de      : nullable.  The contraction is cut along K into counts[4] slabs (chosen on the
 *               device from the live item count, <= rk_plan_t.dw3_max_splits); G_de != NULL receives
 *               their sum, G_de == NULL leaves them at workspace + rk_plan_t.dw3_slabs_offset_bytes as arrays of
 *               n_cap*h floats for rk_adam_multi (rk_adam_job_t.gparts_dev = counts + 4).
 */
int rk_decode_bwd_dw3(const float *dO, const float *Z, int32_t B, int32_t h,
                      const rk_block_t *tgt, float *G_de, float *gb_de, void *workspace,
                      const void *zt_planes /* nullable: made from Z into workspace */,
                      void *stream);
/*
 * rk_decode_bwd_dw2 -- the same contraction with the operands cut into fp16 PAIRS (s.x = hi + lo,
 * three products lo.hi + hi.lo + hi.hi on v_mfma_f32_32x32x16_f16: half the MFMAs of the bf16
 * triples, two planes instead of three) -- the arithmetic of rk_decode_loss / rk_decode_bwd_dz.
 * Power-of-two scales from published maxima: dO from tgt->counts[8..71] (the loss kernels), Z from
 * ranges[0..63] (rk_amax notes) when the planes are made here; a caller-provided zt_planes holds
 * pairs written with the static scale (rk_ae_train_step: bounded activations).  Workspace and
 * slab conventions as rk_decode_bwd_dw3.  rk_plan_t.dw_pairs != 0: the training step uses this entry
 * (default; the probe header's RK_TUNE_DW_BF16X3 keeps the triples).
 */
int rk_decode_bwd_dw2(const float *dO, const float *Z, int32_t B, int32_t h,
                      const rk_block_t *tgt, float *G_de, float *gb_de, void *workspace,
                      const void *zt_planes /* nullable */, const int32_t *ranges /* nullable */,
                      void *stream);
/*
 * rk_decode_bwd_dw2 (the K slabs stay in the workspace, as with G_de == NULL) and rk_ae_encode_bwd in
 * ONE launch: its first workgroups run the dW tiles, the others the encoder backward's columns.  Both
 * depend only on what is in front of them in the step and write disjoint outputs; as two launches
 * they cost the chain 16 + 14 us in line, or a side stream's two cross-queue edges.  Domain:
 * rk_plan_t.dw_encode_bwd_fused_ok (fp16-pair dW, row window of <= 64 bitmap words).
 */
int rk_decode_bwd_dw2_encode_bwd(const float *dO, const float *Z /* nullable with zt_planes */, int32_t B,
                                 int32_t h, const rk_block_t *tgt, void *workspace,
                                 const void *zt_planes /* nullable */, const int32_t *ranges /* nullable */,
                                 int32_t row_off, const float *dZ0pre, float *G_en,
                                 float *gb_en /* nullable */, void *stream);
/* rk_decode_bwd_dw2 (G_de == NULL: the K slabs stay in `workspace`) and rk_decode_dz_reduce (the column-
 * tile slabs rk_decode_loss_dz_planes left in dz_workspace -> dZ, * act'(Zact) if given) in ONE launch:
 * MatrixFactorization steps, where nothing between the decode and the Adam sweep reads dZ
 * (nn.py:344-362: the user rows' gradient goes straight to the optimizer). */
int rk_decode_bwd_dw2_dz_reduce(const float *dO, const float *Z /* nullable with zt_planes */, int32_t B,
                                int32_t h, const rk_block_t *tgt, void *workspace,
                                const void *zt_planes /* nullable */, const int32_t *ranges /* nullable */,
                                const float *dz_workspace, const float *Zact /* nullable */, int32_t act,
                                float *dZ, void *stream);
/* the same launch with the decoder bias gradient gb_de[c] = sum_r dO[r][c] as a third workgroup range
 * (multinomial loss: dO comes from rk_mnll_finish, no decode epilogue has summed its columns) --
 * the rk_colsum launch of those steps */
int rk_decode_bwd_dw2_encode_bwd_colsum(const float *dO, const float *Z /* nullable with zt_planes */,
                                        int32_t B, int32_t h, const rk_block_t *tgt, void *workspace,
                                        const void *zt_planes, const int32_t *ranges, int32_t row_off,
                                        const float *dZ0pre, float *G_en, float *gb_en, float *gb_de,
                                        void *stream);
/* The fused call writes G_en as rk_plan_t.encode_bwd_segments partial arrays of n_cap*h floats
 * each (row segments of long item columns; 1 below 513 rows) and gb_en as as many partial
 * vectors of h floats: the gradients are their sums in segment order -- rk_adam_multi
 * consumes them through g_parts / g_stride. */

/*
 * Pre-split operand planes of the decoder contractions (csrc/planes.h, csrc/decode16.hip).
 * rk_decode_loss / rk_decode_bwd_dz cut every fp32 operand into an fp16 pair (s.x = hi + lo) while
 * they stage it -- each W_de row once per row tile, Z once per column tile.  The *_planes entry
 * points take the operands ALREADY split, once per step, as compact plane images (one 128-byte
 * line = 32 hi + 32 lo values of one row's k-tile) and run the same arithmetic (lo.hi + hi.lo +
 * hi.hi in fp32 on v_mfma_f32_32x32x16_f16, same k order: bit-identical results for equal tile
 * shapes) with a copy -> LDS -> MFMA k-loop:
 *   z  : image of Z [B, h]                (rk_split_wz, or the encoder forward of rk_ae_train_step)
 *   w  : image of W_de[items[c]] [n_b, h] (rk_split_wz)          -- B operand of the decode
 *   wt : image of its transpose [h, n_b]  (rk_split_wz)          -- B operand of dZ
 * rk_planes_layout carves the three images + 4 scale slots out of ONE caller-allocated, ZEROED,
 * 256-byte aligned buffer of rk_plan_t.planes_bytes bytes (the K padding is never
 * written and must read as zero).
 */
typedef struct rk_planes {
  float *scales;      /* [4] dev: [0] split scale of Z, [1] of W_de (written by the split passes) */
  void *z, *w, *wt;
  int32_t h, B_cap, n_cap, n_ld;   /* n_ld = round_up(n_cap, 32) */
} rk_planes_t;
/*
 * The decode + loss of rk_decode_loss_planes (MSE / BCE, 64 x 128 tiles) with dZ FUSED: every workgroup
 * multiplies the dO tile it has just computed (kept in LDS, cut into fp16 pairs with the TILE's own
 * power-of-two scale) with its 128 items of the W^T image and writes the partial dZ[64 rows, h] into
 * slab (column tile) of dz_workspace [ceil(n_cap / 128)][B][h]; rk_decode_dz_reduce sums the live
 * slabs (* act'(Zact) if given) into dZ.  Replaces rk_decode_loss_planes + rk_decode_bwd_dz_planes
 * (one launch, one pass over dO and one cross-queue edge less); same mathematics, other split scales
 * and summation grouping than the stand-alone kernel (agreement ~1e-7 relative).  Shapes / losses:
 * rk_plan_t.decode_dz_fused_ok (h <= 256, B < 1024, slab workspace <= 4 GB; always 64-row tiles).
 */
int rk_decode_loss_dz_planes(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                             const float *b_de, int32_t loss_kind, float confidence, float inv_B, float *dO,
                             float *loss_part, float *gb_part, float *dz_workspace, void *stream);
int rk_decode_dz_reduce(const float *dz_workspace, int32_t B, int32_t h, const rk_block_t *tgt,
                        const float *Zact /* nullable */, int32_t act, float *dZ, void *stream);
/*
 * Round 4: the decoder's three contractions as ONE pipelined kernel family (csrc/pgemm.h) on THREE plane
 * images -- Z, W_de[items] and dLoss/dLogits -- and no transposed copy of any of them: an operand whose
 * contraction index runs along its image's rows is read from LDS with the transpose read of gfx950
 * (ds_read_b64_tr_b16); k-tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), the copy of
 * tile t + 1 in flight under the MFMAs of tile t, one barrier per k-tile.  Replaces, where it applies
 * (MSE / logistic loss), rk_decode_loss_planes + rk_decode_bwd_dz_planes + rk_decode_bwd_dw2 (reference
 * nn.py:271-280, losses.py:43-47, and autograd of F.linear).
 *
 * rk_pg_decode_loss: decode + loss; dLoss/dLogits leaves as a plane image of fp16 pairs (row m at byte
 *   m * ld * 4 of dO_img, ld = counts[2] -- the footprint of the fp32 matrix it replaces; rows
 *   [B, round_up(B, 32)) and columns [n_t, ld) are written as zeros) cut with the TILE's own power-of-two
 *   scale, published in dO_scales[(row / gr) * ceil(n_cap / gc) + col / gc] with (gr, gc) =
 *   rk_plan_t.pg_granule_rows / _cols (rk_plan_t.pg_scale_floats floats fit every producer).  dO_f32 (nullable):
 *   the fp32 matrix as well (tests).  loss_part / gb_part as rk_decode_loss.
 * rk_pg_dz: dZ[B, h] = dO . W_de[T] (* act'(Zact) if given) from the image and pl->w; workspace:
 *   rk_plan_t.pg_dz_workspace_bytes.  rk_pg_dw: the K slabs [rk_plan_t.pg_dw_splits][n_cap][h] of dW[n_t, h] = dO^T . Z
 *   from the image and pl->z (rk_adam_multi adds them: g_parts).  (gr, gc): the granule of the producer
 *   of the image -- rk_plan_t.pg_granule_rows / _cols for rk_pg_decode_loss / rk_pg_decode_mnll, 32 x 64 for rk_fdec_loss_dz.
 */
/* The fused decode + loss + dZ partials of SMALL hidden sizes, register resident (csrc/fdecode.hip): the
 * 128-item x 128-user tile is computed transposed, so that the loss runs in the accumulator layout and
 * the gradient tile becomes the MFMA operand of the dZ product through v_permlane32_swap -- no LDS round
 * trip; the tile's W rows stay resident in LDS and are read a second time along their rows (no W^T image).
 * dLoss/dLogits leaves as a plane image (scale granule 32 users x 64 items: dO_scales[(m / 32) *
 * ceil(n_cap / 64) + n / 64]).  Round 5, the STREAMING form: a workgroup walks a GROUP of column tiles of its
 * 128-user row tile with the dZ accumulators in registers and leaves ONE slab per group (row tiles x groups
 * ~ one workgroup per CU) -- rk_fdec_dz_reduce sums them (* act'(Zact) if given) into dZ[B, h].  The decoder
 * bias gradient is NOT produced here: rk_pg_dw_encode_bwd takes it from the image.
 * Domain: rk_plan_t.fdec_ok (mse / logistic, h <= 224; RK_FDEC=0: off). */
int rk_fdec_loss_dz(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off, const float *b_de,
                    int32_t loss_kind, float confidence, float inv_B, void *dO_img, int32_t rows_img,
                    float *dO_scales, float *loss_part, float *dz_workspace, void *stream);
int rk_fdec_dz_reduce(const float *dz_workspace, int32_t B, int32_t h, const rk_block_t *tgt, const float *Zact,
                      int32_t act, float *dZ, void *stream);
/* rk_pg_dw (dW slabs from the dO image, + the decoder bias gradient from its columns) with rk_fdec_dz_reduce riding on
 * the SAME launch as a workgroup range -- steps in which nothing between the decode and the optimizer reads dZ
 * (MatrixFactorization: the user rows' gradient); < 1024 rows.  dense != 0: no split-K -- `slabs` is the ONE dense
 * gradient array [n_cap][h] (the data-parallel exchange ships it). */
int rk_pg_dw_dz_reduce(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                       const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, float *gb_de /* nullable */,
                       const float *dz_workspace, const float *Zact /* nullable */, int32_t act, float *dZ,
                       int32_t dense, void *stream);
/* (RK_PG=0 in the environment switches the family off: the round-3 plane kernels run instead) */
int rk_pg_decode_loss(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                      const float *b_de, int32_t loss_kind, float confidence, float inv_B, void *dO_img,
                      int32_t rows_img, float *dO_scales, float *dO_f32 /* nullable */, float *loss_part,
                      float *gb_part, void *stream);
/* The multinomial NLL (losses.py:68-71) the same way, as two passes over the decode: a statistics pass
 * that writes 8 bytes per (row, column tile) -- {max, sum exp} of its live columns -- and the decode +
 * loss pass that merges them into the row's log-sum-exp; no logits matrix is written, re-read and
 * rewritten (rk_decode_loss_planes + rk_mnll_finish).  mnll_ws: rk_plan_t.pg_mnll_workspace_floats floats;
 * loss_part: one partial per tile (rk_plan_t.loss_partials slots), gb_part as rk_decode_loss. */
int rk_pg_decode_mnll(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                      const float *b_de, float inv_B, float *mnll_ws, void *dO_img, int32_t rows_img,
                      float *dO_scales, float *dO_f32 /* nullable */, float *loss_part, float *gb_part,
                      void *stream);
int rk_pg_dz(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
             const rk_planes_t *pl, const rk_block_t *tgt, const float *Zact /* nullable */, int32_t act,
             float *dZ, float *workspace, void *stream);
int rk_pg_dw(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
             const rk_planes_t *pl, const rk_block_t *tgt, float *slabs,
             float *gb_de /* nullable: the decoder bias gradient [n_t] = column sums of dO, from the image, by a
                             second workgroup range of the launch */, void *stream);
/* rk_pg_dw || rk_ae_encode_bwd (G_en, gb_en as there; nothing accumulated) in ONE launch: the dW tiles
 * first in the grid, then a wave per item column (domain: rk_plan_t.dw_encode_bwd_fused_ok) */
int rk_pg_dw_encode_bwd(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                        const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, int32_t row_off,
                        const float *dZ0pre, float *G_en, float *gb_en,
                        float *gb_de /* nullable: the decoder bias gradient [n_t] = column sums of dO, from
                                        the image, by a third workgroup range */, void *stream);
int rk_planes_layout(void *buffer, int32_t B_cap, int32_t h, int32_t n_cap, rk_planes_t *out);
/* The operand splits of a decode in ONE launch (same images, bit for bit, as separate launches would make):
 *   W_de[tgt->items[0 .. n_b)] -> pl->w and pl->wt (pl->wt == NULL: no W^T image); scale from ranges[64..127]
 *     (rk_amax notes).  W_de NULL: the W images are already there (rk_ae_encode_fwd_split_w).
 *   Z[B, h] -> pl->z; the scale from ranges[0..63].  Z NULL: only W is cut.
 *   dw_workspace (nullable): ALSO Z^T as the fp16 pair planes (+ their scale) of the dW kernel at the head of
 *     dw_workspace -- where rk_decode_bwd_dw2 makes them in a launch of its own when it is called with
 *     zt_planes == NULL; call it with zt_planes == workspace == dw_workspace afterwards (the kernel then reads
 *     the scale the planes were written with).  Only where rk_plan_t.split_zt_ok (dW on fp16 pairs, not the
 *     plain-bf16 data point). */
int rk_split_wz(const float *W_de /* nullable */, const float *Z /* nullable */, int32_t B, int32_t h,
                const rk_block_t *tgt, const int32_t *ranges, const rk_planes_t *pl,
                void *dw_workspace /* nullable */, void *stream);
/* rk_decode_loss on pl->z x pl->w (arguments as there).  MSE / BCE: the padding columns
 * [n_t, ld) of every dO row are written as zeros. */
int rk_decode_loss_planes(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                          const float *b_de, int32_t loss_kind, float confidence, float inv_B,
                          float *dO, int32_t ld_out, float *loss_part, float *gb_part, void *stream);
/* rk_decode_bwd_dz on dO x pl->wt (arguments as there; dO rows are ld = counts[2] apart). */
int rk_decode_bwd_dz_planes(const float *dO, int32_t B, const rk_planes_t *pl, const rk_block_t *tgt,
                            const float *Zact /* nullable */, int32_t act, float *dZ,
                            float *workspace, void *stream);

/*
 * Hidden nn.Linear stack (nn.py:242-249): Y = act(X W^T + b), and backward.
 *   rk_linear_fwd : Y[B,N] = act(X[B,K] . W[N,K]^T + b)      (wT: W is [K,N])
 *   rk_linear_bwd : dYpre = dY * act'(Y) (in place), dX = dYpre . W,
 *                   dW (+)= dYpre^T . X, db = colsum(dYpre)
 */
int rk_linear_fwd(const float *X, const float *W, const float *b, int32_t B,
                  int32_t N, int32_t K, int32_t w_transposed, int32_t act,
                  float *Y, void *stream);
int rk_linear_bwd(float *dY, const float *Y, const float *X, const float *W,
                  int32_t B, int32_t N, int32_t K, int32_t w_transposed,
                  int32_t act, float *dX /* nullable */, float *dW,
                  int32_t dw_accumulate, float *db, void *stream);
/* rk_linear_bwd for a dY that already IS dYpre = dY * act'(Y) (its producer multiplied act' in: the slab
 * reduce of the decode's dZ, or the dX epilogue of the layer behind it -- dx_act_y): no pass over it;
 * dX (* act'(dx_act_y) if given), dW and db = colsum(dYpre) in ONE launch where the small kernel
 * applies (else rk_colsum + the two products).  `act` is the derivative dx_act_y is taken with. */
int rk_linear_bwd_pre(const float *dYpre, const float *X, const float *W, int32_t B, int32_t N, int32_t K,
                      int32_t w_transposed, int32_t act, float *dX /* nullable */, float *dW,
                      int32_t dw_accumulate, float *db, const float *dx_act_y /* nullable */, void *stream);
/* rk_linear_bwd whose dX leaves multiplied by act'(dx_act_y[B,K]) (nullable): the backward of a
 * stack's FIRST Linear layer hands its gradient to the embedding layer's activation -- the
 * rk_act_grad launch that followed, folded into the dX epilogue (same product, same bits). */
int rk_linear_bwd_dact(float *dY, const float *Y, const float *X, const float *W,
                       int32_t B, int32_t N, int32_t K, int32_t w_transposed, int32_t act,
                       float *dX, float *dW, int32_t dw_accumulate, float *db,
                       const float *dx_act_y, void *stream);

/* elementwise helpers */
int rk_act_grad(float *dY, const float *Y, int64_t n, int32_t act, void *stream);
int rk_dropout(float *X, const uint8_t *keep, int64_t n, int32_t ncols, float p,
               uint64_t seed, uint64_t rng_step, void *stream);
/* out[c] = sum_r X[r,c]; if counts_dev (a block's counts array) is given the
 * column count and ld are read from it (counts[0], counts[2]) */
int rk_colsum(const float *X, int32_t rows, int32_t cols, int32_t ld,
              const int32_t *counts_dev /* nullable */, float *out,
              void *stream);
/* out[r, :] = act(E[rows[r], :]) (nn.py:344-362: MatrixFactorization's user rows), which also
 *   slots  (nullable): publishes max |out| in slots[0..63] the way rk_amax does (64 workgroups, one
 *                      slot each): gather + rk_amax in one launch for unbounded activations;
 *   rows32 (nullable, B + 1 ints): rows32[0] <- B, rows32[1 + r] <- rows[r] -- the index array and
 *                      device-resident count of a SparseAdam job (rk_adam_job_t.rows / n_dev): the
 *                      user table's update then rides on the step's rk_adam_multi launch instead
 *                      of an rk_adam_rows launch of its own */
int rk_gather_rows_amax(const float *E, const int64_t *rows, int32_t B, int32_t d,
                        int32_t act, float *out, int32_t *slots, int32_t *rows32, void *stream);

/*
 * Optimisers (exact formulas of torch.optim.Adam `_single_tensor_adam` and
 * torch.optim.SparseAdam, model.py:135,138,398-402).  `step` is the 1-based
 * step count after increment.  Hyper-parameters are doubles because torch
 * evaluates `1 - beta`, the bias corrections and the step size in Python
 * doubles before the one conversion to fp32.
 *   rk_adam_rows  : SparseAdam on the rows idx[0..n_dev) (K14b); no decay.
 * (Dense Adam -- a full sweep of a [n_rows, h] table whose row i gets gradient G[pos[i]] if pos[i] >= 0 else 0 with L2
 * weight decay, K14a, or a plain dense tensor -- is a job of rk_adam_multi below; rounds 1-5 also exported it
 * as rk_adam_table / rk_adam_dense.)
 */
int rk_adam_rows(float *W, float *m, float *v, int32_t h, const int32_t *idx32,
                 const int64_t *idx64, const int32_t *n_dev, int32_t n_cap,
                 const float *G, double lr, double beta1, double beta2,
                 double eps, int32_t step, void *stream);
/* user-row position map for MF dense Adam: pos[users[r]] = r (or -1 to clear) */
int rk_scatter_pos(int32_t *pos, const int64_t *rows, int32_t B, int32_t clear,
                   void *stream);

/*
 * rk_ae_train_step -- one call = one optimisation step of Recoder._train's hot
 * loop (model.py:383-404) for DynamicAutoencoder(hidden_layers=[h]) without
 * bottleneck dropout: zero_grad / __compute_loss (model.py:454-485) / backward /
 * optimizer.step + sparse_optimizer.step as a chain of 8 launches on one HIP
 * stream: encode_fwd, decode+loss, dW, dZ (split-K GEMM + reduce), encode_bwd
 * (+ encoder-bias gradient), rk_adam_multi (all updates + the loss scalar).  It
 * launches exactly the entry points of this header; it exists so that the host
 * pays one FFI call per step instead of ~35.
 */
enum { RK_PAR_W_EN = 0, RK_PAR_B_EN = 1, RK_PAR_W_DE = 2, RK_PAR_B_DE = 3, RK_PAR_COUNT = 4 };
enum { RK_ENTRY_NONE = 0, RK_ENTRY_ENCODE_FWD = 1, RK_ENTRY_DECODE_LOSS = 2,
       RK_ENTRY_DECODE_BWD_DZ = 3, RK_ENTRY_DECODE_BWD_DW = 4, RK_ENTRY_ENCODE_BWD = 5,
       RK_ENTRY_ADAM_MULTI = 6,
       RK_ENTRY_COUNT = 7,
       RK_ENTRY_ALL = -1 /* bracket every entry: rk_ae_step_t.time_all */ };
/* rk_ae_step_t.phase: which part of the step to enqueue (0 = all).  Data parallel
 * callers run FWD_DW, all-reduce the decoder-side gradients, DZ_ENC, all-reduce the
 * encoder side, then UPDATE. */
enum { RK_STEP_FWD_DW = 1, RK_STEP_DZ_ENC = 2, RK_STEP_UPDATE = 4, RK_STEP_ALL = 7 };
/* Item-parallel segments (item i owned by rank i % own_world; the block holds every
 * user of the global batch restricted to the rank's items): IP_ENC, all-reduce(SUM) of
 * Z0[B,h], IP_MID, all-reduce(SUM) of dZ0[B,h], IP_TAIL.  See step.hip. */
enum { RK_STEP_IP_ENC = 8, RK_STEP_IP_MID = 16, RK_STEP_IP_TAIL = 32, RK_STEP_IP_ALL = 56 };

typedef struct rk_adam_param {
  float *p, *m, *v;          /* parameter and its Adam moments */
  double lr, beta1, beta2, eps, weight_decay;
  int32_t step;              /* 1-based step of THIS update */
  int32_t sparse;            /* tables only: SparseAdam on the touched rows */
} rk_adam_param_t;

/*
 * rk_adam_multi -- all parameter updates of one step (optimizer.step +
 * sparse_optimizer.step, model.py:398-402) and, optionally, the loss scalar in
 * ONE launch: up to RK_ADAM_MULTI_MAX jobs, each the exact arithmetic of
 * a dense-Adam table sweep (pos != NULL), a plain dense tensor (pos == NULL) or rk_adam_rows
 * (par.sparse: rows / n_dev / n_cap).  The gradient of a dense job may be the
 * sum of g_parts arrays g + t * stride (t ascending; stride = *gstride_dev when
 * given, else g_stride) -- the decoder-bias gradient is consumed straight from
 * rk_decode_loss's per-row-tile partials this way.  loss_part != NULL: the last
 * workgroup does what rk_loss_reduce does.
 */
#define RK_ADAM_MULTI_MAX 10
typedef struct rk_adam_job {
  rk_adam_param_t par;
  int32_t n_rows, h;           /* table shape ([1, n] for a flat tensor) */
  const int32_t *pos;          /* dense table job: row -> compact gradient row or -1 */
  const int32_t *rows;         /* SparseAdam job: compact row -> table row */
  const int32_t *n_dev;        /* SparseAdam job: device-resident live row count */
  int32_t n_cap;               /* SparseAdam job: capacity of rows (launch size) */
  int32_t g_parts, g_stride;
  const int32_t *gstride_dev;
  const float *g;
  int32_t row0, row_step;      /* dense jobs: only rows row0, row0 + row_step, ... are updated
                                  (item-parallel ownership: item i lives on rank i % N);
                                  row_step 0 = 1 */
  int32_t *amax_out;           /* nullable: 64 slots (fp32 bit patterns): running maximum of |p|
                                  over everything this job writes (atomicMax, never reset) */
  const int32_t *gparts_dev;   /* nullable: the number of gradient parts is read from the device
                                  (rk_decode_bwd_dw3 publishes its slab count in counts[4]);
                                  g_parts is then the capacity */
  /* LAZY dense Adam (lazy_stamp != NULL; dense table jobs with pos, h % 4 == 0, constants from the replay
   * table, i.e. a cursor): optim.Adam with a dense gradient (model.py:135,398-399) updates EVERY row of the
   * table every step -- rows outside the block's item set with g = 0 (+ weight_decay * p).  That update is a
   * recurrence of the row's own (p, m, v) and the per-step constants only, so it can be applied LATER by
   * replaying the identical per-step arithmetic.  lazy_stamp[row] = global index of the first step NOT yet
   * applied to the row.  A launch of step T brings up to date (missed steps with g = 0, then step T with its
   * gradient) exactly the rows that
   *   have a gradient (pos[row] >= 0), or are read by the NEXT step (lazy_pos_next[row] >= 0 -- that step's
   *   forward needs them current), or lie in chunk T % lazy_period of the table (bounds every lag by
   *   lazy_period steps);
   * lazy_pos_next == NULL: every row (what the last step in front of a checkpoint / validation / the end of
   * train() does).  Row by row the result is BITWISE the dense sweep's.  rk_adam_lazy_flush below brings
   * every row up to date without a step. */
  const int32_t *lazy_pos_next;
  int32_t *lazy_stamp;
  int32_t lazy_period;         /* >= 1 */
  /* nullable (lazy_pos_next != NULL): the rows that have a gradient or are read by the next step (pos >= 0 or
   * lazy_pos_next >= 0), ascending, and their number -- built ahead of the step by rk_lazy_need_lists; the sweep then
   * takes one wave per listed row + one per row of the round-robin chunk instead of one per table row (a third of
   * C2's waves, two thirds of C3's, found nothing to do) */
  const int32_t *lazy_need_list;
  const int32_t *lazy_need_count;
} rk_adam_job_t;

int rk_adam_multi(const rk_adam_job_t *jobs, int32_t n_jobs, float *loss_part,
                  int32_t n_part, float denom, float *loss_out, void *stream);
/* Replay the missed steps [lazy_stamp[row], next_step) of every row of the jobs' tables (g = 0 + weight
 * decay) and set lazy_stamp[row] = next_step: afterwards the tables are what the dense sweeps would have
 * left.  table / tab_stride / tab_slots as rk_replay_t.adam_table (entry of global step s and job j at
 * table[(s - epoch_base) * tab_stride + tab_slots[j]]); only p / m / v, n_rows, h, lazy_stamp and amax_out
 * of a job are read. */
int rk_adam_lazy_flush(const rk_adam_job_t *jobs, int32_t n_jobs, const void *table, int32_t tab_stride,
                       const int32_t *tab_slots, int64_t next_step, int64_t epoch_base, void *stream);
/* D[row, :] = pos[row] >= 0 ? G[pos[row], :] : 0 for row < n_items, 0 for n_items <= row < rows_pad: the
 * compact gradient rows of a block laid out by ITEM ID -- the layout a reduce-scatter over equal row
 * ranges needs (sharded dense Adam, rk_ae_step_t.zero_lo).  h % 4 == 0 and 16-byte aligned, or h == 1 (a gathered
 * bias gradient). */
int rk_rows_to_dense(const float *G, const int32_t *pos, int32_t n_items, int32_t rows_pad, int32_t h,
                     float *D, void *stream);
/* X_k[n_b * h_k .. top * h_k) <- 0 for n_arrays <= 4 compact [n_cap, h_k] gradient arrays, n_b = counts[0] read on the
 * device.  Replayed data-parallel steps exchange the blocks' whole capacity (a captured collective has a fixed size)
 * summed in place: the rows past the live items, which no kernel rewrites, must be zero going in.  high_water
 * (nullable device int32; 0 for freshly zeroed arrays): rows at or past it are known to be zero; top = max(*high_water,
 * n_b) is stored back -- a step clears what a larger earlier item set left behind, not the whole tail.  NULL: top = n_cap. */
int rk_zero_tail_rows(float *const *X, const int32_t *h, int32_t n_arrays, const int32_t *counts, int32_t n_cap,
                      int32_t *high_water, void *stream);

typedef struct rk_ae_step {
  const rk_block_t *blk;
  int32_t row_off, B, h, act, loss_kind, tied;
  float confidence, inv_B, denom, noise_p;
  uint64_t seed, rng_step;
  const uint8_t *keep;       /* nullable per-nnz keep flags (parity tests) */
  const int64_t *users;      /* global user ids of the block rows (RNG key) */
  rk_adam_param_t par[RK_PAR_COUNT];
  /* workspaces (sizes as FusedEngine.ensure_capacity allocates them) */
  float *Z0, *dZ0, *dO, *G_de, *G_en, *gb_de, *gb_part, *gb_en, *ws, *loss_part, *loss_out;
  /* (gb_part: ceil(B / rk_plan_t.decode_row_tile) rows of round_up(n_cap, 32) floats.  Whole steps on the fused decode
   * with h % 32 != 0 leave the decoder bias gradient THERE as one slab of n_cap floats per K slab of dW -- see
   * rk_ae_step_uses_pg bit 4 -- when that fits, i.e. rk_plan_t.pg_dw_splits * n_cap floats) */
  void *stream;              /* hipStream_t: every kernel of the step goes here, in order */
  int32_t time_entry;        /* RK_ENTRY_*: bracket that entry with the two events below */
  int32_t phase;             /* mask of RK_STEP_* (0 = RK_STEP_ALL) */
  void *time_ev0, *time_ev1; /* rk_event_create(1) */
  /* item-parallel segments only */
  const float *user_norm;    /* [n_users] L2 norm of every user's whole row (by global user id) */
  int32_t own_rank, own_world;
  void *zt_planes;           /* nullable: rk_plan_t.dw3_planes_bytes bytes, zeroed once: the
                                encoder forward writes the Z^T planes of the dW kernel there */
  /* HIP-graph replay (whole steps only): a replayed launch cannot take new arguments, so what
   * changes per step is derived ON THE DEVICE from cursor = {global index of the next step,
   * global index of the epoch's first step} (device memory) and this launch's offset in the
   * replayed group: rng_step = cursor[0] + off + 1, Adam constants = adam_table[(cursor[0] -
   * cursor[1] + off) * RK_PAR_COUNT + par] (8 floats each, rk_adam_consts), the loss goes to
   * loss_out[cursor[0] - cursor[1] + off].  cursor == NULL: the host values above are used.
   * Phased (data-parallel) steps replay too: FWD_DW leaves the local loss in loss_out[0] as ever
   * (the caller points it at a scalar of its own and all-reduces it with the gradients); for the
   * UPDATE call the caller sets loss_part = that scalar and loss_out = the epoch's loss buffer --
   * the Adam launch files loss_part[0] under the step's slot and publishes cursor_next. */
  int32_t *ranges;           /* nullable: 128 slots, operand ranges of the decoder contractions
                                (rk_decode_loss): the step keeps [64..127] up to date from its Adam
                                sweep of the decoder table and fills [0..63] with rk_amax(Z) when
                                the activation is unbounded */
  const int64_t *cursor;
  int32_t cursor_off;
  int32_t cursor_advance;    /* with cursor_next: the last step of a replayed group publishes the
                                cursor of the NEXT group, {cursor[0] + cursor_advance, cursor[1]},
                                into cursor_next (a second buffer -- launches of this group may
                                still be reading `cursor`); done by the Adam launch, no extra kernel */
  const void *adam_table;
  int64_t *cursor_next;
  void *const *time_all;     /* time_entry == RK_ENTRY_ALL: host array of 2 * RK_ENTRY_COUNT timing
                                events, entry e is bracketed by [2e] and [2e + 1] */
  /* dW as a branch of the step (whole untied MSE / BCE steps on the 16-bit pipe): dW = dO^T . Z
   * needs nothing of the dZ -> encoder-backward chain, so with dw_stream set it is enqueued on that
   * stream behind dw_fork (recorded after the decode) AFTER the chain's launches -- in a stream
   * capture the branch captured first keeps the launching queue, and the chain is the critical
   * one -- and the Adam sweep waits for dw_join.  It then needs a workspace of its own, ws_dw
   * (rk_plan_t.dw3_workspace_bytes).  All NULL: dW runs in line on `stream`, in `ws`. */
  float *ws_dw;
  void *dw_stream, *dw_fork, *dw_join;
  /* nullable: operand planes (rk_planes_layout) -- the step then splits W_de[items] inside its
   * encoder-forward launch, Z in that kernel's epilogue (unbounded activations: rk_amax +
   * rk_split_z), and runs the decode and dZ through the *_planes kernels */
  const rk_planes_t *planes;
  /* nullable: scale table of the dO IMAGE (rk_plan_t.pg_scale_floats floats).  Given, whole untied MSE / BCE
   * steps outside the fused decode's domain (h > 256, >= 1024 rows) run their three contractions on the
   * pipelined pair-plane kernels (rk_pg_decode_loss / rk_pg_dz / rk_pg_dw): `dO` then holds the image
   * (do_rows >= round_up(B, 32) rows of ld * 4 bytes -- the fp32 matrix's footprint), the W^T image and
   * the Z^T planes are not made at all, `ws` holds rk_pg_dz's slabs and ws_dw (or ws) rk_pg_dw's. */
  float *do_scales;
  int32_t do_rows;
  /* Sharded dense Adam under users-DP (ZeRO-1; phased steps, RK_STEP_UPDATE): this rank updates rows
   * [zero_lo, zero_hi) of the dense-Adam embedding tables only -- their moments live nowhere else -- from
   * the reduce-scattered DENSE gradient shards zero_g_en / zero_g_de ([zero_hi - zero_lo, h], row zero_lo
   * first; rk_rows_to_dense laid the compact rows out by item id before the reduce-scatter); the caller
   * then all-gathers the updated rows.  The update of a row is the replicated one bit for bit (a dense
   * Adam sweep treats every row independently, reference model.py:135,398-399).  zero_hi == 0: off. */
  int32_t zero_lo, zero_hi;
  const float *zero_g_en, *zero_g_de;
  /* nullable: the decoder bias gradient as a DENSE vector [n_items] (users-DP with per-rank item sets,
   * RK_DP_ITEMSETS=local: the ranks' compact columns differ, so every gradient travels laid out by item id);
   * the bias job of RK_STEP_UPDATE then reads it row by row instead of gb_de through the block's pos map */
  const float *zero_gb_de;
  /* Lazy dense Adam of the two embedding tables (rk_adam_job_t.lazy_stamp; whole replayed steps, dense Adam):
   * stamps of W_en / W_de ([n_items] int32 each; lazy_stamp_de unused with tied weights), the NEXT step's
   * pos map (NULL: this step leaves every row up to date), the round-robin period.  lazy_stamp_en == NULL: off. */
  int32_t *lazy_stamp_en, *lazy_stamp_de;
  const int32_t *lazy_pos_next;
  int32_t lazy_period;
  /* nullable, with lazy_pos_next: the need list of this block and the next one (rk_lazy_need_lists; both tables share
   * it: they are swept through the same two maps) */
  const int32_t *lazy_need_list, *lazy_need_count;
} rk_ae_step_t;

/*
 * The collectives of the data-parallel step (users sharded over the GPUs of a node; one process per GPU): thin RCCL
 * wrappers, enqueued IN ORDER on the caller's stream -- inside a stream capture too (they are kernels) -- so that a
 * phased step is gradient launches and exchanges on one stream with nothing in between.  The reference has no
 * multi-device code (its backward + update: model.py:397-402).  RCCL is bound at run time (dlopen): `librccl` names
 * the librccl.so to use (a PyTorch process: the one torch loaded), NULL: the loaded / default one.
 *   rk_comm_unique_id  rank 0 draws the 128-byte id; the caller hands it to the other ranks (any side channel)
 *   rk_comm_init       every rank, with the same id -> communicator handle (NULL: rk_last_error)
 *   rk_allreduce_bucket  n buckets in place; several buckets = ONE RCCL group (one fused launch)
 *   rk_reduce_scatter  recv[i] = sum over ranks of send[rank * recv_count + i]
 *   rk_all_gather      recv[r * send_count + i] = rank r's send[i]
 *   rk_exchange        grouped point-to-point: sends[q] (send_counts[q] elements) to rank q, recvs[q] from rank q
 */
enum { RK_COMM_F32 = 0, RK_COMM_I32 = 1 };
enum { RK_COMM_SUM = 0, RK_COMM_MAX = 1 };
int rk_comm_unique_id(void *id128, const char *librccl);
void *rk_comm_init(const void *id128, int32_t world, int32_t rank, const char *librccl);
void rk_comm_destroy(void *comm);
int rk_allreduce_bucket(void *comm, void *const *bufs, const int64_t *counts, int32_t n, int32_t dtype, int32_t op,
                        void *stream);
int rk_reduce_scatter(void *comm, const void *send, void *recv, int64_t recv_count, int32_t dtype, void *stream);
int rk_all_gather(void *comm, const void *send, void *recv, int64_t send_count, int32_t dtype, void *stream);
int rk_exchange(void *comm, int32_t world, void *const *sends, const int64_t *send_counts, void *const *recvs,
                const int64_t *recv_counts, int32_t dtype, void *stream);

void *rk_event_create(int32_t timing); /* 0: ordering-only (no timing, device-scope fence); 1: for time_ev0 /
                                          time_ev1 and rk_event_elapsed_ms */
void rk_event_destroy(void *event);
float rk_event_elapsed_ms(void *ev0, void *ev1);   /* synchronises on ev1 */
int rk_ae_train_step(const rk_ae_step_t *step);
/* != 0: the step as described runs rk_pg_decode_loss / rk_pg_dz / rk_pg_dw (rk_ae_step_t.do_scales).  Bits 0-3: 1 = the
 * pipelined pair-plane kernels, 3 = the register-resident fused decode + rk_pg_dw; bit 4 (16):
 * the decoder bias gradient of the step is left as K slabs in gb_part -- gb_part[s * n_cap + c], s < counts[4] -- (the
 * dW tiles' output column h against a ones column of the Z image), not in gb_de */
int32_t rk_ae_step_uses_pg(const rk_ae_step_t *step);

/*
 * Graph replay of the hot loop.  rk_collate_at = rk_collate (phase 0) on the users
 * users_base[(cursor[0] - cursor[1] + off) * S ...] with the stamp of global step cursor[0] + off;
 * rk_cursor_set maintains the cursor with a 1-thread launch (in order on the
 * stream, capturable); rk_adam_consts fills one entry of the per-step constants table on the
 * host.  rk_graph_* wrap hipStreamBeginCapture / EndCapture / hipGraphInstantiate / Launch for a
 * caller that holds raw hipStream_t values: everything enqueued on `stream` (and on streams
 * forked from it with events) between begin and end becomes one replayable launch.
 */
int rk_collate_at(const int64_t *ds_indptr, const int32_t *ds_indices, const float *ds_data,
                  const int64_t *users_base, int32_t S, int32_t negative_sampling,
                  const int64_t *cursor, int32_t off, const rk_block_t *blk, void *stream);
/* the same for n_blk <= RK_COLLATE_MULTI equally shaped blocks in ONE set of launches; block g
 * takes cursor offset off0 + g */
#define RK_COLLATE_MULTI 8
/* phase as rk_collate (1: row pointers + item marking; 2: the rest): data-parallel replay puts the
 * MAX all-reduce of the blocks' mark arrays between the two, inside the capture */
int rk_collate_at_multi(const int64_t *ds_indptr, const int32_t *ds_indices, const float *ds_data,
                        const int64_t *users_base, int32_t S, int32_t negative_sampling,
                        const int64_t *cursor, int32_t off0, const rk_block_t *const *blks,
                        int32_t n_blk, int32_t phase /* 0: everything */, void *stream);
/* The need lists of the lazy dense Adam (rk_adam_job_t.lazy_need_list) for the steps of n_blk consecutive collated
 * blocks: lists[i] (capacity n_items) = the item ids that block i OR block i + 1 holds (pos >= 0), ascending, and
 * counts[i][0] their number, i = 0 .. n_blk - 2 -- the rows step i's sweep has to touch besides its round-robin chunk.
 * One launch, no temporaries; catalogues of at most RK_NEED_LIST_MAX_ITEMS items (beyond: no lists, the sweep scans). */
#define RK_NEED_LIST_MAX_ITEMS (64 * RK_SCAN_CHUNK)
int rk_lazy_need_lists(const rk_block_t *const *blks, int32_t n_blk, int32_t *const *lists, int32_t *const *counts,
                       void *stream);
int rk_cursor_set(int64_t *cursor, int64_t step, int64_t epoch_base, void *stream);
int rk_adam_consts(double lr, double beta1, double beta2, double eps, double weight_decay,
                   int32_t step, int32_t n_steps, int32_t stride_floats, float *out_host);
/* (entries of steps step .. step + n_steps - 1, 8 floats each, stride_floats apart) */
/*
 * Graph replay of steps that are sequenced ENTRY BY ENTRY from the host (hidden stacks, bottleneck
 * dropout, MatrixFactorization: everything outside rk_ae_train_step).  Between rk_replay_set and
 * rk_replay_set(NULL) (thread-local) the entry points below take what changes from step to step from
 * the device-resident cursor instead of their host arguments, so that a captured sequence of them
 * can be replayed:
 *   rk_ae_encode_fwd                       rng_step = cursor[0] + off + 1; users = users_base + local * B
 *   rk_dropout                             rng_step = cursor[0] + off + 1
 *   rk_gather_rows_amax / rk_scatter_pos / rows / idx64 = users_base + local * B  (the pointer argument
 *   rk_adam_rows (idx64)                   is ignored)
 *   rk_adam_rows / rk_adam_multi           Adam constants = adam_table[local * tab_stride + slot]
 *                                          (slot: the call's `step` argument / rk_adam_job_t.par.step
 *                                          are read as the parameter's SLOT, >= 1 -> slot = value - 1);
 *                                          rk_adam_multi: loss_out is the BASE of the epoch's loss
 *                                          buffer (slot `local`), and the launch carrying the loss
 *                                          publishes cursor_next = {cursor[0] + advance, cursor[1]}
 * with local = cursor[0] - cursor[1] + off (the step's index in its epoch).
 */
typedef struct rk_replay {
  const int64_t *cursor;
  int32_t off;
  int32_t B;                 /* rows of a step (whole batches only) */
  const int64_t *users_base; /* the epoch's user order */
  const void *adam_table;    /* [steps][tab_stride] entries of 8 floats (rk_adam_consts) */
  int32_t tab_stride;
  int32_t advance;           /* with cursor_next */
  int64_t *cursor_next;      /* nullable */
} rk_replay_t;
void rk_replay_set(const rk_replay_t *ctx);   /* NULL: leave the replay context */
int rk_graph_begin(void *stream);
void *rk_graph_end(void *stream);              /* -> executable graph handle, NULL on error */
int rk_graph_launch(void *graph_exec, void *stream);
/* != 0: rk_ae_train_step's timing events (time_ev0 / time_all) may be used INSIDE a capture -- they
 * become event-record nodes that every replay re-records.  Depends on the HIP runtime in the process:
 * hipEventRecordWithFlags(hipEventRecordExternal) where it is accepted (ROCm 7.2); with
 * the probe header's RK_TUNE_GRAPH_EVENT_NODES: also nodes added with hipGraphAddEventRecordNode at the capture's current
 * dependencies (works on the 7.0 runtime PyTorch bundles; off by default: no faster than the eager
 * brackets there and its intervals read longer); rk_graph_event_node_probe runs that route once on a scratch stream and returns the
 * interval (ms) it read between two such nodes, or a negative code. */
void rk_graph_destroy(void *graph_exec);
/* cross-stream edges inside a capture (fork / join): event from rk_event_create */
int rk_event_record(void *event, void *stream);
int rk_stream_wait_event(void *stream, void *event);

/*
 * rk_topk_masked -- Recoder.recommend (model.py:525-544): scores[B, ld] with the seen items (bits_rc of
 * the non-sampled block; only positive stored interactions, model.py:537 `output[input > 0]`) set to -inf,
 * top-k sorted descending (ties: lower index first, as torch.topk on CPU).  Score column c is item
 * col_off + c * col_stride (mask lookup and returned indices are global); row r's k results go to
 * out_idx / out_val [r * out_ld ...].
 *   whole rows          : col_off 0, col_stride 1, out_ld k
 *   a STRIP of the catalogue (col_stride 1): Recoder.recommend decodes a bounded strip of items at a
 *     time, keeps each strip's top k and merges them with one more call (seen == NULL) -- the
 *     [B, n_items] score matrix never exists
 *   a STRIDED SAMPLE (col_stride > 1): step 1 of the fused form below
 */
int rk_topk_masked(const float *scores, int32_t B, int32_t n, int32_t ld, const rk_block_t *seen,
                   int32_t row_off, int32_t k, int32_t col_off, int32_t col_stride,
                   int64_t *out_idx /* nullable */, float *out_val /* nullable */, int32_t out_ld, void *stream);
/*
 * The fused form (catalogues of more than one strip): no score matrix at all.
 *   1. rk_topk_masked on the scores of a STRIDED SAMPLE of the catalogue (score column c is
 *      item col_off + c * col_stride): the k-th best sampled score of a row is a lower bound of the
 *      k-th best score of its whole row;
 *   2. rk_decode_filter_planes: the decode over the whole catalogue from plane images (rk_split_image
 *      of the encoder output and -- once per evaluation -- of the decoder table), whose epilogue
 *      keeps only the unseen entries that reach their row's bound: (score, item id) pairs appended
 *      to the row's candidate list (cand_cnt[r] counts them, also past cand_cap);
 *   3. rk_topk_pairs: the k best pairs of every row by (score descending, id ascending) -- exactly
 *      torch.topk's choice on the masked full row.  status |= 1: a list overflowed, |= 2: a row has
 *      fewer than k candidates (k above its unseen items); the caller falls back to the strips.
 */
int rk_split_image(const float *X, int64_t rows, int32_t K, int64_t ld, const int32_t *amax /* nullable */,
                   float dflt_scale, void *image, float *scales, int32_t slot, void *stream);
int rk_decode_filter_planes(const void *zimg, const void *wimg, const float *scales, int32_t h, int32_t B,
                            int32_t n, int32_t col_off, const float *b_de /* nullable */,
                            const rk_block_t *seen /* nullable */, int32_t row_off, const float *thr,
                            float *cand_val, int32_t *cand_idx, int32_t *cand_cnt, int32_t cand_cap,
                            const int32_t *n_dev, void *stream);
int rk_topk_pairs(const float *val, const int32_t *idx, const int32_t *cnt, int32_t B, int32_t cap,
                  int32_t k, int64_t *out_idx, int32_t out_ld, int32_t *status, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RECODER_HIP_H */
