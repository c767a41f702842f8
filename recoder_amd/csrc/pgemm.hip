// C ABI of the pipelined pair-plane contractions (csrc/pgemm.h, csrc/pgemm_epi.h): the decoder's three
// products of a training step on THREE operand images and nothing else,
//
//   rk_pg_decode_loss : O = Z . W_de[T]^T + b (reference nn.py:271-280) + MSE / BCE loss epilogue
//                       (losses.py:43-47, torch BCEWithLogits); dLoss/dLogits leaves as a plane image of
//                       fp16 pairs with per-tile scales (never as fp32)
//   rk_pg_dz          : dZ = dO . W_de[T]      (autograd of F.linear w.r.t. its input)   A = dO image,
//                       B = the W image read along its rows (LDS transpose reads): no W^T image
//   rk_pg_dw          : dW = dO^T . Z          (autograd of F.linear w.r.t. the weight)  both operands
//                       read along their rows: no transposed dO, no Z^T planes
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "encoder_bwd.h"
#include "pgemm_epi.h"
#include "planes.h"
#include "splitk_reduce.h"

int rk_splitk_reduce_tiles(const float *ws, int M, int N, const int32_t *Kdev, int max_splits, int tile_k,
                           const float *Zact, int act, float *out, void *stream);

namespace {

inline bool al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

// Tile shapes follow the BATCH size alone: the item capacity of a block says nothing about its live item
// set (C3: 41 k items of capacity, 8.4 k live per batch -- 128 x 256 decode tiles left half the chip
// idle, a 256-row dW tile's 128 KB of LDS left the encoder backward of the same launch one workgroup
// per CU: 0.290 vs 0.233 ms per step), and the live count only exists on the device.
inline void decode_tile(int B, int n_cap, int &bm, int &bn) {
  const int force = rk_tune_get(RK_TUNE_PG_TILE);   // (tuning)
  (void)n_cap;
  if (force == 256 || (force == 0 && B >= 1024)) { bm = 256; bn = 256; return; }
  if (force == 1282) { bm = 128; bn = 256; return; }
  bm = 128; bn = 128;
}

inline int dz_splits_for(int B, int h, int bm, int bn) {
  const int tiles = rk_cdiv(B, bm) * rk_cdiv(h, bn);
  int s = 512 / std::max(1, tiles);
  return std::min(64, std::max(1, s));
}
inline void dz_tile(int B, int h, int &bm, int &bn) {
  // (below 1024 rows 128 x 128 with twice the split-K: C5-shaped at B = 500, h = 512: 83 vs 103 us stand-alone
  // -- profiles/r04_pgemm_probe.txt c5b500 dZ cfg 1 / cfg 2 -- and 0.736 vs 0.756 ms per step)
  bn = (h <= 128 || B < 1024) ? 128 : 256;
  bm = B >= 1024 ? 256 : 128;
}
inline void dw_tile(int B, int h, int n_cap, int &bm, int &bn) {
  // large batches (K = B): 256 x 256 (or x 128) tiles; else 64 x 128 on 4 waves -- 48 KB of LDS, so that
  // the encoder backward's workgroups of the same launch (rk_pg_dw_encode_bwd) still run three to a CU
  (void)n_cap;
  if (B >= 1024) { bm = 256; bn = h <= 128 ? 128 : 256; return; }
  bm = 64; bn = 128;
}

// dW tiles || encoder-backward columns in ONE launch (as dw3.hip's dw_encbwd_kernel): they need only what
// the launches in front of them left (the dO image + Z image; dZ0 + the block) and write disjoint outputs
struct EncBwd {
  rk_block_t b;
  int row_off, B;
  const float *dZ;
  int h;
  float *G, *gb;
  int n_gb;
};
__global__ __launch_bounds__(256) void mnll_merge_kernel(const pg::MnllMerge a) { pg::mnll_merge_body(a, (int)blockIdx.x); }

// the decoder bias gradient gb[n] = sum_m dO[m][n] straight from the dO IMAGE (the register-resident fused
// decode -- csrc/fdecode.hip -- has no column sums to give): a third workgroup range of the launch, 32
// columns (one image line per row) per workgroup; a thread = (row slot of 64, 8 columns), rows in fixed
// order, the 64 row slots combined in fixed order through LDS
struct ColsumImg {
  const char *img;
  const int32_t *counts;      // [0] live columns, [2] ld
  const float *tab;
  int gr, gc, pitch, rows;    // scale granule, table pitch, rows (users)
  float *out;                 // [n_cap]
};
__device__ __forceinline__ void colsum_img_body(const ColsumImg &c, const int block, char *smem) {
  const int n_live = c.counts[0], ld = c.counts[2];
  const int n0 = block * 32;
  if (n0 >= n_live) return;
  const int tid = threadIdx.x, rs = tid >> 2, q = tid & 3;       // row slot, 8-column chunk
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  const int64_t pitch = (int64_t)ld * 4;
  for (int m = rs; m < c.rows; m += 64) {
    const char *line = c.img + m * pitch + (int64_t)block * 128 + q * 16;
    const uint4 hi = *reinterpret_cast<const uint4 *>(line), lo = *reinterpret_cast<const uint4 *>(line + 64);
    const float is = 1.0f / c.tab[(m / c.gr) * c.pitch + n0 / c.gc];
    const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const rkp::f16x2 h2 = __builtin_bit_cast(rkp::f16x2, hw[u]), l2 = __builtin_bit_cast(rkp::f16x2, lw[u]);
      acc[2 * u] += ((float)h2[0] + (float)l2[0]) * is;
      acc[2 * u + 1] += ((float)h2[1] + (float)l2[1]) * is;
    }
  }
  float *part = reinterpret_cast<float *>(smem);            // [64 row slots][32 columns]
#pragma unroll
  for (int u = 0; u < 8; ++u) part[rs * 32 + q * 8 + u] = acc[u];
  __syncthreads();
  if (tid < 32 && n0 + tid < n_live) {
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += part[k * 32 + tid];
    c.out[n0 + tid] = s;
  }
}

// RING: 0 = the two-stage loop as the compiler schedules it; 2 / 4 = LDS stages of the ring loop (csrc/pgemm.h)
// PLAIN: RK_GEMM_PREC=bf16 -- both images hold one bf16 value per element (csrc/pgemm.h)
template <int BM, int BN, int WM, int WN, int HV, int RING = 0, bool PLAIN = false>
__global__ __launch_bounds__(WM * WN * 64) void dw_encbwd_kernel(const pg::Core p, const pg::EpiSlab::Args e,
                                                                const int n_dw, const EncBwd enc,
                                                                const ColsumImg cs, const int n_cs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < n_dw) {
    pg::gemm_body<BM, BN, WM, WN, true, true, pg::EpiSlab, RING == 2 ? 256 : 0, true, RING < 3 ? 2 : RING, PLAIN>(p, e, (int)blockIdx.x, smem);
    return;
  }
  if (threadIdx.x >= 256) return;
  if ((int)blockIdx.x < n_dw + n_cs) {
    colsum_img_body(cs, (int)blockIdx.x - n_dw, smem);
    return;
  }
  ae_encode_bwd_cols_body<HV, true>(enc.b, enc.row_off, enc.B, enc.dZ, enc.h, enc.G, 0, enc.gb, enc.n_gb,
                                    (int)blockIdx.x - n_dw - n_cs, smem);
}
// dW tiles || the image's column sums || the dZ SLAB REDUCE of the fused decode in ONE launch (MatrixFactorization
// steps: nothing between the decode and the Adam sweep reads dZ -- the user rows' gradient --, so its reduce need
// not be a link of the chain; csrc/dw3.hip's dw_reduce_kernel did the same for the round-3 decode): 256-thread
// workgroups throughout, the reduce body with 4 waves per 64 outputs
template <int BM, int BN, int WM, int WN, int RING = 0>
__global__ __launch_bounds__(WM * WN * 64) void dw_red_kernel(const pg::Core p, const pg::EpiSlab::Args e, const int n_dw,
                                                             const ColsumImg cs, const int n_cs, const rkred::Args red) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(WM * WN * 64 == 256, "the reduce range runs rkred::body<4>");
  if ((int)blockIdx.x < n_dw) {
    pg::gemm_body<BM, BN, WM, WN, true, true, pg::EpiSlab, RING == 2 ? 256 : 0, true, RING < 3 ? 2 : RING>(p, e, (int)blockIdx.x, smem);
    return;
  }
  if ((int)blockIdx.x < n_dw + n_cs) {
    colsum_img_body(cs, (int)blockIdx.x - n_dw, smem);
    return;
  }
  rkred::body<4>(red, (int)blockIdx.x - n_dw - n_cs, reinterpret_cast<float4 (*)[64]>(smem));
}
constexpr int DW_MAX_SPLITS = 4, DW_SLOTS = 256;    // (the slab count follows the LIVE item count: pg::Core.auto_slots;
                                                    // 512 / 768 slots, round 5: C2's merged launch 24.7 -> 54-57 us, C4's 20.9 -> 23.5-25.7)

}  // namespace

// RK_PG=0: the round-3 plane kernels everywhere (A/B switch)
extern "C" int32_t rk_pg_enabled(void) {
  static const int on = [] { const char *e = getenv("RK_PG"); return (e && atoi(e) == 0) ? 0 : 1; }();
  return on;
}

// scale granule of rk_pg_decode_loss's image: 64 rows x 32 columns
extern "C" void rk_pg_decode_granule(int32_t B, int32_t n_cap, int32_t *gr, int32_t *gc) {
  (void)B; (void)n_cap;
  *gr = 64; *gc = 32;
}

// floats of a scale table that fits every producer (64-row x 32-column granules at the least)
extern "C" int64_t rk_pg_scale_floats(int32_t B_cap, int32_t n_cap) {
  return (int64_t)rk_cdiv(B_cap, 64) * rk_cdiv(n_cap, 32) + 64;
}

extern "C" int rk_pg_decode_loss(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                                 const float *b_de, int32_t loss_kind, float confidence, float inv_B,
                                 void *dO_img, int32_t rows_img, float *dO_scales, float *dO_f32,
                                 float *loss_part, float *gb_part, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && B <= pl->B_cap && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(loss_kind == RK_LOSS_MSE || loss_kind == RK_LOSS_BCE, "mse / logistic epilogues");
  RK_REQUIRE(tgt->implicit || tgt->pref_rc != nullptr, "explicit values need pref_rc");
  RK_REQUIRE(dO_img && dO_scales && al16(dO_img) && rows_img >= ((B + 31) & ~31), "dO image: 16-byte aligned, round_up(B, 32) rows");
  if (B == 0) return 0;
  int bm, bn;
  decode_tile(B, tgt->n_cap, bm, bn);
  // (the logistic epilogue over 8 accumulator tiles per wave is more code than hipcc unrolls: its
  // accumulators went to scratch memory -- 128 x 256 tiles there)
  if (bm == 256 && loss_kind == RK_LOSS_BCE) bm = 128;
  const int KT = rkp::kp_of(pl->h) / 32;
  pg::Core p = {};
  p.a.img = (const char *)pl->z; p.a.pitch = (int64_t)KT * pg::LINE; p.a.lines = KT; p.a.rows = pl->B_cap;
  p.b.img = (const char *)pl->w; p.b.pitch = (int64_t)KT * pg::LINE; p.b.lines = KT; p.b.rows = pl->n_cap;
  p.M = B; p.N = tgt->n_cap; p.K = pl->h; p.Ndev = tgt->counts; p.splits = 1;
  pg::LossArgs e = {};
  e.blk = *tgt; e.row_off = row_off; e.confidence = confidence; e.inv_B = inv_B;
  e.bias = b_de; e.bidx = tgt->items; e.scales = pl->scales;
  e.loss_part = loss_part; e.gb_part = gb_part;
  e.dimg = (char *)dO_img; e.ld_dev = tgt->counts + 2; e.rows_img = rows_img;
  e.dscale = dO_scales; e.ds_pitch = rk_cdiv(tgt->n_cap, 32);
  e.C = dO_f32;
  const int tiles = rk_cdiv(B, bm) * rk_cdiv(tgt->n_cap, bn);
  hipError_t rc;
#define GO(BM, BN, WM, WN)                                                                                       \
  rc = loss_kind == RK_LOSS_MSE                                                                                  \
           ? pg::launch<BM, BN, WM, WN, false, false, pg::EpiLoss<pg::LOSS_MSE>>(p, e, tiles, stream)            \
           : pg::launch<BM, BN, WM, WN, false, false, pg::EpiLoss<pg::LOSS_BCE>>(p, e, tiles, stream)
  if (bm == 256) rc = pg::launch<256, 256, 2, 4, false, false, pg::EpiLoss<pg::LOSS_MSE>>(p, e, tiles, stream);
  else if (bn == 256) GO(128, 256, 2, 4); else GO(128, 128, 2, 2);
#undef GO
  if (rc != hipSuccess) { rk_set_error("pg_decode_loss: %s", hipGetErrorString(rc)); return -1; }
  return 0;
}

// ---- multinomial NLL: statistics pass + decode / loss pass (pgemm_epi.h EpiStats, EpiLoss<LOSS_MNLL>) ----
extern "C" int64_t rk_pg_mnll_workspace_floats(int32_t B, int32_t n_cap) {
  return (int64_t)B * (2 * rk_cdiv(n_cap, 128) + 2) + 64;
}

extern "C" int rk_pg_decode_mnll(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                                 const float *b_de, float inv_B, float *mnll_ws, void *dO_img, int32_t rows_img,
                                 float *dO_scales, float *dO_f32, float *loss_part, float *gb_part,
                                 void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && B <= pl->B_cap && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(tgt->implicit || tgt->pref_rc != nullptr, "explicit values need pref_rc");
  RK_REQUIRE(mnll_ws && dO_img && dO_scales && al16(dO_img) && rows_img >= ((B + 31) & ~31),
             "dO image: 16-byte aligned, round_up(B, 32) rows; a statistics workspace");
  if (B == 0) return 0;
  int bm, bn;
  decode_tile(B, tgt->n_cap, bm, bn);
  if (bm == 256) bm = 128;               // (the multinomial epilogue: as the logistic one, 4 tiles per wave)
  const int KT = rkp::kp_of(pl->h) / 32;
  pg::Core p = {};
  p.a.img = (const char *)pl->z; p.a.pitch = (int64_t)KT * pg::LINE; p.a.lines = KT; p.a.rows = pl->B_cap;
  p.b.img = (const char *)pl->w; p.b.pitch = (int64_t)KT * pg::LINE; p.b.lines = KT; p.b.rows = pl->n_cap;
  p.M = B; p.N = tgt->n_cap; p.K = pl->h; p.Ndev = tgt->counts; p.splits = 1;
  const int pitch = rk_cdiv(tgt->n_cap, bn);
  float *stats = mnll_ws, *rowst = mnll_ws + (int64_t)B * pitch * 2;
  pg::StatsArgs sa = {};
  sa.blk = *tgt; sa.row_off = row_off; sa.bias = b_de; sa.bidx = tgt->items; sa.scales = pl->scales;
  sa.stats = stats; sa.pitch = pitch;
  pg::MnllMerge mm = {};
  mm.blk = *tgt; mm.row_off = row_off; mm.B = B; mm.Ndev = tgt->counts; mm.bn = bn; mm.stats = stats; mm.pitch = pitch;
  mm.rowst = rowst;
  pg::LossArgs e = {};
  e.blk = *tgt; e.row_off = row_off; e.confidence = 0.f; e.inv_B = inv_B;
  e.bias = b_de; e.bidx = tgt->items; e.scales = pl->scales;
  e.loss_part = loss_part; e.gb_part = gb_part;
  e.dimg = (char *)dO_img; e.ld_dev = tgt->counts + 2; e.rows_img = rows_img;
  e.dscale = dO_scales; e.ds_pitch = rk_cdiv(tgt->n_cap, 32);
  e.C = dO_f32;
  e.rowst_g = rowst;
  const int tiles = rk_cdiv(B, bm) * rk_cdiv(tgt->n_cap, bn);
  hipError_t rc;
  if (bn == 256) rc = pg::launch<128, 256, 2, 4, false, false, pg::EpiStats>(p, sa, tiles, stream);
  else rc = pg::launch<128, 128, 2, 2, false, false, pg::EpiStats>(p, sa, tiles, stream);
  if (rc == hipSuccess) {
    hipLaunchKernelGGL(mnll_merge_kernel, dim3(rk_cdiv(B, 4)), dim3(256), 0, stream, mm);
    rc = hipGetLastError();
  }
  if (rc == hipSuccess) {
    if (bn == 256) rc = pg::launch<128, 256, 2, 4, false, false, pg::EpiLoss<pg::LOSS_MNLL>>(p, e, tiles, stream);
    else rc = pg::launch<128, 128, 2, 2, false, false, pg::EpiLoss<pg::LOSS_MNLL>>(p, e, tiles, stream);
  }
  if (rc != hipSuccess) { rk_set_error("pg_decode_mnll: %s", hipGetErrorString(rc)); return -1; }
  return 0;
}

extern "C" int64_t rk_pg_dz_workspace_bytes(int32_t B, int32_t h) {
  // the slab count is not monotone in the batch size (a ragged last batch of 1024 rows takes 64 slabs where
  // 1100 rows take 51): the workspace of a capacity B covers every batch size up to it
  int64_t rows = 0;
  for (int b = 1; b <= B; ++b) {
    int bm, bn;
    dz_tile(b, h, bm, bn);
    rows = std::max(rows, (int64_t)dz_splits_for(b, h, bm, bn) * b);
  }
  return rows * h * (int64_t)sizeof(float);
}

extern "C" int rk_pg_dz(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                        const rk_planes_t *pl, const rk_block_t *tgt, const float *Zact, int32_t act,
                        float *dZ, float *workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(al16(dO_img) && al16(workspace) && al16(dZ) && dO_scales, "operands must be 16-byte aligned");
  RK_REQUIRE(gr >= 32 && gc >= 32 && gr % 32 == 0 && gc % 32 == 0, "scale granule: multiples of 32");
  if (B == 0) return 0;
  const int h = pl->h;
  int bm, bn;
  dz_tile(B, h, bm, bn);
  const int KT = rkp::kp_of(h) / 32;
  pg::Core p = {};
  p.a.img = (const char *)dO_img; p.a.rows = B; p.a_ld_dev = tgt->counts + 2;
  p.a.lines = rk_cdiv(tgt->n_cap, 32); p.a.pitch = (int64_t)p.a.lines * pg::LINE;      // (replaced on the device)
  p.b.img = (const char *)pl->w; p.b.pitch = (int64_t)KT * pg::LINE; p.b.lines = KT; p.b.rows = pl->n_ld;
  p.M = B; p.N = h; p.K = tgt->n_cap; p.Kdev = tgt->counts;
  p.splits = dz_splits_for(B, h, bm, bn);
  p.rs.tab = dO_scales; p.rs.gr = gr; p.rs.gc = gc; p.rs.pitch = rk_cdiv(tgt->n_cap, gc); p.rs.mode = 1;
  pg::EpiSlab::Args e = {};
  e.C = workspace; e.ldc = h; e.slab_stride = (int64_t)B * h; e.bscale = pl->scales + 1;
  const int tiles = rk_cdiv(B, bm) * rk_cdiv(h, bn);
  hipError_t rc;
  // (RK_TUNE_DW_RING != 0: the ring k-loop of csrc/pgemm.h -- the W image's transpose reads as asm, so that the next
  // tile's DMA lands under the MFMAs instead of being drained in front of the first read)
  const bool ring = rk_tune_get(RK_TUNE_DW_RING) != 0;
  if (bm == 256 && bn == 256) rc = ring ? pg::launch<256, 256, 2, 4, false, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                                        : pg::launch<256, 256, 2, 4, false, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  else if (bm == 256) rc = ring ? pg::launch<256, 128, 4, 2, false, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                                : pg::launch<256, 128, 4, 2, false, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  else if (bn == 256) rc = ring ? pg::launch<128, 256, 2, 4, false, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                                : pg::launch<128, 256, 2, 4, false, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  else rc = ring ? pg::launch<128, 128, 2, 2, false, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                 : pg::launch<128, 128, 2, 2, false, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  if (rc != hipSuccess) { rk_set_error("pg_dz: %s", hipGetErrorString(rc)); return -1; }
  // every slab is written (an empty K range leaves zeros): summing min(K, splits) of them is the sum
  return rk_splitk_reduce_tiles(workspace, B, h, tgt->counts, p.splits, 1, Zact, act, dZ, stream_);
}

// the most K slabs rk_pg_dw writes; the number it did write is published in tgt->counts[4]
extern "C" int32_t rk_pg_dw_splits(int32_t B, int32_t h, int32_t n_cap) {
  (void)B; (void)h; (void)n_cap;
  return DW_MAX_SPLITS;
}
extern "C" int64_t rk_pg_dw_workspace_bytes(int32_t B, int32_t h, int32_t n_cap) {
  return (int64_t)rk_pg_dw_splits(B, h, n_cap) * n_cap * h * sizeof(float);
}

// slabs [counts[4] <= rk_pg_dw_splits][n_cap][h]: the Adam sweep adds them as it reads the gradient
// (g_parts / gparts_dev)
static int pg_dw_impl(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                      const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, const EncBwd *enc,
                      void *stream_, float *gb_de = nullptr, bool dense = false, const rkred::Args *red = nullptr,
                      bool ones = false);

extern "C" int rk_pg_dw(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                        const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, float *gb_de, void *stream_) {
  return pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, slabs, nullptr, stream_, gb_de);
}

// rk_pg_dw as ONE dense array G_de[n_cap][h] (no split-K: the data-parallel exchange ships it) + the
// decoder bias gradient gb_de[n_t] = column sums of dO from the image, one launch (csrc/step.hip: the
// phased step on the register-resident fused decode); internal
int rk_pg_dw_dense(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                   const rk_planes_t *pl, const rk_block_t *tgt, float *G_de, float *gb_de, void *stream_) {
  return pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, G_de, nullptr, stream_, gb_de, true);
}

// rk_pg_dw and rk_ae_encode_bwd (rows [row_off, row_off + B) of the block `tgt`; G_en / gb_en as there,
// nothing accumulated) in ONE launch
extern "C" int rk_pg_dw_encode_bwd(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc,
                                   int32_t B, const rk_planes_t *pl, const rk_block_t *tgt, float *slabs,
                                   int32_t row_off, const float *dZ0pre, float *G_en, float *gb_en,
                                   float *gb_de, void *stream_) {
  RK_REQUIRE(rk_dw_encode_bwd_fused_ok(row_off, B) || (rk_gemm_plain_bf16() && (((row_off + B + 31) >> 5) - (row_off >> 5) <= 64)),
             "outside the fused launch's domain (rk_dw_encode_bwd_fused_ok)");
  RK_REQUIRE(pl && pl->h % 4 == 0 && pl->h <= 1024, "h must be a multiple of 4, <= 1024");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(tgt->bits_cr != nullptr && tgt->pref_rc != nullptr,
             "block was built without the transposed bitmap / prefix index");
  {
    // 256 x 256 dW tiles with h > 512 (>= 1024 rows): the merged instantiation needed 266 VGPRs (10 spilled, 44
    // bytes of scratch -- VERDICT r4 weak 10); dW (+ the image's column sums) and the encoder backward as two
    // launches there
    int bm, bn;
    dw_tile(B, pl->h, tgt->n_cap, bm, bn);
    if (bm == 256 && bn == 256 && rk_cdiv(pl->h, 256) > 2) {
      const int rc = pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, slabs, nullptr, stream_, gb_de);
      if (rc) return rc;
      return rk_ae_encode_bwd(tgt, row_off, B, dZ0pre, pl->h, G_en, 0, gb_en, stream_);
    }
  }
  EncBwd enc = {};
  enc.b = *tgt; enc.row_off = row_off; enc.B = B; enc.dZ = dZ0pre; enc.h = pl->h; enc.G = G_en; enc.gb = gb_en;
  enc.n_gb = gb_en ? rk_cdiv(pl->h, 64) : 0;
  return pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, slabs, &enc, stream_, gb_de);
}

// the Z image carries a ones column (csrc/internal.h): the output column h of the dW tiles is the bias gradient
int rk_pg_dw_ones_ok(int32_t B, int32_t h, int32_t n_cap) {
  int bm, bn;
  dw_tile(B, h, n_cap, bm, bn);
  // (a free padding column inside the image's last line; the merged launch -- not the two-launch form of the
  // 256 x 256 tiles with h > 512; fp16-pair operands)
  return (h % 32 != 0 && !(bm == 256 && bn == 256 && rk_cdiv(h, 256) > 2)) ? 1 : 0;
}
int rk_pg_dw_encode_bwd_ones(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                             const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, int32_t row_off,
                             const float *dZ0pre, float *G_en, float *gb_en, float *gb_slabs, void *stream_) {
  RK_REQUIRE(rk_dw_encode_bwd_fused_ok(row_off, B) || (rk_gemm_plain_bf16() && (((row_off + B + 31) >> 5) - (row_off >> 5) <= 64)),
             "outside the fused launch's domain (rk_dw_encode_bwd_fused_ok)");
  RK_REQUIRE(pl && rk_pg_dw_ones_ok(B, pl->h, tgt->n_cap) && gb_slabs, "rk_pg_dw_ones_ok");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(tgt->bits_cr != nullptr && tgt->pref_rc != nullptr,
             "block was built without the transposed bitmap / prefix index");
  EncBwd enc = {};
  enc.b = *tgt; enc.row_off = row_off; enc.B = B; enc.dZ = dZ0pre; enc.h = pl->h; enc.G = G_en; enc.gb = gb_en;
  enc.n_gb = gb_en ? rk_cdiv(pl->h, 64) : 0;
  return pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, slabs, &enc, stream_, gb_slabs, false, nullptr, true);
}

// rk_pg_dw (+ gb_de) with the slab reduce of rk_fdec_loss_dz riding on the launch (< 1024 rows: 64 x 128 tiles)
extern "C" int rk_pg_dw_dz_reduce(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                                  const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, float *gb_de,
                                  const float *dz_workspace, const float *Zact, int32_t act, float *dZ,
                                  int32_t dense, void *stream_) {
  RK_REQUIRE(B < 1024, "rk_pg_dw_dz_reduce: batches below 1024 rows (the 256-thread dW tiles)");
  RK_REQUIRE(al16(dz_workspace) && al16(dZ), "operands must be 16-byte aligned");
  const rkred::Args red = {dz_workspace, B, pl ? pl->h : 0, tgt->counts, 128, rk_fdec_slabs(B, tgt->n_cap), Zact, act, dZ};
  return pg_dw_impl(dO_img, dO_scales, gr, gc, B, pl, tgt, slabs, nullptr, stream_, gb_de, dense != 0, &red);
}

static int pg_dw_impl(const void *dO_img, const float *dO_scales, int32_t gr, int32_t gc, int32_t B,
                      const rk_planes_t *pl, const rk_block_t *tgt, float *slabs, const EncBwd *enc_,
                      void *stream_, float *gb_de, bool dense, const rkred::Args *red, bool ones) {
  hipStream_t stream = (hipStream_t)stream_;
  // (column sums without an encoder backward: the same launch with an empty encoder range)
  EncBwd no_enc = {};
  no_enc.h = pl ? pl->h : 0;
  const EncBwd *enc = enc_ ? enc_ : (gb_de ? &no_enc : nullptr);
  RK_REQUIRE(pl && tgt->n_cap <= pl->n_cap && B <= pl->B_cap, "planes were laid out for another shape");
  RK_REQUIRE(al16(dO_img) && al16(slabs) && dO_scales, "operands must be 16-byte aligned");
  RK_REQUIRE(gr >= 32 && gc >= 32 && gr % 32 == 0 && gc % 32 == 0, "scale granule: multiples of 32");
  if (B == 0) return 0;
  const int h = pl->h;
  int bm, bn;
  dw_tile(B, h, tgt->n_cap, bm, bn);
  const int KT = rkp::kp_of(h) / 32;
  pg::Core p = {};
  p.a.img = (const char *)dO_img; p.a.rows = (B + 31) & ~31; p.a_ld_dev = tgt->counts + 2;
  p.a.lines = rk_cdiv(tgt->n_cap, 32); p.a.pitch = (int64_t)p.a.lines * pg::LINE;
  p.b.img = (const char *)pl->z; p.b.pitch = (int64_t)KT * pg::LINE; p.b.lines = KT; p.b.rows = (pl->B_cap + 31) & ~31;
  p.M = tgt->n_cap; p.Mdev = tgt->counts; p.N = h; p.K = B;
  p.splits = dense ? 1 : DW_MAX_SPLITS; p.auto_slots = dense ? 0 : DW_SLOTS; p.splits_out = tgt->counts + 4;
  p.rs.tab = dO_scales; p.rs.gr = gr; p.rs.gc = gc; p.rs.pitch = rk_cdiv(tgt->n_cap, gc); p.rs.mode = 2;
  pg::EpiSlab::Args e = {};
  e.C = slabs; e.ldc = h; e.slab_stride = (int64_t)tgt->n_cap * h; e.bscale = pl->scales;
  if (ones) { e.gb = gb_de; e.gb_stride = tgt->n_cap; e.gb_col = h; }     // (gb_de: one slab per K slab here)
  const int tiles = rk_cdiv(tgt->n_cap, bm) * rk_cdiv(h, bn);
  hipError_t rc;
  if (red) {
    const int n_dw = pg::grid_of(p, tiles);
    ColsumImg cs = {};
    int n_cs = 0;
    if (gb_de) {
      cs.img = (const char *)dO_img; cs.counts = tgt->counts; cs.tab = dO_scales; cs.gr = gr; cs.gc = gc;
      cs.pitch = rk_cdiv(tgt->n_cap, gc); cs.rows = B; cs.out = gb_de;
      n_cs = rk_cdiv(tgt->n_cap, 32);
    }
    const int n_red = rkred::blocks(red->M, red->N);
    const int ring = rk_tune_get(RK_TUNE_DW_RING);
#define GO_RED(NS)                                                                                            \
  do {                                                                                                        \
    auto k = dw_red_kernel<64, 128, 2, 2, NS>;                                                                \
    constexpr int ST = NS < 3 ? 2 : NS;                                                                       \
    static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (attr != hipSuccess) { rk_set_error("LDS attribute"); return -1; }                                     \
    hipLaunchKernelGGL(k, dim3(n_dw + n_cs + n_red), dim3(256), ST * (64 + 128) * pg::LINE, stream, p, e, n_dw, cs, n_cs, *red); \
    rc = hipGetLastError();                                                                                   \
  } while (0)
    if (ring == 2) GO_RED(2); else if (ring == 4) GO_RED(4); else GO_RED(0);
#undef GO_RED
  } else if (enc) {
    const int n_dw = pg::grid_of(p, tiles);
    const int n_enc = enc_ ? rk_cdiv(tgt->n_cap, 4) + enc->n_gb : 0;
    const int hv = rk_cdiv(h, 256);
    ColsumImg cs = {};
    int n_cs = 0;
    if (gb_de && !ones) {
      cs.img = (const char *)dO_img; cs.counts = tgt->counts; cs.tab = dO_scales; cs.gr = gr; cs.gc = gc;
      cs.pitch = rk_cdiv(tgt->n_cap, gc); cs.rows = B; cs.out = gb_de;
      n_cs = rk_cdiv(tgt->n_cap, 32);
    }
#define GO(BM, BN, WM, WN, HV)                                                                               \
  do {                                                                                                       \
    auto k = dw_encbwd_kernel<BM, BN, WM, WN, HV>;                                                           \
    static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (attr != hipSuccess) { rc = attr; break; }                                                            \
    hipLaunchKernelGGL(k, dim3(n_dw + n_cs + n_enc), dim3(WM * WN * 64), 2 * (BM + BN) * pg::LINE, stream, p, e, n_dw, *enc, cs, n_cs); \
    rc = hipGetLastError();                                                                                  \
  } while (0)
#define BY_HV(BM, BN, WM, WN) do { if (hv == 1) GO(BM, BN, WM, WN, 1); else if (hv == 2) GO(BM, BN, WM, WN, 2); else GO(BM, BN, WM, WN, 4); } while (0)
#define GO_NS(BM, BN, WM, WN, HV, NS)                                                                        \
  do {                                                                                                       \
    auto k = dw_encbwd_kernel<BM, BN, WM, WN, HV, NS>;                                                       \
    constexpr int ST = NS < 3 ? 2 : NS;                                                                      \
    static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (attr != hipSuccess) { rc = attr; break; }                                                            \
    hipLaunchKernelGGL(k, dim3(n_dw + n_cs + n_enc), dim3(WM * WN * 64), ST * (BM + BN) * pg::LINE, stream, p, e, n_dw, *enc, cs, n_cs); \
    rc = hipGetLastError();                                                                                  \
  } while (0)
    const int ring = rk_tune_get(RK_TUNE_DW_RING);
    if (rk_gemm_plain_bf16()) {
      // plain bf16 images (RK_GEMM_PREC=bf16; the fused decode's domain: < 1024 rows, h <= 224): the two-stage ring loop of
      // the 64 x 128 tiles, one product; the bias gradient as the tiles' output column h (no column-sum range: that
      // range reads fp16 pairs)
      if (!(bm == 64 && hv == 1 && n_cs == 0)) { rk_set_error("pg_dw: plain bf16 operands cover the 64 x 128 tiles, h <= 256, bias gradient as column h"); return -1; }
      auto k = dw_encbwd_kernel<64, 128, 2, 2, 1, 2, true>;
      static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (attr != hipSuccess) { rk_set_error("LDS attribute"); return -1; }
      hipLaunchKernelGGL(k, dim3(n_dw + n_cs + n_enc), dim3(256), 2 * (64 + 128) * pg::LINE, stream, p, e, n_dw, *enc, cs, n_cs);
      rc = hipGetLastError();
    } else
    if (bm == 256 && bn == 256) {            // (hv == 4 never gets here: rk_pg_dw_encode_bwd)
      if (ring != 0) { if (hv == 1) GO_NS(256, 256, 2, 4, 1, 2); else GO_NS(256, 256, 2, 4, 2, 2); }
      else { if (hv == 1) GO(256, 256, 2, 4, 1); else GO(256, 256, 2, 4, 2); }
    }
    else if (bm == 256 && ring != 0) { if (hv == 1) GO_NS(256, 128, 4, 2, 1, 2); else if (hv == 2) GO_NS(256, 128, 4, 2, 2, 2); else GO_NS(256, 128, 4, 2, 4, 2); }
    else if (bm == 256) BY_HV(256, 128, 4, 2);
    else if ((ring == 2 || ring == 4) && hv <= 2) {          // (the ring loop: csrc/pgemm.h)
      if (hv == 1) { if (ring == 2) GO_NS(64, 128, 2, 2, 1, 2); else GO_NS(64, 128, 2, 2, 1, 4); }
      else { if (ring == 2) GO_NS(64, 128, 2, 2, 2, 2); else GO_NS(64, 128, 2, 2, 2, 4); }
    }
    else BY_HV(64, 128, 2, 2);
#undef GO_NS
#undef BY_HV
#undef GO
  } else if (bm == 256 && bn == 256) rc = rk_tune_get(RK_TUNE_DW_RING) != 0 ? pg::launch<256, 256, 2, 4, true, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                                                                             : pg::launch<256, 256, 2, 4, true, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  else if (bm == 256) rc = rk_tune_get(RK_TUNE_DW_RING) != 0 ? pg::launch<256, 128, 4, 2, true, true, pg::EpiSlab, 256, true>(p, e, tiles, stream)
                                                              : pg::launch<256, 128, 4, 2, true, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  else if (rk_tune_get(RK_TUNE_DW_RING) == 2) rc = pg::launch<64, 128, 2, 2, true, true, pg::EpiSlab, 256, true, 2>(p, e, tiles, stream);
  else if (rk_tune_get(RK_TUNE_DW_RING) == 4) rc = pg::launch<64, 128, 2, 2, true, true, pg::EpiSlab, 0, true, 4>(p, e, tiles, stream);
  else rc = pg::launch<64, 128, 2, 2, true, true, pg::EpiSlab, 0, true>(p, e, tiles, stream);
  if (rc != hipSuccess) { rk_set_error("pg_dw: %s", hipGetErrorString(rc)); return -1; }
  return 0;
}
