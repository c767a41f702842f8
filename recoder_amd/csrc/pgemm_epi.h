// Epilogues of the pipelined pair-plane contraction (csrc/pgemm.h) that need the library's types:
//   EpiSlab : fp32 slab store of a split-K partial (dZ, dW), the A operand's granule scale divided out
//   EpiLoss : the decoder's fused loss epilogue (reference losses.py:43-47, torch BCEWithLogits) --
//             bias gather, target lookup through the block's bitmap, loss partial, column sums (decoder
//             bias gradient) and dLoss/dLogits written as a PLANE IMAGE of fp16 pairs cut with the
//             TILE's own power-of-two scale (published in a scale table, pg::Rescale): the dZ and dW
//             contractions copy that image into LDS like any other operand -- no fp32 dO is written,
//             no operand is split inside a k-loop any more.
#pragma once
#include "common.h"
#include "pgemm.h"
#include "planes.h"

namespace pg {

struct EpiSlab {
  struct Args {
    float *C;                // [split][M][ldc]
    int64_t ldc;
    int64_t slab_stride;     // floats between K slabs
    const float *bscale;     // device: the B operand's split scale
    float *gb;               // nullable: output column gb_col (a padding column: n == N) goes here instead --
    int64_t gb_stride;       // gb[split * gb_stride + m]: with a ones column in the B image the column sums of A
    int gb_col;
  };
  template <int BM, int BN, int TM, int TN>
  static __device__ __forceinline__ void run(const Args &e, const Tile &T, f32x16 (&acc)[TM][TN], char *,
                                             const float (&cur)[TM]) {
    const float sb = *e.bscale;
    float *C = e.C + (int64_t)T.split * e.slab_stride;
    const int l31 = T.lane & 31, lh = T.lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float inv = 1.0f / (cur[i] * sb);                   // exact: powers of two
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = T.n0 + (T.wn * TN + j) * 32 + l31;
        if (n < T.N) {
          float *col = C + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = T.m0 + (T.wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < T.M) col[(int64_t)m * e.ldc] = acc[i][j][r] * inv;
          }
        } else if (e.gb && n == e.gb_col) {
          float *g = e.gb + (int64_t)T.split * e.gb_stride;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = T.m0 + (T.wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < T.M) g[m] = acc[i][j][r] * inv;
          }
        }
      }
    }
  }
};

enum { LOSS_MSE = 1, LOSS_MNLL = 2, LOSS_BCE = 3 };

// Multinomial NLL (reference losses.py:68-71: -sum t * log_softmax(o)), two passes over the decode:
//   pass 1 (EpiStats): nothing of the logits is written -- every (row, column tile) leaves the pair
//           {max, sum exp(o - max)} of its live columns (8 bytes), tiles of column tile 0 also the row's
//           target sum;
//   pass 2 (EpiLoss<LOSS_MNLL>): the decode again, the row's log-sum-exp merged from the pairs (fixed
//           order), loss + dLoss/dLogits = (softmax * sum_t - t) / B as the plane image like MSE / BCE.
// 3.2 GF of MFMA work a second time instead of writing, re-reading and rewriting a B x n_b fp32 matrix.
struct StatsArgs {
  rk_block_t blk;
  int row_off;
  const float *bias;
  const int32_t *bidx;
  const float *scales;
  float *stats;              // [row][pitch][2]
  int pitch;                 // column tiles of the capacity
};

struct EpiStats {
  typedef StatsArgs Args;
  template <int BM, int BN, int TM, int TN>
  static __device__ __forceinline__ void run(const Args &e, const Tile &T, f32x16 (&acc)[TM][TN], char *smem,
                                             const float (&)[TM]) {
    constexpr int WN = BN / (TN * 32);
    const int M = T.M, N = T.N, lane = T.lane, l31 = lane & 31, lh = lane >> 5;
    const float inv = 1.0f / (e.scales[0] * e.scales[1]);
    float bv[TN];
    bool live[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = T.n0 + (T.wn * TN + j) * 32 + l31;
      live[j] = n < N;
      bv[j] = e.bias[e.bidx[min(n, N - 1)]];
    }
    float *pm = reinterpret_cast<float *>(smem);           // [WN][BM] maxima, then [WN][BM] sums
    float *ps = pm + WN * BM;
    __syncthreads();                                        // (the k-loop's last stage is free)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o[TN];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          o[j] = live[j] ? acc[i][j][r] * inv + bv[j] : -INFINITY;
          m = fmaxf(m, o[j]);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));     // the half-wave's 32 columns
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) se += live[j] ? expf(o[j] - m) : 0.f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
        const int lr = (T.wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (l31 == 0) { pm[T.wn * BM + lr] = m; ps[T.wn * BM + lr] = se; }
      }
    __syncthreads();
    // merge the WN column ranges of the tile (fixed order), one thread per row
    for (int lr = threadIdx.x; lr < BM; lr += (BM / (TM * 32)) * WN * 64) {
      const int m_ = T.m0 + lr;
      if (m_ >= M) continue;
      float mx = pm[lr];
#pragma unroll
      for (int w = 1; w < WN; ++w) mx = fmaxf(mx, pm[w * BM + lr]);
      float se = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) se += pm[w * BM + lr] == -INFINITY ? 0.f : ps[w * BM + lr] * expf(pm[w * BM + lr] - mx);
      float *d = e.stats + ((int64_t)m_ * e.pitch + T.nt) * 2;
      d[0] = mx; d[1] = se;
    }
  }
};

// between the passes: one WAVE per row merges the row's pairs (lanes over the column tiles, online-softmax
// merge in a fixed order) into rowst[row] = {log-sum-exp, sum of the row's targets}
struct MnllMerge {
  rk_block_t blk;
  int row_off, B;
  const int32_t *Ndev;
  int bn;                    // columns of a statistics tile
  const float *stats;
  int pitch;
  float *rowst;              // [B][2]
};
__device__ __forceinline__ void mnll_merge_body(const MnllMerge &a, const int block) {
  const int lane = threadIdx.x & 63, r = block * 4 + (threadIdx.x >> 6);
  if (r >= a.B) return;
  const int ntl = (*a.Ndev + a.bn - 1) / a.bn;
  const float *st = a.stats + (int64_t)r * a.pitch * 2;
  const rk_block_t &b = a.blk;
  const int row = a.row_off + r;
  const int beg = b.indptr[row], end = b.indptr[row + 1];
  float mx = -INFINITY, se = 0.f;
  for (int t = lane; t < ntl; t += 64) {
    const float2 p = *reinterpret_cast<const float2 *>(st + 2 * t);
    if (p.x > mx) { se = se * expf(mx - p.x) + p.y; mx = p.x; }        // (expf(-inf) = 0 the first time)
    else if (p.x > -INFINITY) se += p.y * expf(p.x - mx);
  }
  float ts = 0.f;
  if (!b.implicit)
    for (int k = beg + lane; k < end; k += 64) ts += b.vals[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(mx, off, 64), os = __shfl_xor(se, off, 64);
    const float nm = fmaxf(mx, om);
    se = (mx > -INFINITY ? se * expf(mx - nm) : 0.f) + (om > -INFINITY ? os * expf(om - nm) : 0.f);
    mx = nm;
    ts += __shfl_xor(ts, off, 64);
  }
  if (lane == 0) {
    a.rowst[2 * r] = mx + logf(se);
    a.rowst[2 * r + 1] = b.implicit ? (float)(end - beg) : ts;
  }
}


struct LossArgs {
    rk_block_t blk;
    int row_off;
    float confidence, inv_B;
    const float *bias;         // decoder bias table
    const int32_t *bidx;       // compact column -> table row
    const float *scales;       // [0] Z, [1] W split scales
    float *loss_part, *gb_part;
    char *dimg;                // dO image: row m at m * ld * 4 bytes (ld = *ld_dev)
    const int32_t *ld_dev;
    int rows_img;              // rows of the image that may be written (>= round_up(M, 32))
    float *dscale;             // scale table [row / 64][ds_pitch], one entry per 32 columns
    int ds_pitch;
    float *C;                  // nullable: dO as fp32 too (tests), leading dimension ld
    // LOSS_MNLL: [row][2] = {the row's log-sum-exp, the sum of its targets} (mnll_merge_body)
    const float *rowst_g;
};

template <int LOSS>
struct EpiLoss {
  typedef LossArgs Args;

  template <int BM, int BN, int TM, int TN>
  static __device__ __forceinline__ void run(const Args &e, const Tile &T, f32x16 (&acc)[TM][TN], char *smem,
                                             const float (&)[TM]) {
    constexpr int TLD = 36;
    constexpr int NWAVES = (BM / (TM * 32)) * (BN / (TN * 32));
    const int M = T.M, N = T.N, lane = T.lane;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rr = lane >> 2, c8 = lane & 3;               // row-major role: rows rr, rr + 16; 8 columns
    const rk_block_t &b = e.blk;
    const bool implicit = b.implicit != 0;
    const int ldc = *e.ld_dev;
    float *fsm = reinterpret_cast<float *>(smem);
    float *wlds = fsm + T.wave * (32 * TLD);               // per-wave transposition area
    float *lred = fsm + NWAVES * (32 * TLD);               // [2 * NWAVES] loss / max partials
    float *cpart = lred + 2 * NWAVES;                      // [BM / 32][BN] column sums per 32-row block
    // operands of the loss: gathered bias and bitmap words, all loads in flight before the first use
    float bv[TN][8];
    uint32_t bw[TM][TN][2];
    {
      int gi[TN][8];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          gi[j][u] = e.bidx[min(T.n0 + (T.wn * TN + j) * 32 + c8 * 8 + u, N - 1)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int m = T.m0 + (T.wm * TM + i) * 32 + rr + 16 * it;
            const int row = e.row_off + min(m, M - 1);
            const int nb = T.n0 + (T.wn * TN + j) * 32;
            bw[i][j][it] = b.bits_rc[(int64_t)row * b.ldw_rc + min(nb >> 5, b.ldw_rc - 1)];
          }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[j][u] = e.bias[gi[j][u]];
    }
    const float inv = 1.0f / (e.scales[0] * e.scales[1]);   // exact: powers of two
    __syncthreads();                                        // (the k-loop's last stage is free)
    float *rowst = cpart + (BM / 32) * BN;                  // LOSS_MNLL: [BM][2] = {log-sum-exp, target sum}
    if (LOSS == LOSS_MNLL) {
      for (int lr = threadIdx.x; lr < BM; lr += NWAVES * 64) {
        const float2 v = *reinterpret_cast<const float2 *>(e.rowst_g + 2 * (int64_t)min(T.m0 + lr, M - 1));
        rowst[2 * lr] = v.x; rowst[2 * lr + 1] = v.y;
      }
      __syncthreads();
    }
    // Scale granule = 64 rows x 32 columns (two 32 x 32 blocks of one wave): the maximum of a granule is
    // a WAVE reduction, and only 2 x 16 gradient registers are alive at a time
    static_assert(TM % 2 == 0, "wave tiles of at least 64 rows");
    const int64_t pitch = (int64_t)ldc * 4;
    float lsum = 0.f, gmax_all = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = T.n0 + (T.wn * TN + j) * 32 + c8 * 8;
#pragma unroll
     for (int ih = 0; ih < TM / 2; ++ih) {
      float G[2][16];
      float gmax = 0.f;
#pragma unroll
      for (int i = 2 * ih; i < 2 * ih + 2; ++i) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) wlds[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + l31] = acc[i][j][r] * inv;
        __builtin_amdgcn_wave_barrier();
        float cs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cs[u] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const float4 v0 = *reinterpret_cast<const float4 *>(wlds + (rr + 16 * it) * TLD + c8 * 8);
          const float4 v1 = *reinterpret_cast<const float4 *>(wlds + (rr + 16 * it) * TLD + c8 * 8 + 4);
          const float ov[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          const int m = T.m0 + (T.wm * TM + i) * 32 + rr + 16 * it;
          const uint32_t w = bw[i][j][it];
          float lse = 0.f, tsum = 0.f;
          if (LOSS == LOSS_MNLL) {
            lse = rowst[2 * ((T.wm * TM + i) * 32 + rr + 16 * it)];
            tsum = rowst[2 * ((T.wm * TM + i) * 32 + rr + 16 * it) + 1];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool ok = (m < M) && (n + u < N);
            const float o = ov[u] + bv[j][u];
            float tv = 0.f;
            if (ok && ((w >> (c8 * 8 + u)) & 1u)) {
              tv = 1.0f;
              if (!implicit) tv = b.vals[rk_entry_index(b, e.row_off + m, n + u, w)];
            }
            float l, g;
            if (LOSS == LOSS_MSE) {
              const float wgt = (tv > 0.f) ? (1.0f + e.confidence) : 1.0f;
              const float d = o - tv;
              l = wgt * (d * d);
              g = (2.0f * d) * (wgt * e.inv_B);
            } else if (LOSS == LOSS_MNLL) {       // -t * log_softmax(o); d/do = softmax * sum_t - t
              const float lsm = o - lse;
              l = -tv * lsm;
              g = (expf(ok ? lsm : -INFINITY) * tsum - tv) * e.inv_B;
            } else {  // BCE with logits: (1 - t) * o - logsigmoid(o)
              const float ls = fminf(o, 0.f) - log1pf(expf(-fabsf(o)));
              l = (1.0f - tv) * o - ls;
              const float sg = 1.0f / (1.0f + expf(-o));
              g = (sg - tv) * e.inv_B;
            }
            if (ok) { lsum += l; cs[u] += g; gmax = fmaxf(gmax, fabsf(g)); }
            else g = 0.f;                 // rows past M / the padding columns [N, ld) are ZEROS in the image
            G[i & 1][it * 8 + u] = g;
          }
        }
        __builtin_amdgcn_wave_barrier();
        // column sums over the 32 rows of this block: lanes with equal c8 hold different rows
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          cs[u] += __shfl_xor(cs[u], 4, 64);
          cs[u] += __shfl_xor(cs[u], 8, 64);
          cs[u] += __shfl_xor(cs[u], 16, 64);
          cs[u] += __shfl_xor(cs[u], 32, 64);
        }
        if (rr == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) cpart[(T.wm * TM + i) * BN + (T.wn * TN + j) * 32 + c8 * 8 + u] = cs[u];
        }
      }
      // the granule's power-of-two scale: max . s in [2^13, 2^14)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
      gmax_all = fmaxf(gmax_all, gmax);
      float s_do = 1.0f;
      if (gmax > 0.f) {
        const int ex = min(max((int)(__float_as_uint(gmax) >> 23) - 127, -100), 100);
        s_do = __uint_as_float((uint32_t)(13 - ex + 127) << 23);
      }
      // (a tile may overhang the item set / the batch: granule columns past the table's pitch belong to
      // the NEXT row, granule rows past the batch lie outside the table; neither is ever read)
      if (lane == 0 && (T.n0 >> 5) + T.wn * TN + j < e.ds_pitch && (T.m0 >> 6) + T.wm * (TM / 2) + ih < ((M + 63) >> 6))
        e.dscale[(int64_t)((T.m0 >> 6) + T.wm * (TM / 2) + ih) * e.ds_pitch + ((T.n0 >> 5) + T.wn * TN + j)] = s_do;
      // the image: lane = (2 rows, 8 consecutive columns) of each 32 x 32 block: 16 bytes of hi, 16 of lo
#pragma unroll
      for (int i = 2 * ih; i < 2 * ih + 2; ++i)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int m = T.m0 + (T.wm * TM + i) * 32 + rr + 16 * it;
          if (m < e.rows_img && n < ldc) {
            uint2 h0, l0, h1, l1;
            const float *g = &G[i & 1][it * 8];
            rkp::split4(make_float4(g[0], g[1], g[2], g[3]), s_do, h0, l0);
            rkp::split4(make_float4(g[4], g[5], g[6], g[7]), s_do, h1, l1);
            char *d = e.dimg + (int64_t)m * pitch + (n >> 5) * LINE + (n & 31) * 2;
            *reinterpret_cast<uint4 *>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4 *>(d + 64) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            if (e.C && m < M) {
              float *c = e.C + (int64_t)m * ldc + n;
              *reinterpret_cast<float4 *>(c) = make_float4(g[0], g[1], g[2], g[3]);
              *reinterpret_cast<float4 *>(c + 4) = make_float4(g[4], g[5], g[6], g[7]);
            }
          }
        }
     }
    }
    lsum = rk_wave_sum(lsum);
    if (lane == 0) { lred[T.wave] = lsum; lred[NWAVES + T.wave] = gmax_all; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ls = 0.f, gm = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) { ls += lred[w]; gm = fmaxf(gm, lred[NWAVES + w]); }
      e.loss_part[T.t] = ls;
      // (the launch-wide maximum, for consumers of an fp32 dO: rk_decode_bwd_dz / dw2)
      atomicMax(reinterpret_cast<unsigned int *>(b.counts) + 8 + ((int)blockIdx.x & 63), __float_as_uint(gm));
    }
    if (e.gb_part) {
      // one gb_part row per 64 rows of dO (rk_decode_row_tile)
      for (int c = threadIdx.x; c < BN; c += NWAVES * 64) {
        const int n = T.n0 + c;
        if (n < N) {
#pragma unroll
          for (int g = 0; g < BM / 64; ++g)
            if (T.m0 + g * 64 < M)
              e.gb_part[(int64_t)(T.m0 / 64 + g) * ldc + n] = cpart[(2 * g) * BN + c] + cpart[(2 * g + 1) * BN + c];
        }
      }
    }
  }
};

}  // namespace pg
