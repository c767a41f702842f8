// Small dense contractions of the hidden nn.Linear stack (reference nn.py:242-249, autograd of
// F.linear): [B, h_i] x [h_i, h_{i+1}] with B = a few hundred rows and h = a few hundred columns.
//
// The LDS-tiled kernel of gemm.hip gives such a problem 16-32 workgroups of 64 x 64 and walks the
// whole K range in every wave: 13 / 13 / 22 us for forward / dX / dW at 500 x 200 x 200, of which
// 5-16 us is the serial k-loop on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 64 cycles per 2 k)
// and 5 us an epilogue that starts by fetching the bias (tools/probes/linear_probe.py).  Here:
//
//   * one workgroup per 32 x 32 output tile (4x the workgroups), its four waves SPLIT K: wave w
//     takes the 8-deep k-blocks w, w + 4, ...; the four accumulators are summed through LDS in wave
//     order (deterministic) and every wave finishes 8 rows of the tile;
//   * operands in the layout the fp32 MFMA wants (ONE value per lane: A row = lane % 32, k = lane / 32;
//     B k = lane / 32, column = lane % 32; a lane takes the four k's of its half of a k-block):
//     a k-major operand ([K, rows]) straight from global memory -- 32 consecutive lanes read 128
//     consecutive bytes -- and a K-contiguous one ([rows, K]) through a [32, 128]-float LDS panel
//     that is fetched with consecutive lanes on consecutive 16 bytes (a lane-per-row fetch of it
//     put 64 cache lines in flight per load instruction: 11 us for the forward product instead of
//     7 for dW); the next chunk's loads are issued before the MFMAs of the current one;
//   * the bias is fetched before the k-loop.
//
// Which k a (step, lane half) pair carries is free as long as both operands agree; the sums are
// fp32 fma chains in a fixed order, so results are reproducible run to run.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SG_KC = 128;             // k per chunk: 16 k-blocks of 8, four per wave
constexpr int SG_LD = SG_KC + 4;       // LDS row stride in floats (an odd number of 16-byte slots)

// one K-contiguous operand panel [32 rows, SG_KC] of chunk `kc0`, fetched with consecutive lanes on
// consecutive 16 bytes of a row (thread t, i: float4 number t + 256 i of the panel); k >= K -> 0
__device__ __forceinline__ void panel_load(const float *__restrict__ X, int64_t ld, int row0, int rows,
                                           int kc0, int K, bool vec, int tid, float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    const int rr = idx >> 5, k = kc0 + ((idx & 31) << 2);
    const float *row = X + (int64_t)min(row0 + rr, rows - 1) * ld;
    if (vec) {
      const float4 q = *reinterpret_cast<const float4 *>(row + min(k, K - 4));
      v[i] = (k < K) ? q : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float e[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) e[s] = (k + s < K) ? row[min(k + s, K - 1)] : 0.f;
      v[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}
__device__ __forceinline__ void panel_store(float *lds, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    *reinterpret_cast<float4 *>(lds + (idx >> 5) * SG_LD + ((idx & 31) << 2)) = v[i];
  }
}
// a k-major operand X[k * ld + rc]: the lane's quads of the wave's four k-blocks of chunk `kc0`,
// straight from global memory (32 consecutive lanes = 128 consecutive bytes)
__device__ __forceinline__ void direct_load(const float *__restrict__ X, int64_t ld, int rc, int kc0, int K,
                                            int w, int hh, float (&v)[4][4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kc0 + 8 * (w + 4 * u) + 4 * hh + s;
      const float x = X[(int64_t)min(k, K - 1) * ld + rc];
      v[u][s] = (k < K) ? x : 0.f;
    }
}

constexpr int SG_PANEL = 32 * SG_LD;
template <int AMODE, int BMODE>
constexpr int sg_smem_floats() {
  // staging panels of the K-contiguous operands; the cross-wave reduction reuses the space
  constexpr int N_PANEL = (AMODE == 0 ? 1 : 0) + (BMODE == 0 ? 1 : 0);
  return (N_PANEL * SG_PANEL > 4 * 16 * 64) ? N_PANEL * SG_PANEL : 4 * 16 * 64;
}

// one 32 x 32 output tile (number `tile` of g's tile grid) by the calling workgroup
template <int AMODE, int BMODE>
__device__ __forceinline__ void small_gemm_tile(const rk_small_gemm_t &g, const int tile, const int tiles_n,
                                                const int vec_a, const int vec_b, float *smem) {
  constexpr int PANEL = SG_PANEL;
  float *As = smem, *Bs = smem + (AMODE == 0 ? PANEL : 0);
  const int mt = tile / tiles_n, nt = tile % tiles_n;
  const int m0 = mt * 32, n0 = nt * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int K = g.K;
  const int col = n0 + r;
  const float bv = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
  const int am = min(m0 + r, g.M - 1), bn = min(col, g.N - 1);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float4 pa[4], pb[4];          // the next chunk's panels on their way to LDS
  float da[4][4], db[4][4];     // the next chunk's quads of a k-major operand
  if (AMODE == 0) panel_load(g.A, g.lda, m0, g.M, 0, K, vec_a != 0, tid, pa);
  else direct_load(g.A, g.lda, am, 0, K, w, hh, da);
  if (BMODE == 0) panel_load(g.B, g.ldb, n0, g.N, 0, K, vec_b != 0, tid, pb);
  else direct_load(g.B, g.ldb, bn, 0, K, w, hh, db);
  for (int kc0 = 0; kc0 < K; kc0 += SG_KC) {
    if (AMODE == 0 || BMODE == 0) {
      if (kc0) __syncthreads();                 // the previous chunk's reads are done
      if (AMODE == 0) panel_store(As, tid, pa);
      if (BMODE == 0) panel_store(Bs, tid, pb);
      __syncthreads();
    }
    float ca[4][4], cb[4][4];
    if (AMODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) ca[u][s] = da[u][s];
    }
    if (BMODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) cb[u][s] = db[u][s];
    }
    // (past the end: clamped addresses, all zeros, never used)
    if (AMODE == 0) panel_load(g.A, g.lda, m0, g.M, kc0 + SG_KC, K, vec_a != 0, tid, pa);
    else direct_load(g.A, g.lda, am, kc0 + SG_KC, K, w, hh, da);
    if (BMODE == 0) panel_load(g.B, g.ldb, n0, g.N, kc0 + SG_KC, K, vec_b != 0, tid, pb);
    else direct_load(g.B, g.ldb, bn, kc0 + SG_KC, K, w, hh, db);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kb = 8 * (w + 4 * u) + 4 * hh;        // the lane's quad inside the chunk
      if (AMODE == 0) {
        const float4 q = *reinterpret_cast<const float4 *>(As + r * SG_LD + kb);
        ca[u][0] = q.x; ca[u][1] = q.y; ca[u][2] = q.z; ca[u][3] = q.w;
      }
      if (BMODE == 0) {
        const float4 q = *reinterpret_cast<const float4 *>(Bs + r * SG_LD + kb);
        cb[u][0] = q.x; cb[u][1] = q.y; cb[u][2] = q.z; cb[u][3] = q.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[u][s], cb[u][s], acc, 0, 0, 0);
  }
  // ---- the four K-slices of the tile: summed in wave order, wave w finishes registers 4w .. 4w+3
  __syncthreads();
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem);
#pragma unroll
  for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[i];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * w + q;
    const float s = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
    // accumulator register i of lane (r, hh) = C[(i & 3) + 8 (i >> 2) + 4 hh][r]
    const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
    if (m < g.M && col < g.N) {
      float *dst = g.C + (int64_t)m * g.ldc + col;
      float o = rk_act(s + bv, g.act);
      // (backward of a stack's first layer: the gradient leaves multiplied by act'(the layer's
      // input activation) -- the rk_act_grad pass that followed, same product)
      if (g.dact_y) o = o * rk_act_dy(g.dact_y[(int64_t)m * g.ldc + col], g.dact);
      *dst = g.accumulate ? (o + *dst) : o;
    }
  }
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void small_gemm_kernel(rk_small_gemm_t g, int tiles_n, int vec_a,
                                                         int vec_b) {
  __shared__ __attribute__((aligned(16))) float smem[sg_smem_floats<AMODE, BMODE>()];
  small_gemm_tile<AMODE, BMODE>(g, (int)blockIdx.x, tiles_n, vec_a, vec_b, smem);
}

// out[c] = sum_r X[r * cols + c] for 32 columns per workgroup (256 threads: 32 columns x 8 row
// slices, 4 accumulators per thread, combined in a fixed order) -- the bias gradient of a Linear
// layer whose dYpre is already complete, as a third workgroup range of its dX / dW launch
struct SgColsum {
  const float *X;
  int rows, cols;
  float *out;
};
__device__ __forceinline__ void sg_colsum_tile(const SgColsum &c, const int block, float *smem) {
  const int lc = threadIdx.x & 31, s = threadIdx.x >> 5;
  const int col = block * 32 + lc;
  const int per = (c.rows + 7) >> 3;
  const int r0 = s * per, r1 = min(c.rows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < c.cols) {
    const float *x = c.X + col;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      a0 += x[(int64_t)r * c.cols];
      a1 += x[(int64_t)(r + 1) * c.cols];
      a2 += x[(int64_t)(r + 2) * c.cols];
      a3 += x[(int64_t)(r + 3) * c.cols];
    }
    for (; r < r1; ++r) a0 += x[(int64_t)r * c.cols];
  }
  smem[s * 32 + lc] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (s == 0 && col < c.cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += smem[k * 32 + lc];
    c.out[col] = t;
  }
}

// TWO independent contractions in one launch (rk_linear_bwd: dX and dW both need dYpre only):
// workgroups [0, tiles1) take g1's tiles, the rest g2's (both operands k-major: <1, 1>)
template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void small_gemm_pair_kernel(rk_small_gemm_t g1, int tiles_n1, int vec_a1,
                                                              int vec_b1, int tiles1, rk_small_gemm_t g2,
                                                              int tiles_n2, SgColsum cs, int n_pair) {
  constexpr int SM1 = sg_smem_floats<AMODE, BMODE>(), SM2 = sg_smem_floats<1, 1>();
  __shared__ __attribute__((aligned(16))) float smem[SM1 > SM2 ? SM1 : SM2];
  if ((int)blockIdx.x >= n_pair) {       // (behind both products' tiles: the bias gradient's column sums)
    sg_colsum_tile(cs, (int)blockIdx.x - n_pair, smem);
    return;
  }
  // (tiles1 < 0: g2's tiles come first in the grid, RK_PAIR_ORDER=1 -- the order makes no difference
  // (13.09 vs 13.12 us per rk_linear_bwd call), but the FORM does: written as a single if / else over the
  // two inlined bodies this kernel took the SUM of their times (16.0 us in rocprofv3 against 6.6 + 8.6 for
  // the two launches) instead of their maximum (7.2 us as compiled from this source); if a toolchain
  // change brings that back, tools/probes/linear_bwd_probe.py shows it and RK_LINEAR_PAIR=0 avoids it)
  if (tiles1 < 0) {
    const int t2 = -tiles1;
    if ((int)blockIdx.x < t2) small_gemm_tile<1, 1>(g2, (int)blockIdx.x, tiles_n2, 0, 0, smem);
    else small_gemm_tile<AMODE, BMODE>(g1, (int)blockIdx.x - t2, tiles_n1, vec_a1, vec_b1, smem);
    return;
  }
  if ((int)blockIdx.x < tiles1) small_gemm_tile<AMODE, BMODE>(g1, (int)blockIdx.x, tiles_n1, vec_a1, vec_b1, smem);
  else small_gemm_tile<1, 1>(g2, (int)blockIdx.x - tiles1, tiles_n2, 0, 0, smem);
}

// dY <- dY * act'(Y) in place and db[c] = sum_r dY[r][c] of the result, in one pass: block = 32
// columns x 32 row slices; a thread issues the loads of all its (<= 16 per round) rows before the
// first store -- dY is read and written, so a load behind a store would wait for it -- and the
// slices are combined in a fixed order
constexpr int AG_R = 16;
__global__ __launch_bounds__(1024) void act_grad_colsum_kernel(float *__restrict__ dY,
                                                               const float *__restrict__ Y, int rows,
                                                               int cols, int act, float *__restrict__ db) {
  __shared__ float part[32][33];
  const int lc = threadIdx.x & 31, s = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lc;
  const int per = (rows + 31) >> 5;
  const int r1 = min(rows, (s + 1) * per);
  float a0 = 0.f, a1 = 0.f;
  if (c < cols) {
    for (int r0 = s * per; r0 < r1; r0 += AG_R) {
      float g[AG_R], y[AG_R];
#pragma unroll
      for (int e = 0; e < AG_R; ++e) {
        const int64_t o = (int64_t)min(r0 + e, r1 - 1) * cols + c;
        g[e] = dY[o];
        y[e] = Y[o];
      }
#pragma unroll
      for (int e = 0; e < AG_R; ++e) {
        const float v = g[e] * rk_act_dy(y[e], act);
        if (r0 + e < r1) {
          dY[(int64_t)(r0 + e) * cols + c] = v;
          if (e & 1) a1 += v; else a0 += v;
        }
      }
    }
  }
  part[s][lc] = a0 + a1;
  __syncthreads();
  if (s == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += part[i][lc];
    db[c] = t;
  }
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool rk_small_gemm_fits(int M, int N, int K) {
  // every workgroup reads its 32 rows of A and 32 columns of B over the whole K from L2: fine for
  // the hidden sizes the reference is used with, not for a large dense layer
  return M > 0 && N > 0 && K > 0 && N <= 1024 && K <= 4096 && (int64_t)rk_cdiv(M, 32) * rk_cdiv(N, 32) <= 4096;
}

int rk_small_gemm(const rk_small_gemm_t *g, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (g->M <= 0 || g->N <= 0) return 0;
  RK_REQUIRE(g->K > 0, "K must be positive");
  const int tiles_n = rk_cdiv(g->N, 32);
  const int grid = rk_cdiv(g->M, 32) * tiles_n;
  // 16-byte loads along a contiguous K: base and row stride aligned, K a multiple of 4
  const int va = (g->amode == 0 && al16(g->A) && g->lda % 4 == 0 && g->K % 4 == 0) ? 1 : 0;
  const int vb = (g->bmode == 0 && al16(g->B) && g->ldb % 4 == 0 && g->K % 4 == 0) ? 1 : 0;
#define SG(AM, BM) RK_LAUNCH((small_gemm_kernel<AM, BM>), dim3(grid), dim3(256), 0, stream, *g, tiles_n, va, vb)
  if (g->amode == 0) { if (g->bmode == 0) SG(0, 0); else SG(0, 1); }
  else               { if (g->bmode == 0) SG(1, 0); else SG(1, 1); }
#undef SG
  RK_CHECK_LAUNCH("small_gemm");
  return 0;
}

// g1 (amode 0) and g2 (amode 1, bmode 1) as one launch; results identical to two rk_small_gemm calls
int rk_small_gemm_pair(const rk_small_gemm_t *g1, const rk_small_gemm_t *g2, void *stream_) {
  return rk_small_gemm_pair_colsum(g1, g2, nullptr, 0, 0, nullptr, stream_);
}

// ... and db[c] = sum_r dYpre[r * cols + c] (cs_X nullable) as a third workgroup range
int rk_small_gemm_pair_colsum(const rk_small_gemm_t *g1, const rk_small_gemm_t *g2, const float *cs_X,
                              int cs_rows, int cs_cols, float *cs_out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SgColsum cs = {cs_X, cs_rows, cs_cols, cs_out};
  const int n_cs = cs_X ? rk_cdiv(cs_cols, 32) : 0;
  RK_REQUIRE(g1->amode == 0 && g2->amode == 1 && g2->bmode == 1, "pair: dX-shaped and dW-shaped operands");
  RK_REQUIRE(g1->M > 0 && g1->N > 0 && g1->K > 0 && g2->M > 0 && g2->N > 0 && g2->K > 0, "pair: empty problem");
  const int tn1 = rk_cdiv(g1->N, 32), tn2 = rk_cdiv(g2->N, 32);
  const int t1 = rk_cdiv(g1->M, 32) * tn1, t2 = rk_cdiv(g2->M, 32) * tn2;
  const int va = (al16(g1->A) && g1->lda % 4 == 0 && g1->K % 4 == 0) ? 1 : 0;
  const int vb = (g1->bmode == 0 && al16(g1->B) && g1->ldb % 4 == 0 && g1->K % 4 == 0) ? 1 : 0;
  const int swap = rk_tune_get(RK_TUNE_PAIR_ORDER) == 1;
  const int t1a = swap ? -t2 : t1;
  if (g1->bmode == 0)
    RK_LAUNCH((small_gemm_pair_kernel<0, 0>), dim3(t1 + t2 + n_cs), dim3(256), 0, stream, *g1, tn1, va, vb, t1a, *g2,
              tn2, cs, t1 + t2);
  else
    RK_LAUNCH((small_gemm_pair_kernel<0, 1>), dim3(t1 + t2 + n_cs), dim3(256), 0, stream, *g1, tn1, va, vb, t1a, *g2,
              tn2, cs, t1 + t2);
  RK_CHECK_LAUNCH("small_gemm_pair");
  return 0;
}

int rk_act_grad_colsum(float *dY, const float *Y, int rows, int cols, int act, float *db, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cols == 0) return 0;
  RK_LAUNCH(act_grad_colsum_kernel, dim3(rk_cdiv(cols, 32)), dim3(1024), 0, stream, dY, Y, rows, cols, act, db);
  RK_CHECK_LAUNCH("act_grad_colsum");
  return 0;
}
