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
//   * no LDS staging: the fp32 MFMA takes ONE value per lane and operand (A: row = lane % 32,
//     k = lane / 32; B: k = lane / 32, column = lane % 32), so a lane loads the four k's of its
//     half of a k-block straight from global memory -- one 16-byte load where K is the contiguous
//     dimension, four coalesced 4-byte loads where the row / column index is -- and eight k-blocks
//     are in flight per wave before the first MFMA issues;
//   * the bias is fetched before the k-loop.
//
// Which k a (step, lane half) pair carries is free as long as both operands agree; the sums are
// fp32 fma chains in a fixed order, so results are reproducible run to run.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SG_U = 8;      // k-blocks (8 k each) in flight per wave

// the lane's 4 values k .. k+3 of one operand row / column `rc` (clamped by the caller):
//   MODE 0: X[rc * ld + k]  (K contiguous; one 16-byte load when `vec`)
//   MODE 1: X[k * ld + rc]  (k-major)
// entries with k >= K read a clamped (valid) address and are zeroed
template <int MODE>
__device__ __forceinline__ void load_quad(const float *__restrict__ X, int64_t ld, int rc, int k, int K,
                                          bool vec, float (&v)[4]) {
  if (MODE == 0) {
    const float *row = X + (int64_t)rc * ld;
    if (vec) {
      const float4 q = *reinterpret_cast<const float4 *>(row + min(k, K - 4));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = row[min(k + s, K - 1)];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = X[(int64_t)min(k + s, K - 1) * ld + rc];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = (k + s < K) ? v[s] : 0.f;
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void small_gemm_kernel(rk_small_gemm_t g, int tiles_n, int vec_a,
                                                         int vec_b) {
  __shared__ float red[4][16][64];
  const int mt = blockIdx.x / tiles_n, nt = blockIdx.x % tiles_n;
  const int m0 = mt * 32, n0 = nt * 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int K = g.K;
  const int nblk = (K + 7) >> 3;
  const int col = n0 + r;
  const float bv = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
  const int am = min(m0 + r, g.M - 1), bn = min(col, g.N - 1);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int j0 = w; j0 < nblk; j0 += 4 * SG_U) {
    float a[SG_U][4], b[SG_U][4];
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {
      const int k = 8 * (j0 + 4 * u) + 4 * hh;       // (blocks past the end: all four k >= K -> zeros)
      load_quad<AMODE>(g.A, g.lda, am, k, K, vec_a != 0, a[u]);
      load_quad<BMODE>(g.B, g.ldb, bn, k, K, vec_b != 0, b[u]);
    }
#pragma unroll
    for (int u = 0; u < SG_U; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s], b[u][s], acc, 0, 0, 0);
  }
  // ---- the four K-slices of the tile: summed in wave order, wave w finishes registers 4w .. 4w+3
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
      const float o = rk_act(s + bv, g.act);
      *dst = g.accumulate ? (o + *dst) : o;
    }
  }
}

// dY <- dY * act'(Y) in place and db[c] = sum_r dY[r][c] of the result, in one pass: block = 64
// columns x 16 row slices, 4 accumulators per thread, combined in a fixed order
__global__ __launch_bounds__(1024) void act_grad_colsum_kernel(float *__restrict__ dY,
                                                               const float *__restrict__ Y, int rows,
                                                               int cols, int act, float *__restrict__ db) {
  __shared__ float part[16][64];
  const int lc = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  const int per = (rows + 15) >> 4;
  const int r0 = s * per, r1 = min(rows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t o = (int64_t)(r + e) * cols + c;
        v[e] = dY[o] * rk_act_dy(Y[o], act);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dY[(int64_t)(r + e) * cols + c] = v[e];
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    }
    for (; r < r1; ++r) {
      const int64_t o = (int64_t)r * cols + c;
      const float v = dY[o] * rk_act_dy(Y[o], act);
      dY[o] = v;
      a0 += v;
    }
  }
  part[s][lc] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (s == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][lc];
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

int rk_act_grad_colsum(float *dY, const float *Y, int rows, int cols, int act, float *db, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cols == 0) return 0;
  RK_LAUNCH(act_grad_colsum_kernel, dim3(rk_cdiv(cols, 64)), dim3(1024), 0, stream, dY, Y, rows, cols, act, db);
  RK_CHECK_LAUNCH("act_grad_colsum");
  return 0;
}
