// The split-K / column-tile slab reduce (gemm.hip splitk_reduce_kernel) as a workgroup body, so that
// it can also ride on another launch as a workgroup range (dw3.hip: MatrixFactorization steps, where
// nothing between the decode and the Adam sweep reads dZ).
#pragma once
#include "common.h"

namespace rkred {

constexpr int RED_W = 8;       // waves per workgroup: each sums 1/RED_W of the splits
struct Args {
  const float *ws;
  int M, N;
  const int32_t *Kdev;
  int tile_k, max_splits;
  const float *Zact;
  int act;
  float *out;
};
inline int blocks(int M, int N) { return (int)(((int64_t)M * N / 4 + 63) / 64); }

// ws[split][M][N] -> out[M][N] (fixed split order), optional * act'(Zact); workgroup `block` of
// RW * 64 threads (RW = RED_W = 8 in the stand-alone launch; 4 where it rides on a launch of 256-thread
// workgroups: csrc/pgemm.hip), `part`: (RW - 1) * 64 float4 of LDS.  The sum is taken in ascending slab order
// within a wave's share and the shares are combined in wave order: a different RW groups the same additions
// differently (agreement to rounding, each form deterministic).
template <int RW = RED_W>
__device__ __forceinline__ void body(const Args &a, const int block, float4 (*part)[64]) {
  // 64 float4 outputs per workgroup; the waves each sum a contiguous share of the splits (all of
  // its loads in flight at once: the slabs come from the Infinity Cache / HBM, and the launch is as
  // long as one wave's chain of load batches), combined in fixed order through LDS
  const int K = *a.Kdev;
  // (tile_k > 0: one slab per tile_k-wide column tile of the decode -- the fused dZ of decode16.hip)
  const int kchunk = a.tile_k > 0 ? a.tile_k : ((K + a.max_splits - 1) / a.max_splits + 31) & ~31;   // as the GEMM derives it
  int ns = (K + kchunk - 1) / kchunk;
  if (ns > a.max_splits) ns = a.max_splits;
  const int64_t tot4 = ((int64_t)a.M * a.N) >> 2;      // M*N is a multiple of 4 (N = h)
  const float4 *ws4 = reinterpret_cast<const float4 *>(a.ws);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)block * 64 + lane;
  const int per = (ns + RW - 1) / RW;
  const int z1 = min(ns, (q + 1) * per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the activation values of the fused act' are fetched with the slabs, not behind the barrier)
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.Zact && q == 0 && i < tot4) y = reinterpret_cast<const float4 *>(a.Zact)[i];
  if (i < tot4) {
    int z = q * per;
    for (; z + 8 <= z1; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws4[(int64_t)(z + u) * tot4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < z1; ++z) {
      const float4 v = ws4[(int64_t)z * tot4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (q > 0) part[q - 1][lane] = s;
  __syncthreads();
  if (q == 0 && i < tot4) {
#pragma unroll
    for (int w = 0; w < RW - 1; ++w) {
      const float4 b = part[w][lane];
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (a.Zact) {
      s.x *= rk_act_dy(y.x, a.act); s.y *= rk_act_dy(y.y, a.act);
      s.z *= rk_act_dy(y.z, a.act); s.w *= rk_act_dy(y.w, a.act);
    }
    reinterpret_cast<float4 *>(a.out)[i] = s;
  }
}

}  // namespace rkred
