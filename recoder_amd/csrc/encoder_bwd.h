// Encoder backward workgroup body (see encoder.hip); a header because the fused
// dW + encoder-backward launch in gemm.hip runs it next to the dW GEMM tiles.
#ifndef RK_ENCODER_BWD_H
#define RK_ENCODER_BWD_H
#include "common.h"

namespace {

template <int HV>
__device__ __forceinline__ void ae_encode_bwd_body(
    const rk_block_t &b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb, int bid,
    int n_seg = 1, int64_t seg_stride = 0) {
  // one workgroup per sampled item column; its 4 waves take the 64-row groups
  // round-robin (popular items hold hundreds of entries -- a single wave per
  // column serialised them into the kernel's tail) and combine in fixed order
  __shared__ __attribute__((aligned(16))) float part[3][HV * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (bid < n_gb) {
    // encoder-bias gradient, 64 columns x one of n_seg row segments per workgroup:
    // gb[seg][j] = sum over the segment's rows of dZ[r, j] (the Adam sweep adds the n_seg
    // partial vectors in order).  A lane owns 4 columns (one 16-byte load) of every 16th row: the
    // 16 (wave, lane / 16) pairs take the rows round-robin, 8 independent loads in flight each --
    // 4 workgroups are all this part has at h = 200, so its chain of load batches (not its
    // bytes) set the length of the whole launch when it was 31 batches long.
    const int gseg = bid % n_seg, j0 = (bid / n_seg) * 64 + (lane & 15) * 4;
    const int srows = ((B + n_seg - 1) / n_seg + 63) & ~63;
    const int r_lo = gseg * srows, r_hi = min(B, r_lo + srows);
    const int jj = min(j0, h - 4);                 // (clamped: columns past h are never stored)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = r_lo + wid * 4 + (lane >> 4);
    for (; r + 7 * 16 < r_hi; r += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(dZ + (int64_t)(r + u * 16) * h + jj);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < r_hi; r += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(dZ + (int64_t)r * h + jj);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    // the 4 row-phases of the wave (lanes l, l^16, l^32, l^48), then the 4 waves, fixed order
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      a.x += __shfl_xor(a.x, off, 64); a.y += __shfl_xor(a.y, off, 64);
      a.z += __shfl_xor(a.z, off, 64); a.w += __shfl_xor(a.w, off, 64);
    }
    if (lane < 16) *reinterpret_cast<float4 *>(&part[0][wid * 64 + lane * 4]) = a;
    __syncthreads();
    if (wid == 0 && lane < 16 && j0 < h) {
      float4 o = a;
      for (int w = 1; w < 4; ++w) {
        const float4 q = *reinterpret_cast<const float4 *>(&part[0][w * 64 + lane * 4]);
        o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
      }
      *reinterpret_cast<float4 *>(gb + gseg * h + j0) = o;        // (h % 4 == 0: whole quads)
    }
    return;
  }
  const int n_b = b.counts[0];
  // long columns (large batches: a popular item is held by thousands of rows) are cut into
  // n_seg row segments, one workgroup each, writing partial gradient rows G + seg*seg_stride
  // that the Adam sweep adds up in segment order
  const int c = (bid - n_gb) / n_seg, seg = (bid - n_gb) % n_seg;
  if (c >= n_b) return;
  const uint32_t *colbits = b.bits_cr + (int64_t)c * b.ldw_cr;
  float4 acc[HV];
#pragma unroll
  for (int k = 0; k < HV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

  const int seg_rows = ((B + n_seg - 1) / n_seg + 63) & ~63;
  const int rbeg = row_off + seg * seg_rows;
  const int rend = min(row_off + B, rbeg + seg_rows);
  G += seg * seg_stride;
  for (int r0 = (rbeg & ~63) + wid * 64; r0 < rend; r0 += 256) {
    const int row = r0 + lane;
    bool on = false;
    float s = 0.f;
    if (row >= rbeg && row < rend) {
      on = (colbits[row >> 5] >> (row & 31)) & 1u;
      if (on) {
        const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (c >> 5)];
        s = b.svals[rk_entry_index(b, row, c, word)];
      }
    }
    unsigned long long mask = __ballot(on);
    // ascending-row order, 8 row loads in flight per pass
    while (mask) {
      int kk[8];
      float sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool have = mask != 0ull;
        const int k = have ? __builtin_ctzll(mask) : 0;
        mask = have ? (mask & (mask - 1)) : 0ull;
        kk[u] = have ? (r0 + k - row_off) : 0;
        sv[u] = have ? __shfl(s, k, 64) : 0.f;
      }
#pragma unroll
      for (int v = 0; v < HV; ++v) {
        const int hh = min((v * 64 + lane) * 4, h - 4);
        float4 d4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          d4[u] = *reinterpret_cast<const float4 *>(dZ + (int64_t)kk[u] * h + hh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc[v].x = fmaf(sv[u], d4[u].x, acc[v].x);
          acc[v].y = fmaf(sv[u], d4[u].y, acc[v].y);
          acc[v].z = fmaf(sv[u], d4[u].z, acc[v].z);
          acc[v].w = fmaf(sv[u], d4[u].w, acc[v].w);
        }
      }
    }
  }
  if (wid > 0) {
#pragma unroll
    for (int v = 0; v < HV; ++v)
      *reinterpret_cast<float4 *>(&part[wid - 1][(v * 64 + lane) * 4]) = acc[v];
  }
  __syncthreads();
  if (wid == 0) {
    float *grow = G + (int64_t)c * h;
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int hh = (v * 64 + lane) * 4;
      if (hh < h) {
        float4 a = acc[v];
        for (int w = 0; w < 3; ++w) {
          const float4 o = *reinterpret_cast<const float4 *>(&part[w][hh]);
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        if (accumulate) {
          const float4 o = *reinterpret_cast<const float4 *>(grow + hh);
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        *reinterpret_cast<float4 *>(grow + hh) = a;
      }
    }
  }
}

}  // namespace
#endif
