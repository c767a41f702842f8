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
    // partial vectors in order).  The 4 waves take interleaved quarters of the rows, 4
    // independent loads in flight each, combined in fixed order.
    const int gseg = bid % n_seg, j = (bid / n_seg) * 64 + lane;
    const int srows = ((B + n_seg - 1) / n_seg + 63) & ~63;
    const int r_lo = gseg * srows, r_hi = min(B, r_lo + srows);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < h) {
      int r = r_lo + wid;
      for (; r + 12 < r_hi; r += 16) {
        a0 += dZ[(int64_t)r * h + j];
        a1 += dZ[(int64_t)(r + 4) * h + j];
        a2 += dZ[(int64_t)(r + 8) * h + j];
        a3 += dZ[(int64_t)(r + 12) * h + j];
      }
      for (; r < r_hi; r += 4) a0 += dZ[(int64_t)r * h + j];
    }
    part[0][wid * 64 + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wid == 0 && j < h)
      gb[gseg * h + j] =
          (part[0][lane] + part[0][64 + lane]) + (part[0][128 + lane] + part[0][192 + lane]);
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
