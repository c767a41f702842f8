// Encoder backward workgroup body (see encoder.hip); a header because the fused
// dW + encoder-backward launch in gemm.hip runs it next to the dW GEMM tiles.
#ifndef RK_ENCODER_BWD_H
#define RK_ENCODER_BWD_H
#include "common.h"

namespace {

// DYN: the body's LDS scratch is carved from `sm` (a kernel with dynamic LDS: csrc/pgemm.hip) instead of
// static arrays
template <int HV, bool DYN = false>
__device__ __forceinline__ void ae_encode_bwd_body(
    const rk_block_t &b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb, int bid,
    int n_seg = 1, int64_t seg_stride = 0, char *sm = nullptr) {
  // one workgroup per sampled item column; its 4 waves take the 64-row groups
  // round-robin (popular items hold hundreds of entries -- a single wave per
  // column serialised them into the kernel's tail) and combine in fixed order
  float (*part)[HV * 256];
  if constexpr (DYN) {
    part = reinterpret_cast<float (*)[HV * 256]>(sm);
  } else {
    __shared__ __attribute__((aligned(16))) float part_s[3][HV * 256];
    part = part_s;
  }
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
    // (entries the input noise dropped -- s == 0 -- are skipped: exact zeros either way)
    unsigned long long mask = __ballot(on && s != 0.f);
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

// Encoder backward, one WAVE per sampled item column (4 columns per workgroup).  A C2 batch holds
// 3.4 stored entries per column on average (99 % of the columns hold <= 41, 20 of ~7 900 more than
// 128): a workgroup per column spent the launch on dispatch and on a chain of dependent loads per
// 64-row group.  Here a wave reads the column's whole bitmap row at once (one word per lane: B <=
// 2048), lists its rows in ascending order, fetches their values in parallel and then streams the
// dZ rows, 8 loads in flight.  Columns with more than 64 entries are done afterwards by the 4 waves
// of the workgroup together (64-row groups round-robin, combined in fixed order).
// WW: bitmap words per lane (1: a row window of <= 2048 rows; 2: <= 4096 -- the stand-alone launch only);
// U: dZ rows of a column in flight (8; 2 = the variant for blocks of LIGHT columns -- one or two stored
// entries each, C5's uniform catalogue -- whose smaller register footprint doubles the columns in flight).
template <int HV, bool DYN = false, int WW = 1, int U = 8>
__device__ __forceinline__ void ae_encode_bwd_cols_body(
    const rk_block_t &b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb, const int bid,
    char *sm = nullptr) {
  static_assert(WW == 1 || !DYN, "the two-word window exists in the stand-alone kernel only");
  if (bid < n_gb) {                 // encoder-bias gradient: the shared body's first branch
    ae_encode_bwd_body<HV, DYN>(b, row_off, B, dZ, h, G, accumulate, gb, n_gb, bid, 1, 0, sm);
    return;
  }
  float (*part)[HV * 256];
  uint16_t (*rows_l)[64];
  uint16_t *rows_h;
  int *heavy_l;                       // entries of wave w's column if it is a heavy one, else 0
  if constexpr (DYN) {
    part = reinterpret_cast<float (*)[HV * 256]>(sm);
    rows_l = reinterpret_cast<uint16_t (*)[64]>(sm + 3 * HV * 256 * 4);
    rows_h = reinterpret_cast<uint16_t *>(sm + 3 * HV * 256 * 4 + 4 * 64 * 2);
    heavy_l = reinterpret_cast<int *>(sm + 3 * HV * 256 * 4 + 4 * 64 * 2 + (2048 + 64) * 2);
  } else {
    __shared__ __attribute__((aligned(16))) float part_s[3][HV * 256];
    __shared__ uint16_t rows_l_s[4][64];
    __shared__ uint16_t rows_h_s[WW * 2048 + 64];
    __shared__ int heavy_l_s[4];
    part = part_s; rows_l = rows_l_s; rows_h = rows_h_s; heavy_l = heavy_l_s;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n_b = b.counts[0];
  // wave w of workgroup i takes column i + w * Q, Q = ceil(n_b / 4): columns that are neighbours
  // by item id land in different workgroups (a catalogue sorted by popularity would otherwise put
  // all its heavy columns into the first few workgroups, one after the other)
  const int Q = (n_b + 3) >> 2;
  const int c0 = bid - n_gb;
  if (c0 >= Q) return;
  const int c = c0 + wid * Q;
  const bool live = c < n_b;
  const int w0 = row_off >> 5, rend = row_off + B;
  const int nw = ((rend + 31) >> 5) - w0;              // <= 64 WW (host-checked)
  uint32_t mw[WW];
  int off_w[WW];                                       // where this lane's rows of word set j start in the list
  int total = 0;
#pragma unroll
  for (int j = 0; j < WW; ++j) {
    const int wj = lane + 64 * j;
    uint32_t m = 0;
    if (live && wj < nw) {
      m = b.bits_cr[(int64_t)c * b.ldw_cr + w0 + wj];
      const int base = (w0 + wj) << 5;
      if (base < row_off) m &= ~0u << (row_off - base);
      if (base + 32 > rend) m &= (1u << (rend - base)) - 1u;     // (rend - base is in 1..31 here)
    }
    mw[j] = m;
  }
#pragma unroll
  for (int j = 0; j < WW; ++j) {
    const int cnt = __popc(mw[j]);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    off_w[j] = total + incl - cnt;                     // (ascending rows: word set 0 first)
    total += __shfl(incl, 63, 64);
  }
  const bool heavy = total > 64;
  if (lane == 0) heavy_l[wid] = heavy ? total : 0;
  if (!heavy) {
#pragma unroll
    for (int j = 0; j < WW; ++j) {
      uint32_t m = mw[j];
      int o = off_w[j];
      while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        rows_l[wid][o++] = (uint16_t)(((w0 + lane + 64 * j) << 5) + k - row_off);    // (relative: < 2048 WW + 32)
      }
    }
  }
  __syncthreads();
  float4 acc[HV];
#pragma unroll
  for (int k = 0; k < HV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && !heavy && total > 0) {
    int row = row_off;
    float sv = 0.f;
    if (lane < total) {
      row = row_off + rows_l[wid][lane];
      const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (c >> 5)];
      sv = b.svals[rk_entry_index(b, row, c, word)];
    }
    // entries the input noise dropped (sv == 0: half of them at noise_prob 0.5) are skipped: the wave walks
    // the set bits of the ballot of its live entries (exact: fmaf(0, d, acc) == acc, acc never -0)
    unsigned long long lv = __ballot(sv != 0.f);
    while (lv) {
      int kk[U];
      float s8[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {       // (a pass short of U live entries pads with row 0 / 0: exact zeros)
        const bool has = lv != 0ull;
        const int k = has ? __builtin_ctzll(lv) : 0;
        lv &= lv - 1ull;
        kk[u] = has ? __builtin_amdgcn_readlane(row, k) - row_off : 0;
        s8[u] = has ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), k)) : 0.f;
      }
#pragma unroll
      for (int v = 0; v < HV; ++v) {
        const int hh = min((v * 64 + lane) * 4, h - 4);
        float4 d4[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          d4[u] = *reinterpret_cast<const float4 *>(dZ + (int64_t)kk[u] * h + hh);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc[v].x = fmaf(s8[u], d4[u].x, acc[v].x);
          acc[v].y = fmaf(s8[u], d4[u].y, acc[v].y);
          acc[v].z = fmaf(s8[u], d4[u].z, acc[v].z);
          acc[v].w = fmaf(s8[u], d4[u].w, acc[v].w);
        }
      }
    }
  }
  // (a later row window -- `accumulate` -- of a column without entries in it leaves the row as it is)
  if (live && !heavy && !(accumulate && total == 0)) {
    float *grow = G + (int64_t)c * h;
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int hh = (v * 64 + lane) * 4;
      if (hh < h) {
        float4 a = acc[v];
        if (accumulate) {
          const float4 o = *reinterpret_cast<const float4 *>(grow + hh);
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        *reinterpret_cast<float4 *>(grow + hh) = a;
      }
    }
  }
  // ---- the workgroup's heavy columns, one after the other, all 4 waves each: the owner wave
  // lists the column's rows in LDS, every wave takes a contiguous quarter of the ENTRIES (balanced
  // whatever the rows are), fetches their values 64 at a time and streams the dZ rows 16 loads deep
  for (int j = 0; j < 4; ++j) {
    const int tot_j = heavy_l[j];                   // (uniform: read from LDS behind a barrier)
    if (tot_j == 0) continue;
    const int cj = c0 + j * Q;
    if (wid == j) {
#pragma unroll
      for (int jw = 0; jw < WW; ++jw) {
        uint32_t m = mw[jw];
        int o = off_w[jw];
        while (m) {
          const int k = __builtin_ctz(m);
          m &= m - 1;
          rows_h[o++] = (uint16_t)(((w0 + lane + 64 * jw) << 5) + k - row_off);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int per = (tot_j + 3) >> 2;
    const int e1 = min(tot_j, (wid + 1) * per);
    for (int e0 = wid * per; e0 < e1; e0 += 64) {
      const int ne = min(64, e1 - e0);
      int row = row_off;
      float sv = 0.f;
      if (lane < ne) {
        row = row_off + rows_h[e0 + lane];
        const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (cj >> 5)];
        sv = b.svals[rk_entry_index(b, row, cj, word)];
      }
      constexpr int UH = U >= 8 ? 16 : 2 * U;     // (dZ rows of a heavy column in flight)
      unsigned long long lv = __ballot(sv != 0.f);      // (dropped entries skipped, as above)
      while (lv) {
        int kk[UH];
        float s16[UH];
#pragma unroll
        for (int u = 0; u < UH; ++u) {    // (a short pass pads with row 0 / 0: exact zeros)
          const bool has = lv != 0ull;
          const int k = has ? __builtin_ctzll(lv) : 0;
          lv &= lv - 1ull;
          kk[u] = has ? __builtin_amdgcn_readlane(row, k) - row_off : 0;
          s16[u] = has ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), k)) : 0.f;
        }
#pragma unroll
        for (int v = 0; v < HV; ++v) {
          const int hh = min((v * 64 + lane) * 4, h - 4);
          float4 d4[UH];
#pragma unroll
          for (int u = 0; u < UH; ++u)
            d4[u] = *reinterpret_cast<const float4 *>(dZ + (int64_t)kk[u] * h + hh);
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            acc[v].x = fmaf(s16[u], d4[u].x, acc[v].x);
            acc[v].y = fmaf(s16[u], d4[u].y, acc[v].y);
            acc[v].z = fmaf(s16[u], d4[u].z, acc[v].z);
            acc[v].w = fmaf(s16[u], d4[u].w, acc[v].w);
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
      float *grow = G + (int64_t)cj * h;
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
    __syncthreads();                                // (part / rows_h are reused by the next one)
  }
}

}  // namespace
#endif
