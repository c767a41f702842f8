// On-device sparse-batch collation.
//
// Replaces RecommendationDataset._extract (reference data.py:64-83: CSR row
// gather) and BatchCollator.collate (data.py:203-251: nonzero() ->
// np.unique(return_inverse) -> per-slice COO) with HBM-resident integer work:
//   phase1 : ONE launch -- stamp every touched item id (scatter, 4 B / nnz),
//            exclusive scan of the S row degrees (block CSR pointers), clear the
//            transposed bitmap
//   scan   : stamps -> pos[item] / items[] in ascending item id (== np.unique
//            order) and n_b; one workgroup for catalogues <= 64k items, a
//            count + assign pair of launches above that
//   build  : per row (one wave): cols[j] = pos[item_j], vals[j], the row's bitmap
//            words and their prefix popcounts assembled in LDS, transposed bits
// Three launches (each tiny kernel costs ~4.5 us of dependent round trips + launch
// tail, so the original seven were 1/6 of the step); all HBM/latency-bound integer
// work: one pass over the group's nnz + two passes over n_items ints.
#include "common.h"

namespace {

constexpr int SMALL_SCAN_MAX = 1 << 16;   // catalogues up to 64k items: single-workgroup scan
constexpr int SEG_WORDS = 2048;           // bitmap words a wave builds in LDS at a time, at most
// (the launch asks for what the block's capacity needs: 629 words per wave at ML-20M's 20 k items = 10 KB
// per workgroup instead of 32 -- the look-ahead collation runs beside the fused decode launch, which
// leaves 22 KB of a CU's LDS)
inline int seg_words_for(int n_cap) {
  const int w = ((n_cap + 31) / 32 + 63) & ~63;
  return w < 64 ? 64 : (w > SEG_WORDS ? SEG_WORDS : w);
}

// ---- phase 1 (one launch): mark the touched items, scan the row degrees, clear
//      the transposed bitmap.  Roles by block index:
//        [0, nrow_blk)            one wave per sampled row: mark[item] = stamp
//        nrow_blk                 block CSR row pointers (exclusive scan of degrees)
//        (nrow_blk, gridDim.x)    zero bits_cr (capacity-sized, no dependency on n_b)
// up to RK_COLLATE_MULTI blocks collated by ONE set of launches (rk_collate_at_multi): block g =
// blockIdx.y, cursor offset off0 + g
struct MultiBlk {
  rk_block_t b[RK_COLLATE_MULTI];
};

__device__ __forceinline__ void collate_phase1_body(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const int64_t *__restrict__ users, int S, int32_t stamp, int all, int nrow_blk, const rk_block_t &b,
    rk_cur_t cur, const int bx, const int nbx) {
  // (bx of nbx: the role index -- blockIdx.x of a launch with one workgroup per role, or a workgroup's turn)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (cur.cursor) { users += rk_cur_local(cur) * S; stamp = rk_cur_stamp(cur); }
  if (bx < nrow_blk) {
    if (all) return;
    const int row = bx * 4 + wid;
    if (row >= S) return;
    const int64_t u = users[row];
    const int64_t beg = ds_indptr[u], end = ds_indptr[u + 1];
    for (int64_t e = beg + lane; e < end; e += 64) b.mark[ds_indices[e]] = stamp;
    return;
  }
  if (bx == nrow_blk) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry_s;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < S; base += 256) {
      const int i = base + tid;
      int32_t d = 0;
      if (i < S) {
        const int64_t u = users[i];
        d = (int32_t)(ds_indptr[u + 1] - ds_indptr[u]);
      }
      int32_t x = d;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        int32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
      }
      if (lane == 63) wsum[wid] = x;
      __syncthreads();
      int32_t woff = 0;
      for (int w = 0; w < wid; ++w) woff += wsum[w];
      const int32_t carry = carry_s;
      if (i < S) b.indptr[i] = min(carry + woff + x - d, b.nnz_cap);    // (clamped: see counts[6])
      __syncthreads();
      if (tid == 255) carry_s = carry + woff + x;
      __syncthreads();
    }
    if (tid == 0) {
      // more stored interactions than the block is sized for (never by construction): every row
      // range is clamped into the arrays, the true count goes to counts[6] and the host raises
      b.indptr[S] = min(carry_s, b.nnz_cap);
      b.counts[1] = min(carry_s, b.nnz_cap);
      b.counts[6] = carry_s > b.nnz_cap ? carry_s : 0;
      b.counts[3] = S;
      for (int i = 8; i < 72; ++i) b.counts[i] = 0;   // max |dLoss/dLogit| slots (gemm.hip)
    }
    return;
  }
  if (b.bits_cr) {
    const int zb = bx - nrow_blk - 1, nz = nbx - nrow_blk - 1;
    const int64_t tot4 = ((int64_t)b.n_cap * b.ldw_cr) >> 2;     // uint4 granules
    uint4 *p4 = reinterpret_cast<uint4 *>(b.bits_cr);
    for (int64_t i = (int64_t)zb * 256 + tid; i < tot4; i += (int64_t)nz * 256)
      p4[i] = make_uint4(0u, 0u, 0u, 0u);
    const int64_t tail0 = tot4 << 2, tot = (int64_t)b.n_cap * b.ldw_cr;
    if (zb == 0 && tail0 + tid < tot) b.bits_cr[tail0 + tid] = 0u;
  }
}

__global__ __launch_bounds__(256) void collate_phase1_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const int64_t *__restrict__ users, int S, int32_t stamp, int all, int nrow_blk, rk_block_t b,
    rk_cur_t cur) {
  collate_phase1_body(ds_indptr, ds_indices, users, S, stamp, all, nrow_blk, b, cur, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256) void collate_phase1_multi_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const int64_t *__restrict__ users, int S, int all, int nrow_blk, MultiBlk mb, rk_cur_t cur, int nbx) {
  cur.off += (int)blockIdx.y;
  // the look-back slots of this collation's scan (collate_scan_lb_multi_kernel) start EMPTY: a slot is
  // valid once its chunk's workgroup of THIS launch set wrote it -- whatever stamp the previous collation
  // of the block ran under (the stamp as the ready flag let a block collated twice with one
  // stamp read the first run's totals)
  if (blockIdx.x == 0 && (int)threadIdx.x < min(mb.b[blockIdx.y].n_chunks, 32))
    reinterpret_cast<unsigned long long *>(mb.b[blockIdx.y].scan_tmp)[threadIdx.x] = 0ull;
  // (the grid may hold fewer workgroups than roles: the look-ahead collation of a replayed group runs BESIDE the
  // training chain and has eight steps to finish -- a small grid takes few of the chain's wave slots; rk_collate_at_multi)
  for (int bx = blockIdx.x; bx < nbx; bx += gridDim.x) {
    collate_phase1_body(ds_indptr, ds_indices, users, S, 1, all, nrow_blk, mb.b[blockIdx.y], cur, bx, nbx);
    __syncthreads();
  }
}

// ---- count: marked items per chunk (large catalogues) ----
__device__ __forceinline__ void collate_count_body(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    int32_t *__restrict__ scan_tmp, rk_cur_t cur) {
  __shared__ int32_t ws[4];
  if (cur.cursor) stamp = rk_cur_stamp(cur);
  const int base = blockIdx.x * RK_SCAN_CHUNK;
  int32_t c = 0;
  for (int i = threadIdx.x; i < RK_SCAN_CHUNK; i += 256) {
    const int it = base + i;
    if (it < n_items) c += (all || mark[it] == stamp) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) scan_tmp[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void collate_count_kernel(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    int32_t *__restrict__ scan_tmp, rk_cur_t cur) {
  collate_count_body(mark, n_items, stamp, all, scan_tmp, cur);
}
__global__ __launch_bounds__(256) void collate_count_multi_kernel(int all, MultiBlk mb, rk_cur_t cur) {
  cur.off += (int)blockIdx.y;
  const rk_block_t &b = mb.b[blockIdx.y];
  collate_count_body(b.mark, b.n_items, 1, all, b.scan_tmp, cur);
}

// ---- assign: pos[] / items[] in ascending item order (large catalogues) ----
__device__ __forceinline__ void collate_assign_body(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    const int32_t *__restrict__ scan_tmp, int n_chunks, int32_t *__restrict__ pos,
    int32_t *__restrict__ items, int32_t *__restrict__ counts, int n_cap, int nnz_cap, rk_cur_t cur) {
  __shared__ int32_t red[4];
  if (cur.cursor) stamp = rk_cur_stamp(cur);
  __shared__ int32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int32_t part = 0;
  for (int i = tid; i < (int)blockIdx.x; i += 256) part += scan_tmp[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if (lane == 0) red[wid] = part;
  __syncthreads();
  const int32_t base_cnt = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (blockIdx.x == (unsigned)(n_chunks - 1) && tid == 0) {
    // a block smaller than its item set (never by construction: recoder_amd sizes n_cap from the
    // dataset) is truncated IN BOUNDS and flagged in counts[5] -- the host raises on it
    const int32_t n_all = base_cnt + scan_tmp[blockIdx.x], n_b = min(n_all, n_cap);
    counts[0] = n_b;
    counts[2] = (n_b + 31) & ~31;
    counts[5] = n_all > n_cap ? n_all : 0;
  }
  const int it0 = blockIdx.x * RK_SCAN_CHUNK + tid * 8;
  int32_t f[8];
  int32_t local = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int it = it0 + k;
    f[k] = (it < n_items && (all || mark[it] == stamp)) ? 1 : 0;
    local += f[k];
  }
  int32_t x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  int32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  int32_t p = base_cnt + woff + x - local;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int it = it0 + k;
    if (it < n_items) {
      if (f[k]) {
        if (p < n_cap) { pos[it] = p; items[p] = it; } else { pos[it] = -1; }
        ++p;
      } else {
        pos[it] = -1;
      }
    }
  }
}
__global__ __launch_bounds__(256) void collate_assign_kernel(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    const int32_t *__restrict__ scan_tmp, int n_chunks, int32_t *__restrict__ pos,
    int32_t *__restrict__ items, int32_t *__restrict__ counts, int n_cap, int nnz_cap, rk_cur_t cur) {
  collate_assign_body(mark, n_items, stamp, all, scan_tmp, n_chunks, pos, items, counts, n_cap, nnz_cap, cur);
}
__global__ __launch_bounds__(256) void collate_assign_multi_kernel(int all, MultiBlk mb, rk_cur_t cur) {
  cur.off += (int)blockIdx.y;
  const rk_block_t &b = mb.b[blockIdx.y];
  collate_assign_body(b.mark, b.n_items, 1, all, b.scan_tmp, b.n_chunks, b.pos, b.items, b.counts, b.n_cap,
                      b.nnz_cap, cur);
}

// ---- scan (small catalogues): ONE workgroup does count + assign in one launch ----
// ONE pass: every thread loads its 4 items of EVERY 4096-item round up front (independent 16-byte
// loads: one round trip instead of one per round -- the five dependent rounds of a 20 k-item
// catalogue took 33 us), counts them, a wave scan per round, then one exclusive scan over the
// (round, wave) totals in LDS gives every thread its output offsets.
__device__ __forceinline__ void collate_scan_small_body(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    int32_t *__restrict__ pos, int32_t *__restrict__ items, int32_t *__restrict__ counts,
    int n_cap, int nnz_cap, rk_cur_t cur) {
  constexpr int R = SMALL_SCAN_MAX / 4096;          // 16 rounds at most
  __shared__ int32_t wsum[R * 16];                  // [round][wave]: marked items
  __shared__ int32_t wtot[4];
  __shared__ int32_t total_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (cur.cursor) stamp = rk_cur_stamp(cur);
  const int rounds = (n_items + 4095) >> 12;
  int4 m[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int it0 = r * 4096 + tid * 4;
    m[r] = make_int4(0, 0, 0, 0);
    if (r < rounds) {
      if (it0 + 3 < n_items) {
        m[r] = *reinterpret_cast<const int4 *>(mark + it0);
      } else {
        if (it0 + 0 < n_items) m[r].x = mark[it0 + 0];
        if (it0 + 1 < n_items) m[r].y = mark[it0 + 1];
        if (it0 + 2 < n_items) m[r].z = mark[it0 + 2];
      }
    }
  }
  uint32_t fl[R];                                   // 4 flag bits per round
  int32_t incl[R];                                  // inclusive wave scan of the per-thread counts
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int it0 = r * 4096 + tid * 4;
    uint32_t f = 0;
    if (r < rounds) {
      if (it0 + 0 < n_items && (all || m[r].x == stamp)) f |= 1u;
      if (it0 + 1 < n_items && (all || m[r].y == stamp)) f |= 2u;
      if (it0 + 2 < n_items && (all || m[r].z == stamp)) f |= 4u;
      if (it0 + 3 < n_items && (all || m[r].w == stamp)) f |= 8u;
    }
    fl[r] = f;
    int32_t x = __popc(f);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    incl[r] = x;
    if (lane == 63) wsum[r * 16 + wid] = x;
  }
  __syncthreads();
  // exclusive scan of the R * 16 (round, wave) totals, in item order, by the first 4 waves
  int32_t v = 0, vin = 0;
  if (tid < R * 16) {
    v = (tid >> 4) < rounds ? wsum[tid] : 0;
    vin = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t y = __shfl_up(vin, off, 64);
      if (lane >= off) vin += y;
    }
    if (lane == 63) wtot[wid] = vin;
  }
  __syncthreads();
  if (tid < R * 16) {
    int32_t before = 0;
    for (int w = 0; w < wid; ++w) before += wtot[w];
    wsum[tid] = before + vin - v;                   // exclusive offset of (round, wave)
    if (tid == R * 16 - 1) total_s = before + vin;
  }
  __syncthreads();
  const int32_t total = total_s;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < rounds) {
      const int it0 = r * 4096 + tid * 4;
      int32_t p = wsum[r * 16 + wid] + incl[r] - __popc(fl[r]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int it = it0 + k;
        if (it < n_items) {
          if ((fl[r] >> k) & 1u) {
            if (p < n_cap) { pos[it] = p; items[p] = it; } else { pos[it] = -1; }
            ++p;
          } else {
            pos[it] = -1;
          }
        }
      }
    }
  }
  if (tid == 0) {
    const int32_t n_b = min(total, n_cap);          // (see collate_assign_kernel)
    counts[0] = n_b;
    counts[2] = (n_b + 31) & ~31;
    counts[5] = total > n_cap ? total : 0;
  }
}
__global__ __launch_bounds__(1024) void collate_scan_small_kernel(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    int32_t *__restrict__ pos, int32_t *__restrict__ items, int32_t *__restrict__ counts,
    int n_cap, int nnz_cap, rk_cur_t cur) {
  collate_scan_small_body(mark, n_items, stamp, all, pos, items, counts, n_cap, nnz_cap, cur);
}
__global__ __launch_bounds__(1024) void collate_scan_small_multi_kernel(int all, MultiBlk mb, rk_cur_t cur) {
  cur.off += (int)blockIdx.x;
  const rk_block_t &b = mb.b[blockIdx.x];
  collate_scan_small_body(b.mark, b.n_items, 1, all, b.pos, b.items, b.counts, b.n_cap, b.nnz_cap, cur);
}

// ---- scan (small catalogues, batched collation): count + assign of a 2048-item chunk in ONE workgroup, the
// chunks of a block chained by a decoupled look-back -- every workgroup publishes {ready, marked items of its
// chunk} in its 64-bit slot of scan_tmp (zeroed by phase 1 of the same collation) and adds up the slots in front of it (they belong to workgroups
// dispatched before it; agent-scope atomics: the slots cross XCDs within the launch).  n_chunks x n_blk light
// workgroups instead of n_blk workgroups of 1024 threads that hold a CU for 17-30 us each.
__global__ __launch_bounds__(256) void collate_scan_lb_multi_kernel(int all, MultiBlk mb, rk_cur_t cur) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t base_s;
  cur.off += (int)blockIdx.y;
  const rk_block_t &b = mb.b[blockIdx.y];
  const int32_t stamp = rk_cur_stamp(cur);
  const int c = (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_items = b.n_items;
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(b.scan_tmp);
  const int it0 = c * RK_SCAN_CHUNK + tid * 8;
  int32_t mk[8];
  if (it0 + 7 < n_items) {
    const int4 m0 = *reinterpret_cast<const int4 *>(b.mark + it0), m1 = *reinterpret_cast<const int4 *>(b.mark + it0 + 4);
    mk[0] = m0.x; mk[1] = m0.y; mk[2] = m0.z; mk[3] = m0.w; mk[4] = m1.x; mk[5] = m1.y; mk[6] = m1.z; mk[7] = m1.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) mk[k] = it0 + k < n_items ? b.mark[it0 + k] : 0;
  }
  uint32_t f = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (it0 + k < n_items && (all || mk[k] == stamp)) f |= 1u << k;
  const int32_t local = __popc(f);
  int32_t x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  const int32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  constexpr unsigned long long READY = 1ull << 63;      // (phase 1 of this collation zeroed the slots)
  if (tid == 0)
    __hip_atomic_store(slots + c, READY | (uint32_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // look-back: the chunks in front of this one (at most 31: one wave)
  if (wid == 0) {
    int32_t v = 0;
    if (lane < c) {
      unsigned long long s;
      do {
        s = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(s & READY)) __builtin_amdgcn_s_sleep(1);
      } while (!(s & READY));
      v = (int32_t)(uint32_t)s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) base_s = v;
  }
  __syncthreads();
  const int32_t base_cnt = base_s;
  int32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  int32_t p = base_cnt + woff + x - local;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int it = it0 + k;
    if (it < n_items) {
      if ((f >> k) & 1u) {
        if (p < b.n_cap) { b.pos[it] = p; b.items[p] = it; } else { b.pos[it] = -1; }
        ++p;
      } else {
        b.pos[it] = -1;
      }
    }
  }
  if (c == (int)gridDim.x - 1 && tid == 0) {
    const int32_t n_all = base_cnt + total, n_b = min(n_all, b.n_cap);      // (see collate_assign_kernel)
    b.counts[0] = n_b;
    b.counts[2] = (n_b + 31) & ~31;
    b.counts[5] = n_all > b.n_cap ? n_all : 0;
  }
}

// ---- build: one wave per sampled row -- relabelled columns, values, the row's
//      bitmap words + their exclusive prefix popcounts (assembled in LDS and
//      written out whole: bits_rc needs no clearing), transposed-bitmap bits ----
// entries of rows past this go to all four waves of the row's workgroup (below)
constexpr int BUILD_HEAVY = 512;

// one row's entries k = first, first + stride, ... (four per thread and round: the item ids, then their
// positions, are fetched as independent loads -- a 3 000-entry row was 47 rounds of ids -> positions -> stores
// one after the other, the tail of the launch)
__device__ __forceinline__ void build_entries(const rk_block_t &b, const int32_t *__restrict__ ds_indices,
                                              const float *__restrict__ ds_data, const int64_t beg, const int n,
                                              const int out0, const int row, const int first, const int stride) {
  for (int k0 = first; k0 < n; k0 += 4 * stride) {
    int32_t gi[4], c[4];
    float val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * stride;
      gi[u] = k < n ? ds_indices[beg + k] : -1;
      val[u] = (k < n && ds_data) ? ds_data[beg + k] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = gi[u] >= 0 ? b.pos[gi[u]] : -1;   // (< 0 only in a truncated block, counts[5] != 0)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * stride;
      if (k < n) {
        b.cols[out0 + k] = max(c[u], 0);
        if (b.gcols) b.gcols[out0 + k] = gi[u];
        b.vals[out0 + k] = c[u] < 0 ? 0.0f : val[u];
        if (b.bits_cr && c[u] >= 0) atomicOr(&b.bits_cr[(int64_t)c[u] * b.ldw_cr + (row >> 5)], 1u << (row & 31));
      }
    }
  }
}
// the row's bits of bitmap words [w0, w0 + nw) into the LDS segment wb (positions looked up again: the cols this
// pass could read back were written by other threads of the workgroup when the row is shared)
__device__ __forceinline__ void build_bits(const rk_block_t &b, const int32_t *__restrict__ ds_indices, const int64_t beg,
                                           const int n, uint32_t *wb, const int w0, const int nw, const int first,
                                           const int stride) {
  for (int k0 = first; k0 < n; k0 += 4 * stride) {
    int32_t gi[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) gi[u] = k0 + u * stride < n ? ds_indices[beg + k0 + u * stride] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = gi[u] >= 0 ? b.pos[gi[u]] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int w = (c[u] >> 5) - w0;
      if (c[u] >= 0 && w >= 0 && w < nw) atomicOr(&wb[w], 1u << (c[u] & 31));
    }
  }
}
// words [w0, w0 + nw) of the row's bitmap + their exclusive prefix popcounts, by ONE wave
__device__ __forceinline__ void build_words(const uint32_t *wb, const int w0, const int nw, uint32_t *bits, int32_t *pref,
                                            int32_t &carry, const int lane) {
  for (int wq = 0; wq < nw; wq += 64) {
    const int w = wq + lane;
    const uint32_t word = (w < nw) ? wb[w] : 0u;
    const int32_t c = __popc(word);
    int32_t x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (w < nw) {
      bits[w0 + w] = word;
      if (pref) pref[w0 + w] = carry + x - c;
    }
    carry += __shfl(x, 63, 64);
  }
}

__device__ __forceinline__ void collate_build_body(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const float *__restrict__ ds_data, const int64_t *__restrict__ users, int S, const rk_block_t &b,
    rk_cur_t cur, const int seg, const int bx) {
  extern __shared__ uint32_t wbits[];               // [4 waves][seg words]
  __shared__ int heavy_n[4];
  if (cur.cursor) users += rk_cur_local(cur) * S;
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const int row = bx * 4 + wid;
  const bool live = row < S;
  const int wr = (b.counts[0] + 31) >> 5;          // bitmap words in use
  int64_t beg = 0;
  int n = 0, out0 = 0;
  if (live) {
    const int64_t u = users[row];
    beg = ds_indptr[u];
    n = (int)(ds_indptr[u + 1] - beg);
    out0 = b.indptr[row];
    n = min(n, b.indptr[row + 1] - out0);          // (the clamped range of a truncated block)
  }
  // a wave per row; rows past BUILD_HEAVY entries wait for all four waves (the launch was as long as its
  // longest row: 12 rounds of dependent loads for 3 000 entries on one wave, 3 on four)
  const bool heavy = live && n > BUILD_HEAVY;
  if (lane == 0) heavy_n[wid] = heavy ? 1 : 0;
  if (live && !heavy) {
    build_entries(b, ds_indices, ds_data, beg, n, out0, row, lane, 64);
    uint32_t *wb = wbits + wid * seg;
    int32_t carry = 0;
    for (int w0 = 0; w0 < wr; w0 += seg) {
      const int nw = min(seg, wr - w0);
      for (int w = lane; w < nw; w += 64) wb[w] = 0u;
      __builtin_amdgcn_wave_barrier();
      build_bits(b, ds_indices, beg, n, wb, w0, nw, lane, 64);
      __builtin_amdgcn_wave_barrier();
      build_words(wb, w0, nw, b.bits_rc + (int64_t)row * b.ldw_rc,
                  b.pref_rc ? b.pref_rc + (int64_t)row * b.ldw_rc : nullptr, carry, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int j = 0; j < 4; ++j) {
    if (heavy_n[j] == 0) continue;                   // (uniform)
    const int rj = bx * 4 + j;
    const int64_t u = users[rj];
    const int64_t bj = ds_indptr[u];
    const int oj = b.indptr[rj];
    const int nj = min((int)(ds_indptr[u + 1] - bj), b.indptr[rj + 1] - oj);
    build_entries(b, ds_indices, ds_data, bj, nj, oj, rj, tid, 256);
    uint32_t *wb = wbits + j * seg;
    int32_t carry = 0;
    for (int w0 = 0; w0 < wr; w0 += seg) {
      const int nw = min(seg, wr - w0);
      for (int w = tid; w < nw; w += 256) wb[w] = 0u;
      __syncthreads();
      build_bits(b, ds_indices, bj, nj, wb, w0, nw, tid, 256);
      __syncthreads();
      if (wid == 0)
        build_words(wb, w0, nw, b.bits_rc + (int64_t)rj * b.ldw_rc,
                    b.pref_rc ? b.pref_rc + (int64_t)rj * b.ldw_rc : nullptr, carry, lane);
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(256) void collate_build_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const float *__restrict__ ds_data, const int64_t *__restrict__ users, int S, rk_block_t b,
    rk_cur_t cur, int seg) {
  collate_build_body(ds_indptr, ds_indices, ds_data, users, S, b, cur, seg, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void collate_build_multi_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const float *__restrict__ ds_data, const int64_t *__restrict__ users, int S, MultiBlk mb,
    rk_cur_t cur, int seg, int nbx) {
  cur.off += (int)blockIdx.y;
  for (int bx = blockIdx.x; bx < nbx; bx += gridDim.x) {        // (see collate_phase1_multi_kernel)
    collate_build_body(ds_indptr, ds_indices, ds_data, users, S, mb.b[blockIdx.y], cur, seg, bx);
    __syncthreads();
  }
}

}  // namespace

// ---- densify: rows [row_off, row_off+B) of the block as a dense [B, ld] fp32 matrix
//      (model.py:457-458 `torch.sparse.FloatTensor(...).to_dense()`); only the generic
//      torch-autograd path (user-defined models / losses / optimizers) needs it ----
namespace {
__global__ __launch_bounds__(256) void densify_kernel(rk_block_t b, int row_off, int n, int ld,
                                                      float *__restrict__ out) {
  const int r = blockIdx.x;
  float *orow = out + (int64_t)r * ld;
  for (int c = threadIdx.x; c < n; c += 256) orow[c] = 0.f;
  __syncthreads();
  const int beg = b.indptr[row_off + r], end = b.indptr[row_off + r + 1];
  for (int j = beg + threadIdx.x; j < end; j += 256) orow[b.cols[j]] = b.vals[j];
}
}  // namespace

extern "C" int rk_densify(const rk_block_t *blk, int32_t row_off, int32_t B, int32_t n, float *out,
                          int32_t ld, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= blk->S_cap, "row slice out of range");
  RK_REQUIRE(n >= 0 && n <= ld, "n must not exceed the leading dimension");
  if (B == 0 || n == 0) return 0;
  RK_LAUNCH(densify_kernel, dim3(B), dim3(256), 0, stream, *blk, row_off, n, ld, out);
  RK_CHECK_LAUNCH("densify");
  return 0;
}

static int collate_impl(const int64_t *ds_indptr, const int32_t *ds_indices,
                        const float *ds_data, const int64_t *users, int32_t S,
                        int32_t negative_sampling, int32_t stamp, int32_t phase,
                        const rk_block_t *blk, rk_cur_t cur, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(blk != nullptr, "null block");
  RK_REQUIRE(S >= 0 && S <= blk->S_cap, "S exceeds block capacity");
  RK_REQUIRE(blk->n_chunks == rk_cdiv(blk->n_items, RK_SCAN_CHUNK), "n_chunks mismatch");
  RK_REQUIRE(blk->ldw_rc * 32 >= blk->n_cap && blk->ldw_cr * 32 >= blk->S_cap, "bitmap ld");
  RK_REQUIRE(stamp != 0, "stamp must be non-zero");
  RK_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0, 1 or 2");
  if (S == 0) return 0;
  const int all = negative_sampling ? 0 : 1;
  if (phase != 2) {
    const int nrow_blk = rk_cdiv(S, 4);
    int nzero = 0;
    if (blk->bits_cr) {
      nzero = rk_cdiv((int64_t)blk->n_cap * blk->ldw_cr, 4 * 256 * 8);
      if (nzero < 1) nzero = 1;
      if (nzero > 512) nzero = 512;
    }
    RK_LAUNCH(collate_phase1_kernel, dim3(nrow_blk + 1 + nzero), dim3(256), 0, stream,
                       ds_indptr, ds_indices, users, S, stamp, all, nrow_blk, *blk, cur);
    RK_CHECK_LAUNCH("collate_phase1");
  }
  if (phase == 1) return 0;
  if (blk->n_items <= SMALL_SCAN_MAX) {
    RK_LAUNCH(collate_scan_small_kernel, dim3(1), dim3(1024), 0, stream, blk->mark,
                       blk->n_items, stamp, all, blk->pos, blk->items, blk->counts, blk->n_cap,
                       blk->nnz_cap, cur);
    RK_CHECK_LAUNCH("collate_scan_small");
  } else {
    RK_LAUNCH(collate_count_kernel, dim3(blk->n_chunks), dim3(256), 0, stream,
                       blk->mark, blk->n_items, stamp, all, blk->scan_tmp, cur);
    RK_CHECK_LAUNCH("collate_count");
    RK_LAUNCH(collate_assign_kernel, dim3(blk->n_chunks), dim3(256), 0, stream,
                       blk->mark, blk->n_items, stamp, all, blk->scan_tmp, blk->n_chunks,
                       blk->pos, blk->items, blk->counts, blk->n_cap, blk->nnz_cap, cur);
    RK_CHECK_LAUNCH("collate_assign");
  }
  const int seg = seg_words_for(blk->n_cap);
  RK_LAUNCH(collate_build_kernel, dim3(rk_cdiv(S, 4)), dim3(256), 4 * seg * sizeof(uint32_t), stream,
                     ds_indptr, ds_indices, ds_data, users, S, *blk, cur, seg);
  RK_CHECK_LAUNCH("collate_build");
  return 0;
}

extern "C" int rk_collate(const int64_t *ds_indptr, const int32_t *ds_indices,
                          const float *ds_data, const int64_t *users, int32_t S,
                          int32_t negative_sampling, int32_t stamp, int32_t phase,
                          const rk_block_t *blk, void *stream_) {
  const rk_cur_t none = {nullptr, 0};
  return collate_impl(ds_indptr, ds_indices, ds_data, users, S, negative_sampling, stamp, phase, blk,
                      none, stream_);
}

extern "C" int rk_collate_at(const int64_t *ds_indptr, const int32_t *ds_indices,
                             const float *ds_data, const int64_t *users_base, int32_t S,
                             int32_t negative_sampling, const int64_t *cursor, int32_t off,
                             const rk_block_t *blk, void *stream_) {
  RK_REQUIRE(cursor != nullptr, "null cursor");
  const rk_cur_t cur = {cursor, off};
  return collate_impl(ds_indptr, ds_indices, ds_data, users_base, S, negative_sampling, 1, 0, blk, cur,
                      stream_);
}

// rk_collate_at for n_blk blocks in ONE set of launches (block g: cursor offset off0 + g): the G
// look-ahead blocks of a replayed group, or the G blocks behind a cut, cost 3 launches on one queue
// instead of 3 G on G queues (a 4-branch graph took 140 us to start them under the profiler)
// phase as rk_collate: 1 = row pointers + item marking, 2 = the rest (data-parallel replay: the
// MAX all-reduce of the n_blk mark arrays goes between them, captured with the launches)
extern "C" int rk_collate_at_multi(const int64_t *ds_indptr, const int32_t *ds_indices,
                                   const float *ds_data, const int64_t *users_base, int32_t S,
                                   int32_t negative_sampling, const int64_t *cursor, int32_t off0,
                                   const rk_block_t *const *blks, int32_t n_blk, int32_t phase,
                                   void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0, 1 or 2");
  RK_REQUIRE(cursor != nullptr && blks != nullptr, "null cursor / blocks");
  RK_REQUIRE(n_blk >= 1 && n_blk <= RK_COLLATE_MULTI, "1 .. RK_COLLATE_MULTI blocks");
  MultiBlk mb = {};
  const rk_block_t *b0 = blks[0];
  for (int g = 0; g < n_blk; ++g) {
    const rk_block_t *blk = blks[g];
    RK_REQUIRE(blk != nullptr, "null block");
    RK_REQUIRE(S >= 0 && S <= blk->S_cap, "S exceeds block capacity");
    RK_REQUIRE(blk->n_chunks == rk_cdiv(blk->n_items, RK_SCAN_CHUNK), "n_chunks mismatch");
    RK_REQUIRE(blk->ldw_rc * 32 >= blk->n_cap && blk->ldw_cr * 32 >= blk->S_cap, "bitmap ld");
    RK_REQUIRE(blk->n_items == b0->n_items && blk->n_cap == b0->n_cap && blk->ldw_cr == b0->ldw_cr &&
               (blk->bits_cr != nullptr) == (b0->bits_cr != nullptr), "the blocks must have one shape");
    mb.b[g] = *blk;
  }
  if (S == 0) return 0;
  const rk_cur_t cur = {cursor, off0};
  const int all = negative_sampling ? 0 : 1;
  const int nrow_blk = rk_cdiv(S, 4);
  int nzero = 0;
  if (b0->bits_cr) {
    nzero = rk_cdiv((int64_t)b0->n_cap * b0->ldw_cr, 4 * 256 * 8);
    if (nzero < 1) nzero = 1;
    if (nzero > 512) nzero = 512;
  }
  // workgroups per block of the two row-parallel launches.  These launches run BESIDE the training chain of a replayed
  // group (graph.GraphStepper's look-ahead collation), which pays for every wave slot they hold: 8 x 32 looping workgroups
  // instead of 8 x 125-150 short ones -- the collation takes 81 instead of 60 us of its stream, the chain loses less to it
  // (C2, alternating on one box, four boxes: 0.0933-0.0948 vs 0.0954-0.0970 ms per step; caps of 16 .. 64 within 0.5 us
  // of each other, 4 no better than none)
  constexpr int cap = 32;
  if (phase != 2) {
    const int nbx = nrow_blk + 1 + nzero;
    RK_LAUNCH(collate_phase1_multi_kernel, dim3(std::min(cap, nbx), n_blk), dim3(256), 0, stream, ds_indptr,
              ds_indices, users_base, S, all, nrow_blk, mb, cur, nbx);
    RK_CHECK_LAUNCH("collate_phase1_multi");
  }
  if (phase == 1) return 0;
  if (b0->n_items <= SMALL_SCAN_MAX) {
    // (<= 32 chunks: the look-back fits one wave; scan_tmp holds 64-bit slots -- rk_block_t.scan_tmp)
    RK_LAUNCH(collate_scan_lb_multi_kernel, dim3(b0->n_chunks, n_blk), dim3(256), 0, stream, all, mb, cur);
    RK_CHECK_LAUNCH("collate_scan_lb_multi");
  } else {
    RK_LAUNCH(collate_count_multi_kernel, dim3(b0->n_chunks, n_blk), dim3(256), 0, stream, all, mb, cur);
    RK_CHECK_LAUNCH("collate_count_multi");
    RK_LAUNCH(collate_assign_multi_kernel, dim3(b0->n_chunks, n_blk), dim3(256), 0, stream, all, mb, cur);
    RK_CHECK_LAUNCH("collate_assign_multi");
  }
  const int seg = seg_words_for(b0->n_cap);
  RK_LAUNCH(collate_build_multi_kernel, dim3(std::min(cap, rk_cdiv(S, 4)), n_blk), dim3(256),
            4 * seg * sizeof(uint32_t), stream, ds_indptr, ds_indices, ds_data, users_base, S, mb, cur, seg, rk_cdiv(S, 4));
  RK_CHECK_LAUNCH("collate_build_multi");
  return 0;
}

// ---- need lists of the lazy dense Adam (include/recoder_hip.h rk_lazy_need_lists): pair y = (block y, block y + 1),
//      workgroup x = chunk x of RK_SCAN_CHUNK item ids.  The chunk's offset in the list is counted by the workgroup
//      itself over the ids in front of it (at most 63 chunks of two L2-resident maps: no temporaries, no second launch) ----
namespace {
struct NeedLists {
  const int32_t *pos_a[RK_COLLATE_MULTI], *pos_b[RK_COLLATE_MULTI];
  int32_t *list[RK_COLLATE_MULTI], *count[RK_COLLATE_MULTI];
  int n_items, n_chunks;
};
__global__ __launch_bounds__(256) void need_lists_kernel(NeedLists p) {
  __shared__ int32_t red[4], wsum[4];
  const int y = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int32_t *pa = p.pos_a[y], *pb = p.pos_b[y];
  // (16-byte loads, four per map in flight: one id per thread and trip was a chain of dependent round trips --
  // 47-61 us per launch at C2's ten chunks)
  int32_t part = 0;
  const int4 *pa4 = reinterpret_cast<const int4 *>(pa), *pb4 = reinterpret_cast<const int4 *>(pb);
  const int n4 = c * (RK_SCAN_CHUNK / 4);
  for (int i0 = tid; i0 < n4; i0 += 4 * 256) {
    int4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      va[u] = i < n4 ? pa4[i] : make_int4(-1, -1, -1, -1);
      vb[u] = i < n4 ? pb4[i] : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      part += ((va[u].x >= 0 || vb[u].x >= 0) ? 1 : 0) + ((va[u].y >= 0 || vb[u].y >= 0) ? 1 : 0) +
              ((va[u].z >= 0 || vb[u].z >= 0) ? 1 : 0) + ((va[u].w >= 0 || vb[u].w >= 0) ? 1 : 0);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if (lane == 0) red[wid] = part;
  __syncthreads();
  const int32_t base = red[0] + red[1] + red[2] + red[3];
  const int it0 = c * RK_SCAN_CHUNK + tid * 8;
  int32_t f[8], local = 0;
  int32_t va8[8], vb8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {               // (loads first: independent)
    const int it = it0 + k;
    va8[k] = it < p.n_items ? pa[it] : -1;
    vb8[k] = it < p.n_items ? pb[it] : -1;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f[k] = (va8[k] >= 0 || vb8[k] >= 0) ? 1 : 0;
    local += f[k];
  }
  int32_t x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t v = __shfl_up(x, off, 64);
    if (lane >= off) x += v;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  int32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  int32_t o = base + woff + x - local;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (f[k]) p.list[y][o++] = it0 + k;
  if (c == p.n_chunks - 1 && tid == 255) p.count[y][0] = o;      // (the last thread of the last chunk: the total)
}
}  // namespace

extern "C" int rk_lazy_need_lists(const rk_block_t *const *blks, int32_t n_blk, int32_t *const *lists,
                                  int32_t *const *counts, void *stream_) {
  RK_REQUIRE(blks != nullptr && n_blk >= 1 && n_blk <= RK_COLLATE_MULTI, "1 .. RK_COLLATE_MULTI blocks");
  if (n_blk < 2) return 0;
  RK_REQUIRE(lists != nullptr && counts != nullptr, "null lists / counts");
  NeedLists p = {};
  p.n_items = blks[0]->n_items; p.n_chunks = blks[0]->n_chunks;
  RK_REQUIRE(p.n_items >= 1 && p.n_items <= RK_NEED_LIST_MAX_ITEMS && p.n_chunks == rk_cdiv(p.n_items, RK_SCAN_CHUNK),
             "catalogue too large for need lists (RK_NEED_LIST_MAX_ITEMS)");
  for (int i = 0; i + 1 < n_blk; ++i) {
    RK_REQUIRE(blks[i] && blks[i + 1] && blks[i + 1]->n_items == p.n_items && lists[i] && counts[i], "null block / list");
    RK_REQUIRE((((uintptr_t)blks[i]->pos | (uintptr_t)blks[i + 1]->pos) & 15) == 0, "pos maps must be 16-byte aligned");
    p.pos_a[i] = blks[i]->pos; p.pos_b[i] = blks[i + 1]->pos; p.list[i] = lists[i]; p.count[i] = counts[i];
  }
  RK_LAUNCH(need_lists_kernel, dim3(p.n_chunks, n_blk - 1), dim3(256), 0, (hipStream_t)stream_, p);
  RK_CHECK_LAUNCH("lazy_need_lists");
  return 0;
}

namespace {
__global__ void cursor_set_kernel(int64_t *cursor, int64_t step, int64_t epoch_base, int64_t add) {
  if (add) { cursor[0] += add; return; }
  cursor[0] = step;
  cursor[1] = epoch_base;
}
}  // namespace

extern "C" int rk_cursor_set(int64_t *cursor, int64_t step, int64_t epoch_base, void *stream_) {
  RK_LAUNCH(cursor_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, cursor, step, epoch_base,
            (int64_t)0);
  RK_CHECK_LAUNCH("cursor_set");
  return 0;
}
