// On-device sparse-batch collation.
//
// Replaces RecommendationDataset._extract (reference data.py:64-83: CSR row
// gather) and BatchCollator.collate (data.py:203-251: nonzero() ->
// np.unique(return_inverse) -> per-slice COO) with HBM-resident integer work:
//   rows     : degrees of the S sampled users -> exclusive scan (block CSR)
//   mark     : stamp every touched item id            (scatter, 4 B / nnz)
//   count    : per-2048-chunk population of the stamp array
//   assign   : chunk-ordered exclusive scan -> pos[item] / items[] (ascending
//              item id == np.unique order) and n_b
//   zero     : clear the (row,col) and (col,row) bitmaps for n_b columns
//   relabel  : cols[j] = pos[item_j], vals[j], set bitmap bits
// All HBM-bound; one pass over the group's nnz + two passes over n_items ints.
#include "common.h"

namespace {

constexpr int ROWS_THREADS = 1024;

// ---- rows: block CSR row pointers of the S sampled users (single block) ----
__global__ __launch_bounds__(ROWS_THREADS) void collate_rows_kernel(
    const int64_t *__restrict__ ds_indptr, const int64_t *__restrict__ users, int S,
    int32_t *__restrict__ indptr, int32_t *__restrict__ counts) {
  __shared__ int32_t wsum[ROWS_THREADS / 64];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < S; base += ROWS_THREADS) {
    const int i = base + tid;
    int32_t d = 0;
    if (i < S) {
      const int64_t u = users[i];
      d = (int32_t)(ds_indptr[u + 1] - ds_indptr[u]);
    }
    // inclusive wave scan
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
    if (i < S) indptr[i] = carry + woff + x - d;
    __syncthreads();
    if (tid == ROWS_THREADS - 1) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    indptr[S] = carry_s;
    counts[1] = carry_s;
    counts[3] = S;
  }
}

// ---- mark: one wave per sampled row ----
__global__ __launch_bounds__(256) void collate_mark_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const int64_t *__restrict__ users, int S, int32_t stamp, int32_t *__restrict__ mark) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const int lane = threadIdx.x & 63;
  const int64_t u = users[row];
  const int64_t beg = ds_indptr[u], end = ds_indptr[u + 1];
  for (int64_t e = beg + lane; e < end; e += 64) mark[ds_indices[e]] = stamp;
}

// ---- count: marked items per chunk ----
__global__ __launch_bounds__(256) void collate_count_kernel(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    int32_t *__restrict__ scan_tmp) {
  __shared__ int32_t ws[4];
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

// ---- assign: pos[] / items[] in ascending item order ----
__global__ __launch_bounds__(256) void collate_assign_kernel(
    const int32_t *__restrict__ mark, int n_items, int32_t stamp, int all,
    const int32_t *__restrict__ scan_tmp, int n_chunks, int32_t *__restrict__ pos,
    int32_t *__restrict__ items, int32_t *__restrict__ counts) {
  __shared__ int32_t red[4];
  __shared__ int32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // base = sum of the chunk counts before this chunk (fixed order, integer)
  int32_t part = 0;
  for (int i = tid; i < (int)blockIdx.x; i += 256) part += scan_tmp[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if (lane == 0) red[wid] = part;
  __syncthreads();
  const int32_t base_cnt = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (blockIdx.x == (unsigned)(n_chunks - 1) && tid == 0) {
    const int32_t n_b = base_cnt + scan_tmp[blockIdx.x];
    counts[0] = n_b;
    counts[2] = (n_b + 31) & ~31;
  }
  // each thread owns 8 consecutive items of the chunk
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
        pos[it] = p;
        items[p] = it;
        ++p;
      } else {
        pos[it] = -1;
      }
    }
  }
}

// ---- zero the two bitmaps for the live region ----
__global__ __launch_bounds__(256) void collate_zero_bits_kernel(rk_block_t b) {
  const int n_b = b.counts[0];
  const int S = b.counts[3];
  const int wr = (n_b + 31) >> 5;   // words per row in use
  const int wc = (S + 31) >> 5;     // words per column in use
  const int64_t tot_rc = (int64_t)S * wr;
  const int64_t tot_cr = b.bits_cr ? (int64_t)n_b * wc : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot_rc; i += stride) {
    const int r = (int)(i / wr), w = (int)(i % wr);
    b.bits_rc[(int64_t)r * b.ldw_rc + w] = 0u;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot_cr; i += stride) {
    const int c = (int)(i / wc), w = (int)(i % wc);
    b.bits_cr[(int64_t)c * b.ldw_cr + w] = 0u;
  }
}

// ---- relabel: one wave per row ----
__global__ __launch_bounds__(256) void collate_relabel_kernel(
    const int64_t *__restrict__ ds_indptr, const int32_t *__restrict__ ds_indices,
    const float *__restrict__ ds_data, const int64_t *__restrict__ users, int S, rk_block_t b) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const int lane = threadIdx.x & 63;
  const int64_t u = users[row];
  const int64_t beg = ds_indptr[u];
  const int n = (int)(ds_indptr[u + 1] - beg);
  const int out0 = b.indptr[row];
  for (int k = lane; k < n; k += 64) {
    const int32_t it = ds_indices[beg + k];
    const int32_t c = b.pos[it];
    b.cols[out0 + k] = c;
    b.vals[out0 + k] = ds_data ? ds_data[beg + k] : 1.0f;
    atomicOr(&b.bits_rc[(int64_t)row * b.ldw_rc + (c >> 5)], 1u << (c & 31));
    if (b.bits_cr) atomicOr(&b.bits_cr[(int64_t)c * b.ldw_cr + (row >> 5)], 1u << (row & 31));
  }
}

// ---- prefix: per-row exclusive prefix popcount of the (row,col) bitmap ----
__global__ __launch_bounds__(256) void collate_prefix_kernel(rk_block_t b, int S) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const int lane = threadIdx.x & 63;
  const int wr = (b.counts[0] + 31) >> 5;
  const uint32_t *bits = b.bits_rc + (int64_t)row * b.ldw_rc;
  int32_t *pref = b.pref_rc + (int64_t)row * b.ldw_rc;
  int32_t carry = 0;
  for (int w0 = 0; w0 < wr; w0 += 64) {
    const int w = w0 + lane;
    const int32_t c = (w < wr) ? __popc(bits[w]) : 0;
    int32_t x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (w < wr) pref[w] = carry + x - c;
    carry += __shfl(x, 63, 64);
  }
}

}  // namespace

extern "C" int rk_collate(const int64_t *ds_indptr, const int32_t *ds_indices,
                          const float *ds_data, const int64_t *users, int32_t S,
                          int32_t negative_sampling, int32_t stamp, int32_t phase,
                          const rk_block_t *blk, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(blk != nullptr, "null block");
  RK_REQUIRE(S >= 0 && S <= blk->S_cap, "S exceeds block capacity");
  RK_REQUIRE(blk->n_chunks == rk_cdiv(blk->n_items, RK_SCAN_CHUNK), "n_chunks mismatch");
  RK_REQUIRE(blk->ldw_rc * 32 >= blk->n_cap && blk->ldw_cr * 32 >= blk->S_cap, "bitmap ld");
  RK_REQUIRE(stamp != 0, "stamp must be non-zero");
  if (S == 0) return 0;
  const int all = negative_sampling ? 0 : 1;
  RK_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0, 1 or 2");
  if (phase != 2) {
  hipLaunchKernelGGL(collate_rows_kernel, dim3(1), dim3(ROWS_THREADS), 0, stream, ds_indptr,
                     users, S, blk->indptr, blk->counts);
  RK_CHECK_LAUNCH("collate_rows");
  if (!all) {
    hipLaunchKernelGGL(collate_mark_kernel, dim3(rk_cdiv(S, 4)), dim3(256), 0, stream,
                       ds_indptr, ds_indices, users, S, stamp, blk->mark);
    RK_CHECK_LAUNCH("collate_mark");
  }
  }
  if (phase == 1) return 0;
  hipLaunchKernelGGL(collate_count_kernel, dim3(blk->n_chunks), dim3(256), 0, stream,
                     blk->mark, blk->n_items, stamp, all, blk->scan_tmp);
  RK_CHECK_LAUNCH("collate_count");
  hipLaunchKernelGGL(collate_assign_kernel, dim3(blk->n_chunks), dim3(256), 0, stream,
                     blk->mark, blk->n_items, stamp, all, blk->scan_tmp, blk->n_chunks,
                     blk->pos, blk->items, blk->counts);
  RK_CHECK_LAUNCH("collate_assign");
  {
    const int64_t words = (int64_t)S * blk->ldw_rc + (int64_t)blk->n_cap * blk->ldw_cr;
    int grid = (int)((words + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(collate_zero_bits_kernel, dim3(grid), dim3(256), 0, stream, *blk);
    RK_CHECK_LAUNCH("collate_zero_bits");
  }
  hipLaunchKernelGGL(collate_relabel_kernel, dim3(rk_cdiv(S, 4)), dim3(256), 0, stream,
                     ds_indptr, ds_indices, ds_data, users, S, *blk);
  RK_CHECK_LAUNCH("collate_relabel");
  if (blk->pref_rc) {
    hipLaunchKernelGGL(collate_prefix_kernel, dim3(rk_cdiv(S, 4)), dim3(256), 0, stream, *blk, S);
    RK_CHECK_LAUNCH("collate_prefix");
  }
  return 0;
}
