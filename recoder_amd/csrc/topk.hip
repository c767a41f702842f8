// Masked top-k for Recoder.recommend (reference model.py:525-544):
//   output[input > 0] = -inf ; torch.topk(output, k, dim=1, sorted=True)
// One workgroup per user row: 4-pass 8-bit radix select of the k-th largest
// order-preserving key, ordered collection of the survivors (ties resolved to
// the lower item id), then a bitonic sort of the <= KMAX winners in LDS.
// HBM-bound: the score row is read 5 times from L2 (n * 4 B per row).
#include "common.h"

namespace {

constexpr int KMAX = 1024;

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void topk_masked_kernel(const float *scores, int n, int ld,
                                                          rk_block_t seen, int has_seen,
                                                          int row_off, int k, int64_t *out_idx,
                                                          float *out_val, int col_off, int out_ld,
                                                          int col_stride) {
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long cand[KMAX];
  __shared__ uint32_t s_prefix, s_need, s_cnt, s_tie;
  __shared__ uint32_t wtot[4];
  const int r = blockIdx.x, row = row_off + r;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float *srow = scores + (int64_t)r * ld;
  const uint32_t *bits = has_seen ? seen.bits_rc + (int64_t)row * seen.ldw_rc : nullptr;

  // score column c is item col_off + c of the (unsampled) input block; as the reference does
  // (`output[input > 0] = -inf`), only POSITIVE stored interactions are masked
  const bool implicit = seen.implicit != 0;
  auto key_at = [&](int c) -> uint32_t {
    float f = srow[c];
    if (bits) {
      const int gc = col_off + c * col_stride;
      const uint32_t word = bits[gc >> 5];
      if ((word >> (gc & 31)) & 1u) {
        const float v = implicit ? 1.0f : seen.vals[rk_entry_index(seen, row, gc, word)];
        if (v > 0.f) f = -INFINITY;
      }
    }
    return f2key(f);
  };

  // ---- radix select: find key T of the k-th largest element ----
  uint32_t prefix = 0, pmask = 0, need = (uint32_t)k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
      const uint32_t key = key_at(c);
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= need) break;
        acc += hist[d];
      }
      s_prefix = prefix | ((uint32_t)d << shift);
      s_need = need - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    pmask |= 255u << shift;
    __syncthreads();
  }
  const uint32_t T = prefix;       // k-th largest key; `need` = how many ties of T to take
  if (tid == 0) { s_cnt = 0; s_tie = 0; }
  __syncthreads();
  // ---- collect: key > T (any order), key == T in ascending index order ----
  for (int base = 0; base < n; base += 256) {
    const int c = base + tid;
    uint32_t key = 0;
    bool gt = false, eq = false;
    if (c < n) {
      key = key_at(c);
      gt = key > T;
      eq = key == T;
    }
    // ordered rank of the ties inside this 256-chunk
    const unsigned long long bm = __ballot(eq);
    const uint32_t wrank = __popcll(bm & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wid] = __popcll(bm);
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    const uint32_t chunk_tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const uint32_t tie_base = s_tie;
    bool take = gt;
    if (eq && (tie_base + woff + wrank) < need) take = true;
    if (take) {
      const uint32_t slot = atomicAdd(&s_cnt, 1u);
      // composite: key descending, then index ascending
      if (slot < KMAX) cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)c);
    }
    __syncthreads();
    if (tid == 0) s_tie = tie_base + chunk_tot;
    __syncthreads();
  }
  const int cnt = (int)min(s_cnt, (uint32_t)KMAX);   // == k
  // ---- bitonic sort (descending) of cand[0..P) padded with 0 ----
  int P = 1;
  while (P < cnt) P <<= 1;
  for (int i = cnt + tid; i < P; i += 256) cand[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = ((i & size) == 0);
          const unsigned long long a = cand[i], b = cand[j];
          if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 256) {
    const unsigned long long e = cand[i];
    const uint32_t c = ~(uint32_t)(e & 0xffffffffull);
    out_idx[(int64_t)r * out_ld + i] = (int64_t)c * col_stride + col_off;
    if (out_val) out_val[(int64_t)r * out_ld + i] = key2f((uint32_t)(e >> 32));
  }
}

// Final pass of the fused-filter recommend: row r holds cnt[r] (<= cap) candidate (score, item id)
// pairs in ANY order; its k best by (score descending, id ascending) -> out_idx[r][0..k).  One
// workgroup per row: bitonic sort of the composite keys in LDS.  status[0] |= 1 if a row overflowed
// its list (cnt > cap), |= 2 if it has fewer than k candidates: the caller then takes the strip path.
constexpr int PAIRS_CAP = 8192;
__global__ __launch_bounds__(256) void topk_pairs_kernel(const float *__restrict__ val,
                                                         const int32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ cnt, int cap, int k,
                                                         int64_t *out_idx, int out_ld, int32_t *status) {
  extern __shared__ unsigned long long keys[];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int c0 = cnt[r];
  if (c0 > cap || c0 < k) {
    if (tid == 0) atomicOr(status, c0 > cap ? 1 : 2);
    return;
  }
  int P = 1;
  while (P < c0) P <<= 1;
  for (int i = tid; i < P; i += 256) {
    unsigned long long e = 0ull;
    if (i < c0) e = ((unsigned long long)f2key(val[(int64_t)r * cap + i]) << 32) |
                    (uint32_t)(~(uint32_t)idx[(int64_t)r * cap + i]);
    keys[i] = e;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = ((i & size) == 0);
          const unsigned long long a = keys[i], b = keys[j];
          if ((a < b) == desc) { keys[i] = b; keys[j] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 256)
    out_idx[(int64_t)r * out_ld + i] = (int64_t)(~(uint32_t)(keys[i] & 0xffffffffull));
}

}  // namespace

extern "C" int rk_topk_pairs(const float *val, const int32_t *idx, const int32_t *cnt, int32_t B,
                             int32_t cap, int32_t k, int64_t *out_idx, int32_t out_ld, int32_t *status,
                             void *stream_) {
  RK_REQUIRE(cap >= 1 && cap <= PAIRS_CAP && (cap & (cap - 1)) == 0, "cap: a power of two <= 8192");
  RK_REQUIRE(k >= 1 && k <= cap && out_ld >= k, "k / out_ld");
  if (B == 0) return 0;
  RK_LAUNCH(topk_pairs_kernel, dim3(B), dim3(256), cap * 8, (hipStream_t)stream_, val, idx, cnt, cap, k,
            out_idx, out_ld, status);
  RK_CHECK_LAUNCH("topk_pairs");
  return 0;
}

extern "C" int32_t rk_topk_pairs_max_cap(void) { return PAIRS_CAP; }

extern "C" int rk_topk_masked(const float *scores, int32_t B, int32_t n, int32_t ld,
                              const rk_block_t *seen, int32_t row_off, int32_t k,
                              int32_t col_off, int32_t col_stride, int64_t *out_idx,
                              float *out_val, int32_t out_ld, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(k >= 1 && k <= KMAX, "k must be in [1, 1024]");
  RK_REQUIRE(k <= n, "k larger than the number of score columns");
  RK_REQUIRE(out_ld >= k && col_off >= 0 && col_stride >= 1, "bad output layout");
  RK_REQUIRE(seen == nullptr || seen->implicit || seen->pref_rc != nullptr, "explicit values need pref_rc");
  if (B == 0) return 0;
  rk_block_t dummy = {};
  RK_LAUNCH(topk_masked_kernel, dim3(B), dim3(256), 0, stream, scores, n, ld,
                     seen ? *seen : dummy, seen ? 1 : 0, row_off, k, out_idx, out_val, col_off, out_ld,
                     col_stride);
  RK_CHECK_LAUNCH("topk_masked");
  return 0;
}

extern "C" int32_t rk_topk_max_k(void) { return KMAX; }
