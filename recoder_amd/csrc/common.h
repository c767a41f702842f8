// Shared device helpers for librecoder_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/recoder_hip.h"
#include "../../include/recoder_hip_probe.h"
#include "internal.h"
int rk_tune_get(int knob);    // capi.hip: the value of an RK_TUNE_* knob (include/recoder_hip_probe.h)

void rk_set_error(const char *fmt, ...);

#define RK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)

#define RK_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      rk_set_error("%s: %s", name, hipGetErrorString(e__));                \
      return -1;                                                           \
    }                                                                      \
  } while (0)

#define RK_REQUIRE(cond, msg)                                              \
  do {                                                                     \
    if (!(cond)) {                                                         \
      rk_set_error("%s: %s", __func__, msg);                               \
      return -2;                                                           \
    }                                                                      \
  } while (0)

static inline int rk_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// internal entry points shared between translation units (not part of the C ABI)
// (es: nullable, planes.h -- split jobs riding on the launch; rng_step is used when cursor == null)
struct rk_enc_split_t;
int rk_ae_encode_fwd_at(const rk_block_t *blk, int32_t row_off, int32_t B, const float *W_en,
                        const float *b_en, int32_t h, const uint8_t *keep, float p, uint64_t seed,
                        const int64_t *cursor, int32_t cursor_off, const int64_t *users, int32_t act,
                        float *Z0, void *zt_planes, void *stream, const rk_enc_split_t *es,
                        uint64_t rng_step);
// The replay context of the per-entry sequencing (rk_replay_set, capi.hip): thread-local, null outside
// a bracketed step
const rk_replay_t *rk_replay_get(void);
// gemm.hip: the split-K reduce (ws[split][M][N] -> out, * act'(Zact) if given) and the split-K factor
int rk_splitk_reduce(const float *ws, int M, int N, const int32_t *Kdev, int splits, const float *Zact,
                     int act, float *out, void *stream);
int rk_dz_splits(int B);
extern "C" int32_t rk_gemm_plain_bf16(void);
extern "C" int32_t rk_dw_pairs(void);
extern "C" int32_t rk_dw_encode_bwd_fused_ok(int32_t row_off, int32_t B);
int rk_adam_multi_at(const rk_adam_job_t *jobs, int32_t n_jobs, float *loss_part, int32_t n_part,
                     float denom, float *loss_out, const int64_t *cursor, int32_t cursor_off,
                     const void *table, int32_t tab_stride, const int32_t *tab_slots,
                     int64_t *cursor_next, int32_t advance, void *stream);

// small dense contraction C[M,N] = act(A . B + bias) (csrc/linear.hip: the hidden nn.Linear stack)
//   amode 0: A(m,k) = A[m * lda + k]     amode 1: A(m,k) = A[k * lda + m]
//   bmode 0: B(k,n) = B[n * ldb + k]     bmode 1: B(k,n) = B[k * ldb + n]
struct rk_small_gemm_t {
  const float *A; int lda, amode;
  const float *B; int ldb, bmode;
  int M, N, K;
  float *C; int ldc;
  const float *bias;          // per output column, nullable
  int act, accumulate;        // C = act(...) (+ C if accumulate)
  const float *dact_y;        // nullable: C = (...) * act'(dact_y[m * ldc + n]) (derivative `dact`)
  int dact;
};
bool rk_small_gemm_fits(int M, int N, int K);
int rk_small_gemm(const rk_small_gemm_t *g, void *stream);
int rk_small_gemm_pair(const rk_small_gemm_t *g1, const rk_small_gemm_t *g2, void *stream);
int rk_small_gemm_pair_colsum(const rk_small_gemm_t *g1, const rk_small_gemm_t *g2, const float *cs_X,
                              int cs_rows, int cs_cols, float *cs_out, void *stream);
// dY <- dY * act'(Y) in place, db[c] = column sums of the result ([rows, cols] row-major)
int rk_act_grad_colsum(float *dY, const float *Y, int rows, int cols, int act, float *db, void *stream);

// ---------------------------------------------------------------- activations
// y = act(x);  derivative expressed through y (what autograd of torch.tanh /
// sigmoid / relu / selu / elu evaluates to for the saved output).
__device__ __forceinline__ float rk_act(float x, int act) {
  switch (act) {
    case RK_ACT_TANH: return tanhf(x);
    case RK_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case RK_ACT_RELU: return x > 0.f ? x : 0.f;
    case RK_ACT_SELU: {
      const float a = 1.6732632423543772848170429916717f;
      const float s = 1.0507009873554804934193349852946f;
      return s * (x > 0.f ? x : a * (expf(x) - 1.0f));
    }
    case RK_ACT_ELU: return x > 0.f ? x : (expf(x) - 1.0f);
    default: return x;
  }
}

__device__ __forceinline__ float rk_act_dy(float y, int act) {
  switch (act) {
    case RK_ACT_TANH: return 1.0f - y * y;
    case RK_ACT_SIGMOID: return (1.0f - y) * y;
    case RK_ACT_RELU: return y > 0.f ? 1.0f : 0.f;
    case RK_ACT_SELU: {
      const float a = 1.6732632423543772848170429916717f;
      const float s = 1.0507009873554804934193349852946f;
      return y > 0.f ? s : (y + s * a);
    }
    case RK_ACT_ELU: return y > 0.f ? 1.0f : (y + 1.0f);
    default: return 1.0f;
  }
}

// ------------------------------------------------------------- counter RNG
// Stateless keep/drop draw for dropout: a 64-bit mix of (seed, step, a, b).
// Not the reference's stream (a GPU cannot reproduce torch's CPU Bernoulli
// stream); parity tests inject masks instead.
__device__ __forceinline__ uint64_t rk_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool rk_keep_draw(uint64_t seed, uint64_t step, uint64_t a,
                                             uint64_t b, float p) {
  uint64_t z = rk_mix64(seed + 0x9e3779b97f4a7c15ULL * (step + 1));
  z = rk_mix64(z ^ (a * 0xd1342543de82ef95ULL + b + 0x632be59bd9b4e019ULL));
  // 24 uniform bits -> [0,1)
  float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p;
}

// ------------------------------------------------------- device-resident step cursor
// HIP-graph replay bakes every kernel argument in; what changes from step to step (which users,
// which stamp, the RNG step, Adam's bias corrections, where the loss goes) is therefore derived
// IN the kernels from a cursor in device memory: cursor[0] = global index of the next step,
// cursor[1] = global index of the first step of the current epoch.  `off` = position of the
// launch inside the replayed group.  cursor == null: the host-provided values are used as given.
struct rk_cur_t {
  const int64_t *cursor;
  int32_t off;
};
__device__ __forceinline__ int64_t rk_cur_global(const rk_cur_t &c) { return c.cursor[0] + c.off; }
__device__ __forceinline__ int64_t rk_cur_local(const rk_cur_t &c) {
  return c.cursor[0] - c.cursor[1] + c.off;
}
// the collation stamp of global step s (non-zero, unique over 2^30 steps)
__device__ __forceinline__ int32_t rk_cur_stamp(const rk_cur_t &c) {
  return (int32_t)(rk_cur_global(c) & 0x3fffffff) + 1;
}

// --------------------------------------------------- fp32 -> three bf16 pieces
// (a, b) -> packed bf16 pairs hi / mid / lo with a = hi.x + mid.x + lo.x exactly (same for b):
// round to nearest at each level, the residuals are exact in fp32 (csrc/dw3.hip)
typedef __bf16 rk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rk_split_bf16_pair(float a, float b, uint32_t &h, uint32_t &m,
                                                   uint32_t &l) {
  const rk_f32x2 x = {a, b};
  const rk_bf16x2 hh = __builtin_convertvector(x, rk_bf16x2);
  const rk_f32x2 r1 = x - __builtin_convertvector(hh, rk_f32x2);
  const rk_bf16x2 mm = __builtin_convertvector(r1, rk_bf16x2);
  const rk_f32x2 r2 = r1 - __builtin_convertvector(mm, rk_f32x2);
  const rk_bf16x2 ll = __builtin_convertvector(r2, rk_bf16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  m = __builtin_bit_cast(uint32_t, mm);
  l = __builtin_bit_cast(uint32_t, ll);
}

// ------------------------------------------------------------ wave helpers
__device__ __forceinline__ float rk_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float rk_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}

// index j into the block's cols/vals/svals of the stored entry (row, c), given
// the bitmap word that holds its (set) bit: rows are column-sorted, so j is the
// row start plus the number of set bits before c (prefix popcount, no search)
__device__ __forceinline__ int rk_entry_index(const rk_block_t &b, int row, int c, uint32_t word) {
  return b.indptr[row] + b.pref_rc[(int64_t)row * b.ldw_rc + (c >> 5)] +
         __popc(word & ((1u << (c & 31)) - 1u));
}

// binary search of column c in the (ascending) relabelled columns of one row
__device__ __forceinline__ int rk_find_col(const int32_t *cols, int beg, int end, int c) {
  int lo = beg, hi = end;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cols[mid] < c) lo = mid + 1; else hi = mid;
  }
  return (lo < end && cols[lo] == c) ? lo : -1;
}
