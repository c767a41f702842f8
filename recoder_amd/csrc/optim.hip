// Optimiser + small elementwise kernels (all HBM-bound, 16-B vectorised).
//
// Adam     : torch.optim.Adam `_single_tensor_adam` as driven by the reference
//            (model.py:135,398-399): L2 weight decay folded into the gradient,
//            lerp first moment, bias-corrected step, eps added after the
//            bias-corrected sqrt.  A table job with pos is the "dense gradient"
//            (sparse=False) form for an embedding table: the reference
//            materialises a full [n_items,h] gradient that is zero outside the
//            sampled rows (K13/K14a of SURVEY 2.3); here the sweep reads the
//            compact gradient rows through pos[] instead.
// SparseAdam: torch.optim.SparseAdam (`_functional.sparse_adam`) on the touched
//            rows only (model.py:138,401-402): no weight decay, eps added to
//            the raw sqrt, bias correction folded into the step size.
#include <stdlib.h>

#include <algorithm>
#include <string.h>

#include "common.h"

namespace {

struct AdamC {
  float one_m_b1, b2, one_m_b2, eps, wd;
  float inv_bc2_sqrt; // 1 / sqrt(1 - beta2^t)
  float neg_step;     // -(lr / (1 - beta1^t))           (Adam)
  float neg_step_sp;  // -(lr * sqrt(1-beta2^t) / (1-beta1^t))   (SparseAdam)
};

// One Adam update of one element.  EVERY dense-Adam path of this file goes through this one function with every
// operation spelled out (explicit fma, no contraction left to the compiler): the lazy sweep replays missed steps
// with it and must reproduce the dense sweep bit for bit, whatever code surrounds the call.
// Round 6: sqrt(v) / bc2_sqrt + eps as fma(sqrt(v), 1 / bc2_sqrt, eps) with the hardware square root, and the final
// quotient as a multiplication by the hardware reciprocal (both within 1 ulp) -- the IEEE sqrt + two IEEE divisions
// of rounds 1-5 were ~45 of the update's ~60 VALU issue slots per element, which made the sweep (and above all the
// replay of missed steps) VALU-bound next to its HBM traffic: tools/probes/lazy_adam_probe.hip.
__device__ __forceinline__ void adam1(float &p, float &m, float &v, float g, const AdamC &c) {
#pragma clang fp contract(off)
  if (c.wd != 0.f) g = fmaf(c.wd, p, g);             // grad.add(param, alpha=wd)
  m = fmaf(c.one_m_b1, g - m, m);                    // exp_avg.lerp_(grad, 1-beta1)
  v = fmaf(c.one_m_b2 * g, g, v * c.b2);             // mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), c.inv_bc2_sqrt, c.eps);
  p = fmaf(c.neg_step * m, __builtin_amdgcn_rcpf(denom), p);   // addcdiv_(exp_avg, denom, -step_size)
}

__device__ __forceinline__ void sadam1(float &p, float &m, float &v, float g, const AdamC &c) {
  const float um = (g - m) * c.one_m_b1;
  const float uv = (g * g - v) * c.one_m_b2;
  const float numer = um + m;
  const float dsq = uv + v;
  m = m + um;
  v = v + uv;
  const float denom = sqrtf(dsq) + c.eps;
  p = p + (numer / denom) * c.neg_step_sp;
}

__global__ __launch_bounds__(256) void adam_rows_kernel(float *W, float *m, float *v, int h,
                                                        const int32_t *idx32, const int64_t *idx64,
                                                        const int32_t *n_dev, int n_host,
                                                        const float *G, AdamC c, rk_cur_t cur,
                                                        const AdamC *ctab, int tab_stride, int tab_slot) {
  if (cur.cursor) {            // replayed step: this step's users and constants
    if (idx64) idx64 += rk_cur_local(cur) * n_host;
    if (ctab) c = ctab[rk_cur_local(cur) * tab_stride + tab_slot];
  }
  const int n = n_dev ? *n_dev : n_host;
  const int64_t tot = (int64_t)n * h;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / h), q = (int)(i % h);
    const int64_t row = idx32 ? (int64_t)idx32[r] : idx64[r];
    const int64_t o = row * h + q;
    float p1 = W[o], m1 = m[o], v1 = v[o];
    sadam1(p1, m1, v1, G[i], c);
    W[o] = p1; m[o] = m1; v[o] = v1;
  }
}

// X[r, c] = act(X[r, c] + bias[c])   (item-parallel: after the all-reduce of the partial
// encoder sums)
__global__ __launch_bounds__(256) void bias_act_kernel(float *X, const float *bias, int64_t n,
                                                       int cols, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    X[i] = rk_act(X[i] + (bias ? bias[i % cols] : 0.f), act);
}

__global__ __launch_bounds__(256) void act_grad_kernel(float *dY, const float *Y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dY[i] = dY[i] * rk_act_dy(Y[i], act);
}

__global__ __launch_bounds__(256) void dropout_kernel(float *X, const uint8_t *keep, int64_t n,
                                                      int ncols, float p, float scale,
                                                      uint64_t seed, uint64_t step, rk_cur_t cur) {
  if (cur.cursor) step = (uint64_t)(rk_cur_global(cur) + 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool k = keep ? (keep[i] != 0)
                        : rk_keep_draw(seed, step, (uint64_t)(i / ncols) + 0x51ed270b1ULL,
                                       (uint64_t)(i % ncols), p);
    X[i] = k ? X[i] * scale : X[i] * 0.f;
  }
}

// out[c] = sum_r X[r*ld + c]; block = 64 columns x 16 row slices (1024 threads),
// 4 independent accumulators per thread, everything combined in a fixed order
__global__ __launch_bounds__(1024) void colsum_kernel(const float *X, int rows, int cols_host,
                                                      int ld_host, const int32_t *counts,
                                                      float *out) {
  __shared__ float part[16][64];
  const int cols = counts ? counts[0] : cols_host;
  const int ld = counts ? counts[2] : ld_host;
  const int lc = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lc;
  const int s = threadIdx.x >> 6;
  const int per = (rows + 15) >> 4;
  const int r0 = s * per, r1 = min(rows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    const float *x = X + c;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      a0 += x[(int64_t)r * ld];
      a1 += x[(int64_t)(r + 1) * ld];
      a2 += x[(int64_t)(r + 2) * ld];
      a3 += x[(int64_t)(r + 3) * ld];
    }
    for (; r < r1; ++r) a0 += x[(int64_t)r * ld];
  }
  part[s][lc] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (s == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lc];
    out[c] = t;
  }
}

// out[r, :] = act(E[rows[r], :]) with max |out| published for the split contractions (rk_amax's contract: the
// maximum over the 64 slots is what the kernels use): 64 workgroups, workgroup b files ITS maximum
// under slots[b] -- no atomics, no zeroing pass, one launch instead of two in front of an MF decode
// rows32 (nullable): rows32[0] <- B, rows32[1 + r] <- rows[r] -- the step's user rows as the int32
// index array (+ device-resident count) a SparseAdam job of rk_adam_multi takes, so that the user
// table's update rides on the step's one Adam launch (replayed steps: these ARE the cursor's users)
// V4 (d % 4 == 0, 16-byte aligned tables): float4 elements, FOUR (row index -> table row) chains in flight
// per thread -- element by element the 64 workgroups walked B*d / 16 k dependent round trips one after
// another (C4, 500 x 200: six of them, 8.4 us for 400 KB)
template <bool V4>
__global__ __launch_bounds__(256) void gather_rows_amax_kernel(const float *E, const int64_t *rows, int B,
                                                               int d, int act, float *out,
                                                               uint32_t *__restrict__ slots,
                                                               int32_t *__restrict__ rows32, rk_cur_t cur) {
  __shared__ float red[4];
  if (cur.cursor) rows += rk_cur_local(cur) * B;
  if (rows32 && blockIdx.x == 0) {
    if (threadIdx.x == 0) rows32[0] = B;
    for (int r = threadIdx.x; r < B; r += 256) rows32[1 + r] = (int32_t)rows[r];
  }
  float m = 0.f;
  if constexpr (V4) {
    const int d4 = d >> 2;
    const int64_t tot = (int64_t)B * d4, G = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < tot; i0 += 4 * G) {
      int64_t src[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * G;
        src[u] = i < tot ? rows[i / d4] * d + (i % d4) * 4 : -1;
      }
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = src[u] >= 0 ? *reinterpret_cast<const float4 *>(E + src[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * G;
        if (i >= tot) break;
        float4 w = v[u];
        w.x = rk_act(w.x, act); w.y = rk_act(w.y, act); w.z = rk_act(w.z, act); w.w = rk_act(w.w, act);
        *reinterpret_cast<float4 *>(out + i * 4) = w;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
      }
    }
  } else {
    const int64_t tot = (int64_t)B * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
      const int r = (int)(i / d), q = (int)(i % d);
      const float v = rk_act(E[rows[r] * d + q], act);
      out[i] = v;
      m = fmaxf(m, fabsf(v));
    }
  }
  m = rk_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0 && slots)
    slots[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

__global__ __launch_bounds__(256) void scatter_pos_kernel(int32_t *pos, const int64_t *rows, int B,
                                                          int clear, rk_cur_t cur) {
  if (cur.cursor) rows += rk_cur_local(cur) * B;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) pos[rows[i]] = clear ? -1 : i;
}

// ---------------------------------------------------------------------------
// rk_adam_multi: every Adam / SparseAdam update of one optimisation step (and the
// loss scalar) in ONE launch.  The step is a serial chain of kernels on one
// stream, so each separate small launch (bias tables, column sums, the loss
// reduction: 4-10 us apiece) was pure critical path.  Workgroup ranges select the
// job; within a job the arithmetic is exactly adam1 / sadam1 above.
// ---------------------------------------------------------------------------
struct UJob {
  float *p, *m, *v;
  const float *g;
  const int32_t *pos;          // dense table: row -> compact gradient row or -1
  const int32_t *rows;         // SparseAdam: compact row -> table row
  const int32_t *n_dev;        // SparseAdam: live compact rows
  const int32_t *gstride_dev;  // device-resident stride between gradient parts (or null)
  const int32_t *gparts_dev;   // device-resident number of gradient parts (or null: g_parts)
  int n_rows, h, g_parts, g_stride, sparse, blk0, nblk;
  int row0, row_step;          // the job covers rows row0, row0 + row_step, ... (owned rows)
  int tab_slot;                // >= 0: the constants come from the device table (UArgs.ctab)
  uint32_t *amax_out;          // nullable: 64 slots, running max |p| after the update (bit patterns)
  AdamC c;
};

// The BIG jobs of a launch -- dense Adam sweeps of [n_rows, h] embedding tables through pos, h % 4 == 0 -- as lean
// records of their own, at most two per launch, LAST in the grid: their fields sit at static offsets of the kernel
// argument (a few wide scalar loads), the element index is 32-bit and nothing of the general job's options
// (row strides, SparseAdam, scalar elements) is in their way.
struct UTab {
  float4 *p, *m, *v;
  const float4 *g;             // compact gradient rows (+ g_parts - 1 more arrays, gs4 float4s apart)
  const int32_t *pos;          // row -> compact gradient row or -1
  const int32_t *pos_next;     // lazy: the next step's map (null: every row is brought up to date)
  int32_t *stamp;              // lazy: row -> first step not yet applied; null: the plain sweep
  const int32_t *gparts_dev;
  uint32_t *amax_out;
  int64_t gs4;
  int n_rows, hq, g_parts, tab_slot, lazy_period, blk0, nblk;
  const int32_t *need_list, *need_count;   // lazy: the need-set outside the chunk as a list (nullable)
  AdamC c;
};

struct UArgs {
  int blk0[RK_ADAM_MULTI_MAX];   // first workgroup of job k (INT_MAX past n_jobs): the kernel's job lookup
  UJob job[RK_ADAM_MULTI_MAX];
  int n_jobs;
  UTab tab[2];
  int n_tab;
  float *loss_part;
  int n_part;
  float denom;
  float *loss_out;
  // graph replay (rk_cur_t, common.h): per-step constants table [step in epoch][tab_stride] and
  // the loss slot loss_out[step in epoch]
  rk_cur_t cur;
  const AdamC *ctab;
  int tab_stride;
  // last step of a replayed group: the cursor the NEXT group reads (a second buffer: this
  // group's other launches may still be reading `cur`) <- this one advanced by `advance`
  int64_t *cursor_next;
  int advance;
  // lazy jobs without a cursor (rk_adam_lazy_flush): the step the rows are brought up to (exclusive) and the
  // epoch's first step; flush_only: no step of its own is applied
  int64_t host_T, host_base;
  int flush_only;
};

template <typename T> struct VecOps;
template <> struct VecOps<float4> {
  static constexpr int W = 4;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void add(float4 &a, const float4 &b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  static __device__ __forceinline__ void adam(float4 &p, float4 &m, float4 &v, const float4 &g,
                                              const AdamC &c) {
    adam1(p.x, m.x, v.x, g.x, c); adam1(p.y, m.y, v.y, g.y, c);
    adam1(p.z, m.z, v.z, g.z, c); adam1(p.w, m.w, v.w, g.w, c);
  }
  static __device__ __forceinline__ void sadam(float4 &p, float4 &m, float4 &v, const float4 &g,
                                               const AdamC &c) {
    sadam1(p.x, m.x, v.x, g.x, c); sadam1(p.y, m.y, v.y, g.y, c);
    sadam1(p.z, m.z, v.z, g.z, c); sadam1(p.w, m.w, v.w, g.w, c);
  }
};
template <> struct VecOps<float> {
  static constexpr int W = 1;
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ void add(float &a, const float &b) { a += b; }
  static __device__ __forceinline__ void adam(float &p, float &m, float &v, const float &g,
                                              const AdamC &c) { adam1(p, m, v, g, c); }
  static __device__ __forceinline__ void sadam(float &p, float &m, float &v, const float &g,
                                               const AdamC &c) { sadam1(p, m, v, g, c); }
};

template <typename T> __device__ __forceinline__ float absmax_of(const T &v);
template <> __device__ __forceinline__ float absmax_of<float4>(const float4 &v) {
  return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}
template <> __device__ __forceinline__ float absmax_of<float>(const float &v) { return fabsf(v); }

// running maximum of |p| over everything this job has ever written: the decoder GEMMs take the
// fp16 split scale of their weight operand from it (gemm.hip b_amax).  The slot is read at the START
// of the workgroup's sweep (with its other loads: a dependent read at the end costs the short-lived
// workgroups 6-12 us per sweep, and so do thousands of unconditional atomics on 64 addresses) and
// the atomic is only issued by a wave that would raise it -- the bound moves rarely
__device__ __forceinline__ uint32_t *pmax_slot(uint32_t *slots) {
  return slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63);
}
__device__ __forceinline__ void publish_pmax(uint32_t *slots, uint32_t seen, float m) {
  m = rk_wave_max(m);
  const uint32_t bits = __float_as_uint(m);
  if ((threadIdx.x & 63) == 0 && bits > seen) atomicMax(pmax_slot(slots), bits);
}

template <typename T>
__device__ __forceinline__ void update_job(const UJob &J, int lb, const AdamC &C) {
  using V = VecOps<T>;
  float pmax = 0.f;
  const uint32_t seen = J.amax_out ? *pmax_slot(J.amax_out) : 0u;
  const int hq = J.h / V::W;
  const int64_t stride = J.gstride_dev ? (int64_t)*J.gstride_dev : (int64_t)J.g_stride;
  const int g_parts = J.gparts_dev ? min(*J.gparts_dev, J.g_parts) : J.g_parts;
  T *P = reinterpret_cast<T *>(J.p), *M = reinterpret_cast<T *>(J.m), *Vv = reinterpret_cast<T *>(J.v);
  const int64_t step = (int64_t)J.nblk * 256;
  if (J.sparse) {
    const int n = *J.n_dev;
    const int64_t tot = (int64_t)n * hq;
    const bool idx32 = tot < ((int64_t)1 << 31);
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < tot; i += step) {
      int r, q;
      if (idx32) { r = (int)((uint32_t)i / (uint32_t)hq); q = (int)((uint32_t)i - (uint32_t)r * (uint32_t)hq); }
      else { r = (int)(i / hq); q = (int)(i % hq); }
      const int64_t o = (int64_t)J.rows[r] * hq + q;
      T g = *reinterpret_cast<const T *>(J.g + i * V::W);
      for (int t = 1; t < g_parts; ++t)
        V::add(g, *reinterpret_cast<const T *>(J.g + t * stride + i * V::W));
      T p1 = P[o], m1 = M[o], v1 = Vv[o];
      V::sadam(p1, m1, v1, g, C);
      P[o] = p1; M[o] = m1; Vv[o] = v1;
      pmax = fmaxf(pmax, absmax_of<T>(p1));
    }
    if (J.amax_out) publish_pmax(J.amax_out, seen, pmax);
    return;
  }
  const int rows_live = J.row0 < J.n_rows ? (J.n_rows - J.row0 + J.row_step - 1) / J.row_step : 0;
  const int64_t tot = (int64_t)rows_live * hq;
  const bool by_row = J.pos != nullptr || J.row_step != 1 || J.row0 != 0;
  const bool idx32 = tot < ((int64_t)1 << 31);       // (a 64-bit division per element costs as much as the update itself)
  for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < tot; i += step) {
    int64_t e = i;                      // element (in units of T) of the parameter
    int64_t go = i * V::W;              // gradient offset (floats)
    bool have = true;
    if (by_row) {
      int r0, q;
      if (idx32) { r0 = (int)((uint32_t)i / (uint32_t)hq); q = (int)((uint32_t)i - (uint32_t)r0 * (uint32_t)hq); }
      else { r0 = (int)(i / hq); q = (int)(i % hq); }
      const int row = J.row0 + r0 * J.row_step;
      e = (int64_t)row * hq + q;
      go = e * V::W;
      if (J.pos) {
        const int pr = J.pos[row];
        have = pr >= 0;
        go = (int64_t)pr * J.h + q * V::W;
      }
    }
    T g = V::zero();
    if (have) {
      g = *reinterpret_cast<const T *>(J.g + go);
      int t = 1;
      constexpr int U = V::W == 4 ? 4 : 8;         // (float4: the K slabs of dW, at most 4; float: 8 row-tile partials)
      for (; t + U <= g_parts; t += U) {           // partial gradients, fixed order; the
        T v[U];                                      // loads of a group are independent
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const T *>(J.g + (t + u) * stride + go);
#pragma unroll
        for (int u = 0; u < U; ++u) V::add(g, v[u]);
      }
      for (; t < g_parts; ++t)
        V::add(g, *reinterpret_cast<const T *>(J.g + t * stride + go));
    }
    T p1 = P[e], m1 = M[e], v1 = Vv[e];
    V::adam(p1, m1, v1, g, C);
    P[e] = p1; M[e] = m1; Vv[e] = v1;
    pmax = fmaxf(pmax, absmax_of<T>(p1));
  }
  if (J.amax_out) publish_pmax(J.amax_out, seen, pmax);
}

// Lazy dense Adam of an embedding table (include/recoder_hip.h rk_adam_job_t.lazy_stamp).  One WAVE per table
// row (its lanes the row's float4s): the row is skipped as a whole -- three 4-byte loads -- or brought up to
// date by replaying the steps it missed, stamp[row] .. T - 1, with g = 0 and those steps' constants from the
// table, then step T with the gradient.  Every replayed step is the same adam1 call the dense sweep makes for a
// row without a gradient (g = 0 + wd * p), so p / m / v come out bit for bit.  The stamp is read by all lanes
// and written by lane 0 behind the row's last store: nobody else touches the row in this launch.
__device__ __forceinline__ float4 tab_gradient(const UTab &J, const uint32_t go, const int g_parts) {
  const float4 *gp = J.g + go;
  float4 g = *gp;
  int t = 1;
  for (; t + 2 <= g_parts; t += 2) {            // the K slabs of dW, summed in slab order
    const float4 v0 = gp[t * J.gs4], v1 = gp[(t + 1) * J.gs4];
    VecOps<float4>::add(g, v0);
    VecOps<float4>::add(g, v1);
  }
  for (; t < g_parts; ++t) VecOps<float4>::add(g, gp[t * J.gs4]);
  return g;
}

__device__ __forceinline__ void table_sweep_lazy(const UTab &J, const int lb, const UArgs &a) {
  using V = VecOps<float4>;
  const int lane = threadIdx.x & 63;
  const int n_waves = J.nblk * 4;
  const uint32_t hq = J.hq;
  const int64_t T = a.cur.cursor ? rk_cur_global(a.cur) : a.host_T;
  const int64_t base = a.cur.cursor ? a.cur.cursor[1] : a.host_base;
  const int end = (int)(a.flush_only ? T : T + 1);               // rows leave current up to here (exclusive)
  const int Tl = (int)(T - base);                                 // this step's row of the constants table
  const AdamC *tab = a.ctab + J.tab_slot;
  const int tstride = a.tab_stride;
  const int L = J.lazy_period;
  const int c = (int)(T % L);
  const int lo = (int)((int64_t)c * J.n_rows / L), hi = (int)((int64_t)(c + 1) * J.n_rows / L);
  const int g_parts = J.gparts_dev ? min(*J.gparts_dev, J.g_parts) : J.g_parts;
  float pmax = 0.f;
  const uint32_t seen = J.amax_out ? *pmax_slot(J.amax_out) : 0u;
  // (the row index is wave-uniform: through readfirstlane its pos / stamp loads and the replayed steps' constants
  // are scalar loads, the replay loop's trip count a scalar.)  The rows in rotated order, from the round-robin
  // chunk's first row: the long replays start with the launch
  // the constants of the LAST step a row is brought through (this step's; a flush: the one before): every row needs them
  const AdamC C_last = tab[(int64_t)(Tl - (a.flush_only ? 1 : 0)) * tstride];
  // with a need list (rk_lazy_need_lists: the rows with a gradient or read by the next step, whatever the step): work
  // items 0 .. hi - lo - 1 are the chunk's rows -- those of them the list does not hold -- the others the listed rows;
  // no wave for a row outside both
  const bool listed = J.need_list != nullptr;
  const int n_chunk = hi - lo;
  const int n_work = listed ? n_chunk + *J.need_count : J.n_rows;
  for (int w = __builtin_amdgcn_readfirstlane(lb * 4 + (int)(threadIdx.x >> 6)); w < n_work; w += n_waves) {
    int row;
    if (listed) row = w < n_chunk ? lo + w : J.need_list[w - n_chunk];
    else row = w + lo < J.n_rows ? w + lo : w + lo - J.n_rows;
    // the row's three words in ONE round trip (a short-circuited chain of them was three): its gradient row, the next
    // step's, its stamp; the row's vectors are then fetched before the stamp is looked at
    const int pr = (a.flush_only || J.pos == nullptr) ? -1 : J.pos[row];
    const int pn = J.pos_next ? J.pos_next[row] : 0;
    const int nx = J.stamp[row];
    const bool have = pr >= 0;
    const bool in_list = have || pn >= 0;
    const bool need = listed ? (w < n_chunk ? !in_list : true) : (in_list || (row >= lo && row < hi));
    if (!need) continue;
    const int lag = end - nx;                                   // steps to apply: [nx, end); <= 0: a flush of a current row
    for (uint32_t q = lane; q < hq; q += 64) {
      const uint32_t e = (uint32_t)row * hq + q;
      float4 p1 = J.p[e], m1 = J.m[e], v1 = J.v[e];
      float4 g = V::zero();
      if (have) g = tab_gradient(J, (uint32_t)pr * hq + q, g_parts);
      if (lag <= 0) continue;
      for (int k = lag - 1; k > 0; --k) {                      // the missed steps end - 1 - k, oldest first: g = 0
        const AdamC C = tab[(int64_t)(Tl - (a.flush_only ? 1 : 0) - k) * tstride];
        V::adam(p1, m1, v1, V::zero(), C);
        pmax = fmaxf(pmax, absmax_of<float4>(p1));
      }
      V::adam(p1, m1, v1, g, C_last);                          // step T with its gradient (a flush: the last missed step)
      pmax = fmaxf(pmax, absmax_of<float4>(p1));
      J.p[e] = p1; J.m[e] = m1; J.v[e] = v1;
    }
    if (lane == 0 && lag > 0) J.stamp[row] = end;
  }
  if (J.amax_out) publish_pmax(J.amax_out, seen, pmax);
}

// the plain sweep of a big table: one float4 per thread, every row
__device__ __forceinline__ void table_sweep_dense(const UTab &J, const int lb, const UArgs &a) {
  using V = VecOps<float4>;
  AdamC C = J.c;
  if (a.ctab && J.tab_slot >= 0) C = a.ctab[rk_cur_local(a.cur) * a.tab_stride + J.tab_slot];
  const uint32_t hq = J.hq, tot = (uint32_t)J.n_rows * hq, step = (uint32_t)J.nblk * 256u;
  const int g_parts = J.gparts_dev ? min(*J.gparts_dev, J.g_parts) : J.g_parts;
  float pmax = 0.f;
  const uint32_t seen = J.amax_out ? *pmax_slot(J.amax_out) : 0u;
  for (uint32_t i = (uint32_t)lb * 256u + threadIdx.x; i < tot; i += step) {
    const uint32_t row = i / hq, q = i - row * hq;
    const int pr = J.pos[row];
    float4 g = V::zero();
    if (pr >= 0) g = tab_gradient(J, (uint32_t)pr * hq + q, g_parts);
    float4 p1 = J.p[i], m1 = J.m[i], v1 = J.v[i];
    V::adam(p1, m1, v1, g, C);
    J.p[i] = p1; J.m[i] = m1; J.v[i] = v1;
    pmax = fmaxf(pmax, absmax_of<float4>(p1));
  }
  if (J.amax_out) publish_pmax(J.amax_out, seen, pmax);
}

template <bool LAZY>
__device__ __forceinline__ void table_sweep(const UTab &J, const int b, const UArgs &a) {
  if constexpr (LAZY) {
    if (J.stamp) { table_sweep_lazy(J, b - J.blk0, a); return; }
  }
  table_sweep_dense(J, b - J.blk0, a);
}

__device__ __forceinline__ void run_job(const UJob &J, int b, const UArgs &a) {
  AdamC C = J.c;
  if (a.ctab && J.tab_slot >= 0) C = a.ctab[rk_cur_local(a.cur) * a.tab_stride + J.tab_slot];
  if ((J.h & 3) == 0) update_job<float4>(J, b - J.blk0, C);
  else update_job<float>(J, b - J.blk0, C);
}

// LAZY: the instantiation of launches that hold a lazy job (the dense sweeps keep their registers)
template <bool LAZY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void adam_multi_kernel(UArgs a) {
  int b = blockIdx.x;
  if (a.loss_part) {
    if (b != 0) {
      --b;
    } else {
    // workgroup 0 (dispatched first: its dependent-load chain hides behind the sweeps):
    // loss = sum(partials) / denom in double, fixed order; loads first (they pipeline), then
    // the partials are re-zeroed for the next rk_decode_loss
    __shared__ double red[256];
    double s = 0.0;
    int i = threadIdx.x;
    for (; i + 7 * 256 < a.n_part; i += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.loss_part[i + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; i < a.n_part; i += 256) s += (double)a.loss_part[i];
    __syncthreads();
    for (i = threadIdx.x; i < a.n_part; i += 256) a.loss_part[i] = 0.f;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      a.loss_out[a.cur.cursor ? rk_cur_local(a.cur) : 0] = (float)red[0] / a.denom;
      if (a.cursor_next) {
        a.cursor_next[0] = a.cur.cursor[0] + a.advance;
        a.cursor_next[1] = a.cur.cursor[1];
      }
    }
    return;
    }
  }
  // the big table sweeps: the last workgroups of the grid
  if (a.n_tab > 0 && b >= a.tab[0].blk0) {
    if (a.n_tab > 1 && b >= a.tab[1].blk0) table_sweep<LAZY>(a.tab[1], b, a);
    else table_sweep<LAZY>(a.tab[0], b, a);
    return;
  }
  // The workgroup's job: the first blocks of the jobs sit side by side in the argument (ONE scalar load; slots past
  // n_jobs hold INT_MAX), the job record is then COPIED out under a static index (a dynamically indexed by-value
  // struct would go to scratch) and ONE copy of the update code runs on it.  Round 6: ten inlined copies of the
  // update code behind a chain of ten dependent scalar loads cost every (short-lived: one element per thread) wave
  // 28 scalar memory instructions and ~190 scalar ALU instructions before its first vector load -- 6.5 x the
  // probe kernel's (tools/probes/lazy_adam_probe.hip), and the sweep ran at 5.5 TB/s where that kernel reads 7.
  int j = 0;
#pragma unroll
  for (int k = 1; k < RK_ADAM_MULTI_MAX; ++k) j += b >= a.blk0[k] ? 1 : 0;
  UJob J;
  switch (j) {
    case 9: J = a.job[9]; break;
    case 8: J = a.job[8]; break;
    case 7: J = a.job[7]; break;
    case 6: J = a.job[6]; break;
    case 5: J = a.job[5]; break;
    case 4: J = a.job[4]; break;
    case 3: J = a.job[3]; break;
    case 2: J = a.job[2]; break;
    case 1: J = a.job[1]; break;
    default: J = a.job[0]; break;
  }
  run_job(J, b, a);
}

AdamC make_consts(double lr, double b1, double b2, double eps, double wd, int step) {
  AdamC c;
  c.one_m_b1 = (float)(1.0 - b1);
  c.b2 = (float)b2;
  c.one_m_b2 = (float)(1.0 - b2);
  c.eps = (float)eps;
  c.wd = (float)wd;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  c.neg_step = (float)(-(lr / bc1));
  c.neg_step_sp = (float)(-(lr * sqrt(bc2) / bc1));
  return c;
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int rk_adam_rows(float *W, float *m, float *v, int32_t h, const int32_t *idx32,
                            const int64_t *idx64, const int32_t *n_dev, int32_t n_cap,
                            const float *G, double lr, double beta1, double beta2, double eps,
                            int32_t step, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(step >= 1, "step must be >= 1");
  RK_REQUIRE((idx32 != nullptr) != (idx64 != nullptr), "exactly one index array");
  if (n_cap == 0) return 0;
  const AdamC c = make_consts(lr, beta1, beta2, eps, 0.0, step);
  rk_cur_t cur = {nullptr, 0};
  const AdamC *ctab = nullptr;
  int tab_stride = 0, tab_slot = 0;
  if (const rk_replay_t *rp = rk_replay_get()) {     // replayed step: `step` is the parameter's slot + 1
    cur = {rp->cursor, rp->off};
    ctab = (const AdamC *)rp->adam_table; tab_stride = rp->tab_stride; tab_slot = step - 1;
    if (idx64) { idx64 = rp->users_base; RK_REQUIRE(n_cap == rp->B, "replay: idx64 covers one batch"); }
  }
  RK_LAUNCH(adam_rows_kernel, dim3(grid_for((int64_t)n_cap * h)), dim3(256), 0, stream, W,
                     m, v, h, idx32, idx64, n_dev, n_cap, G, c, cur, ctab, tab_stride, tab_slot);
  RK_CHECK_LAUNCH("adam_rows");
  return 0;
}

// cursor / table: graph replay -- job j takes its constants from table[(step in epoch) *
// tab_stride + tab_slots[j]] (tab_slots[j] < 0: its own par) and the loss goes to
// loss_out[step in epoch]
int rk_adam_multi_at(const rk_adam_job_t *jobs, int32_t n_jobs, float *loss_part, int32_t n_part,
                     float denom, float *loss_out, const int64_t *cursor, int32_t cursor_off,
                     const void *table, int32_t tab_stride, const int32_t *tab_slots,
                     int64_t *cursor_next, int32_t advance, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(n_jobs >= 0 && n_jobs <= RK_ADAM_MULTI_MAX, "too many jobs for one launch");
  RK_REQUIRE(n_jobs == 0 || jobs != nullptr, "null jobs");
  RK_REQUIRE((loss_part == nullptr) == (loss_out == nullptr), "loss_part and loss_out go together");
  UArgs a = {};
  int blocks = 0;
  int big[2] = {-1, -1};
  for (int j = 0; j < n_jobs; ++j) {
    const rk_adam_job_t &s = jobs[j];
    RK_REQUIRE(s.par.step >= 1, "step must be >= 1");
    RK_REQUIRE(s.h >= 1 && s.n_rows >= 0, "bad job shape");
    RK_REQUIRE(!(s.par.sparse && (s.rows == nullptr || s.n_dev == nullptr)),
               "SparseAdam job needs rows and n_dev");
    RK_REQUIRE(s.g_parts >= 1, "g_parts must be >= 1");
    const bool vec = s.h % 4 == 0;
    RK_REQUIRE(!vec || ((((uintptr_t)s.par.p | (uintptr_t)s.par.m | (uintptr_t)s.par.v |
                          (uintptr_t)s.g) & 15) == 0 && (s.g_parts == 1 || s.gstride_dev == nullptr) &&
                        s.g_stride % 4 == 0),
               "vector jobs need 16-byte aligned operands / strides");
    const int row_step = s.row_step > 0 ? s.row_step : 1;
    RK_REQUIRE(s.row0 >= 0 && (s.par.sparse == 0 || (s.row0 == 0 && row_step == 1)),
               "row0 / row_step apply to dense jobs");
    const int64_t rows = s.par.sparse ? s.n_cap
                                      : (s.row0 < s.n_rows ? (s.n_rows - s.row0 + row_step - 1) / row_step : 0);
    if (rows == 0) continue;
    const int slot = (table && tab_slots) ? tab_slots[j] : -1;
    // a BIG table sweep (UTab): dense Adam through pos over every row of a float4 table whose element and
    // gradient indices fit 32 bits -- at most two per launch, behind the other jobs
    const bool table_job = !s.par.sparse && s.pos != nullptr && vec && row_step == 1 && s.row0 == 0 &&
                           s.gstride_dev == nullptr && ((int64_t)s.n_rows * s.h >= 65536 || s.lazy_stamp != nullptr) &&
                           (int64_t)s.n_rows * s.h < ((int64_t)1 << 32) && a.n_tab < 2;
    if (s.lazy_stamp) {
      RK_REQUIRE(table_job && s.lazy_period >= 1,
                 "lazy jobs: dense Adam through pos on a [n_rows, h % 4 == 0] table, all rows, lazy_period >= 1, at most "
                 "two per launch");
      RK_REQUIRE(cursor != nullptr && table != nullptr && slot >= 0,
                 "lazy jobs take their constants from the replay table (cursor + table + slot)");
    }
    if (table_job) {
      UTab &t = a.tab[a.n_tab];
      t.p = reinterpret_cast<float4 *>(s.par.p); t.m = reinterpret_cast<float4 *>(s.par.m);
      t.v = reinterpret_cast<float4 *>(s.par.v); t.g = reinterpret_cast<const float4 *>(s.g);
      t.pos = s.pos; t.pos_next = s.lazy_pos_next; t.stamp = s.lazy_stamp; t.lazy_period = s.lazy_stamp ? s.lazy_period : 0;
      t.gparts_dev = s.gparts_dev; t.amax_out = reinterpret_cast<uint32_t *>(s.amax_out);
      t.gs4 = s.g_stride / 4; t.n_rows = s.n_rows; t.hq = s.h / 4; t.g_parts = s.g_parts; t.tab_slot = slot;
      t.c = make_consts(s.par.lr, s.par.beta1, s.par.beta2, s.par.eps, s.par.weight_decay, s.par.step);
      t.nblk = grid_for((int64_t)s.n_rows * s.h / 4);
      t.need_list = s.lazy_stamp ? s.lazy_need_list : nullptr;
      t.need_count = s.lazy_stamp ? s.lazy_need_count : nullptr;
      RK_REQUIRE((t.need_list == nullptr) == (t.need_count == nullptr) && (t.need_list == nullptr || s.lazy_pos_next != nullptr),
                 "lazy_need_list and lazy_need_count go together, with lazy_pos_next");
      big[a.n_tab++] = j;
      continue;
    }
    UJob &d = a.job[a.n_jobs];
    d.p = s.par.p; d.m = s.par.m; d.v = s.par.v; d.g = s.g;
    d.pos = s.par.sparse ? nullptr : s.pos;
    d.rows = s.rows; d.n_dev = s.n_dev; d.gstride_dev = s.gstride_dev; d.gparts_dev = s.gparts_dev;
    d.n_rows = s.n_rows; d.h = s.h; d.g_parts = s.g_parts; d.g_stride = s.g_stride;
    d.sparse = s.par.sparse ? 1 : 0;
    d.row0 = s.row0; d.row_step = row_step;
    d.tab_slot = slot;
    d.amax_out = reinterpret_cast<uint32_t *>(s.amax_out);
    d.c = make_consts(s.par.lr, s.par.beta1, s.par.beta2, s.par.eps,
                      s.par.sparse ? 0.0 : s.par.weight_decay, s.par.step);
    d.blk0 = blocks;
    d.nblk = grid_for(rows * s.h / (vec ? 4 : 1));
    blocks += d.nblk;
    ++a.n_jobs;
  }
  for (int k = 0; k < a.n_tab; ++k) { a.tab[k].blk0 = blocks; blocks += a.tab[k].nblk; }
  if (loss_part) {
    a.loss_part = loss_part; a.n_part = n_part; a.denom = denom; a.loss_out = loss_out;
    blocks += 1;
  }
  if (blocks == 0) return 0;
  if (a.n_jobs == 0) { a.n_jobs = 1; a.job[0].nblk = 0; a.job[0].h = 1; a.job[0].blk0 = 0; }   // no general job
  for (int k = 0; k < RK_ADAM_MULTI_MAX; ++k) a.blk0[k] = k < a.n_jobs ? a.job[k].blk0 : 0x7fffffff;
  a.cur.cursor = cursor; a.cur.off = cursor_off;
  a.ctab = (const AdamC *)table; a.tab_stride = tab_stride;
  RK_REQUIRE(cursor_next == nullptr || (cursor != nullptr && loss_part != nullptr && cursor_next != cursor),
             "cursor_next needs a cursor, the loss block and a buffer of its own");
  a.cursor_next = cursor_next; a.advance = advance;
  bool lazy = false;
  for (int k = 0; k < a.n_tab; ++k) lazy = lazy || a.tab[k].stamp != nullptr;
  if (lazy) RK_LAUNCH(adam_multi_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
  else RK_LAUNCH(adam_multi_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
  RK_CHECK_LAUNCH("adam_multi");
  return 0;
}

extern "C" int rk_adam_multi(const rk_adam_job_t *jobs, int32_t n_jobs, float *loss_part,
                             int32_t n_part, float denom, float *loss_out, void *stream_) {
  if (const rk_replay_t *rp = rk_replay_get()) {
    // replayed per-entry step: par.step of every job is its parameter's slot + 1 in the constants table
    int32_t slots[RK_ADAM_MULTI_MAX];
    rk_adam_job_t tmp[RK_ADAM_MULTI_MAX];
    RK_REQUIRE(n_jobs >= 0 && n_jobs <= RK_ADAM_MULTI_MAX, "too many jobs for one launch");
    for (int j = 0; j < n_jobs; ++j) { tmp[j] = jobs[j]; slots[j] = jobs[j].par.step - 1; tmp[j].par.step = 1; }
    return rk_adam_multi_at(tmp, n_jobs, loss_part, n_part, denom, loss_out, rp->cursor, rp->off,
                            rp->adam_table, rp->tab_stride, slots,
                            loss_part ? rp->cursor_next : nullptr, rp->advance, stream_);
  }
  return rk_adam_multi_at(jobs, n_jobs, loss_part, n_part, denom, loss_out, nullptr, 0, nullptr, 0,
                          nullptr, nullptr, 0, stream_);
}

extern "C" int rk_adam_lazy_flush(const rk_adam_job_t *jobs, int32_t n_jobs, const void *table, int32_t tab_stride,
                                  const int32_t *tab_slots, int64_t next_step, int64_t epoch_base, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(n_jobs >= 0 && n_jobs <= RK_ADAM_MULTI_MAX, "too many jobs for one launch");
  RK_REQUIRE(n_jobs == 0 || (jobs != nullptr && table != nullptr && tab_slots != nullptr), "null jobs / table / slots");
  RK_REQUIRE(next_step >= epoch_base, "next_step >= epoch_base");
  for (int j0 = 0; j0 < n_jobs; j0 += 2) {              // two tables per launch
    UArgs a = {};
    int blocks = 0;
    for (int j = j0; j < n_jobs && j < j0 + 2; ++j) {
      const rk_adam_job_t &s = jobs[j];
      RK_REQUIRE(s.lazy_stamp != nullptr && s.h >= 4 && s.h % 4 == 0 && s.n_rows >= 0 && tab_slots[j] >= 0 &&
                 (int64_t)s.n_rows * s.h < ((int64_t)1 << 32),
                 "flush jobs: lazy_stamp, h % 4 == 0, a table slot");
      RK_REQUIRE((((uintptr_t)s.par.p | (uintptr_t)s.par.m | (uintptr_t)s.par.v) & 15) == 0, "16-byte aligned tables");
      if (s.n_rows == 0) continue;
      UTab &t = a.tab[a.n_tab++];
      t.p = reinterpret_cast<float4 *>(s.par.p); t.m = reinterpret_cast<float4 *>(s.par.m);
      t.v = reinterpret_cast<float4 *>(s.par.v);
      t.n_rows = s.n_rows; t.hq = s.h / 4; t.g_parts = 1;
      t.tab_slot = tab_slots[j];
      t.amax_out = reinterpret_cast<uint32_t *>(s.amax_out);
      t.stamp = s.lazy_stamp; t.lazy_period = 1;
      t.blk0 = blocks;
      t.nblk = grid_for((int64_t)s.n_rows * s.h / 4);
      blocks += t.nblk;
    }
    if (blocks == 0) continue;
    a.n_jobs = 1; a.job[0].nblk = 0; a.job[0].h = 1;
    for (int k = 0; k < RK_ADAM_MULTI_MAX; ++k) a.blk0[k] = k < a.n_jobs ? 0 : 0x7fffffff;
    a.ctab = (const AdamC *)table; a.tab_stride = tab_stride;
    a.host_T = next_step; a.host_base = epoch_base; a.flush_only = 1;
    RK_LAUNCH(adam_multi_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    RK_CHECK_LAUNCH("adam_lazy_flush");
  }
  return 0;
}

// one entry of the per-step constants table (8 floats): exactly what rk_adam_multi derives from
// the same hyper-parameters for step `step`
extern "C" int rk_adam_consts(double lr, double beta1, double beta2, double eps, double weight_decay,
                              int32_t step, int32_t n_steps, int32_t stride_floats, float *out) {
  RK_REQUIRE(step >= 1 && n_steps >= 0 && out != nullptr && (n_steps <= 1 || stride_floats >= 8),
             "step >= 1, out != NULL, stride >= 8 floats");
  static_assert(sizeof(AdamC) == 32, "AdamC is 8 floats");
  for (int i = 0; i < n_steps; ++i) {
    const AdamC c = make_consts(lr, beta1, beta2, eps, weight_decay, step + i);
    memcpy(out + (int64_t)i * stride_floats, &c, sizeof(c));
  }
  return 0;
}

extern "C" int rk_scatter_pos(int32_t *pos, const int64_t *rows, int32_t B, int32_t clear,
                              void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return 0;
  rk_cur_t cur = {nullptr, 0};
  if (const rk_replay_t *rp = rk_replay_get()) { cur = {rp->cursor, rp->off}; rows = rp->users_base; }
  RK_LAUNCH(scatter_pos_kernel, dim3(rk_cdiv(B, 256)), dim3(256), 0, stream, pos, rows, B,
                     clear, cur);
  RK_CHECK_LAUNCH("scatter_pos");
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void rows_to_dense_kernel(const float4 *__restrict__ G, const int32_t *__restrict__ pos,
                                                            int n_items, int rows_pad, int hq, float4 *__restrict__ D) {
  const int64_t tot = (int64_t)rows_pad * hq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / hq), q = (int)(i % hq);
    const int pr = row < n_items ? pos[row] : -1;
    D[i] = pr >= 0 ? G[(int64_t)pr * hq + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__global__ __launch_bounds__(256) void rows_to_dense1_kernel(const float *__restrict__ G, const int32_t *__restrict__ pos,
                                                             int n_items, int rows_pad, float *__restrict__ D) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows_pad; i += gridDim.x * 256) {
    const int pr = i < n_items ? pos[i] : -1;
    D[i] = pr >= 0 ? G[pr] : 0.f;
  }
}
}  // namespace

extern "C" int rk_rows_to_dense(const float *G, const int32_t *pos, int32_t n_items, int32_t rows_pad, int32_t h,
                                float *D, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(rows_pad >= n_items && n_items >= 0, "rows_pad >= n_items");
  if (rows_pad == 0) return 0;
  if (h == 1) {
    RK_LAUNCH(rows_to_dense1_kernel, dim3(grid_for(rows_pad)), dim3(256), 0, stream, G, pos, n_items, rows_pad, D);
    RK_CHECK_LAUNCH("rows_to_dense");
    return 0;
  }
  RK_REQUIRE(h > 0 && h % 4 == 0, "h % 4 == 0 (or 1)");
  RK_REQUIRE((((uintptr_t)G | (uintptr_t)D) & 15) == 0, "operands must be 16-byte aligned");
  RK_LAUNCH(rows_to_dense_kernel, dim3(grid_for((int64_t)rows_pad * (h / 4))), dim3(256), 0, stream,
            reinterpret_cast<const float4 *>(G), pos, n_items, rows_pad, h / 4, reinterpret_cast<float4 *>(D));
  RK_CHECK_LAUNCH("rows_to_dense");
  return 0;
}

namespace {
struct ZeroTails { float *x[4]; int h[4]; int n; };
__global__ __launch_bounds__(256) void zero_tail_rows_kernel(ZeroTails z, const int32_t *counts, int n_cap, int32_t *hwm) {
  const int n_b = min(max(counts[0], 0), n_cap);
  // (hwm: rows at or past it are known to be zero; whichever value a workgroup reads -- the old one or the one
  // workgroup 0 is about to write -- max(hwm, n_b) is the same bound)
  const int top = hwm ? min(max(*hwm, n_b), n_cap) : n_cap;
  for (int k = 0; k < z.n; ++k) {
    float *x = k == 0 ? z.x[0] : k == 1 ? z.x[1] : k == 2 ? z.x[2] : z.x[3];
    const int h = k == 0 ? z.h[0] : k == 1 ? z.h[1] : k == 2 ? z.h[2] : z.h[3];
    const int64_t lo = (int64_t)n_b * h, hi = (int64_t)top * h;
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) x[i] = 0.f;
  }
  if (hwm && blockIdx.x == 0 && threadIdx.x == 0) *hwm = top;
}
}  // namespace

// X_k[n_b * h_k .. top * h_k) <- 0 for up to four [n_cap, h_k] arrays, n_b = counts[0] read on the device: the
// rows of the compact gradient arrays past the block's live items.  A REPLAYED data-parallel step exchanges the
// blocks' whole capacity (a captured collective has a fixed size) and sums in place: rows nobody rewrites would be
// multiplied by the world size every step.  high_water (nullable, device int32, 0 for freshly zeroed
// arrays): the rows at or past it are known to be zero -- top = max(*high_water, n_b), which the call stores back;
// a step then clears the few rows a LARGER earlier item set left behind, not the whole tail (NULL: top = n_cap).
extern "C" int rk_zero_tail_rows(float *const *X, const int32_t *h, int32_t n_arrays, const int32_t *counts,
                                 int32_t n_cap, int32_t *high_water, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(n_arrays >= 0 && n_arrays <= 4 && (n_arrays == 0 || (X && h && counts)), "at most four arrays");
  if (n_arrays == 0 || n_cap == 0) return 0;
  ZeroTails z = {};
  int hmax = 1;
  for (int k = 0; k < n_arrays; ++k) {
    RK_REQUIRE(X[k] != nullptr && h[k] >= 1, "null array / bad width");
    z.x[k] = X[k]; z.h[k] = h[k]; hmax = h[k] > hmax ? h[k] : hmax;
  }
  z.n = n_arrays;
  // (with a high-water mark the range is a few hundred rows: a small grid)
  const int grid = high_water ? std::min(grid_for((int64_t)n_cap * hmax / 8), 128) : grid_for((int64_t)n_cap * hmax / 8);
  RK_LAUNCH(zero_tail_rows_kernel, dim3(grid), dim3(256), 0, stream, z, counts, n_cap, high_water);
  RK_CHECK_LAUNCH("zero_tail_rows");
  return 0;
}

extern "C" int rk_bias_act(float *X, const float *bias, int32_t rows, int32_t cols, int32_t act,
                           void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = (int64_t)rows * cols;
  if (n == 0) return 0;
  RK_LAUNCH(bias_act_kernel, dim3(grid_for(n)), dim3(256), 0, stream, X, bias, n, cols, act);
  RK_CHECK_LAUNCH("bias_act");
  return 0;
}

extern "C" int rk_act_grad(float *dY, const float *Y, int64_t n, int32_t act, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0 || act == RK_ACT_NONE) return 0;
  RK_LAUNCH(act_grad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dY, Y, n, act);
  RK_CHECK_LAUNCH("act_grad");
  return 0;
}

extern "C" int rk_dropout(float *X, const uint8_t *keep, int64_t n, int32_t ncols, float p,
                          uint64_t seed, uint64_t rng_step, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(p >= 0.f && p < 1.f, "dropout prob must be in [0,1)");
  if (n == 0 || p == 0.f) return 0;
  const float scale = 1.0f / (float)(1.0 - (double)p);
  rk_cur_t cur = {nullptr, 0};
  if (const rk_replay_t *rp = rk_replay_get()) cur = {rp->cursor, rp->off};
  RK_LAUNCH(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, stream, X, keep, n, ncols, p,
                     scale, seed, rng_step, cur);
  RK_CHECK_LAUNCH("dropout");
  return 0;
}

extern "C" int rk_colsum(const float *X, int32_t rows, int32_t cols, int32_t ld,
                         const int32_t *counts_dev, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cols == 0) return 0;
  RK_LAUNCH(colsum_kernel, dim3(rk_cdiv(cols, 64)), dim3(1024), 0, stream, X, rows, cols, ld,
                     counts_dev, out);
  RK_CHECK_LAUNCH("colsum");
  return 0;
}

extern "C" int rk_gather_rows_amax(const float *E, const int64_t *rows, int32_t B, int32_t d,
                                   int32_t act, float *out, int32_t *slots, int32_t *rows32,
                                   void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(B > 0, "B > 0");
  rk_cur_t cur = {nullptr, 0};
  if (const rk_replay_t *rp = rk_replay_get()) { cur = {rp->cursor, rp->off}; rows = rp->users_base; }
  if (d % 4 == 0 && (((uintptr_t)E | (uintptr_t)out) & 15) == 0)
    RK_LAUNCH(gather_rows_amax_kernel<true>, dim3(64), dim3(256), 0, stream, E, rows, B, d, act, out,
              reinterpret_cast<uint32_t *>(slots), rows32, cur);
  else
    RK_LAUNCH(gather_rows_amax_kernel<false>, dim3(64), dim3(256), 0, stream, E, rows, B, d, act, out,
              reinterpret_cast<uint32_t *>(slots), rows32, cur);
  RK_CHECK_LAUNCH("gather_rows_amax");
  return 0;
}
