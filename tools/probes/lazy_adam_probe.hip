// What can a dense-Adam sweep that SKIPS rows get out of the chip?  (round 6: lazy dense Adam, csrc/optim.hip)
// Both C2 tables (2 x 20108 rows x 200 floats; p, m, v) with the row set a lazy step touches: the union of two
// consecutive Zipf item sets + a 1/16 round-robin chunk (~59 % of the rows).  Variants of the row -> thread mapping
// and of the memory layout, all with the real adam1 arithmetic:
//   A  dense flat sweep, every row                         (the kernel of rounds 1-5)
//   B  flat mapping, rows without work skipped by a mask
//   C  one wave per row, scalar mask / stamp loads          (round 6's first form)
//   D  compact list of the rows with work, flat mapping over the list
//   E  D + the replay of missed steps (lags as a lazy run has them), constants of a step from a table
//   F  D on an interleaved layout [row][p | m | v]         (one 2400-byte piece per row instead of three 800-byte ones)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lazy_probe tools/probes/lazy_adam_probe.hip && /tmp/lazy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct AdamC { float one_m_b1, b2, one_m_b2, eps, wd, bc2_sqrt, neg_step, pad; };

// ARITH 0: IEEE sqrt and two IEEE divisions (rounds 1-5); 1: sqrt(v) * (1 / bc2_sqrt) with the hardware sqrt, ONE IEEE
// division; 2: hardware sqrt and reciprocal (1 ulp each), no division
#ifndef ARITH
#define ARITH 0
#endif
__device__ __forceinline__ void adam1(float &p, float &m, float &v, float g, const AdamC &c) {
  g = fmaf(c.wd, p, g);
  m = fmaf(c.one_m_b1, g - m, m);
  v = fmaf(c.one_m_b2 * g, g, v * c.b2);
#if ARITH == 0
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p + (c.neg_step * m) / denom;
#elif ARITH == 1
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), c.bc2_sqrt, c.eps);      // (bc2_sqrt holds the reciprocal here)
  p = p + (c.neg_step * m) / denom;
#else
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), c.bc2_sqrt, c.eps);
  p = fmaf(c.neg_step * m, __builtin_amdgcn_rcpf(denom), p);
#endif
}
__device__ __forceinline__ void adam4(float4 &p, float4 &m, float4 &v, const float4 &g, const AdamC &c) {
  adam1(p.x, m.x, v.x, g.x, c); adam1(p.y, m.y, v.y, g.y, c); adam1(p.z, m.z, v.z, g.z, c); adam1(p.w, m.w, v.w, g.w, c);
}

struct Args {
  float4 *p, *m, *v;          // [rows][hq]  (F: p = base of the interleaved array)
  const float4 *g;            // compact gradient rows
  const int *pos;             // row -> gradient row or -1
  const int *need;            // row -> 0 / 1
  const int *list;            // compact list of the rows with work
  const int *lag;             // row -> steps to replay (>= 1)
  const AdamC *tab;           // constants of the last 17 steps
  int rows, hq, n_list;
  AdamC c;
};

__global__ __launch_bounds__(256) void kA(Args a) {
  const size_t tot = (size_t)a.rows * a.hq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const unsigned row = (unsigned)i / (unsigned)a.hq, q = (unsigned)i - row * (unsigned)a.hq;
    const int pr = a.pos[row];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr >= 0) g = a.g[(size_t)pr * a.hq + q];
    float4 p = a.p[i], m = a.m[i], v = a.v[i];
    adam4(p, m, v, g, a.c);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
__global__ __launch_bounds__(256) void kB(Args a) {
  const size_t tot = (size_t)a.rows * a.hq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const unsigned row = (unsigned)i / (unsigned)a.hq, q = (unsigned)i - row * (unsigned)a.hq;
    if (!a.need[row]) continue;
    const int pr = a.pos[row];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr >= 0) g = a.g[(size_t)pr * a.hq + q];
    float4 p = a.p[i], m = a.m[i], v = a.v[i];
    adam4(p, m, v, g, a.c);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
__global__ __launch_bounds__(256) void kC(Args a) {
  const int lane = threadIdx.x & 63;
  const int n_waves = gridDim.x * 4;
  for (int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (int)(threadIdx.x >> 6)); row < a.rows; row += n_waves) {
    if (!a.need[row]) continue;
    const int pr = a.pos[row];
    for (int q = lane; q < a.hq; q += 64) {
      const size_t i = (size_t)row * a.hq + q;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pr >= 0) g = a.g[(size_t)pr * a.hq + q];
      float4 p = a.p[i], m = a.m[i], v = a.v[i];
      adam4(p, m, v, g, a.c);
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
  }
}
template <bool REPLAY, bool INTER>
__global__ __launch_bounds__(256) void kD(Args a) {
  __shared__ AdamC tab[17];
  if (REPLAY) {
    if (threadIdx.x < 17) tab[threadIdx.x] = a.tab[threadIdx.x];
    __syncthreads();
  }
  const size_t tot = (size_t)a.n_list * a.hq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const unsigned k = (unsigned)i / (unsigned)a.hq, q = (unsigned)i - k * (unsigned)a.hq;
    const int row = a.list[k];
    const int pr = a.pos[row];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr >= 0) g = a.g[(size_t)pr * a.hq + q];
    float4 p, m, v;
    size_t e;
    if (INTER) {
      e = (size_t)row * 3 * a.hq + q;
      p = a.p[e]; m = a.p[e + a.hq]; v = a.p[e + 2 * a.hq];
    } else {
      e = (size_t)row * a.hq + q;
      p = a.p[e]; m = a.m[e]; v = a.v[e];
    }
    if (REPLAY) {
      const int lag = a.lag[row];
      for (int s = 17 - lag; s < 17; ++s) {
        const float4 gs = s == 16 ? g : make_float4(0.f, 0.f, 0.f, 0.f);
        adam4(p, m, v, gs, tab[s]);
      }
    } else {
      adam4(p, m, v, g, a.c);
    }
    if (INTER) { a.p[e] = p; a.p[e + a.hq] = m; a.p[e + 2 * a.hq] = v; }
    else { a.p[e] = p; a.m[e] = m; a.v[e] = v; }
  }
}

// H / I: one wave per row WITH the stamp logic of csrc/optim.hip (stamp read by the wave, written by lane 0) and the
// replay; constants of the replayed steps from LDS (H) or by scalar loads from the global table (I).  FIRST: the rows
// of the round-robin chunk (the long replays) are handled by the first waves of the grid.
template <bool LDS_TAB, bool CHUNK_FIRST>
__global__ __launch_bounds__(256) void kH(Args a, int *stamp, int T, int lo, int hi) {
  __shared__ AdamC tabs[17];
  if (LDS_TAB) {
    if (threadIdx.x < 17) tabs[threadIdx.x] = a.tab[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int n_waves = gridDim.x * 4;
  const int N = a.rows / 2;
  for (int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (int)(threadIdx.x >> 6)); w < a.rows; w += n_waves) {
    int row = w;
    if (CHUNK_FIRST) {
      // waves [0, 2 (hi - lo)) take the chunk rows of both tables, the others the rest in order
      const int nc = hi - lo;
      if (w < 2 * nc) row = (w < nc ? lo + w : N + lo + (w - nc));
      else { int r = w - 2 * nc; const int t = r >= N - nc; if (t) r -= N - nc; row = t * N + (r < lo ? r : r + nc); }
    }
    if (!a.need[row]) continue;
    const int pr = a.pos[row];
    const int nx = stamp[row];
    for (int q = lane; q < a.hq; q += 64) {
      const size_t i = (size_t)row * a.hq + q;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pr >= 0) g = a.g[(size_t)pr * a.hq + q];
      float4 p = a.p[i], m = a.m[i], v = a.v[i];
      for (int s = nx; s <= T; ++s) {
        const float4 gs = s == T ? g : make_float4(0.f, 0.f, 0.f, 0.f);
        if (LDS_TAB) adam4(p, m, v, gs, tabs[16 - (T - s)]);
        else { const AdamC c = a.tab[16 - (T - s)]; adam4(p, m, v, gs, c); }
      }
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
    if (lane == 0) stamp[row] = nx;          // (the probe leaves the lags as they are: every launch does the same work)
  }
}

template <typename F> static float time_us(F f, int reps = 30) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main(int argc, char **argv) {
  const int N = 20108, h = 200, hq = h / 4, rows = 2 * N, L = 16;
  const double zipf_a = argc > 1 ? atof(argv[1]) : 1.0;
  std::mt19937_64 rng(1);
  // Zipf item sampler (inverse CDF) -- the synthetic C2 matrix: 500 users x ~73 items per step
  std::vector<double> cum(N);
  double tot = 0;
  for (int i = 0; i < N; ++i) { tot += 1.0 / pow(i + 1.0, zipf_a); cum[i] = tot; }
  auto draw_set = [&](std::vector<char> &in) {
    std::uniform_real_distribution<double> U(0.0, tot);
    in.assign(N, 0);
    for (int k = 0; k < 500 * 73; ++k) in[std::lower_bound(cum.begin(), cum.end(), U(rng)) - cum.begin()] = 1;
  };
  // a short lazy run on the host: stamps -> the lags of step 40
  std::vector<int> stamp(N, 0);
  std::vector<char> cur, nxt;
  draw_set(cur);
  std::vector<int> lag1(N, 0), need1(N, 0), pos1(N, -1);
  int T = 0;
  for (; T < 40; ++T) {
    draw_set(nxt);
    const int c = T % L, lo = (int)((long)c * N / L), hi = (int)((long)(c + 1) * N / L);
    for (int r = 0; r < N; ++r) {
      const bool need = cur[r] || nxt[r] || (r >= lo && r < hi);
      if (T == 39) { need1[r] = need; lag1[r] = need ? T + 1 - stamp[r] : 0; }
      if (need) stamp[r] = T + 1;
    }
    if (T == 39) { int k = 0; for (int r = 0; r < N; ++r) if (cur[r]) pos1[r] = k++; }
    cur.swap(nxt);
  }
  std::vector<int> need(rows), lag(rows), pos(rows), list;
  int n_b = 0;
  for (int r = 0; r < N; ++r) n_b += pos1[r] >= 0;
  long lag_sum = 0; int lag_max = 0;
  for (int t = 0; t < 2; ++t)
    for (int r = 0; r < N; ++r) {
      need[t * N + r] = need1[r]; lag[t * N + r] = lag1[r];
      pos[t * N + r] = pos1[r] >= 0 ? pos1[r] + t * n_b : -1;
      if (need1[r]) { list.push_back(t * N + r); lag_sum += lag1[r]; lag_max = std::max(lag_max, lag1[r]); }
    }
  printf("rows %d, with work %zu (%.1f %%), gradient rows %d, mean lag %.2f max %d\n", rows, list.size(),
         100.0 * list.size() / rows, 2 * n_b, (double)lag_sum / list.size(), lag_max);
  Args a = {};
  const size_t n4 = (size_t)rows * hq;
  float4 *inter;
  hipMalloc(&a.p, n4 * 16); hipMalloc(&a.m, n4 * 16); hipMalloc(&a.v, n4 * 16); hipMalloc(&inter, 3 * n4 * 16);
  hipMalloc((void **)&a.g, (size_t)2 * n_b * hq * 16);
  hipMemset(a.p, 0, n4 * 16); hipMemset(a.m, 0, n4 * 16); hipMemset(a.v, 0x3c, n4 * 16); hipMemset(inter, 0x3c, 3 * n4 * 16);
  hipMemset((void *)a.g, 0, (size_t)2 * n_b * hq * 16);
  int *d_pos, *d_need, *d_list, *d_lag; AdamC *d_tab;
  hipMalloc(&d_pos, rows * 4); hipMalloc(&d_need, rows * 4); hipMalloc(&d_list, list.size() * 4); hipMalloc(&d_lag, rows * 4);
  hipMalloc(&d_tab, 17 * sizeof(AdamC));
  hipMemcpy(d_pos, pos.data(), rows * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_need, need.data(), rows * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_lag, lag.data(), rows * 4, hipMemcpyHostToDevice);
  AdamC c = {0.1f, 0.999f, 0.001f, 1e-8f, 2e-5f, 0.3f, -1e-3f, 0.f};
  std::vector<AdamC> tab(17, c);
  hipMemcpy(d_tab, tab.data(), 17 * sizeof(AdamC), hipMemcpyHostToDevice);
  a.pos = d_pos; a.need = d_need; a.list = d_list; a.lag = d_lag; a.tab = d_tab;
  a.rows = rows; a.hq = hq; a.n_list = (int)list.size(); a.c = c;
  const double dense_mb = (rows * 24.0 * h + 2.0 * n_b * h * 4) / 1e6;
  const double lazy_mb = (list.size() * 24.0 * h + 2.0 * n_b * h * 4) / 1e6;
  auto grid = [](size_t n) { size_t g = (n + 255) / 256; return (int)std::min<size_t>(g, 8192); };
  auto report = [&](const char *name, float us, double mb) {
    printf("%-58s %7.2f us   %6.1f MB   %5.2f TB/s\n", name, us, mb, mb / us);
  };
  report("A dense flat sweep", time_us([&] { hipLaunchKernelGGL(kA, dim3(grid(n4)), dim3(256), 0, 0, a); }), dense_mb);
  report("B flat, rows skipped by a mask", time_us([&] { hipLaunchKernelGGL(kB, dim3(grid(n4)), dim3(256), 0, 0, a); }), lazy_mb);
  report("C wave per row, scalar mask loads", time_us([&] { hipLaunchKernelGGL(kC, dim3(grid(n4)), dim3(256), 0, 0, a); }), lazy_mb);
  const size_t nl4 = (size_t)list.size() * hq;
  report("D compact row list, flat over the list", time_us([&] { hipLaunchKernelGGL((kD<false, false>), dim3(grid(nl4)), dim3(256), 0, 0, a); }), lazy_mb);
  report("E = D + replay of the missed steps (LDS constants)", time_us([&] { hipLaunchKernelGGL((kD<true, false>), dim3(grid(nl4)), dim3(256), 0, 0, a); }), lazy_mb);
  Args f = a; f.p = inter;
  report("F = D on the interleaved layout [row][p|m|v]", time_us([&] { hipLaunchKernelGGL((kD<false, true>), dim3(grid(nl4)), dim3(256), 0, 0, f); }), lazy_mb);
  report("G = E on the interleaved layout", time_us([&] { hipLaunchKernelGGL((kD<true, true>), dim3(grid(nl4)), dim3(256), 0, 0, f); }), lazy_mb);
  // the same row COUNT as one contiguous prefix (what a popularity-sorted table with a sharp head would give)
  std::vector<int> pre(list.size());
  for (size_t k = 0; k < list.size(); ++k) pre[k] = (int)k;
  hipMemcpy(d_list, pre.data(), pre.size() * 4, hipMemcpyHostToDevice);
  report("D on a contiguous prefix of as many rows", time_us([&] { hipLaunchKernelGGL((kD<false, false>), dim3(grid(nl4)), dim3(256), 0, 0, a); }), lazy_mb);
  // H / I with the stamps of the host run (stamp = T + 1 - lag, T = 39)
  std::vector<int> st(rows);
  for (int r = 0; r < rows; ++r) st[r] = need[r] ? 40 - lag[r] : 0;
  int *d_stamp;
  hipMalloc(&d_stamp, rows * 4);
  hipMemcpy(d_stamp, st.data(), rows * 4, hipMemcpyHostToDevice);
  const int cT = 39 % L, lo = (int)((long)cT * N / L), hi = (int)((long)(cT + 1) * N / L);
  report("H wave per row, stamps + replay, LDS constants", time_us([&] { hipLaunchKernelGGL((kH<true, false>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }), lazy_mb);
  report("H' ... the chunk's rows (long replays) first", time_us([&] { hipLaunchKernelGGL((kH<true, true>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }), lazy_mb);
  report("I wave per row, stamps + replay, scalar global constants", time_us([&] { hipLaunchKernelGGL((kH<false, false>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }), lazy_mb);
  report("I' ... the chunk's rows first", time_us([&] { hipLaunchKernelGGL((kH<false, true>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }), lazy_mb);
  // COLD: 1 GB streamed through the chip between two launches (HBM-resident tables, as in the step)
  {
    char *junk; hipMalloc(&junk, (size_t)1 << 30);
    auto cold = [&](const char *name, auto launch, double mb) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float sum = 0.f;
      for (int rep = 0; rep < 8; ++rep) {
        hipMemsetAsync(junk, rep, (size_t)1 << 30, 0);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
      }
      char nm[128]; snprintf(nm, sizeof nm, "COLD %s", name);
      report(nm, sum / 6 * 1000.f, mb);
    };
    cold("A dense flat sweep", [&] { hipLaunchKernelGGL(kA, dim3(grid(n4)), dim3(256), 0, 0, a); }, dense_mb);
    hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice);
    cold("C wave per row, skip", [&] { hipLaunchKernelGGL(kC, dim3(grid(n4)), dim3(256), 0, 0, a); }, lazy_mb);
    cold("D compact list", [&] { hipLaunchKernelGGL((kD<false, false>), dim3(grid(nl4)), dim3(256), 0, 0, a); }, lazy_mb);
    cold("H stamps + replay, LDS constants", [&] { hipLaunchKernelGGL((kH<true, false>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }, lazy_mb);
    cold("H' chunk first", [&] { hipLaunchKernelGGL((kH<true, true>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }, lazy_mb);
    cold("I scalar global constants", [&] { hipLaunchKernelGGL((kH<false, false>), dim3(grid(n4)), dim3(256), 0, 0, a, d_stamp, 39, lo, hi); }, lazy_mb);
  }
  hipDeviceSynchronize();
  return 0;
}
