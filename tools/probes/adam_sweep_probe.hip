// How fast can the dense Adam sweep of the C2 step go?  p, m, v [rows, h] fp32 read + written, the
// gradient read through pos[] from compact rows (2 partial slabs) for ~38 % of the rows, zero for the
// rest -- the traffic of rk_adam_multi's two table jobs (214 MB per launch, 38 us in the step).
// Variants of the loop structure, same arithmetic:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/asp tools/probes/adam_sweep_probe.hip && /tmp/asp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct C { float one_m_b1, b2, one_m_b2, eps, wd, bc2_sqrt, neg_step; };
__device__ __forceinline__ void adam1(float &p, float &m, float &v, float g, const C &c) {
  if (c.wd != 0.f) g = g + c.wd * p;
  m = fmaf(c.one_m_b1, g - m, m);
  v = v * c.b2 + (c.one_m_b2 * g) * g;
  const float d = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p + (c.neg_step * m) / d;
}
__device__ __forceinline__ void adam4(float4 &p, float4 &m, float4 &v, const float4 g, const C &c) {
  adam1(p.x, m.x, v.x, g.x, c); adam1(p.y, m.y, v.y, g.y, c); adam1(p.z, m.z, v.z, g.z, c); adam1(p.w, m.w, v.w, g.w, c);
}
__device__ __forceinline__ float4 grad(const float *G, const int *pos, int row, int q, int h, int64_t stride) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const int pr = pos[row];
  if (pr >= 0) {
    g = *reinterpret_cast<const float4 *>(G + (int64_t)pr * h + q * 4);
    const float4 o = *reinterpret_cast<const float4 *>(G + stride + (int64_t)pr * h + q * 4);
    g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
  }
  return g;
}

// V0: the shipped loop (grid-stride, one float4 per iteration)
__global__ __launch_bounds__(256) void v0(float4 *P, float4 *M, float4 *V, const float *G, const int *pos, int rows, int h, int64_t stride, C c) {
  const int hq = h >> 2;
  const int64_t tot = (int64_t)rows * hq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / hq), q = (int)(i % hq);
    const float4 g = grad(G, pos, row, q, h, stride);
    float4 p = P[i], m = M[i], v = V[i];
    adam4(p, m, v, g, c);
    P[i] = p; M[i] = m; V[i] = v;
  }
}
// V1: two independent elements per iteration (all loads first)
__global__ __launch_bounds__(256) void v1(float4 *P, float4 *M, float4 *V, const float *G, const int *pos, int rows, int h, int64_t stride, C c) {
  const int hq = h >> 2;
  const int64_t tot = (int64_t)rows * hq, step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += 2 * step) {
    const int64_t j = i + step;
    const bool two = j < tot;
    const int64_t jj = two ? j : i;
    float4 p0 = P[i], m0 = M[i], v0_ = V[i], p1 = P[jj], m1 = M[jj], v1_ = V[jj];
    const float4 g0 = grad(G, pos, (int)(i / hq), (int)(i % hq), h, stride);
    const float4 g1 = grad(G, pos, (int)(jj / hq), (int)(jj % hq), h, stride);
    adam4(p0, m0, v0_, g0, c);
    P[i] = p0; M[i] = m0; V[i] = v0_;
    if (two) { adam4(p1, m1, v1_, g1, c); P[j] = p1; M[j] = m1; V[j] = v1_; }
  }
}
// V2: a wave per row slice: pos once per row (scalar), 64 lanes cover 64 float4 (h <= 256)
__global__ __launch_bounds__(256) void v2(float4 *P, float4 *M, float4 *V, const float *G, const int *pos, int rows, int h, int64_t stride, C c) {
  const int hq = h >> 2, lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int row = wave; row < rows; row += nw) {
    if (lane < hq) {
      const int64_t i = (int64_t)row * hq + lane;
      float4 p = P[i], m = M[i], v = V[i];
      const float4 g = grad(G, pos, row, lane, h, stride);
      adam4(p, m, v, g, c);
      P[i] = p; M[i] = m; V[i] = v;
    }
  }
}
// V3: nontemporal stores only (loads normal)
__global__ __launch_bounds__(256) void v3(float4 *P, float4 *M, float4 *V, const float *G, const int *pos, int rows, int h, int64_t stride, C c) {
  const int hq = h >> 2;
  const int64_t tot = (int64_t)rows * hq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / hq), q = (int)(i % hq);
    const float4 g = grad(G, pos, row, q, h, stride);
    float4 p = P[i], m = M[i], v = V[i];
    adam4(p, m, v, g, c);
    __builtin_nontemporal_store(p.x, &P[i].x); __builtin_nontemporal_store(p.y, &P[i].y);
    __builtin_nontemporal_store(p.z, &P[i].z); __builtin_nontemporal_store(p.w, &P[i].w);
    __builtin_nontemporal_store(m.x, &M[i].x); __builtin_nontemporal_store(m.y, &M[i].y);
    __builtin_nontemporal_store(m.z, &M[i].z); __builtin_nontemporal_store(m.w, &M[i].w);
    __builtin_nontemporal_store(v.x, &V[i].x); __builtin_nontemporal_store(v.y, &V[i].y);
    __builtin_nontemporal_store(v.z, &V[i].z); __builtin_nontemporal_store(v.w, &V[i].w);
  }
}
// copy3: the traffic floor -- three arrays read and written, nothing else
__global__ __launch_bounds__(256) void copy3(float4 *P, float4 *M, float4 *V, int64_t tot) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    float4 p = P[i], m = M[i], v = V[i];
    p.x += 1.f; m.x += 1.f; v.x += 1.f;
    P[i] = p; M[i] = m; V[i] = v;
  }
}

int main() {
  const int rows = 2 * 20108, h = 200, n_cap = 2 * 8192;
  const int64_t n = (int64_t)rows * h;
  float *P, *M, *V, *G; int *pos;
  hipMalloc(&P, n * 4); hipMalloc(&M, n * 4); hipMalloc(&V, n * 4); hipMalloc(&G, (int64_t)2 * n_cap * h * 4); hipMalloc(&pos, rows * 4);
  hipMemset(P, 0, n * 4); hipMemset(M, 0, n * 4); hipMemset(V, 0, n * 4); hipMemset(G, 0, (int64_t)2 * n_cap * h * 4);
  std::vector<int> hp(rows);
  int c = 0; unsigned x = 12345;
  for (int r = 0; r < rows; ++r) { x = x * 1664525u + 1013904223u; hp[r] = ((x >> 8) % 100 < 38 && c < n_cap) ? c++ : -1; }
  hipMemcpy(pos, hp.data(), rows * 4, hipMemcpyHostToDevice);
  // something large between timed launches would be the real step; here: back-to-back sweeps (each
  // touches 483 MB of distinct lines, more than L2 + MALL keep)
  C cc = {0.1f, 0.999f, 0.001f, 1e-8f, 2e-5f, 0.5f, -1e-3f};
  const int64_t stride = (int64_t)n_cap * h;
  const double mb = (6.0 * n * 4 + 2.0 * c * h * 4) / 1e6;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 40;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.2f us  %6.2f TB/s\n", name, ms / reps * 1e3, mb / (ms / reps * 1e3));
  };
  printf("%.1f MB per sweep (%d live gradient rows)\n", mb, c);
  for (int grid : {2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "v0 grid %d", grid);
    run(nm, [&] { v0<<<grid, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, G, pos, rows, h, stride, cc); });
  }
  const int full = (int)((n / 4 + 255) / 256);
  run("v0 one element per thread", [&] { v0<<<full, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, G, pos, rows, h, stride, cc); });
  for (int grid : {2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, 64, "v1 (x2) grid %d", grid);
    run(nm, [&] { v1<<<grid, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, G, pos, rows, h, stride, cc); });
  }
  for (int grid : {2048, 4096, 10054}) {
    char nm[64];
    snprintf(nm, 64, "v2 (wave per row) grid %d", grid);
    run(nm, [&] { v2<<<grid, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, G, pos, rows, h, stride, cc); });
  }
  run("v3 (nt stores) grid 8192", [&] { v3<<<8192, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, G, pos, rows, h, stride, cc); });
  for (int grid : {4096, 8192})  {
    char nm[64];
    snprintf(nm, 64, "copy3 grid %d", grid);
    run(nm, [&] { copy3<<<grid, 256>>>((float4 *)P, (float4 *)M, (float4 *)V, n / 4); });
  }
  return 0;
}
