// Per-CU fetch rate from L2 / Infinity Cache / HBM with plain 16-byte loads: how many bytes a
// workgroup can pull per microsecond as a function of the loads it keeps in flight.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2rate tools/probes/l2_fetch_rate.hip && /tmp/l2rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(256) void rd_kernel(const float4 *__restrict__ x, size_t n_vec_per_wg,
                                                 size_t wg_stride, int iters, float *out) {
  const float4 *base = x + (size_t)blockIdx.x * wg_stride;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    for (size_t i = threadIdx.x; i < n_vec_per_wg; i += 256 * UNROLL) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = base[i + u * 256];   // (sizes are multiples of 256 * UNROLL)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 123.456f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int UNROLL>
float run(const float4 *x, size_t per_wg_vec, size_t stride, int wgs, int iters, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rd_kernel<UNROLL><<<wgs, 256>>>(x, per_wg_vec, stride, 1, out);
  hipEventRecord(e0);
  rd_kernel<UNROLL><<<wgs, 256>>>(x, per_wg_vec, stride, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const size_t total = (size_t)2 << 30;      // 2 GiB buffer
  float4 *x; float *out;
  hipMalloc(&x, total); hipMalloc(&out, 4);
  hipMemset(x, 0, total);
  struct Case { const char *name; size_t per_wg_bytes; size_t stride_bytes; int wgs; int iters; };
  // shared small set (every WG reads the same 1 MB: L2 hits after the first pass), private
  // L2-sized sets (256 x 64 KB = 16 MB over 8 XCD L2s of 4 MB), MALL-sized (256 x 512 KB = 128 MB),
  // HBM (1024 x 2 MB)
  Case cases[] = {
    {"L2 shared 1MB, 256 WG", 1 << 20, 0, 256, 40},
    {"L2 shared 1MB, 512 WG", 1 << 20, 0, 512, 40},
    {"L2 shared 1MB, 1024 WG", 1 << 20, 0, 1024, 40},
    {"L2 private 64KB x256", 64 << 10, 64 << 10, 256, 400},
    {"L2 private 32KB x512", 32 << 10, 32 << 10, 512, 400},
    {"MALL private 512KB x256", 512 << 10, 512 << 10, 256, 60},
    {"MALL private 128KB x1024", 128 << 10, 128 << 10, 1024, 60},
    {"HBM private 2MB x1024", 2 << 20, 2 << 20, 1024, 4},
    {"HBM private 8MB x256", 8 << 20, 8 << 20, 256, 2},
  };
  for (auto &c : cases) {
    const size_t vec = c.per_wg_bytes / 16, stride = c.stride_bytes / 16;
    float ms[4];
    ms[0] = run<1>(x, vec, stride, c.wgs, c.iters, out);
    ms[1] = run<2>(x, vec, stride, c.wgs, c.iters, out);
    ms[2] = run<4>(x, vec, stride, c.wgs, c.iters, out);
    ms[3] = run<8>(x, vec, stride, c.wgs, c.iters, out);
    const double bytes = (double)c.per_wg_bytes * c.wgs * c.iters;
    printf("%-28s", c.name);
    for (int k = 0; k < 4; ++k) printf("  U%d: %7.2f TB/s (%6.1f GB/s/CU)", 1 << k, bytes / ms[k] / 1e9, bytes / ms[k] / 1e6 / 256);
    printf("\n");
  }
  return 0;
}
