// Probe: issue rate of v_mfma_f32_32x32x2_f32 (and 16x16x4) with NACC
// independent accumulators, 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float run(F f) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  f(); hipDeviceSynchronize();
  hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
  float *out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int blocks : {256, 512, 1024}) {
    float ms;
    ms = run([&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f); });
    printf("32x32x2 nacc1 blocks %4d: %.3f ms  %.1f TF  (%.1f cyc/mfma/wave @2.4GHz)\n", blocks, ms, blocks*4.0*iters*1*4096/ms/1e9, ms*1e-3*2.4e9/(iters*1));
    ms = run([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f); });
    printf("32x32x2 nacc2 blocks %4d: %.3f ms  %.1f TF  (%.1f cyc/mfma/wave)\n", blocks, ms, blocks*4.0*iters*2*4096/ms/1e9, ms*1e-3*2.4e9/(iters*2));
    ms = run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f); });
    printf("32x32x2 nacc4 blocks %4d: %.3f ms  %.1f TF  (%.1f cyc/mfma/wave)\n", blocks, ms, blocks*4.0*iters*4*4096/ms/1e9, ms*1e-3*2.4e9/(iters*4));
    ms = run([&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f); });
    printf("16x16x4 nacc2 blocks %4d: %.3f ms  %.1f TF  (%.1f cyc/mfma/wave)\n", blocks, ms, blocks*4.0*iters*2*2048/ms/1e9, ms*1e-3*2.4e9/(iters*2));
    ms = run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f); });
    printf("16x16x4 nacc8 blocks %4d: %.3f ms  %.1f TF  (%.1f cyc/mfma/wave)\n", blocks, ms, blocks*4.0*iters*8*2048/ms/1e9, ms*1e-3*2.4e9/(iters*8));
  }
  return 0;
}
