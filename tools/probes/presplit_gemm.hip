// What the decoder contraction could run at if its operands arrived PRE-SPLIT: C[M,N] = A . B^T with
// A, B given as fp16 hi / lo planes (K contiguous), three products per pair (lo.hi + hi.lo + hi.hi)
// accumulated in fp32 on v_mfma_f32_32x32x16_f16 -- the arithmetic of gemm.hip's PREC_H3, without
// the fp32 -> fp16-pair split in the k-loop (~150 VALU instructions per wave and k-tile there).
// 128 x 128 tiles, 4 waves of 64 x 64 (8 LDS reads for 12 MFMAs per 16-deep k-step), BK = 32, register
// prefetch of the next k-tile into a second LDS stage (one barrier per tile), plain fp32 store of C.  Sizes: the C5 shape at B = 4096 by default.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/psg tools/probes/presplit_gemm.hip && /tmp/psg [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 144;                 // LDS row: 64 B hi | 64 B lo | 16 B pad (odd number of 16-B slots)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void presplit_gemm(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al,
                                                     const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl,
                                                     float *__restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // two stages of (BM + BN) rows
  constexpr int STAGE = (BM + BN) * ROWB;
  // XCD-aware order as in gemm.hip: workgroup L runs on XCD L % 8; give each XCD a contiguous chunk
  const int total = gridDim.x, chunk = (total + 7) >> 3;
  const int t = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (t >= total) return;
  const int mt = t / tiles_n, nt = t % tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: per operand and k-tile 128 rows x (4 hi + 4 lo) 16-byte pieces = 1024 pieces, 4 per thread
  const _Float16 *srcA[4], *srcB[4];
  int dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;              // 0..1023
    const int row = idx >> 3, piece = idx & 7;  // piece 0..3 hi, 4..7 lo
    const int ra = min(m0 + row, M - 1), rb = min(n0 + row, N - 1);
    srcA[i] = (piece < 4 ? Ah : Al) + (size_t)ra * K + (piece & 3) * 8;
    srcB[i] = (piece < 4 ? Bh : Bl) + (size_t)rb * K + (piece & 3) * 8;
    dst[i] = row * ROWB + piece * 16;
  }
  uint4 ra[4], rb[4];
  // (a macro, not a lambda capturing the arrays: hipcc put them into scratch memory then)
#define GLOAD(k0)                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                      \
    ra[i] = *reinterpret_cast<const uint4 *>(srcA[i] + (k0));          \
    rb[i] = *reinterpret_cast<const uint4 *>(srcB[i] + (k0));          \
  }
#define SSTORE(buf)                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
    *reinterpret_cast<uint4 *>(smem + (buf) * STAGE + dst[i]) = ra[i];               \
    *reinterpret_cast<uint4 *>(smem + (buf) * STAGE + BM * ROWB + dst[i]) = rb[i];   \
  }
  GLOAD(0);
  SSTORE(0);
  __syncthreads();
  const int a_off = (wm * 64 + l31) * ROWB + lh * 16, b_off = BM * ROWB + (wn * 64 + l31) * ROWB + lh * 16;
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
    GLOAD(min(k0 + BK, K - BK));                 // the next tile, in flight under this one's MFMAs
    const char *S = smem + buf * STAGE;
    f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];          // [k-step][tile]: all LDS reads of the tile up front
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char *q = S + a_off + i * 32 * ROWB + ks * 32;
        ah[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));
        al[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64));
        const char *p = S + b_off + i * 32 * ROWB + ks * 32;
        bh[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(p));
        bl[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(p + 64));
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    }
    SSTORE(buf ^ 1);                             // (last read in the previous iteration: one barrier per tile)
    __syncthreads();
  }
  // plain store: lane (l31, lh) holds rows (r & 3) + 8 (r >> 2) + 4 lh of column l31 of each 32 x 32 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (m < M && n < N) C[(size_t)m * N + n] = acc[i][j][r];
      }
}

__global__ void fill(_Float16 *p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(((int)(x & 1023) - 512) * scale);
  }
}

int main(int argc, char **argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 336000, K = argc > 3 ? atoi(argv[3]) : 512;
  if (K % BK) { printf("K must be a multiple of %d\n", BK); return 1; }
  _Float16 *Ah, *Al, *Bh, *Bl;
  float *C;
  hipMalloc(&Ah, (size_t)M * K * 2); hipMalloc(&Al, (size_t)M * K * 2);
  hipMalloc(&Bh, (size_t)N * K * 2); hipMalloc(&Bl, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 4);
  fill<<<2048, 256>>>(Ah, (size_t)M * K, 1, 1.0f / 512); fill<<<2048, 256>>>(Al, (size_t)M * K, 2, 1.0f / (512 * 2048));
  fill<<<2048, 256>>>(Bh, (size_t)N * K, 3, 1.0f / 512); fill<<<2048, 256>>>(Bl, (size_t)N * K, 4, 1.0f / (512 * 2048));
  const int tiles_n = (N + BN - 1) / BN, tiles = ((M + BM - 1) / BM) * tiles_n;
  const int grid = ((tiles + 7) / 8) * 8;
  hipFuncSetAttribute((const void *)presplit_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * ROWB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) presplit_gemm<<<grid, 256, 2 * (BM + BN) * ROWB>>>(Ah, Al, Bh, Bl, C, M, N, K, tiles_n);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it) presplit_gemm<<<grid, 256, 2 * (BM + BN) * ROWB>>>(Ah, Al, Bh, Bl, C, M, N, K, tiles_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double flop = 2.0 * M * N * K;
  printf("M=%d N=%d K=%d: %.3f ms  %.1f algorithmic TFLOP/s (x3 products = %.0f TFLOP/s on the fp16 pipe), C store %.1f GB\n",
         M, N, K, ms, flop / ms * 1e-9, 3 * flop / ms * 1e-9, (double)M * N * 4e-9);
  // spot check against the host: a few entries of lo.hi + hi.lo + hi.hi in double
  std::vector<_Float16> hAh((size_t)M * K), hAl((size_t)M * K);
  hipMemcpy(hAh.data(), Ah, hAh.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hAl.data(), Al, hAl.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 8; ++s) {
    const int m = (s * 977) % M, n = (int)(((size_t)s * 104729) % N);
    std::vector<_Float16> bh(K), bl(K);
    hipMemcpy(bh.data(), Bh + (size_t)n * K, K * 2, hipMemcpyDeviceToHost); hipMemcpy(bl.data(), Bl + (size_t)n * K, K * 2, hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) {
      const double ah = (double)(float)hAh[(size_t)m * K + k], al = (double)(float)hAl[(size_t)m * K + k];
      ref += al * (double)(float)bh[k] + ah * (double)(float)bl[k] + ah * (double)(float)bh[k];
    }
    float got; hipMemcpy(&got, C + (size_t)m * N + n, 4, hipMemcpyDeviceToHost);
    worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 1e-6));
  }
  printf("spot check: max relative error %.2e\n", worst);
  return 0;
}
