// Stand-alone probe of csrc/pgemm.h (no torch): numerics of the three operand forms against a float64
// product, the LDS transpose read's lane map, and the rate of the contraction shapes of C5 / C2.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/pgemm_probe.hip -o tools/probes/pgemm_probe
//   tools/probes/pgemm_probe check | bench [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../recoder_amd/csrc/pgemm.h"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// image of X[rows][cols] (row-major, ld): rows_pad rows x ceil(cols / 32) lines, s.x = hi + lo
static std::vector<char> make_image(const float *X, int rows, int cols, int ld, float s, int rows_pad) {
  const int lines = (cols + 31) / 32;
  std::vector<char> img((size_t)rows_pad * lines * 128, 0);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const float v = X[(size_t)r * ld + c] * s;
      const uint16_t hi = f2h(v);
      const uint16_t lo = f2h(v - h2f(hi));
      char *line = img.data() + ((size_t)r * lines + (c >> 5)) * 128;
      memcpy(line + (c & 31) * 2, &hi, 2);
      memcpy(line + 64 + (c & 31) * 2, &lo, 2);
    }
  return img;
}

__global__ void tr_probe_kernel(unsigned short *out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  const pg::s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pg::lds_s16x4 *)(sm + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

static int tr_semantic() {
  unsigned short *d, h[256];
  HC(hipMalloc(&d, 512));
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, 0, d);
  HC(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = (l >> 4) * 64 + j * 16 + (l & 15);
      if (h[l * 4 + j] != want) ++bad;
    }
  printf("ds_read_b64_tr_b16 lane map (lane l, elem j reads element (l>>4)*64 + j*16 + (l&15)): %s\n", bad ? "MISMATCH" : "ok");
  if (bad) {
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  HC(hipFree(d));
  return bad;
}

struct Case { int M, N, K, cfg, atr, btr, splits; };

static int g_var = 0;      // tuning variant (pgemm.h VAR): env VAR

template <int BM, int BN, int WM, int WN, int VAR>
static hipError_t dispatch_v(const pg::Core &p, const pg::EpiStore::Args &e, int atr, int btr, int tiles, hipStream_t s) {
  if (!atr && !btr) return pg::launch<BM, BN, WM, WN, false, false, pg::EpiStore, VAR>(p, e, tiles, s);
  if (!atr && btr) return pg::launch<BM, BN, WM, WN, false, true, pg::EpiStore, VAR>(p, e, tiles, s);
  if (atr && btr) return pg::launch<BM, BN, WM, WN, true, true, pg::EpiStore, VAR>(p, e, tiles, s);
  return hipErrorInvalidValue;
}
template <int BM, int BN, int WM, int WN>
static hipError_t dispatch(const pg::Core &p, const pg::EpiStore::Args &e, int atr, int btr, int tiles, hipStream_t s) {
  switch (g_var) {
    case 1: return dispatch_v<BM, BN, WM, WN, 1>(p, e, atr, btr, tiles, s);
    case 2: return dispatch_v<BM, BN, WM, WN, 2>(p, e, atr, btr, tiles, s);
    case 3: return dispatch_v<BM, BN, WM, WN, 3>(p, e, atr, btr, tiles, s);
    case 4: return dispatch_v<BM, BN, WM, WN, 4>(p, e, atr, btr, tiles, s);
    case 8: return dispatch_v<BM, BN, WM, WN, 8>(p, e, atr, btr, tiles, s);
    case 12: return dispatch_v<BM, BN, WM, WN, 12>(p, e, atr, btr, tiles, s);
    case 24: return dispatch_v<BM, BN, WM, WN, 24>(p, e, atr, btr, tiles, s);
    case 32: return dispatch_v<BM, BN, WM, WN, 32>(p, e, atr, btr, tiles, s);
    case 40: return dispatch_v<BM, BN, WM, WN, 40>(p, e, atr, btr, tiles, s);
    case 56: return dispatch_v<BM, BN, WM, WN, 56>(p, e, atr, btr, tiles, s);
    case 72: return dispatch_v<BM, BN, WM, WN, 72>(p, e, atr, btr, tiles, s);
    case 128: return dispatch_v<BM, BN, WM, WN, 128>(p, e, atr, btr, tiles, s);
    case 160: return dispatch_v<BM, BN, WM, WN, 160>(p, e, atr, btr, tiles, s);
    default: return dispatch_v<BM, BN, WM, WN, 0>(p, e, atr, btr, tiles, s);
  }
}

static hipError_t run_cfg(int cfg, const pg::Core &p, const pg::EpiStore::Args &e, int atr, int btr, hipStream_t s) {
  if (cfg == 0) return dispatch<256, 256, 2, 4>(p, e, atr, btr, ((p.M + 255) / 256) * ((p.N + 255) / 256), s);
  if (cfg == 1) return dispatch<128, 128, 2, 2>(p, e, atr, btr, ((p.M + 127) / 128) * ((p.N + 127) / 128), s);
  if (cfg == 2) return dispatch<128, 256, 2, 4>(p, e, atr, btr, ((p.M + 127) / 128) * ((p.N + 255) / 256), s);
  if (cfg == 3) return dispatch<256, 128, 4, 2>(p, e, atr, btr, ((p.M + 255) / 256) * ((p.N + 127) / 128), s);
  if (cfg == 4) {                // the 64 x 128 dW tile of < 1024-row batches; env RING = LDS stages of the deep ring
    const int tiles = ((p.M + 63) / 64) * ((p.N + 127) / 128);
    static const int ring = getenv("RING") ? atoi(getenv("RING")) : 0;
    if (atr && btr && ring == 2) return pg::launch<64, 128, 2, 2, true, true, pg::EpiStore, 256, false, 2>(p, e, tiles, s);
    if (atr && btr && ring == 3) return pg::launch<64, 128, 2, 2, true, true, pg::EpiStore, 0, false, 3>(p, e, tiles, s);
    if (atr && btr && ring == 4) return pg::launch<64, 128, 2, 2, true, true, pg::EpiStore, 0, false, 4>(p, e, tiles, s);
    if (atr && btr && ring == 6) return pg::launch<64, 128, 2, 2, true, true, pg::EpiStore, 0, false, 6>(p, e, tiles, s);
    return dispatch<64, 128, 2, 2>(p, e, atr, btr, tiles, s);
  }
  return hipErrorInvalidValue;
}

struct Dev {
  char *a = nullptr, *b = nullptr;
  float *c = nullptr;
  pg::Core p;
  pg::EpiStore::Args e;
};

// A_math[M][K], B_math[K][N] -> operand images in the requested forms
static Dev setup(const Case &cs, const std::vector<float> &A, const std::vector<float> &B, float sa, float sb) {
  Dev d;
  const int M = cs.M, N = cs.N, K = cs.K;
  std::vector<char> ia, ib;
  pg::Core p = {};
  if (!cs.atr) {
    ia = make_image(A.data(), M, K, K, sa, M);
    p.a.lines = (K + 31) / 32; p.a.rows = M;
  } else {
    std::vector<float> At((size_t)K * M);
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) At[(size_t)k * M + m] = A[(size_t)m * K + k];
    const int rp = (K + 31) / 32 * 32;
    ia = make_image(At.data(), K, M, M, sa, rp);
    p.a.lines = (M + 31) / 32; p.a.rows = rp;
  }
  if (!cs.btr) {
    std::vector<float> Bt((size_t)N * K);
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) Bt[(size_t)n * K + k] = B[(size_t)k * N + n];
    ib = make_image(Bt.data(), N, K, K, sb, N);
    p.b.lines = (K + 31) / 32; p.b.rows = N;
  } else {
    const int rp = (K + 31) / 32 * 32;
    ib = make_image(B.data(), K, N, N, sb, rp);
    p.b.lines = (N + 31) / 32; p.b.rows = rp;
  }
  HC(hipMalloc(&d.a, ia.size())); HC(hipMemcpy(d.a, ia.data(), ia.size(), hipMemcpyHostToDevice));
  HC(hipMalloc(&d.b, ib.size())); HC(hipMemcpy(d.b, ib.data(), ib.size(), hipMemcpyHostToDevice));
  HC(hipMalloc(&d.c, (size_t)cs.splits * M * N * 4));
  p.a.img = d.a; p.a.pitch = (int64_t)p.a.lines * 128;
  p.b.img = d.b; p.b.pitch = (int64_t)p.b.lines * 128;
  p.M = M; p.N = N; p.K = K; p.splits = cs.splits;
  d.p = p;
  d.e = {};
  d.e.C = d.c; d.e.ldc = N; d.e.slab_stride = (int64_t)M * N; d.e.scale = 1.0f / (sa * sb);
  return d;
}

static void release(Dev &d) { HC(hipFree(d.a)); HC(hipFree(d.b)); HC(hipFree(d.c)); }

static int check_case(const Case &cs) {
  const int M = cs.M, N = cs.N, K = cs.K;
  std::vector<float> A((size_t)M * K), B((size_t)K * N);
  srand(1234 + M + 7 * N + 13 * K);
  for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  Dev d = setup(cs, A, B, 32.f, 128.f);
  HC(hipMemset(d.c, 0xff, (size_t)cs.splits * M * N * 4));
  HC(run_cfg(cs.cfg, d.p, d.e, cs.atr, cs.btr, 0));
  HC(hipDeviceSynchronize());
  std::vector<float> C((size_t)cs.splits * M * N);
  HC(hipMemcpy(C.data(), d.c, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  int wm = -1, wn = -1;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double ex = 0, sc = 0;
      for (int k = 0; k < K; ++k) {
        const double t = (double)A[(size_t)m * K + k] * B[(size_t)k * N + n];
        ex += t; sc += fabs(t);
      }
      double got = 0;
      for (int s = 0; s < cs.splits; ++s) got += C[((size_t)s * M + m) * N + n];
      const double err = fabs(got - ex) / (sc + 1e-30);
      if (!(err <= worst)) { worst = err; wm = m; wn = n; }
    }
  const bool ok = worst < 1e-6;
  printf("check M=%d N=%d K=%d cfg=%d A%s B%s splits=%d: max err / sum|products| = %.3e at (%d, %d) %s\n", M, N, K,
         cs.cfg, cs.atr ? "TR" : "KC", cs.btr ? "TR" : "KC", cs.splits, worst, wm, wn, ok ? "ok" : "FAIL");
  release(d);
  return ok ? 0 : 1;
}

static void bench_case(const char *name, const Case &cs, int reps) {
  const int M = cs.M, N = cs.N, K = cs.K;
  // images of random fp16 values straight on the host (no fp32 source: the big shapes would take
  // minutes to split on one core): every 16-bit pattern of a small finite fp16
  pg::Core p = {};
  const int a_rows = cs.atr ? (K + 31) / 32 * 32 : M, a_lines = cs.atr ? (M + 31) / 32 : (K + 31) / 32;
  const int b_rows = cs.btr ? (K + 31) / 32 * 32 : N, b_lines = cs.btr ? (N + 31) / 32 : (K + 31) / 32;
  const size_t ab = (size_t)a_rows * a_lines * 128, bb = (size_t)b_rows * b_lines * 128;
  std::vector<uint16_t> pool(1 << 20);
  srand(7);
  for (auto &v : pool) v = f2h(((float)rand() / RAND_MAX * 2.f - 1.f));
  char *da, *db;
  float *dc;
  HC(hipMalloc(&da, ab)); HC(hipMalloc(&db, bb));
  for (size_t o = 0; o < ab; o += pool.size() * 2) HC(hipMemcpy(da + o, pool.data(), std::min(pool.size() * 2, ab - o), hipMemcpyHostToDevice));
  for (size_t o = 0; o < bb; o += pool.size() * 2) HC(hipMemcpy(db + o, pool.data() + 12345, std::min(pool.size() * 2 - 24690, bb - o), hipMemcpyHostToDevice));
  HC(hipMalloc(&dc, (size_t)cs.splits * M * N * 4));
  p.a.img = da; p.a.lines = a_lines; p.a.rows = a_rows; p.a.pitch = (int64_t)a_lines * 128;
  p.b.img = db; p.b.lines = b_lines; p.b.rows = b_rows; p.b.pitch = (int64_t)b_lines * 128;
  p.M = M; p.N = N; p.K = K; p.splits = cs.splits;
  if (getenv("NLIVE")) {            // the grid covers N (a capacity), the live column count sits on the device
    int32_t *nd;
    const int32_t nl = atoi(getenv("NLIVE"));
    HC(hipMalloc(&nd, 4));
    HC(hipMemcpy(nd, &nl, 4, hipMemcpyHostToDevice));
    p.Ndev = nd;
  }
  pg::EpiStore::Args e = {};
  e.C = dc; e.ldc = N; e.slab_stride = (int64_t)M * N; e.scale = 1.f;
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) HC(run_cfg(cs.cfg, p, e, cs.atr, cs.btr, 0));
  HC(hipDeviceSynchronize());
  HC(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) HC(run_cfg(cs.cfg, p, e, cs.atr, cs.btr, 0));
  HC(hipEventRecord(e1, 0));
  HC(hipEventSynchronize(e1));
  float ms;
  HC(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double fl = 2.0 * M * N * K;
  printf("bench %-28s M=%-7d N=%-7d K=%-7d cfg=%d A%s B%s splits=%-3d %9.3f us  %7.1f algorithmic TFLOP/s (%.1f MFMA TF/s)\n",
         name, M, N, K, cs.cfg, cs.atr ? "TR" : "KC", cs.btr ? "TR" : "KC", cs.splits, ms * 1e3, fl / ms / 1e9,
         3 * fl / ms / 1e9);
  HC(hipFree(da)); HC(hipFree(db)); HC(hipFree(dc));
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "check";
  g_var = getenv("VAR") ? atoi(getenv("VAR")) : 0;
  printf("VAR = %d\n", g_var);
  if (!strcmp(mode, "check")) {
    int bad = tr_semantic();
    const Case cases[] = {
        {300, 520, 200, 1, 0, 0, 1}, {300, 520, 200, 0, 0, 0, 1}, {257, 130, 96, 2, 0, 0, 1}, {257, 130, 96, 3, 0, 0, 1},
        {300, 200, 1000, 1, 0, 1, 1}, {300, 200, 1000, 0, 0, 1, 3}, {130, 260, 700, 2, 0, 1, 2}, {300, 200, 1000, 3, 0, 1, 2},
        {520, 200, 300, 1, 1, 1, 1}, {520, 200, 300, 0, 1, 1, 2}, {700, 260, 130, 2, 1, 1, 1}, {520, 136, 333, 3, 1, 1, 1},
        {64, 64, 32, 1, 0, 0, 1}, {1, 1, 1, 1, 0, 0, 1}, {33, 70, 45, 1, 1, 1, 1},
    };
    for (const Case &c : cases) bad += check_case(c);
    printf(bad ? "FAILED: %d\n" : "all checks passed\n", bad);
    return bad ? 1 : 0;
  }
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  if (!strcmp(mode, "one")) {      // one M N K cfg atr btr splits [reps]: a single case (profiling)
    Case c = {atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8])};
    bench_case("one", c, argc > 9 ? atoi(argv[9]) : 5);
    return 0;
  }
  if (!strcmp(mode, "quick")) {    // the large shapes on the two main tilings only
    const int NBq = getenv("NB") ? atoi(getenv("NB")) : 131072;
    for (int cfg = 0; cfg < 2; ++cfg) bench_case("c5 decode  Z.W^T", {4096, NBq, 512, cfg, 0, 0, 1}, reps);
    for (int cfg = 0; cfg < 2; ++cfg) bench_case("c5 dZ      dO.W", {4096, 512, NBq, cfg, 0, 1, cfg == 0 ? 16 : 4}, reps);
    for (int cfg = 0; cfg < 2; ++cfg) bench_case("c5 dW      dO^T.Z", {NBq, 512, 4096, cfg, 1, 1, 1}, reps);
    for (int cfg = 0; cfg < 2; ++cfg) bench_case("c5b500 decode", {500, 48800, 512, cfg, 0, 0, 1}, reps * 4);
    for (int cfg = 0; cfg < 2; ++cfg) bench_case("c5b500 dW", {48800, 512, 500, cfg, 1, 1, 1}, reps * 4);
    for (int cfg = 1; cfg < 3; ++cfg) bench_case("c2 decode", {500, 7900, 200, cfg, 0, 0, 1}, reps * 10);
    for (int cfg = 1; cfg < 3; ++cfg) bench_case("c2 dW", {7900, 200, 500, cfg, 1, 1, cfg == 1 ? 2 : 4}, reps * 10);
    return 0;
  }
  // C5 at B = 4096 (n_b = 336 k in the real step; 131 072 items here: the same tiles, a third of them)
  const int NB = getenv("NB") ? atoi(getenv("NB")) : 131072;
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5 decode  Z.W^T", {4096, NB, 512, cfg, 0, 0, 1}, reps);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5 dZ      dO.W", {4096, 512, NB, cfg, 0, 1, cfg == 0 ? 16 : (cfg == 1 ? 4 : 8)}, reps);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5 dW      dO^T.Z", {NB, 512, 4096, cfg, 1, 1, 1}, reps);
  // C5 at B = 500 (n_b = 48.8 k)
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5b500 decode", {500, 48800, 512, cfg, 0, 0, 1}, reps * 4);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5b500 dZ", {500, 512, 48800, cfg, 0, 1, cfg == 1 ? 32 : 64}, reps * 4);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c5b500 dW", {48800, 512, 500, cfg, 1, 1, 1}, reps * 4);
  // C2 (B = 500, n_b = 7.9 k, h = 200)
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c2 decode", {500, 7900, 200, cfg, 0, 0, 1}, reps * 10);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c2 dZ", {500, 200, 7900, cfg, 0, 1, cfg == 1 ? 32 : 62}, reps * 10);
  for (int cfg = 0; cfg < 4; ++cfg) bench_case("c2 dW", {7900, 200, 500, cfg, 1, 1, cfg == 1 ? 2 : 4}, reps * 10);
  return 0;
}
