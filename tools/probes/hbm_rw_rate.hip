// What a read-modify-write sweep can get out of the HBM: x[i] = f(x[i]) over three arrays in place
// (the shape of the dense Adam sweep: 3 streams read, 3 written), against a read-only and a
// write-only sweep of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rw tools/probes/hbm_rw_rate.hip && /tmp/rw
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void rmw3(float4 *p, float4 *m, float4 *v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 a = p[i], b = m[i], c = v[i];
    b.x = 0.9f * b.x + 0.1f * a.x; b.y = 0.9f * b.y + 0.1f * a.y; b.z = 0.9f * b.z + 0.1f * a.z; b.w = 0.9f * b.w + 0.1f * a.w;
    c.x = 0.99f * c.x + 0.01f * a.x * a.x; c.y = 0.99f * c.y + 0.01f * a.y * a.y;
    c.z = 0.99f * c.z + 0.01f * a.z * a.z; c.w = 0.99f * c.w + 0.01f * a.w * a.w;
    a.x -= 1e-3f * b.x; a.y -= 1e-3f * b.y; a.z -= 1e-3f * b.z; a.w -= 1e-3f * b.w;
    p[i] = a; m[i] = b; v[i] = c;
  }
}
// the same sweep with the row -> pos -> compact gradient row chain of the dense Adam job: 39 % of the
// rows (every row with pos >= 0) add a 16-byte gradient load from a compact [n_b][h] array
__global__ __launch_bounds__(256) void rmw3g(float4 *p, float4 *m, float4 *v, size_t n, const int *pos,
                                             const float4 *g, int hq) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned row = (unsigned)i / (unsigned)hq, q = (unsigned)i - row * (unsigned)hq;
    const int pr = pos[row];
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr >= 0) gg = g[(size_t)pr * hq + q];
    float4 a = p[i], b = m[i], c = v[i];
    b.x = 0.9f * b.x + 0.1f * (a.x + gg.x); b.y = 0.9f * b.y + 0.1f * (a.y + gg.y);
    b.z = 0.9f * b.z + 0.1f * (a.z + gg.z); b.w = 0.9f * b.w + 0.1f * (a.w + gg.w);
    c.x = 0.99f * c.x + 0.01f * gg.x * gg.x; c.y = 0.99f * c.y + 0.01f * gg.y * gg.y;
    c.z = 0.99f * c.z + 0.01f * gg.z * gg.z; c.w = 0.99f * c.w + 0.01f * gg.w * gg.w;
    a.x -= 1e-3f * b.x / (sqrtf(c.x) + 1e-8f); a.y -= 1e-3f * b.y / (sqrtf(c.y) + 1e-8f);
    a.z -= 1e-3f * b.z / (sqrtf(c.z) + 1e-8f); a.w -= 1e-3f * b.w / (sqrtf(c.w) + 1e-8f);
    p[i] = a; m[i] = b; v[i] = c;
  }
}
__global__ __launch_bounds__(256) void rd3(const float4 *p, const float4 *m, const float4 *v, size_t n, float *out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 a = p[i], b = m[i], c = v[i];
    s += a.x + b.y + c.z;
  }
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void wr3(float4 *p, float4 *m, float4 *v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
    p[i] = z; m[i] = z; v[i] = z;
  }
}

int main() {
  const size_t n = (size_t)2 * 20108 * 200 / 4;      // float4 per array: both C2 tables
  float4 *p, *m, *v; float *out;
  hipMalloc(&p, n * 16); hipMalloc(&m, n * 16); hipMalloc(&v, n * 16); hipMalloc(&out, 4);
  hipMemset(p, 0, n * 16); hipMemset(m, 0, n * 16); hipMemset(v, 0, n * 16);
  // pos map: 39 % of 2 * 20108 rows hold a gradient row (ascending), 200 floats per row
  const int rows = 2 * 20108, hq = 50;
  int *pos_h = (int *)malloc(rows * sizeof(int));
  int nb = 0;
  for (int r = 0; r < rows; ++r) pos_h[r] = ((r * 2654435761u) >> 8) % 100 < 39 ? nb++ : -1;
  int *pos; float4 *g;
  hipMalloc(&pos, rows * sizeof(int)); hipMalloc(&g, (size_t)nb * hq * 16);
  hipMemcpy(pos, pos_h, rows * sizeof(int), hipMemcpyHostToDevice); hipMemset(g, 0, (size_t)nb * hq * 16);
  // something else to stream between two sweeps (what a training step touches: ~170 MB)
  float4 *other; const size_t n_other = (size_t)170 << 20 >> 4;
  hipMalloc(&other, n_other * 16); hipMemset(other, 0, n_other * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int between = 0; between < 2; ++between) {
    float tot = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
      if (between) rd3<<<4096, 256>>>(other, other + n_other / 3, other + 2 * (n_other / 3), n_other / 3, out);
      hipEventRecord(e0);
      rmw3g<<<(int)((n + 255) / 256), 256>>>(p, m, v, n, pos, g, hq);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) tot += ms;
    }
    printf("Adam-shaped sweep (pos -> gradient rows, sqrt / div)%s: %6.1f us = %5.2f TB/s\n",
           between ? " with 170 MB streamed between two sweeps" : "", tot / 3 * 1e3,
           (6.0 * n * 16 + (double)nb * hq * 16) / (tot / 3) / 1e9);
  }
  const int grids[] = {2048, 4096, 8192, 0};
  for (int gi = 0; gi < 4; ++gi) {
    const int g = grids[gi] ? grids[gi] : (int)((n + 255) / 256);
    float ms[3];
    for (int k = 0; k < 3; ++k) {
      for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) hipEventRecord(e0);
        if (k == 0) rmw3<<<g, 256>>>(p, m, v, n);
        if (k == 1) rd3<<<g, 256>>>(p, m, v, n, out);
        if (k == 2) wr3<<<g, 256>>>(p, m, v, n);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[k], e0, e1);
    }
    printf("grid %6d: rmw %6.1f us = %5.2f TB/s | read %6.1f us = %5.2f TB/s | write %6.1f us = %5.2f TB/s   (%.0f MB per array set)\n",
           g, ms[0] * 1e3, 6.0 * n * 16 / ms[0] / 1e9, ms[1] * 1e3, 3.0 * n * 16 / ms[1] / 1e9, ms[2] * 1e3,
           3.0 * n * 16 / ms[2] / 1e9, 3.0 * n * 16 / 1e6);
  }
  return 0;
}
