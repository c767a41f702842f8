// Can timing events live INSIDE a replayed HIP graph?  Capture  k1 ; record(e0) ; k2 ; record(e1) ; k3
// with hipEventRecordWithFlags(..., hipEventRecordExternal) (an event-record NODE instead of a capture
// dependency), replay it a few times and read hipEventElapsedTime(e0, e1) after each replay.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/gep tools/probes/graph_event_probe.hip && /tmp/gep
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(float *x, int n) {
  float v = x[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  x[threadIdx.x] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  float *d;
  CK(hipMalloc(&d, 4096));
  CK(hipMemset(d, 0, 4096));
  hipStream_t s, side;
  const char *mode = argc > 1 ? argv[1] : "plain";
  if (mode[0] == 'n') { CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0)); }
  else { CK(hipStreamCreate(&s)); }
  CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, 0));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  const bool forked = argc > 2;
  printf("stream: %s, forked side stream in the capture: %d\n", mode, (int)forked);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) spin<<<1, 256, 0, s>>>(d, 1000);
  CK(hipStreamSynchronize(s));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  spin<<<1, 256, 0, s>>>(d, 20000);
  if (forked) {
    CK(hipEventRecord(fork, s));
    CK(hipStreamWaitEvent(side, fork, 0));
    spin<<<1, 256, 0, side>>>(d + 512, 20000);
  }
  CK(hipEventRecordWithFlags(e0, s, hipEventRecordExternal));
  spin<<<1, 256, 0, s>>>(d, 200000);
  CK(hipEventRecordWithFlags(e1, s, hipEventRecordExternal));
  spin<<<1, 256, 0, s>>>(d, 20000);
  if (forked) {
    CK(hipEventRecord(join, side));
    CK(hipStreamWaitEvent(s, join, 0));
  }
  hipGraph_t g;
  CK(hipStreamEndCapture(s, &g));
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  printf("graph nodes: %zu\n", nn);
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 4; ++r) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float in = -1, out = -1;
    hipError_t e = hipEventElapsedTime(&in, e0, e1);
    CK(hipEventElapsedTime(&out, a, b));
    printf("replay %d: inside %.3f ms (%s)  whole graph %.3f ms\n", r, in, hipGetErrorString(e), out);
  }
  // reference: the middle kernel alone, eagerly
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s));
  spin<<<1, 256, 0, s>>>(d, 200000);
  CK(hipEventRecord(b, s));
  CK(hipStreamSynchronize(s));
  float t;
  CK(hipEventElapsedTime(&t, a, b));
  printf("eager middle kernel: %.3f ms\n", t);
  return 0;
}
