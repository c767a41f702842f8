// Probe: per-phase cycle stamps of one wave of the hot GEMM kernels.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRK_PROBE -w \
//        tools/probes/gemm_probe.hip recoder_amd/csrc/optim.hip recoder_amd/csrc/capi.hip -o tools/probes/gemm_probe
#include "../../recoder_amd/csrc/gemm.hip"
#include <vector>
#include <stdlib.h>
static void dump(const char *name) {
  unsigned long long h[16];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(rk_dbg), sizeof(h));
  printf("%-8s total %6llu | prologue %6llu | tile2: gload %5llu compute %5llu sstore %5llu barrier %5llu | 2nd half %6llu | loop(all) %6llu | epilogue %6llu\n",
         name, h[9] - h[0], h[1] - h[0], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[1], h[9] - h[8]);
}
int main() {
  const int B = 500, h = 200, n_t = 7628, ld = (n_t + 31) & ~31, n_items = 20108;
  float *dO, *Z, *G, *W, *bias, *ws, *dZ, *part, *gbp; int32_t *counts, *items, *indptr, *pref; uint32_t *bits;
  hipMalloc(&dO, (size_t)B * ld * 4); hipMalloc(&Z, B * h * 4); hipMalloc(&G, (size_t)n_t * h * 4);
  hipMalloc(&W, (size_t)n_items * h * 4); hipMalloc(&bias, n_items * 4); hipMalloc(&dZ, B * h * 4);
  hipMalloc(&ws, rk_dz_workspace_bytes(B, h)); hipMalloc(&part, 65536 * 4); hipMalloc(&gbp, (size_t)16 * ld * 4);
  hipMalloc(&counts, 16); hipMalloc(&items, n_t * 4); hipMalloc(&indptr, (B + 1) * 4);
  hipMalloc(&bits, (size_t)B * (ld / 32) * 4); hipMalloc(&pref, (size_t)B * (ld / 32) * 4);
  hipMemset(dO, 0, (size_t)B * ld * 4); hipMemset(Z, 0, B * h * 4); hipMemset(W, 0, (size_t)n_items * h * 4);
  hipMemset(bias, 0, n_items * 4); hipMemset(part, 0, 65536 * 4); hipMemset(bits, 0, (size_t)B * (ld / 32) * 4);
  hipMemset(pref, 0, (size_t)B * (ld / 32) * 4); hipMemset(indptr, 0, (B + 1) * 4);
  int32_t hc[4] = {n_t, 0, ld, B};
  hipMemcpy(counts, hc, 16, hipMemcpyHostToDevice);
  std::vector<int32_t> it(n_t); for (int i = 0; i < n_t; ++i) it[i] = getenv("CONTIG") ? i : (int)((long long)i * n_items / n_t);
  hipMemcpy(items, it.data(), n_t * 4, hipMemcpyHostToDevice);
  rk_block_t blk = {};
  blk.S_cap = B; blk.nnz_cap = 1; blk.n_cap = n_t; blk.n_items = n_items; blk.ldw_rc = ld / 32; blk.ldw_cr = 16;
  blk.implicit = 1; blk.counts = counts; blk.items = items; blk.indptr = indptr; blk.bits_rc = bits; blk.pref_rc = pref;
  for (int rep = 0; rep < 3; ++rep) {
    rk_decode_loss(Z, B, h, &blk, 0, W, bias, RK_LOSS_MSE, 0.f, 1.f / B, dO, 0, part, gbp, nullptr, nullptr);
    hipDeviceSynchronize(); if (rep == 2) dump("decode");
    rk_decode_bwd_dz(dO, B, h, &blk, W, nullptr, 0, dZ, ws, nullptr, nullptr);
    hipDeviceSynchronize(); if (rep == 2) dump("dz");
    rk_decode_bwd_dw(dO, Z, B, h, &blk, G, nullptr, nullptr);
    hipDeviceSynchronize(); if (rep == 2) dump("dw");
  }
  printf("last error: %s\n", rk_last_error());
  return 0;
}
