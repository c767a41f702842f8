// Decoder contractions on PRE-SPLIT operand planes (csrc/planes.h) -- the round-3 replacement of
// gemm.hip's PREC_H3 kernels on the hot path:
//
//   decode + loss : O[B,n_b] = Z . W_de[T]^T + b_de[T]  (reference nn.py:271-280) + the fused loss /
//                   dLoss/dLogits epilogue (losses.py:43-47, BCEWithLogits) -- both operands arrive
//                   as fp16 hi / lo planes, the k-loop is copy -> LDS -> MFMA: no VALU split at all
//   bwd dZ        : dZ[B,h] = dO[B,n_b] . W_de[T]  (autograd of F.linear) -- the B operand is the
//                   W^T plane image; dO (fp32, read exactly once per column tile) goes global ->
//                   registers of the one wave that owns its 32 rows, split there (no LDS round trip)
//
// Arithmetic identical to gemm.hip PREC_H3: s.x = hi + lo (fp16), a.b = lo.hi + hi.lo + hi.hi
// accumulated in fp32 on v_mfma_f32_32x32x16_f16 in the same k order -- with equal tile shapes the
// two paths agree BIT FOR BIT (tests/test_planes.py).
#include <stdlib.h>

#include "common.h"
#include "planes.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (a first-class vector: HIP's uint4 is a struct)

constexpr int ROWB = 144;            // LDS row: 64 B hi | 64 B lo | 16 B pad (odd number of 16-B slots)
constexpr float SCALE_DO = 1024.0f;  // dO without a published maximum (as gemm.hip)

// EPI_FILTER (Recoder.recommend): nothing of the score tile is stored -- an entry survives only if it
// reaches its row's threshold (a lower bound of the row's k-th best score, from a strided sample of
// the catalogue) and is not a seen item; survivors are appended to the row's candidate list
enum { EPI_STORE = 0, EPI_LOSS_MSE = 1, EPI_LOSS_BCE = 3, EPI_FILTER = 4 };

__device__ __forceinline__ void publish_amax(int32_t *counts, int slot, float v) {
  atomicMax(reinterpret_cast<unsigned int *>(counts) + 8 + (slot & 63), __float_as_uint(v));
}

// (force-inlined helpers with compile-time trip counts: a _Pragma("unroll") loop inside a macro that
// is expanded inside ANOTHER macro was not unrolled, and the register ring went to scratch memory)
template <int N>
__device__ __forceinline__ void ld_pieces(u32x4 (&r)[N], const char *const (&src)[N], const int64_t off) {
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = *reinterpret_cast<const u32x4 *>(src[i] + off);
}
template <int N>
__device__ __forceinline__ void st_pieces(char *base, const int (&dst)[N], const u32x4 (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) *reinterpret_cast<u32x4 *>(base + dst[i]) = r[i];
}

template <int TOTAL, int N>
__device__ __forceinline__ void st_pieces_n(char *base, const int (&dst)[N], const u32x4 (&r)[N], const int tid) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (TOTAL % 256 == 0 || tid + 256 * i < TOTAL) *reinterpret_cast<u32x4 *>(base + dst[i]) = r[i];
}
// this lane's A fragments of one 32-deep k-tile: 8 consecutive k per k-step = 2 x 16 bytes
__device__ __forceinline__ void ld_a4(f32x4 (&r)[4], const float *src) {
#pragma unroll
  for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const f32x4 *>(src + (u >> 1) * 16 + (u & 1) * 4);
}

// rk_planes_probe(buffer): every workgroup of the two contraction kernels records wall_clock64() at
// entry, after the prologue (first tile staged), after the k-loop and after the epilogue, plus its
// tile index (tools/probes/planes_phase_probe.py).  One uniform branch per stamp.
unsigned long long *g_probe = nullptr;
#define RK_STAMP(k)                                                                  \
  do {                                                                               \
    if (p.probe && threadIdx.x == 0) p.probe[(size_t)L * 8 + (k)] = wall_clock64();  \
  } while (0)

struct DecP {
  unsigned long long *probe;
  const char *zp, *wp;        // A / B plane images (row pitch KT * 128 bytes)
  const float *scales;        // [0] scale of Z, [1] scale of W
  int KT;
  int M;                      // rows (B)
  int n_cap;
  const int32_t *Ndev;        // n_t on the device
  // store epilogue
  float *C;
  int ldc;                    // <= 0: read from ld_dev
  const int32_t *ld_dev;
  const float *bias;
  const int32_t *bidx;        // bias index = bidx[n] (gathered decoder bias)
  // loss epilogue
  rk_block_t blk;
  int row_off;
  float confidence, inv_B;
  float *loss_part;
  float *gb_part;
  // fused dZ (DZT > 0): the W^T image, the slabs [column tile][M][h] and h
  const char *wtp;
  float *dz_ws;
  int h;
  // filter epilogue: column n is item col_off + n; blk = the users' INPUT block over the whole
  // catalogue (seen items), has_seen == 0: nothing is masked
  const float *thr;           // [M]
  float *cand_val;            // [M][cand_cap]
  int32_t *cand_idx;          // [M][cand_cap] item ids
  int32_t *cand_cnt;          // [M] survivors seen (may exceed cand_cap: the caller checks)
  int cand_cap, col_off, has_seen;
};

// 4 waves as 2 x 2, wave tile (TM*32) x (TN*32): BM = 64*TM, BN = 64*TN.  BK = 32, two LDS stages,
// register prefetch of the next k-tile, ONE barrier per k-tile, every LDS read of a tile ahead of
// its MFMAs (tools/probes/presplit_gemm.hip: 15 us at the C2 shape against 24 for the in-loop split).
// PLAIN: bf16 images, ONE product (RK_GEMM_PREC=bf16: the separate bf16 data point)
// DZT > 0 (64 x 128 tiles, loss epilogues): dZ FUSED -- the workgroup keeps its dO tile (fp32, in LDS),
// cuts it into fp16 pairs with the TILE's own power-of-two scale (its maximum is known here; the
// stand-alone dZ kernel has to take the launch-wide one the loss epilogues publish) and multiplies it
// with the tile's 128 items of the W^T image: dZ partial [64 rows, h] of column tile nt -> slab nt of
// the split-K workspace, summed by rk_splitk_reduce like the stand-alone kernel's K slabs (62 against
// 64 at C2).  One launch, one pass over dO and one cross-queue edge less.  DZT = ceil(h / 32) <= 8.
template <int TM, int TN, int EPI, int RD = 3, bool PLAIN = false, int DZT = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void decode_planes_kernel(DecP p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  static_assert(DZT == 0 || (TM == 1 && TN == 2 && !PLAIN && (EPI == EPI_LOSS_MSE || EPI == EPI_LOSS_BCE)),
                "the fused dZ rides on the 64 x 128 loss tiles");
  constexpr int DT_LD = 132;                              // floats per row of the dO tile in LDS
  constexpr int DT_OFF = 256 * ROWB;                      // its byte offset (behind the W^T stage: <= 256 rows)
  constexpr int B2_ROWS = 32 * (DZT > 0 ? DZT : 1);       // rows of a W^T k-tile staged (>= kp_of(h))
  constexpr int B2_PT = (B2_ROWS * 8 + 255) / 256;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int A_PT = BM / 32, B_PT = BN / 32;          // 16-byte pieces per thread and k-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = blockIdx.x;
  const int M = p.M, N = *p.Ndev;
  // XCD-aware tile order (gemm.hip): workgroup L runs on XCD L % 8 and takes a contiguous chunk of
  // the LIVE tile list; consecutive tiles share the W panel (mt fastest)
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  const int total = tm * tn;
  const int chunk = (total + 7) >> 3;
  const int t = (L & 7) * chunk + (L >> 3);
  if ((L >> 3) >= chunk || t >= total) return;
  const int mt = t % tm, nt = t / tm;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  RK_STAMP(0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t pitch = (int64_t)p.KT * rkp::LINE;
  const char *srcA[A_PT], *srcB[B_PT];
  int dstA[A_PT], dstB[B_PT];
#pragma unroll
  for (int i = 0; i < A_PT; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, piece = idx & 7;
    srcA[i] = p.zp + (int64_t)min(m0 + row, M - 1) * pitch + piece * 16;
    dstA[i] = row * ROWB + piece * 16;
  }
#pragma unroll
  for (int i = 0; i < B_PT; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, piece = idx & 7;
    srcB[i] = p.wp + (int64_t)min(n0 + row, N - 1) * pitch + piece * 16;
    dstB[i] = BM * ROWB + row * ROWB + piece * 16;
  }
  // register ring, RD k-tiles deep: K = h is SHORT (7 k-tiles at h = 200) and a tile's loads take
  // 1-2 us to come back from L2 -- with one tile in flight the k-loop was 7 round trips long
  u32x4 ra0[A_PT], ra1[A_PT], ra2[A_PT], rb0[B_PT], rb1[B_PT], rb2[B_PT];
#define GLOAD(slot, kt)                                                                  \
  ld_pieces(ra##slot, srcA, (int64_t)(kt) * rkp::LINE);                                  \
  ld_pieces(rb##slot, srcB, (int64_t)(kt) * rkp::LINE);
#define SSTORE(buf, slot)                                                                \
  st_pieces(smem + (buf) * STAGE, dstA, ra##slot);                                       \
  st_pieces(smem + (buf) * STAGE, dstB, rb##slot);

  const int KT = p.KT;
  GLOAD(0, 0);
  GLOAD(1, min(1, KT - 1));
  GLOAD(2, min(2, KT - 1));
  // Loss epilogue operands (gathered bias, bitmap words) are fetched NOW, behind the first tiles'
  // loads, so that their dependent round trips overlap the k-loop instead of the epilogue.  All
  // gather indices first, then all values: independent loads (a `bidx ? bidx[n] : n` select per
  // element made hipcc branch and wait vmcnt(0) for every one of them: 16 serial round trips).
  constexpr bool LOSS = (EPI == EPI_LOSS_MSE || EPI == EPI_LOSS_BCE);
  float pre_bv[LOSS ? TN : 1][4];
  uint32_t pre_w[LOSS ? TM : 1][LOSS ? TN : 1][4];
  if (LOSS) {
    int gi[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        gi[j][e] = p.bidx[min(n0 + (wn * TN + j) * 32 + (lane & 7) * 4 + e, N - 1)];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wn * TN + j) * 32;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + (wm * TM + i) * 32 + (lane >> 3) + 8 * it;
          const int row = p.row_off + min(m, M - 1);
          pre_w[i][j][it] = p.blk.bits_rc[(int64_t)row * p.blk.ldw_rc + min(nb >> 5, p.blk.ldw_rc - 1)];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) pre_bv[j][e] = p.bias[gi[j][e]];
  }
  SSTORE(0, 0);
  __syncthreads();
  RK_STAMP(1);
  const int a_off = ((wm * TM) * 32 + l31) * ROWB + lh * 16;
  const int b_off = BM * ROWB + ((wn * TN) * 32 + l31) * ROWB + lh * 16;
  // tile kt sits in ring slot kt % RD until it is stored into LDS stage kt & 1 (at the end of
  // iteration kt - 1); iteration kt refills that slot with tile kt + RD (clamped: never stored)
  static_assert(RD == 3, "the k-loop below is unrolled by hand for a ring of 3");
#define KTILE(U, UN)                                                                    \
  if (kt0 + (U) < KT) {                                                                  \
    const int kt = kt0 + (U);                                                            \
    GLOAD(U, min(kt + RD, KT - 1));                                                     \
    const char *S = smem + (kt & 1) * STAGE;                                            \
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];                                   \
    _Pragma("unroll")                                                                   \
    for (int ks = 0; ks < 2; ++ks) {                                                    \
    _Pragma("unroll")                                                                   \
      for (int i = 0; i < TM; ++i) {                                                    \
        const char *q = S + a_off + i * 32 * ROWB + ks * 32;                            \
        ah[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));     \
        al[ks][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64)); \
      }                                                                                 \
    _Pragma("unroll")                                                                   \
      for (int j = 0; j < TN; ++j) {                                                    \
        const char *q = S + b_off + j * 32 * ROWB + ks * 32;                            \
        bh[ks][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));     \
        bl[ks][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64)); \
      }                                                                                 \
    }                                                                                   \
    if (PLAIN) {                                                                        \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                        \
                __builtin_bit_cast(bf16x8, ah[ks][i]), __builtin_bit_cast(bf16x8, bh[ks][j]), acc[i][j], 0, 0, 0); \
    } else {                                                                            \
    _Pragma("unroll")                                                                   \
    for (int ks = 0; ks < 2; ++ks) {                                                    \
    _Pragma("unroll")                                                                   \
      for (int i = 0; i < TM; ++i)                                                      \
    _Pragma("unroll")                                                                   \
        for (int j = 0; j < TN; ++j)                                                    \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll")                                                                   \
      for (int i = 0; i < TM; ++i)                                                      \
    _Pragma("unroll")                                                                   \
        for (int j = 0; j < TN; ++j)                                                    \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll")                                                                   \
      for (int i = 0; i < TM; ++i)                                                      \
    _Pragma("unroll")                                                                   \
        for (int j = 0; j < TN; ++j)                                                    \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0); \
    }                                                                                   \
    }                                                                                   \
    SSTORE((kt + 1) & 1, UN);                                                           \
    __syncthreads();                                                                    \
  }
  for (int kt0 = 0; kt0 < KT; kt0 += RD) {
    KTILE(0, 1)
    KTILE(1, 2)
    KTILE(2, 0)
  }
#undef KTILE
#undef GLOAD
#undef SSTORE

  RK_STAMP(2);
  // fused dZ: the first k-tile of the W^T image (the tile's first 32 items, every hidden unit) is
  // fetched NOW, under the epilogue
  const char *srcB2[B2_PT];
  int dstB2[B2_PT];
  u32x4 rw2[B2_PT];
  const int Hp2 = rkp::kp_of(p.h);
  if (DZT > 0) {
#pragma unroll
    for (int i = 0; i < B2_PT; ++i) {
      const int idx = min(tid + 256 * i, B2_ROWS * 8 - 1), row = idx >> 3, piece = idx & 7;
      srcB2[i] = p.wtp + ((int64_t)(n0 >> 5) * Hp2 + min(row, Hp2 - 1)) * rkp::LINE + piece * 16;
      dstB2[i] = row * ROWB + piece * 16;
    }
    ld_pieces(rw2, srcB2, 0);
  }
  // ------------------------------------------------------------- epilogues (as gemm.hip)
  {
    const float inv = 1.0f / (p.scales[0] * p.scales[1]);       // exact: powers of two
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  constexpr int TLD = 36;
  float *fsm = reinterpret_cast<float *>(smem);
  float *wlds = fsm + wid * (32 * TLD);                     // private to this wave
  const int rr0 = lane >> 3, c4 = lane & 7;                 // row-major role of the lane
  auto transpose_tile = [&](const f32x16 &a, float4 (&v)[4]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) wlds[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + l31] = a[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it)
      v[it] = *reinterpret_cast<const float4 *>(wlds + (rr0 + 8 * it) * TLD + c4 * 4);
    __builtin_amdgcn_wave_barrier();
  };

  if (EPI == EPI_STORE) {
    const int ldc = p.ldc > 0 ? p.ldc : *p.ld_dev;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float4 v[4];
        transpose_tile(acc[i][j], v);
        const int n = n0 + (wn * TN + j) * 32 + c4 * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          int gi[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) gi[e] = p.bidx[min(n + e, N - 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = p.bias[gi[e]];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + (wm * TM + i) * 32 + rr0 + 8 * it;
          if (m < M && n < N) {
            const float o[4] = {v[it].x + bv[0], v[it].y + bv[1], v[it].z + bv[2], v[it].w + bv[3]};
            float *dst = p.C + (int64_t)m * ldc + n;
            if (n + 3 < N && (ldc & 3) == 0) {
              *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < N) dst[e] = o[e];
            }
          }
        }
      }
  } else if (EPI == EPI_FILTER) {
    const rk_block_t &b = p.blk;
    const bool implicit = b.implicit != 0;
    float th[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int it = 0; it < 4; ++it)
        th[i][it] = p.thr[min(m0 + (wm * TM + i) * 32 + rr0 + 8 * it, M - 1)];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + c4 * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.bias[min(n + e, N - 1)];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float4 v[4];
        transpose_tile(acc[i][j], v);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + (wm * TM + i) * 32 + rr0 + 8 * it;
          const float ov[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float o = ov[e] + bv[e];
            if (m < M && n + e < N && o >= th[i][it]) {        // (rare: a few hundred per row in all)
              const int gc = p.col_off + n + e;
              bool seen = false;
              if (p.has_seen) {
                const int row = p.row_off + m;
                const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (gc >> 5)];
                if ((word >> (gc & 31)) & 1u)
                  seen = implicit ? true : (b.vals[rk_entry_index(b, row, gc, word)] > 0.f);
              }
              if (!seen) {
                const int slot = atomicAdd(p.cand_cnt + m, 1);
                if (slot < p.cand_cap) {
                  p.cand_val[(int64_t)m * p.cand_cap + slot] = o;
                  p.cand_idx[(int64_t)m * p.cand_cap + slot] = gc;
                }
              }
            }
          }
        }
      }
    }
  } else {
    float *lred = fsm + 4 * (32 * TLD);           // after the 4 per-wave transpose areas
    float *cpart = lred + 8;                      // [2][BN] column partial sums (one row per 64 rows)
    const int ldc = *p.ld_dev;
    const rk_block_t &b = p.blk;
    const bool implicit = b.implicit != 0;
    float lsum = 0.f, gmax = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wn * TN + j) * 32;     // multiple of 32: one bitmap word per row
      const int n = nb + c4 * 4;
      float bv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = pre_bv[LOSS ? j : 0][e];
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float4 v[4];
        transpose_tile(acc[i][j], v);
        uint32_t w[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) w[it] = pre_w[LOSS ? i : 0][LOSS ? j : 0][it];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + (wm * TM + i) * 32 + rr0 + 8 * it;
          const float ov[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = (m < M) && (n + e < N);
            const float o = ov[e] + bv[e];
            float tv = 0.f;
            if (ok && ((w[it] >> (c4 * 4 + e)) & 1u)) {
              tv = 1.0f;
              if (!implicit) tv = b.vals[rk_entry_index(b, p.row_off + m, n + e, w[it])];
            }
            float l;
            if (EPI == EPI_LOSS_MSE) {
              const float wgt = (tv > 0.f) ? (1.0f + p.confidence) : 1.0f;
              const float d = o - tv;
              l = wgt * (d * d);
              g[e] = (2.0f * d) * (wgt * p.inv_B);
            } else {  // BCE with logits: (1-t)*o - logsigmoid(o)
              const float ls = fminf(o, 0.f) - log1pf(expf(-fabsf(o)));
              l = (1.0f - tv) * o - ls;
              const float sg = 1.0f / (1.0f + expf(-o));
              g[e] = (sg - tv) * p.inv_B;
            }
            if (ok) { lsum += l; cs[e] += g[e]; gmax = fmaxf(gmax, fabsf(g[e])); }
            else g[e] = 0.f;     // padding columns [N, ld) of the dO row are ZEROS (rk_decode_bwd_dz_planes)
          }
          // (the tiles cover [0, ld): ld = round_up(N, 32) and BN is a multiple of 32)
          if (m < M && n < ldc)
            *reinterpret_cast<float4 *>(p.C + (int64_t)m * ldc + n) = make_float4(g[0], g[1], g[2], g[3]);
          if (DZT > 0)      // (rows past M / columns past N hold zeros: g was zeroed above)
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(smem + DT_OFF) +
                                        ((wm * TM + i) * 32 + rr0 + 8 * it) * DT_LD + (wn * TN + j) * 32 + c4 * 4) =
                make_float4(g[0], g[1], g[2], g[3]);
        }
      }
      // column sums over this wave's rows: lanes with equal c4 hold different rows
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] += __shfl_xor(cs[e], 8, 64);
        cs[e] += __shfl_xor(cs[e], 16, 64);
        cs[e] += __shfl_xor(cs[e], 32, 64);
      }
      if (rr0 == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cpart[wm * BN + (wn * TN + j) * 32 + c4 * 4 + e] = cs[e];
      }
    }
    lsum = rk_wave_sum(lsum);
    gmax = rk_wave_max(gmax);
    if (lane == 0) { lred[wid] = lsum; lred[4 + wid] = gmax; }
    __syncthreads();
    if (tid == 0) {
      p.loss_part[t] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
      const float gm = fmaxf(fmaxf(lred[4], lred[5]), fmaxf(lred[6], lred[7]));
      publish_amax(b.counts, L, gm);
    }
    if (p.gb_part && tid < BN) {
      // one gb_part row per 64 rows of dO (rk_decode_row_tile): TM = 1: the two waves along M hold
      // 32 rows each of ONE row group; TM = 2: each of them holds a whole group
      const int n = n0 + tid;
      if (n < N) {
        if (TM == 1) {
          if (m0 < M) p.gb_part[(int64_t)mt * ldc + n] = cpart[tid] + cpart[BN + tid];
        } else {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (m0 + g * 64 < M) p.gb_part[(int64_t)(mt * 2 + g) * ldc + n] = cpart[g * BN + tid];
        }
      }
    }
  }
  if (DZT > 0) {
    // ---------------------------------------------------------------- fused dZ partial of this tile
    RK_STAMP(5);
    constexpr int T2A = DZT > 0 ? (DZT + 1) / 2 : 1;     // column tiles (32 hidden units) per wave
    float *lred2 = fsm + 4 * (32 * TLD);
    const float gm = fmaxf(fmaxf(lred2[4], lred2[5]), fmaxf(lred2[6], lred2[7]));   // the tile's max |dO|
    float s_do = 1.0f;
    if (gm > 0.f) {
      const int e = min(max((int)(__float_as_uint(gm) >> 23) - 127, -100), 100);
      s_do = __uint_as_float((uint32_t)(13 - e + 127) << 23);
    }
    const int nk2 = min(BN / 32, (N - n0 + 31) >> 5);    // live k-tiles (32 items each) of this tile
    const int j0 = wn * T2A, nj = wn == 0 ? T2A : DZT - T2A;
    f32x16 acc2[T2A];
#pragma unroll
    for (int j = 0; j < T2A; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    __syncthreads();                                     // the epilogue's LDS (transposes, sums) is free
    st_pieces_n<B2_ROWS * 8>(smem, dstB2, rw2, tid);
    __syncthreads();
    const float *drow = reinterpret_cast<const float *>(smem + DT_OFF) + (wm * 32 + l31) * DT_LD + lh * 8;
    const int b2_off = l31 * ROWB + lh * 16;
    const int64_t b2_step = (int64_t)Hp2 * rkp::LINE;
    for (int kt = 0; kt < nk2; ++kt) {
      if (kt + 1 < nk2) ld_pieces(rw2, srcB2, (int64_t)(kt + 1) * b2_step);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 a0 = *reinterpret_cast<const float4 *>(drow + kt * 32 + ks * 16);
        const float4 a1 = *reinterpret_cast<const float4 *>(drow + kt * 32 + ks * 16 + 4);
        uint2 h0, l0, h1, l1;
        rkp::split4(a0, s_do, h0, l0);
        rkp::split4(a1, s_do, h1, l1);
        const f16x8 ah = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        const f16x8 al = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
        f16x8 bh[T2A], bl[T2A];
#pragma unroll
        for (int j = 0; j < T2A; ++j) {
          const char *q = smem + b2_off + min(j0 + j, DZT - 1) * 32 * ROWB + ks * 32;
          bh[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));
          bl[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64));
        }
#pragma unroll
        for (int j = 0; j < T2A; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T2A; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < T2A; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc2[j], 0, 0, 0);
      }
      __syncthreads();
      if (kt + 1 < nk2) {
        st_pieces_n<B2_ROWS * 8>(smem, dstB2, rw2, tid);
        __syncthreads();
      }
    }
    // slab store (as dz_planes_kernel): per-wave LDS transpose, 16 bytes per lane
    const float inv2 = 1.0f / (s_do * p.scales[1]);
    float *ws = p.dz_ws + (int64_t)nt * M * p.h;
#pragma unroll
    for (int j = 0; j < T2A; ++j) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) wlds[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + l31] = acc2[j][r] * inv2;
      __builtin_amdgcn_wave_barrier();
      const int n = (j0 + j) * 32 + c4 * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 v = *reinterpret_cast<const float4 *>(wlds + (rr0 + 8 * it) * TLD + c4 * 4);
        const int m = m0 + wm * 32 + rr0 + 8 * it;
        if (j < nj && m < M && n < p.h) *reinterpret_cast<float4 *>(ws + (int64_t)m * p.h + n) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  RK_STAMP(3);
  if (p.probe && threadIdx.x == 0) p.probe[(size_t)L * 8 + 4] = (unsigned long long)t + 1;
}

// ------------------------------------------------------------------------------------ dZ
struct DzP {
  unsigned long long *probe;
  const float *dO;            // [M][ld] fp32 (columns [n_t, ld): anything, masked in the kernel)
  const char *wtp;            // W^T image, k-tile major: line (kt, j) at (kt * Hp + j) * 128
  const float *scales;        // [1] scale of W
  const uint32_t *a_amax;     // 64 slots: running max |dO| (counts + 8)
  const int32_t *counts;      // [0] n_t (= K), [2] ld
  int n_ld;
  int M, N;                   // rows (B), h
  int tiles_n;
  float *ws;                  // [split][M][N] slabs
};

// Workgroup = 4 waves x 32 rows (BM = 128) x BN = 32 * TN columns.  B (the W^T image, shared by the
// four waves) goes through LDS as in the decode; A (dO) belongs to ONE wave per row, so its
// fragments go global -> registers directly (a lane: 8 consecutive k of its row per k-step = two
// 16-byte loads) and are split there -- with one column tile (h <= 32 * TN) every dO element is
// split exactly once on the whole chip.
// RD: k-tiles of register prefetch.  1 by default: next to the dW kernel on the side stream a 3-deep
// ring gained dZ 0.5 us and cost dW 11 (22 -> 33 us: it waits on the same L2 fetch path, and the
// Adam sweep waits for dW); 3 where dZ runs alone.
template <int TN, int RD = 1, bool PLAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void dz_planes_kernel(DzP p) {
  constexpr int BM = 128, BN = 32 * TN;
  constexpr int STAGE = BN * ROWB;
  constexpr int B_PT = (BN * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nsplit = gridDim.y;
  const int L = blockIdx.y * gridDim.x + blockIdx.x;
  const int M = p.M, N = p.N, K = p.counts[0], lda = p.counts[2];
  const float a_scale = PLAIN ? 1.0f : rkp::scale_from(p.a_amax, SCALE_DO);
  const float b_scale = p.scales[1];
  const int tm = (M + BM - 1) / BM, tn = p.tiles_n;
  const int per_split = tm * tn;
  const int total = per_split * nsplit;
  const int chunk = (total + 7) >> 3;
  const int t = (L & 7) * chunk + (L >> 3);
  if ((L >> 3) >= chunk || t >= total) return;
  const int split = t / per_split, rt = t % per_split;
  const int mt = rt % tm, nt = rt / tm;
  const int m0 = mt * BM, n0 = nt * BN;
  // split-K: the chunk follows the device-resident K so that every split is live (as gemm.hip)
  const int kchunk = ((K + nsplit - 1) / nsplit + 31) & ~31;
  const int kbeg = split * kchunk;
  const int kend = min(K, kbeg + kchunk);
  if (kbeg >= kend) return;
  const int nk = (kend - kbeg + 31) >> 5;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  RK_STAMP(0);
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // A: this lane's row, 8 consecutive k per k-step (kbeg is a multiple of 32, lda of 32: 16-byte
  // aligned; the tile never reaches past column ld, whose tail [n_t, ld) holds zeros)
  const float *a_src = p.dO + (int64_t)min(m0 + wid * 32 + l31, M - 1) * lda + kbeg + lh * 8;
  // B: rows n0 .. n0 + BN - 1 of the W^T image, k-tile (kbeg >> 5) + kt
  const char *srcB[B_PT];
  int dstB[B_PT];
#pragma unroll
  for (int i = 0; i < B_PT; ++i) {
    const int idx = min(tid + 256 * i, BN * 8 - 1), row = idx >> 3, piece = idx & 7;
    // (the image has round_up(h, 32) rows; a tile may be wider: clamp -- those columns are never stored)
    srcB[i] = p.wtp + ((int64_t)(kbeg >> 5) * rkp::kp_of(N) + min(n0 + row, rkp::kp_of(N) - 1)) * rkp::LINE + piece * 16;
    dstB[i] = row * ROWB + piece * 16;
  }
  // register ring (RD k-tiles) for both operands; B goes on through two LDS stages, A stays in registers
  u32x4 rb0[B_PT], rb1[B_PT], rb2[B_PT];
  f32x4 ra0[4], ra1[4], ra2[4];
  const int64_t b_step = (int64_t)rkp::kp_of(N) * rkp::LINE;      // one k-tile of the image
#define GLOAD(slot, kt)                                                            \
  ld_pieces(rb##slot, srcB, (int64_t)(kt) * b_step);                               \
  ld_a4(ra##slot, a_src + (kt) * 32);
#define SSTOREB(buf, slot) st_pieces_n<BN * 8>(smem + (buf) * STAGE, dstB, rb##slot, tid);
  GLOAD(0, 0);
  if (RD == 3) {
    GLOAD(1, min(1, nk - 1));
    GLOAD(2, min(2, nk - 1));
  }
  SSTOREB(0, 0);
  __syncthreads();
  RK_STAMP(1);
  const int b_off = l31 * ROWB + lh * 16;
#define KTILE(U, UN, D)                                                                              \
  if (kt0 + (U) < nk) {                                                                             \
    const int kt = kt0 + (U);                                                                       \
    f32x4 ac[4];                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) ac[u] = ra##U[u];                                 \
    /* K tail: the columns [K, ld) of dO may hold anything (stale values of an earlier, wider */   \
    /* block: scaled by THIS block's split scale they can overflow fp16, and inf x 0 = nan) -- */  \
    /* zeroed here, in the last k-tile of the last split only (uniform branch) */                   \
    if (kt == nk - 1 && kbeg + nk * 32 > kend) {                                                    \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                               \
        const int k = kbeg + kt * 32 + (u >> 1) * 16 + lh * 8 + (u & 1) * 4;                        \
        ac[u].x = k + 0 < kend ? ac[u].x : 0.f;                                                     \
        ac[u].y = k + 1 < kend ? ac[u].y : 0.f;                                                     \
        ac[u].z = k + 2 < kend ? ac[u].z : 0.f;                                                     \
        ac[u].w = k + 3 < kend ? ac[u].w : 0.f;                                                     \
      }                                                                                             \
    }                                                                                               \
    f16x8 ah[2], al[2];                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
      uint2 h0, l0, h1, l1;                                                                         \
      if (PLAIN) {                                                                                  \
        rkp::plain4(make_float4(ac[2 * ks].x, ac[2 * ks].y, ac[2 * ks].z, ac[2 * ks].w), h0, l0);   \
        rkp::plain4(make_float4(ac[2 * ks + 1].x, ac[2 * ks + 1].y, ac[2 * ks + 1].z, ac[2 * ks + 1].w), h1, l1); \
      } else {                                                                                      \
      rkp::split4(make_float4(ac[2 * ks].x, ac[2 * ks].y, ac[2 * ks].z, ac[2 * ks].w), a_scale, h0, l0); \
      rkp::split4(make_float4(ac[2 * ks + 1].x, ac[2 * ks + 1].y, ac[2 * ks + 1].z, ac[2 * ks + 1].w), a_scale, h1, l1); \
      }                                                                                             \
      ah[ks] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));                       \
      al[ks] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));                       \
    }                                                                                               \
    GLOAD(U, min(kt + (D), nk - 1));                                                                  \
    const char *S = smem + (kt & 1) * STAGE;                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
      f16x8 bh[TN], bl[TN];                                                                         \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                              \
        const char *q = S + b_off + j * 32 * ROWB + ks * 32;                                        \
        bh[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));                     \
        bl[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64));                \
      }                                                                                             \
      if (PLAIN) {                                                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[ks]),      \
                                                           __builtin_bit_cast(bf16x8, bh[j]), acc[j], 0, 0, 0); \
      } else {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[j], acc[j], 0, 0, 0);            \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[j], acc[j], 0, 0, 0);            \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[j], acc[j], 0, 0, 0);            \
      }                                                                                             \
    }                                                                                               \
    SSTOREB((kt + 1) & 1, UN);                                                                      \
    __syncthreads();                                                                                \
  }
  if (RD == 3) {
    for (int kt0 = 0; kt0 < nk; kt0 += 3) {
      KTILE(0, 1, 3)
      KTILE(1, 2, 3)
      KTILE(2, 0, 3)
    }
  } else {
    for (int kt0 = 0; kt0 < nk; ++kt0) {
      KTILE(0, 0, 1)
    }
  }
#undef KTILE
#undef GLOAD
#undef SSTOREB

  RK_STAMP(2);
  const float inv = 1.0f / (a_scale * b_scale);
  // slab store: a lane holds 16 rows of ONE column; through the per-wave LDS transpose a lane owns
  // 4 x (one row, 4 consecutive columns) and stores 16 bytes at a time
  constexpr int TLD = 36;
  float *wlds = reinterpret_cast<float *>(smem) + wid * (32 * TLD);
  const int rr0 = lane >> 3, c4 = lane & 7;
  float *ws = p.ws + (int64_t)split * M * N;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) wlds[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + l31] = acc[j][r] * inv;
    __builtin_amdgcn_wave_barrier();
    const int n = n0 + j * 32 + c4 * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float4 v = *reinterpret_cast<const float4 *>(wlds + (rr0 + 8 * it) * TLD + c4 * 4);
      const int m = m0 + wid * 32 + rr0 + 8 * it;
      if (m < M && n < N) *reinterpret_cast<float4 *>(ws + (int64_t)m * N + n) = v;   // (N % 4 == 0)
    }
    __builtin_amdgcn_wave_barrier();
  }
  RK_STAMP(3);
  if (p.probe && threadIdx.x == 0) p.probe[(size_t)L * 8 + 4] = (unsigned long long)t + 1;
}

// ---------------------------------------------------------------- stand-alone split passes
__global__ __launch_bounds__(256) void split_w_kernel(rkp::SplitW p) {
  __shared__ __attribute__((aligned(16))) char sm[rkp::SPLIT_W_LDS];
  rkp::split_w_job<256>(p, (int)blockIdx.x, sm);
}

// X[rows, K] fp32 (leading dimension ld) -> its plane image; scale from `amax` (64 slots, nullable)
// or the static default; scales[slot] <- the scale used
__device__ __forceinline__ void split_rows_job(const int block, const float *__restrict__ X, int64_t rows,
                                               int K, int64_t ld, const uint32_t *amax, float dflt,
                                               char *img, int KT, float *scales, int slot, int plain) {
  const float s = plain ? 1.0f : rkp::scale_from(amax, dflt);
  if (block == 0 && threadIdx.x == 0) scales[slot] = s;
  const int q4 = KT * 8;                          // float4 per image row
  const int64_t i = (int64_t)block * 256 + threadIdx.x;
  if (i >= (int64_t)rows * q4) return;
  const int64_t r = i / q4;
  const int k = (int)(i % q4) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v = *reinterpret_cast<const float4 *>(X + r * ld + k);     // (K % 4 == 0)
  rkp::store_split4(img + r * KT * rkp::LINE, k, v, s, plain != 0);
}

__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ X, int64_t rows, int K,
                                                         int64_t ld, const uint32_t *amax, float dflt,
                                                         char *img, int KT, float *scales, int slot,
                                                         int plain) {
  split_rows_job((int)blockIdx.x, X, rows, K, ld, amax, dflt, img, KT, scales, slot, plain);
}

// rk_split_wz: both operand splits of an entry-by-entry sequenced decode in ONE launch (the one-call
// step lets them ride on its encoder forward): workgroups [0, w_tiles) cut W_de[items], the rest Z
// ... and behind them (zt_blocks > 0) Z^T as the fp16 pair planes the dW kernel reads, written where
// rk_decode_bwd_dw2 would make them in a launch of its own: at the head of ITS workspace
struct SplitZt {
  uint16_t *P;
  float *scale_out;
  int rows_pad, cols_pad;
};
__global__ __launch_bounds__(256) void split_wz_kernel(rkp::SplitW p, int w_tiles, const float *__restrict__ Z,
                                                       int B, const uint32_t *zmax, char *zimg, int z_blocks,
                                                       SplitZt zt) {
  __shared__ __attribute__((aligned(16))) char sm[rkp::SPLIT_W_LDS];
  const int blk = (int)blockIdx.x;
  if (blk < w_tiles) {
    rkp::split_w_job<256>(p, blk, sm);
    return;
  }
  if (blk < w_tiles + z_blocks) {
    split_rows_job(blk - w_tiles, Z, B, p.h, p.h, zmax, rkp::SCALE_Z, zimg, p.KT, p.scales, 0, p.plain);
    return;
  }
  rkp::split_zt_pairs_job(blk - w_tiles - z_blocks, Z, B, p.h, p.h, zt.rows_pad, zt.cols_pad, zt.P, zmax,
                          zt.scale_out);
}

inline bool aligned16(const void *q) { return ((uintptr_t)q & 15) == 0; }

template <typename Kern>
int set_lds(Kern k, int bytes) {
  // More than 64 KB of LDS per workgroup has to be asked for -- ONCE per kernel and process, for
  // the most any launch will use (160 KB): calling hipFuncSetAttribute next to another thread's
  // launch of the same kernel faulted the GPU (two virtual ranks in one process).
  static const int rc = [k] {
    return hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
  }();
  (void)bytes;
  return rc;
}

// decode tile shape: by the problem's size unless RK_DEC_TILE / rk_planes_tile force 64 (64 x 128,
// the tile shape of gemm.hip's decode: bit-identical results) or 128 (128 x 128)
// 0: by shape; 1 / 2: forced (rk_tune RK_TUNE_PLANES_TILE = 64 / 128)
inline int dec_tm_forced() {
  const int rows = rk_tune_get(RK_TUNE_PLANES_TILE);
  return rows == 64 ? 1 : (rows == 128 ? 2 : 0);
}
inline int dec_tm(int B, int n_cap) {
  if (dec_tm_forced() > 0) return dec_tm_forced();
  // 128 x 128 tiles for large batches; below that 64 x 128 (two workgroups per CU): C2 (B = 500,
  // n_b ~ 7.9 k: one wave of either tiling) 20.4 vs 22.4 us.  (n_b only exists on the device.)
  // (long item sets -- C5: 48.8 k sampled items of 1 M -- take the 128 x 128 tile at any batch size:
  // 122 vs 154 us, its W panel is re-read half as often)
  return (B >= 1024 || n_cap >= 32768) ? 2 : 1;
}

}  // namespace

// ------------------------------------------------------------------------------- C ABI
static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" void rk_planes_probe(unsigned long long *buffer) { g_probe = buffer; }


extern "C" int64_t rk_planes_bytes(int32_t B_cap, int32_t h, int32_t n_cap) {
  // (image rows in whole groups of 32: csrc/pgemm.h reads an image along its rows in 32-row k-tiles)
  const int64_t KT = rkp::kp_of(h) / 32, n_ld = rkp::kp_of(n_cap);
  return 256 + align256((int64_t)rkp::kp_of(B_cap) * KT * rkp::LINE) + align256(n_ld * KT * rkp::LINE) +
         align256((int64_t)rkp::kp_of(h) * (n_ld / 32) * rkp::LINE);
}

extern "C" int rk_planes_layout(void *buffer, int32_t B_cap, int32_t h, int32_t n_cap, rk_planes_t *out) {
  RK_REQUIRE(buffer != nullptr && out != nullptr && (((uintptr_t)buffer) & 255) == 0, "buffer: 256-byte aligned");
  RK_REQUIRE(h > 0 && h % 4 == 0, "h must be a multiple of 4");
  const int64_t KT = rkp::kp_of(h) / 32, n_ld = rkp::kp_of(n_cap);
  char *b = (char *)buffer;
  out->scales = (float *)b;
  out->z = b + 256;
  out->w = (char *)out->z + align256((int64_t)rkp::kp_of(B_cap) * KT * rkp::LINE);
  out->wt = (char *)out->w + align256(n_ld * KT * rkp::LINE);
  out->h = h; out->B_cap = B_cap; out->n_cap = n_cap; out->n_ld = (int32_t)n_ld;
  return 0;
}

static rkp::SplitW split_w_args(const float *W_de, const rk_block_t *tgt, const int32_t *ranges,
                                const rk_planes_t *pl) {
  rkp::SplitW s = {};
  s.W = W_de; s.items = tgt->items; s.counts = tgt->counts;
  s.amax = ranges ? reinterpret_cast<const uint32_t *>(ranges) + 64 : nullptr;
  s.wp = (char *)pl->w; s.wtp = (char *)pl->wt; s.scales = pl->scales;
  s.h = pl->h; s.KT = rkp::kp_of(pl->h) / 32; s.n_ld = pl->n_ld;
  s.plain = rk_gemm_plain_bf16();
  return s;
}

rkp::SplitW rk_split_w_args(const float *W_de, const rk_block_t *tgt, const int32_t *ranges,
                            const rk_planes_t *pl) {
  return split_w_args(W_de, tgt, ranges, pl);
}

extern "C" int rk_split_w(const float *W_de, int32_t h, const rk_block_t *tgt, const int32_t *ranges,
                          const rk_planes_t *pl, void *stream_) {
  RK_REQUIRE(pl && pl->h == h && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(aligned16(W_de), "W_de must be 16-byte aligned");
  const rkp::SplitW s = split_w_args(W_de, tgt, ranges, pl);
  RK_LAUNCH(split_w_kernel, dim3(rk_cdiv(tgt->n_cap, 32)), dim3(256), 0, (hipStream_t)stream_, s);
  RK_CHECK_LAUNCH("split_w");
  return 0;
}

extern "C" int32_t rk_split_zt_ok(void) { return (rk_dw_pairs() && !rk_gemm_plain_bf16()) ? 1 : 0; }

extern "C" int rk_split_wz(const float *W_de, const float *Z, int32_t B, int32_t h, const rk_block_t *tgt,
                           const int32_t *ranges, const rk_planes_t *pl, void *dw_workspace,
                           void *stream_) {
  if (Z == nullptr) return W_de ? rk_split_w(W_de, h, tgt, ranges, pl, stream_) : 0;
  RK_REQUIRE(pl && pl->h == h && tgt->n_cap <= pl->n_cap && B <= pl->B_cap,
             "planes were laid out for another shape");
  RK_REQUIRE(aligned16(W_de) && aligned16(Z), "W_de and Z must be 16-byte aligned");
  RK_REQUIRE(dw_workspace == nullptr || (rk_split_zt_ok() && (((uintptr_t)dw_workspace) & 255) == 0),
             "Z^T pair planes: fp16-pair dW only (rk_split_zt_ok), 256-byte aligned workspace");
  if (B == 0) return W_de ? rk_split_w(W_de, h, tgt, ranges, pl, stream_) : 0;
  const rkp::SplitW s = split_w_args(W_de, tgt, ranges, pl);
  // (W_de == NULL: the W images are already there -- rk_ae_encode_fwd_split_w -- only Z is cut here)
  const int w_tiles = W_de ? rk_cdiv(tgt->n_cap, 32) : 0;
  const int z_blocks = rk_cdiv((int64_t)B * s.KT * 8, 256);
  SplitZt zt = {};
  int zt_blocks = 0;
  if (dw_workspace) {
    zt.rows_pad = rk_dw3_rows_pad(B); zt.cols_pad = rk_dw3_cols_pad(h);
    zt.P = (uint16_t *)dw_workspace;
    zt.scale_out = reinterpret_cast<float *>((char *)dw_workspace + (int64_t)2 * zt.rows_pad * zt.cols_pad * 2);
    zt_blocks = rk_cdiv((int64_t)(zt.rows_pad >> 3) * zt.cols_pad, 256);
  }
  RK_LAUNCH(split_wz_kernel, dim3(w_tiles + z_blocks + zt_blocks), dim3(256), 0, (hipStream_t)stream_, s,
            w_tiles, Z, B, reinterpret_cast<const uint32_t *>(ranges), (char *)pl->z, z_blocks, zt);
  RK_CHECK_LAUNCH("split_wz");
  return 0;
}

extern "C" int rk_split_z(const float *Z, int32_t B, int32_t h, const int32_t *ranges,
                          const rk_planes_t *pl, void *stream_) {
  RK_REQUIRE(pl && pl->h == h && B <= pl->B_cap, "planes were laid out for another shape");
  RK_REQUIRE(aligned16(Z), "Z must be 16-byte aligned");
  if (B == 0) return 0;
  const int KT = rkp::kp_of(h) / 32;
  RK_LAUNCH(split_rows_kernel, dim3(rk_cdiv((int64_t)B * KT * 8, 256)), dim3(256), 0, (hipStream_t)stream_,
            Z, B, h, h, reinterpret_cast<const uint32_t *>(ranges), rkp::SCALE_Z, (char *)pl->z, KT,
            pl->scales, 0, (int)rk_gemm_plain_bf16());
  RK_CHECK_LAUNCH("split_z");
  return 0;
}

extern "C" int rk_decode_loss_planes(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt,
                                     int32_t row_off, const float *b_de, int32_t loss_kind,
                                     float confidence, float inv_B, float *dO, int32_t ld_out,
                                     float *loss_part, float *gb_part, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && B <= pl->B_cap && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  if (B == 0) return 0;
  DecP p = {};
  p.probe = g_probe;
  p.zp = (const char *)pl->z; p.wp = (const char *)pl->w; p.scales = pl->scales;
  p.KT = rkp::kp_of(pl->h) / 32;
  p.M = B; p.n_cap = tgt->n_cap; p.Ndev = tgt->counts;
  p.C = dO; p.bias = b_de; p.bidx = tgt->items;
  p.blk = *tgt; p.row_off = row_off; p.confidence = confidence; p.inv_B = inv_B;
  p.loss_part = loss_part; p.gb_part = gb_part;
  const bool loss = loss_kind == RK_LOSS_MSE || loss_kind == RK_LOSS_BCE;
  if (loss) {
    RK_REQUIRE(tgt->implicit || tgt->pref_rc != nullptr, "explicit values need pref_rc");
    p.ld_dev = tgt->counts + 2;
  } else if (loss_kind == RK_LOSS_MNLL) {
    p.ldc = 0; p.ld_dev = tgt->counts + 2;
  } else {
    RK_REQUIRE(ld_out > 0, "ld_out");
    p.ldc = ld_out;
  }
  const int tm = dec_tm(B, tgt->n_cap);
  const int BM = 64 * tm, BN = 128;
  const int grid = rk_cdiv(rk_cdiv(B, BM) * rk_cdiv(tgt->n_cap, BN), 8) * 8;
  const int lds = 2 * (BM + BN) * ROWB;
#define LAUNCH(TM, EPI)                                                                        \
  do {                                                                                         \
    if (rk_gemm_plain_bf16()) {                                                                \
      if (set_lds(decode_planes_kernel<TM, 2, EPI, 3, true>, lds)) { rk_set_error("LDS attribute"); return -1; } \
      RK_LAUNCH((decode_planes_kernel<TM, 2, EPI, 3, true>), dim3(grid), dim3(256), lds, stream, p); \
    } else {                                                                                   \
      if (set_lds(decode_planes_kernel<TM, 2, EPI>, lds)) { rk_set_error("LDS attribute"); return -1; } \
      RK_LAUNCH((decode_planes_kernel<TM, 2, EPI>), dim3(grid), dim3(256), lds, stream, p);    \
    }                                                                                          \
  } while (0)
  if (tm == 2) {
    if (loss_kind == RK_LOSS_MSE) LAUNCH(2, EPI_LOSS_MSE);
    else if (loss_kind == RK_LOSS_BCE) LAUNCH(2, EPI_LOSS_BCE);
    else LAUNCH(2, EPI_STORE);
  } else {
    if (loss_kind == RK_LOSS_MSE) LAUNCH(1, EPI_LOSS_MSE);
    else if (loss_kind == RK_LOSS_BCE) LAUNCH(1, EPI_LOSS_BCE);
    else LAUNCH(1, EPI_STORE);
  }
#undef LAUNCH
  RK_CHECK_LAUNCH("decode_loss_planes");
  return 0;
}

// X[rows, K] fp32 (leading dimension ld) -> its plane image [rows][kp_of(K) / 32 lines]; the scale from
// `amax` (64 slots of fp32 bit patterns, nullable: dflt_scale), published in scales[slot]
extern "C" int rk_split_image(const float *X, int64_t rows, int32_t K, int64_t ld, const int32_t *amax,
                              float dflt_scale, void *image, float *scales, int32_t slot, void *stream_) {
  RK_REQUIRE(aligned16(X) && aligned16(image) && K % 4 == 0 && ld % 4 == 0, "16-byte aligned operands, K % 4 == 0");
  RK_REQUIRE(slot == 0 || slot == 1, "slot is 0 (Z) or 1 (W)");
  RK_REQUIRE(rows * (int64_t)(rkp::kp_of(K) / 32) * 8 < ((int64_t)1 << 31) * 256, "image too large for one launch");
  if (rows == 0) return 0;
  const int KT = rkp::kp_of(K) / 32;
  RK_LAUNCH(split_rows_kernel, dim3((unsigned)rk_cdiv(rows * KT * 8, 256)), dim3(256), 0, (hipStream_t)stream_,
            X, rows, K, ld, reinterpret_cast<const uint32_t *>(amax), dflt_scale, (char *)image, KT, scales,
            slot, (int)rk_gemm_plain_bf16());
  RK_CHECK_LAUNCH("split_image");
  return 0;
}

// Recoder.recommend's decode over a strip of the catalogue with the top-k FILTER fused (EPI_FILTER):
// zimg [B][KT], wimg [n][KT] (rows = items col_off .. col_off + n), scales {Z, W}
extern "C" int rk_decode_filter_planes(const void *zimg, const void *wimg, const float *scales, int32_t h,
                                       int32_t B, int32_t n, int32_t col_off, const float *b_de,
                                       const rk_block_t *seen, int32_t row_off, const float *thr,
                                       float *cand_val, int32_t *cand_idx, int32_t *cand_cnt,
                                       int32_t cand_cap, const int32_t *n_dev, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(zimg && wimg && scales && thr && cand_val && cand_idx && cand_cnt && n_dev, "null argument");
  RK_REQUIRE(seen == nullptr || (row_off >= 0 && row_off + B <= seen->S_cap && seen->bits_rc != nullptr &&
                                 (seen->implicit || seen->pref_rc != nullptr) &&
                                 (int64_t)seen->ldw_rc * 32 >= col_off + n),
             "seen block: rows / bitmap do not cover the strip");
  if (B == 0 || n == 0) return 0;
  DecP p = {};
  p.zp = (const char *)zimg; p.wp = (const char *)wimg; p.scales = scales;
  p.KT = rkp::kp_of(h) / 32;
  p.M = B; p.n_cap = n; p.Ndev = n_dev;
  p.bias = b_de ? b_de + col_off : nullptr;
  if (seen) { p.blk = *seen; p.has_seen = 1; }
  p.row_off = row_off;
  p.thr = thr; p.cand_val = cand_val; p.cand_idx = cand_idx; p.cand_cnt = cand_cnt;
  p.cand_cap = cand_cap; p.col_off = col_off;
  const int tm = dec_tm(B, n);
  const int BM = 64 * tm, BN = 128;
  const int grid = rk_cdiv(rk_cdiv(B, BM) * rk_cdiv(n, BN), 8) * 8;
  const int lds = 2 * (BM + BN) * ROWB;
  if (tm == 2) {
    if (set_lds(decode_planes_kernel<2, 2, EPI_FILTER>, lds)) { rk_set_error("LDS attribute"); return -1; }
    RK_LAUNCH((decode_planes_kernel<2, 2, EPI_FILTER>), dim3(grid), dim3(256), lds, stream, p);
  } else {
    if (set_lds(decode_planes_kernel<1, 2, EPI_FILTER>, lds)) { rk_set_error("LDS attribute"); return -1; }
    RK_LAUNCH((decode_planes_kernel<1, 2, EPI_FILTER>), dim3(grid), dim3(256), lds, stream, p);
  }
  RK_CHECK_LAUNCH("decode_filter_planes");
  return 0;
}

// ---- decode + loss with the dZ partials of every 128-item column tile fused in (DZT > 0 above) ----
int rk_splitk_reduce_tiles(const float *ws, int M, int N, const int32_t *Kdev, int max_splits, int tile_k,
                           const float *Zact, int act, float *out, void *stream);

// rk_tune(RK_TUNE_DZ_FUSED, 0): the stand-alone dZ kernel everywhere (A/B switch)
static int dz_fused_on() { return rk_tune_get(RK_TUNE_DZ_FUSED) != 0; }

// The fused launch always runs 64 x 128 tiles, whatever the block's CAPACITY: the 128-row tile rule of
// the plain decode (dec_tm: n_cap >= 32768) looks at the capacity because the live item count only
// exists on the device, and sent the MSD-big stand-in (250 k items, 10 k of them live per batch) to
// the two-launch form -- 0.138 vs 0.122 ms per step.  Not with a forced 128-row tile (tests), and not
// when the slab workspace (one slab per 128 items of CAPACITY) would pass 4 GB.
// (decided by the capacity and h alone -- priced at the domain's largest batch -- so that every batch
// size of one engine, its ragged last batch included, takes the same form and finds its workspace)
static inline bool dz_fused_slabs_ok(int h, int n_cap) {
  return (int64_t)rk_cdiv(n_cap, 128) * 1023 * h * (int64_t)sizeof(float) <= ((int64_t)4 << 30);
}
extern "C" int64_t rk_dz_fused_workspace_bytes(int32_t B, int32_t h, int32_t n_cap) {
  if (h > 256 || B >= 1024 || !dz_fused_slabs_ok(h, n_cap)) return 0;    // the two-launch form: no workspace
  return (int64_t)rk_cdiv(n_cap, 128) * B * h * sizeof(float);
}

extern "C" int32_t rk_decode_dz_fused_ok(int32_t B, int32_t h, int32_t n_cap, int32_t loss_kind) {
  return dz_fused_on() && !rk_gemm_plain_bf16() && (loss_kind == RK_LOSS_MSE || loss_kind == RK_LOSS_BCE) &&
         h % 4 == 0 && h <= 256 && B < 1024 && dec_tm_forced() != 2 &&
         dz_fused_slabs_ok(h, n_cap) ? 1 : 0;
}

extern "C" int rk_decode_loss_dz_planes(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt,
                                        int32_t row_off, const float *b_de, int32_t loss_kind,
                                        float confidence, float inv_B, float *dO, float *loss_part,
                                        float *gb_part, float *dz_workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && B <= pl->B_cap && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(rk_decode_dz_fused_ok(B, pl->h, tgt->n_cap, loss_kind), "shape / loss outside the fused dZ (rk_decode_dz_fused_ok)");
  RK_REQUIRE(tgt->implicit || tgt->pref_rc != nullptr, "explicit values need pref_rc");
  RK_REQUIRE(aligned16(dz_workspace), "workspace must be 16-byte aligned");
  if (B == 0) return 0;
  DecP p = {};
  p.probe = g_probe;
  p.zp = (const char *)pl->z; p.wp = (const char *)pl->w; p.scales = pl->scales;
  p.KT = rkp::kp_of(pl->h) / 32;
  p.M = B; p.n_cap = tgt->n_cap; p.Ndev = tgt->counts;
  p.C = dO; p.bias = b_de; p.bidx = tgt->items;
  p.blk = *tgt; p.row_off = row_off; p.confidence = confidence; p.inv_B = inv_B;
  p.loss_part = loss_part; p.gb_part = gb_part;
  p.ld_dev = tgt->counts + 2;
  p.wtp = (const char *)pl->wt; p.dz_ws = dz_workspace; p.h = pl->h;
  const int BM = 64, BN = 128;
  const int grid = rk_cdiv(rk_cdiv(B, BM) * rk_cdiv(tgt->n_cap, BN), 8) * 8;
  const int lds = 256 * ROWB + 64 * 132 * 4;            // W^T stage | dO tile (the k-loop's stages fit below)
  const int dzt = rk_cdiv(pl->h, 32);
#define LAUNCH(EPI, DZT)                                                                          \
  do {                                                                                            \
    if (set_lds(decode_planes_kernel<1, 2, EPI, 3, false, DZT>, lds)) { rk_set_error("LDS attribute"); return -1; } \
    RK_LAUNCH((decode_planes_kernel<1, 2, EPI, 3, false, DZT>), dim3(grid), dim3(256), lds, stream, p); \
  } while (0)
#define BY_H(EPI)                                                                                 \
  do {                                                                                            \
    if (dzt <= 2) LAUNCH(EPI, 2); else if (dzt <= 4) LAUNCH(EPI, 4); else if (dzt <= 7) LAUNCH(EPI, 7); \
    else LAUNCH(EPI, 8);                                                                          \
  } while (0)
  if (loss_kind == RK_LOSS_MSE) BY_H(EPI_LOSS_MSE); else BY_H(EPI_LOSS_BCE);
#undef BY_H
#undef LAUNCH
  RK_CHECK_LAUNCH("decode_loss_dz_planes");
  return 0;
}

// dZ = sum of the column-tile slabs rk_decode_loss_dz_planes wrote (* act'(Zact) if given)
extern "C" int rk_decode_dz_reduce(const float *dz_workspace, int32_t B, int32_t h, const rk_block_t *tgt,
                                   const float *Zact, int32_t act, float *dZ, void *stream_) {
  RK_REQUIRE(aligned16(dz_workspace) && aligned16(dZ), "operands must be 16-byte aligned");
  if (B == 0) return 0;
  return rk_splitk_reduce_tiles(dz_workspace, B, h, tgt->counts, rk_cdiv(tgt->n_cap, 128), 128, Zact, act, dZ,
                                stream_);
}

// the split-K reduce of gemm.hip (ws[split][M][N] -> out, * act'(Zact) if given)
int rk_splitk_reduce(const float *ws, int M, int N, const int32_t *Kdev, int splits, const float *Zact,
                     int act, float *out, void *stream);
int rk_dz_splits(int B);

extern "C" int rk_decode_bwd_dz_planes(const float *dO, int32_t B, const rk_planes_t *pl,
                                       const rk_block_t *tgt, const float *Zact, int32_t act, float *dZ,
                                       float *workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(aligned16(dO) && aligned16(workspace) && aligned16(dZ), "operands must be 16-byte aligned");
  if (B == 0) return 0;
  const int h = pl->h;
  DzP p = {};
  p.probe = g_probe;
  p.dO = dO; p.wtp = (const char *)pl->wt; p.scales = pl->scales;
  p.a_amax = reinterpret_cast<const uint32_t *>(tgt->counts) + 8;
  p.counts = tgt->counts; p.n_ld = pl->n_ld;
  p.M = B; p.N = h; p.ws = workspace;
  const int splits = rk_dz_splits(B);
  // one column tile up to h = 224 (dO read and split once); wider h: 128-column tiles with two
  // resident workgroups per CU (C5, h = 512: 190 vs 282 us for 256-column tiles with one)
  int tn = h <= 64 ? 2 : (h <= 128 ? 4 : (h <= 224 ? 7 : 4));
  {
    const int tn_env = rk_tune_get(RK_TUNE_DZ_TN);   // (tuning)
    if (tn_env == 2 || tn_env == 4 || tn_env == 7 || tn_env == 8) tn = tn_env;
  }
  p.tiles_n = rk_cdiv(h, 32 * tn);
  const int tiles = rk_cdiv(B, 128) * p.tiles_n;
  const int lds = 2 * 32 * tn * ROWB;
#define LAUNCH(TN)                                                                             \
  do {                                                                                         \
    if (rk_gemm_plain_bf16()) {                                                                \
      if (set_lds(dz_planes_kernel<TN, 1, true>, lds)) { rk_set_error("LDS attribute"); return -1; } \
      RK_LAUNCH((dz_planes_kernel<TN, 1, true>), dim3(tiles, splits), dim3(256), lds, stream, p); \
    } else {                                                                                   \
      if (set_lds(dz_planes_kernel<TN>, lds)) { rk_set_error("LDS attribute"); return -1; }    \
      RK_LAUNCH((dz_planes_kernel<TN>), dim3(tiles, splits), dim3(256), lds, stream, p);       \
    }                                                                                          \
  } while (0)
  if (tn == 2) LAUNCH(2); else if (tn == 4) LAUNCH(4); else if (tn == 7) LAUNCH(7); else LAUNCH(8);
#undef LAUNCH
  RK_CHECK_LAUNCH("decode_bwd_dz_planes");
  return rk_splitk_reduce(workspace, B, h, tgt->counts, splits, Zact, act, dZ, stream_);
}
