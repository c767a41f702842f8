// Decoder-weight gradient  G[n_t,h] = dO^T . Z   (autograd of F.linear(z, W_de[T]), reference
// nn.py:280) on the 16-bit matrix pipe, in fp32 accuracy, with NO operand range:
//
//   every fp32 operand x is cut into THREE bf16 pieces, x = hi + mid + lo exactly (8 + 8 + 8
//   significant bits, round to nearest at each level, the residuals are exact in fp32; bf16 has
//   fp32's exponent, so there is no scale to choose and nothing to overflow), and a.b is
//   accumulated in fp32 from the six products  lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi
//   on v_mfma_f32_32x32x16_bf16.  The three dropped products are <= 2^-23 |a.b| together and
//   zero-mean -- the size of ONE fp32 rounding.
//
// Data movement (what the fp32-MFMA dW kernel spent its time on):
//   * dO [B, ld] fp32 is the big operand (read once): its tiles are staged through LDS as they
//     are; the transposition dO^T needs happens for free in the fragment read (a lane reads 8
//     k-rows of ONE item column) and the bf16 split is done in registers by the consuming wave.
//   * Z [B, h] is small and re-read by every workgroup: it is split ONCE per step into bf16 planes
//     stored in fragment order -- [plane][k/8][n][8] -- (split_planes_t_launch below) and goes from
//     L2 straight into the registers of the one wave that owns those columns.
//   * What bounds the kernel is what a CU can FETCH: ~290 cache lines (128 B) per us and CU from
//     L2, whatever the path -- an LDS-DMA ring of any depth, register prefetch, a rotated K order
//     and fp32 instead of pre-split operands were all measured at the same line rate; hence full
//     lines everywhere, the 64 x 256 tile (Z^T is re-read once per 64 items), no loads for
//     all-padding columns.
//   * split-K sized ON THE DEVICE from the live item count (counts[0]); the slabs are consumed by
//     rk_adam_multi (g_parts read from counts[4]) or summed by slab_sum3.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "encoder_bwd.h"
#include "planes.h"
#include "splitk_reduce.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void split_pair(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l) {
  rk_split_bf16_pair(a, b, h, m, l);
}

// X[rows, cols] fp32 (row-major, ld) -> bf16 planes [3][rows_pad/8][cols_pad][8] of X^T's
// k-contiguous image: element (k = row, n = col) of plane p sits at ((k/8)*cols_pad + n)*8 + k%8.
// Rows >= rows and columns >= cols are written as zeros (the GEMM relies on it).
// pairs != 0: TWO fp16 planes hi / lo of s.x (s from `amax`: 64 slots, nullable -> the static scale of
// planes.h; *scale_out <- s) instead of three bf16 ones
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float *__restrict__ X, int rows,
                                                             int cols, int ld, int rows_pad,
                                                             int cols_pad, uint16_t *__restrict__ P,
                                                             int pairs, const uint32_t *amax,
                                                             float *scale_out) {
  if (pairs) {
    rkp::split_zt_pairs_job((int)blockIdx.x, X, rows, cols, ld, rows_pad, cols_pad, P, amax, scale_out);
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (chunk, n)
  const int64_t tot = (int64_t)(rows_pad >> 3) * cols_pad;
  if (i >= tot) return;
  const int c8 = (int)(i / cols_pad), n = (int)(i % cols_pad);
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = c8 * 8 + j;
    x[j] = (r < rows && n < cols) ? X[(int64_t)r * ld + n] : 0.f;
  }
  const int64_t plane = (int64_t)rows_pad * cols_pad;               // 16-bit elements
  uint16_t *d = P + i * 8;
  uint4 h, m, l;
  split_pair(x[0], x[1], h.x, m.x, l.x);
  split_pair(x[2], x[3], h.y, m.y, l.y);
  split_pair(x[4], x[5], h.z, m.z, l.z);
  split_pair(x[6], x[7], h.w, m.w, l.w);
  *reinterpret_cast<uint4 *>(d) = h;
  *reinterpret_cast<uint4 *>(d + plane) = m;
  *reinterpret_cast<uint4 *>(d + 2 * plane) = l;
}

struct Dw3P {
  const float *dO;
  int32_t *counts;            // [0] n_t, [2] ld of dO; [4] <- number of K slabs written
  const uint16_t *planes;     // Z^T planes (split_planes_t_kernel)
  int64_t plane_stride;       // bf16 elements per plane
  int cols_pad;               // padded h (multiple of BN)
  int B, Bp, h;
  int tiles_n;                // cols_pad / BN
  int max_splits;
  int wg_slots;               // workgroups the chip holds at once (split-K sizing)
  float *G;                   // nullable: written directly when one slab suffices
  float *slabs;               // [max_splits][slab_stride]
  int64_t slab_stride;        // floats
  unsigned long long *probe;  // tuning probe (null in production): 16 wall-clock stamps per workgroup
  int plain;                  // != 0 (RK_GEMM_PREC=bf16): only the hi . hi product -- plain bf16 operands
  // fp16-pair mode (rk_decode_bwd_dw2): power-of-two split scales
  const uint32_t *a_amax;     // 64 slots: running max |dO| (counts + 8)
  const float *z_scale_dev;   // scale the Z^T planes were written with (device), or null:
  float z_scale;              //   this host value
};

unsigned long long *g_dw3_probe = nullptr;


}  // namespace

// number of K slabs for n_t live items: fill the chip once, never less than 64 rows of K per slab;
// returns the number of slabs that are really written (every one of them has kbeg < Bp)
__host__ __device__ static inline int dw3_splits(int n_t, int tiles_n, int Bp, int max_splits,
                                                 int wg_slots, int *kchunk) {
  const int tiles = ((n_t + 63) >> 6) * tiles_n;
  int s = tiles > 0 ? wg_slots / tiles : 1;
  const int kmax = Bp >> 6;
  if (s > kmax) s = kmax;
  if (s > max_splits) s = max_splits;
  if (s < 1) s = 1;
  const int kc = (((Bp + s - 1) / s) + 31) & ~31;
  if (kchunk) *kchunk = kc;
  return (Bp + kc - 1) / kc;
}

namespace {

// BN = 256: the 8 waves sit side by side along N (wave tile 64 items x 32 columns, two 32 x 32
// accumulators); BN = 128: 2 x 4 waves (wave tile 32 x 32).
//   A (dO tile, shared by every wave): global -> registers -> LDS as stored (fp32, [k][item]).
//     The bf16 split costs ~500 cycles per fragment (12 v_cvt_pk + 24 other VALU per lane), so it
//     is done ONCE per workgroup: in a pass of its own every thread takes 4 k-values of one item
//     column (ds_read_b32: the transposition dO^T needs is this strided read, conflict free),
//     splits them and writes the three 8-byte half-fragments into a fragment buffer
//     [m-tile][k-step][plane][lane] x 16 B, from which every wave fetches ready MFMA operands with
//     ds_read_b128.  (Each wave converting its own fragments -- 8x redundant -- ran at 0.85 us per
//     16-deep k-step against 0.35 us of MFMA time.)
//   B (Z^T planes): every wave owns its 32 columns, so its fragments go global -> REGISTERS
//     directly (one 16-byte load per plane and k-step, 512 contiguous bytes per half wave),
//     prefetched P k-steps ahead in a rotating register file -- no LDS round trip, each byte
//     enters the CU once.  (An LDS-DMA ring for B measured ~33 GB/s per CU whatever its depth.)
//   One barrier per 32-deep stage: MFMAs of stage s, conversion of stage s+1 and the raw store of
//   stage s+2 touch three different buffers.
// PLAIN (RK_GEMM_PREC=bf16): only the hi . hi product.  A template parameter, not a run-time branch: a
// uniform `if (p.plain)` in front of the six products cost the default kernel 10 us (22 -> 33: the
// MFMA / load interleave the scheduling barriers pin was gone).
// PAIRS (rk_decode_bwd_dw2, the default of the training step since round 3): the operands are cut into
// fp16 PAIRS s.x = hi + lo like the decode / dZ contractions (scales: dO from the maximum the loss
// kernels publish, Z static or from rk_amax) and THREE products lo.hi + hi.lo + hi.hi are accumulated
// on v_mfma_f32_32x32x16_f16 instead of six bf16 ones: half the MFMAs, two planes instead of three
// to fetch and to convert.
template <int BN, bool PLAIN, bool PAIRS>
__device__ __forceinline__ void dw3_body(const Dw3P &p, const int L) {
  constexpr int BM = 64, R = 4, P = R - 1;              // B: k-steps of 16, P of them prefetched
  constexpr int BKA = 32;                               // A: LDS stages of two k-steps
  constexpr int WN = BN / 32, WM = 8 / WN, MT = 2 / WM;
  constexpr int RAW_F = BKA * BM;                       // floats per raw stage
  constexpr int NP = PAIRS ? 2 : 3;                     // planes
  constexpr int FRAG_B = 4 * NP * 64 * 16;              // bytes per fragment stage: 4 jobs x NP planes
  __shared__ __attribute__((aligned(16))) char smem[2 * RAW_F * 4 + 2 * FRAG_B];
  float *raw = reinterpret_cast<float *>(smem);
  char *frag = smem + 2 * RAW_F * 4;

  const int n_t = p.counts[0], ld = p.counts[2];
  const float a_scale = PAIRS ? rkp::scale_from(p.a_amax, 1024.0f) : 1.0f;
  const float z_scale = PAIRS ? (p.z_scale_dev ? *p.z_scale_dev : p.z_scale) : 1.0f;
  int kc;
  const int ns = dw3_splits(n_t, p.tiles_n, p.Bp, p.max_splits, p.wg_slots, &kc);
  if (L == 0 && threadIdx.x == 0) p.counts[4] = ns;
  const int tiles_m = (n_t + BM - 1) / BM;
  if (L >= tiles_m * p.tiles_n * ns) return;
  const int mt0 = L / (p.tiles_n * ns), rem = L % (p.tiles_n * ns);
  const int nt = rem / ns, split = rem % ns;
  const int m0 = mt0 * BM, n0 = nt * BN;
  const int kbeg = split * kc, kend = min(p.Bp, kbeg + kc);
  const int na = (kend - kbeg) / BKA;           // stages (Bp and kc are multiples of 32)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, lh = lane >> 5;
  const bool probing = p.probe != nullptr && tid == 0;
  unsigned long long t_start = 0, t_first = 0, t_loop = 0;
  if (probing) t_start = wall_clock64();

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // A staging: every thread carries one float4 of a 32-row stage (row = tid / 16, 4 items).  Rows
  // past the batch are clamped to the last one (finite data; the matching Z^T entries are zero),
  // columns past the row to its last 16 bytes (they only reach output rows that are never stored).
  const int a_row = tid >> 4;
  const float *a_src = p.dO + min(m0 + (tid & 15) * 4, ld - 4);
  // Every workgroup walks its K range from a different starting stage (rotated by its tile
  // index): the ~240 workgroups of a launch all read the SAME Z^T planes, and in lockstep they
  // hammer the same few L2 channels at any instant (measured: 34 GB/s per CU whatever the path).
  // The summation order of a tile is still a fixed function of its index: deterministic.
  const int rot = na > 0 ? (mt0 * 5 + split * 3) % na : 0;
  auto stage_of = [&](int it) { return (it + rot) % na; };
  auto loadA = [&](int it) -> float4 {
    const int row = min(kbeg + stage_of(it) * BKA + a_row, p.B - 1);
    return *reinterpret_cast<const float4 *>(a_src + (int64_t)row * ld);
  };
  auto storeA = [&](int buf, const float4 v) {
    *reinterpret_cast<float4 *>(raw + buf * RAW_F + a_row * BM + (tid & 15) * 4) = v;
  };
  // conversion pass: wave w = job * 2 + half; job = (m-tile, k-step) of the stage, a lane takes
  // k-values [half*4, half*4 + 4) of ITS fragment (lane = item l31, k-group lh)
  const int cj = wid >> 1, chf = wid & 1;               // job 0..3 = mt * 2 + ks
  auto convert = [&](int buf) {
    const float *r = raw + buf * RAW_F + (((cj & 1) * 16 + lh * 8 + chf * 4) * BM + (cj >> 1) * 32 + l31);
    const float x0 = r[0], x1 = r[BM], x2 = r[2 * BM], x3 = r[3 * BM];
    char *d = frag + buf * FRAG_B + (cj * NP * 64 + lane) * 16 + chf * 8;
    if (PAIRS) {
      uint2 h, l;
      rkp::split4(make_float4(x0, x1, x2, x3), a_scale, h, l);
      *reinterpret_cast<uint2 *>(d) = h;
      *reinterpret_cast<uint2 *>(d + 64 * 16) = l;
    } else {
      uint2 h, m, l;
      split_pair(x0, x1, h.x, m.x, l.x);
      split_pair(x2, x3, h.y, m.y, l.y);
      *reinterpret_cast<uint2 *>(d) = h;
      *reinterpret_cast<uint2 *>(d + 64 * 16) = m;
      *reinterpret_cast<uint2 *>(d + 2 * 64 * 16) = l;
    }
  };
  // B fragments: plane pl, k-step kt -> 8 k-values (chunk 2*kt + lh) of column n0 + wn*32 + l31
  const int nk = 2 * na;
  const uint16_t *b_src = p.planes + ((int64_t)((kbeg >> 3) + lh) * p.cols_pad + n0 + wn * 32 + l31) * 8;
  const int64_t b_step = (int64_t)2 * p.cols_pad * 8;
  // (a wave whose 32 columns all lie past h -- h = 200: the last of the eight -- neither loads
  // nor multiplies: the fetch rate is what bounds the kernel)
  const bool wave_live = n0 + wn * 32 < p.h;
  auto loadB = [&](uint4 (&dst)[NP], int kt) {
    const uint16_t *q = b_src + (int64_t)(2 * stage_of(kt >> 1) + (kt & 1)) * b_step;
    if (wave_live) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const uint4 *>(q + pl * p.plane_stride);
    }
  };

  auto compute = [&](int buf, int ks, const uint4 (&b)[NP]) {
    if (!wave_live) return;
    if (PAIRS) {
      const f16x8 Bh = __builtin_bit_cast(f16x8, b[0]), Bl = __builtin_bit_cast(f16x8, b[1]);
      f16x8 Ah[MT], Al[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const char *q = frag + buf * FRAG_B + ((((wm * MT + t) * 2 + ks) * NP) * 64 + lane) * 16;
        Ah[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q));
        Al[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q + 64 * 16));
      }
      // small terms first, as the decode / dZ contractions
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t], Bl, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t], Bh, acc[t], 0, 0, 0);
      return;
    }
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, b[0]), Bm = __builtin_bit_cast(bf16x8, b[1]),
                 Bl = __builtin_bit_cast(bf16x8, b[NP - 1]);
    bf16x8 Ah[MT], Am[MT], Al[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const char *q = frag + buf * FRAG_B + ((((wm * MT + t) * 2 + ks) * NP) * 64 + lane) * 16;
      Ah[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(q));
      Am[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(q + 64 * 16));
      Al[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(q + (NP - 1) * 64 * 16));
    }
    // small terms first; the accumulators alternate so that no MFMA waits on the one before it
    // (an instruction slipping between two MFMAs on the SAME accumulator costs ~40 cycles)
    if (PLAIN) {
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bh, acc[t], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bl, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[t], Bm, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bm, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bh, acc[t], 0, 0, 0);
  };

  uint4 breg[R][NP];
  if (na > 0) {
#pragma unroll
    for (int r = 0; r < P; ++r) loadB(breg[r], r);
    storeA(0, loadA(0));
    const float4 a1 = loadA(1);
    __syncthreads();
    convert(0);
    storeA(1, a1);
    __syncthreads();
    if (probing) t_first = wall_clock64();
    // Every load is issued unconditionally (past the end: the last tile again, never used) and
    // pinned ahead of the MFMAs by a scheduling barrier: hipcc otherwise sinks a prefetch down to
    // its first use (and then waits for it with the whole queue drained).
    for (int it = 0; it < na; it += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (it + u < na) {
          const int k = 2 * (it + u);                    // k % R == 2 * u
          loadB(breg[(2 * u + P) % R], k + P);
          const float4 a_next = loadA(it + u + 2);
          __builtin_amdgcn_sched_barrier(0);
          compute(u, 0, breg[2 * u]);                    // fragments of stage it+u
          loadB(breg[(2 * u + 1 + P) % R], k + 1 + P);
          __builtin_amdgcn_sched_barrier(0);
          compute(u, 1, breg[2 * u + 1]);
          convert(u ^ 1);                                // raw stage it+u+1 -> its fragments
          __builtin_amdgcn_sched_barrier(0);
          storeA(u, a_next);                             // raw stage it+u+2 (buffer of stage it+u)
          __syncthreads();
        }
      }
    }
  }
  if (probing) t_loop = wall_clock64();

  // ---- epilogue: a lane holds 16 items (rows) of ONE column; 32 lanes = 128 contiguous bytes ----
  float *C = (ns == 1 && p.G) ? p.G : p.slabs + (int64_t)split * p.slab_stride;
  const int n = n0 + wn * 32 + l31;
  const float inv = PAIRS ? 1.0f / (a_scale * z_scale) : 1.0f;       // exact: powers of two
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int item = m0 + (wm * MT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (item < n_t && n < p.h) C[(int64_t)item * p.h + n] = PAIRS ? acc[t][r] * inv : acc[t][r];
    }
  }
  if (probing) {
    unsigned long long *o = p.probe + (size_t)L * 16;
    o[0] = t_start; o[1] = t_first; o[13] = t_loop; o[14] = wall_clock64(); o[15] = (unsigned long long)nk;
  }
}

template <int BN, bool PLAIN = false, bool PAIRS = false>
__global__ __launch_bounds__(512) void dw3_kernel(Dw3P p) {
  dw3_body<BN, PLAIN, PAIRS>(p, (int)blockIdx.x);
}

// dW (fp16 pairs) || the dZ slab reduce in ONE launch (MatrixFactorization steps: nothing between the
// decode and the Adam sweep reads dZ, so its reduce need not be a link of the chain): the first n_dw
// workgroups run the dW tiles, the others rkred::body (512 threads either way)
template <int BN>
__global__ __launch_bounds__(512) void dw_reduce_kernel(Dw3P p, int n_dw, rkred::Args r) {
  if ((int)blockIdx.x < n_dw) {
    dw3_body<BN, false, true>(p, (int)blockIdx.x);
    return;
  }
  __shared__ float4 part[rkred::RED_W - 1][64];
  rkred::body(r, (int)blockIdx.x - n_dw, part);
}

// dW (fp16 pairs) || encoder backward in ONE launch: the first n_dw workgroups run the dW tiles, the
// others the encoder backward's columns (encoder_bwd.h: a wave per column, 4 per workgroup -- waves
// 4 .. 7 of those workgroups leave at once).  Both need only what the launches in front of them left
// (dO and the Z^T planes; dZ0 and the block) and write disjoint outputs: as two launches they cost the
// chain 16 + 14 us in line or two cross-queue edges (12 us fork + 6-11 us join) on a side stream.
struct EncBwdP {
  rk_block_t b;
  int row_off, B;
  const float *dZ;
  int h;
  float *G;
  float *gb;
  int n_gb;
};
// (multinomial loss: dO comes from rk_mnll_finish, no epilogue has summed its columns) the decoder
// bias gradient out[c] = sum_r dO[r][c] as a third workgroup range of the same launch: 64 columns x
// 8 row slices per workgroup, 4 accumulators per thread, combined in a fixed order
struct ColsumP {
  const float *X;
  int rows;
  const int32_t *counts;      // [0] live columns, [2] ld
  float *out;
};
__device__ __forceinline__ void colsum_body_512(const ColsumP &c, const int block) {
  __shared__ float part[8][64];
  const int cols = c.counts[0], ld = c.counts[2];
  const int lc = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int col = block * 64 + lc;
  const int per = (c.rows + 7) >> 3;
  const int r0 = s * per, r1 = min(c.rows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < cols) {
    const float *x = c.X + col;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      a0 += x[(int64_t)r * ld];
      a1 += x[(int64_t)(r + 1) * ld];
      a2 += x[(int64_t)(r + 2) * ld];
      a3 += x[(int64_t)(r + 3) * ld];
    }
    for (; r < r1; ++r) a0 += x[(int64_t)r * ld];
  }
  part[s][lc] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (s == 0 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][lc];
    c.out[col] = t;
  }
}

template <int BN, int HV>
__global__ __launch_bounds__(512) void dw_encbwd_kernel(Dw3P p, EncBwdP e, int n_dw, ColsumP cs, int n_cs) {
  if ((int)blockIdx.x < n_dw) {
    dw3_body<BN, false, true>(p, (int)blockIdx.x);
    return;
  }
  if ((int)blockIdx.x < n_dw + n_cs) {
    colsum_body_512(cs, (int)blockIdx.x - n_dw);
    return;
  }
  if (threadIdx.x >= 256) return;
  ae_encode_bwd_cols_body<HV>(e.b, e.row_off, e.B, e.dZ, e.h, e.G, 0, e.gb, e.n_gb,
                              (int)blockIdx.x - n_dw - n_cs);
}

// G = sum of the ns slabs the kernel above wrote (ns = counts[4]; nothing to do for ns == 1:
// the kernel wrote G itself)
__global__ __launch_bounds__(256) void slab_sum3_kernel(const float *__restrict__ slabs,
                                                        int64_t slab_stride, int h,
                                                        const int32_t *__restrict__ counts,
                                                        float *__restrict__ G) {
  const int ns = counts[4];
  if (ns <= 1) return;
  const int64_t live4 = ((int64_t)counts[0] * h) >> 2;
  const float4 *w = reinterpret_cast<const float4 *>(slabs);
  const int64_t s4 = slab_stride >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < live4; i += (int64_t)gridDim.x * 256) {
    float4 s = w[i];
    for (int z = 1; z < ns; ++z) {
      const float4 v = w[(int64_t)z * s4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4 *>(G)[i] = s;
  }
}

constexpr int DW3_MAX_SPLITS = 4;
inline int dw3_bn(int h) { return h <= 128 ? 128 : 256; }
inline int dw3_cols_pad(int h) { const int bn = dw3_bn(h); return rk_cdiv(h, bn) * bn; }
inline int dw3_rows_pad(int B) { return rk_cdiv(B, 64) * 64; }
inline int64_t dw3_plane_bytes(int B, int h) {
  return (int64_t)3 * dw3_rows_pad(B) * dw3_cols_pad(h) * 2;
}

}  // namespace

extern "C" int64_t rk_dw3_workspace_bytes(int32_t B, int32_t h, int32_t n_cap) {
  const int64_t planes = (dw3_plane_bytes(B, h) + 255) & ~(int64_t)255;
  return planes + (int64_t)DW3_MAX_SPLITS * n_cap * h * sizeof(float);
}

extern "C" int32_t rk_dw3_max_splits(void) { return DW3_MAX_SPLITS; }

extern "C" void rk_dw3_probe(unsigned long long *buffer) { g_dw3_probe = buffer; }

static int split_planes_t_launch(const float *X, int32_t rows, int32_t cols, int32_t ld, int32_t rows_pad,
                                 int32_t cols_pad, void *planes, int pairs, const uint32_t *amax,
                                 float *scale_out, void *stream_);

static int split_planes_t_launch(const float *X, int32_t rows, int32_t cols, int32_t ld, int32_t rows_pad,
                                 int32_t cols_pad, void *planes, int pairs, const uint32_t *amax,
                                 float *scale_out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(rows_pad % 8 == 0 && rows_pad >= rows && cols_pad >= cols, "bad padding");
  RK_REQUIRE((((uintptr_t)planes) & 15) == 0, "planes must be 16-byte aligned");
  const int64_t tot = (int64_t)(rows_pad >> 3) * cols_pad;
  if (tot == 0) return 0;
  RK_LAUNCH(split_planes_t_kernel, dim3(rk_cdiv(tot, 256)), dim3(256), 0, stream, X, rows, cols, ld,
            rows_pad, cols_pad, (uint16_t *)planes, pairs, amax, scale_out);
  RK_CHECK_LAUNCH("split_planes_t");
  return 0;
}

// G_de[n_t,h] = dO^T . Z on the bf16 pipe (see the head of this file and recoder_hip.h)
extern "C" int64_t rk_dw3_planes_bytes(int32_t B, int32_t h) { return dw3_plane_bytes(B, h); }
extern "C" int32_t rk_dw3_cols_pad(int32_t h) { return dw3_cols_pad(h); }
extern "C" int32_t rk_dw3_rows_pad(int32_t B) { return dw3_rows_pad(B); }

struct EncBwdArgs {          // the encoder backward riding on the dW launch (rk_decode_bwd_dw2_encode_bwd)
  int32_t row_off;
  const float *dZ0pre;
  float *G_en, *gb_en;
};

static int dw_impl(const float *dO, const float *Z, int32_t B, int32_t h, const rk_block_t *tgt,
                   float *G_de, float *gb_de, void *workspace, const void *zt_planes, bool pairs,
                   const int32_t *ranges, void *stream_, const EncBwdArgs *enc = nullptr,
                   const rkred::Args *red = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h > 0 && h % 4 == 0, "h must be a multiple of 4");
  RK_REQUIRE(workspace != nullptr && (((uintptr_t)workspace) & 255) == 0, "workspace: 256-byte aligned");
  RK_REQUIRE((((uintptr_t)dO | (uintptr_t)Z | (uintptr_t)G_de) & 15) == 0, "operands must be 16-byte aligned");
  if (B == 0) return 0;
  const int bn = dw3_bn(h);
  const int cols_pad = dw3_cols_pad(h), Bp = dw3_rows_pad(B);
  const int64_t planes_b = (dw3_plane_bytes(B, h) + 255) & ~(int64_t)255;
  if (rk_gemm_plain_bf16()) pairs = false;       // (the plain-bf16 data point rides on the bf16 planes)
  Dw3P p = {};
  // Z^T planes: the caller's (the encoder forward wrote them with Z) or made here.  Pairs made here:
  // the scale from ranges[0..63] (a bound of |Z|; all zero / NULL: the static one), kept in the
  // unused third plane's space for the kernel
  p.z_scale = rkp::SCALE_Z;
  if (zt_planes == nullptr) {
    RK_REQUIRE(Z != nullptr, "Z or zt_planes");
    float *sc = pairs ? reinterpret_cast<float *>((char *)workspace + (int64_t)2 * Bp * cols_pad * 2) : nullptr;
    int rc = split_planes_t_launch(Z, B, h, h, Bp, cols_pad, workspace, pairs ? 1 : 0,
                                   reinterpret_cast<const uint32_t *>(ranges), sc, stream_);
    if (rc) return rc;
    zt_planes = workspace;
    p.z_scale_dev = sc;
  } else if (zt_planes == workspace && pairs) {
    // the pair planes AND their scale were made in place, at the head of this workspace, by an earlier
    // launch (rk_split_wz: the operand splits of the step's decode)
    p.z_scale_dev = reinterpret_cast<float *>((char *)workspace + (int64_t)2 * Bp * cols_pad * 2);
  }
  RK_REQUIRE((((uintptr_t)zt_planes) & 15) == 0, "zt_planes must be 16-byte aligned");
  p.a_amax = reinterpret_cast<const uint32_t *>(tgt->counts) + 8;
  p.dO = dO; p.counts = tgt->counts;
  p.planes = (const uint16_t *)zt_planes; p.plane_stride = (int64_t)Bp * cols_pad;
  p.cols_pad = cols_pad; p.B = B; p.Bp = Bp; p.h = h;
  p.tiles_n = cols_pad / bn; p.max_splits = DW3_MAX_SPLITS;
  p.G = G_de; p.slabs = (float *)((char *)workspace + planes_b);
  p.slab_stride = (int64_t)tgt->n_cap * h;
  p.probe = g_dw3_probe;
  p.plain = rk_gemm_plain_bf16();
  const int tiles_cap = rk_cdiv(tgt->n_cap, 64) * p.tiles_n;
  p.wg_slots = 256;       // one 8-wave workgroup per CU is what the split-K sizing aims at
  // the live workgroups are the first tiles_m(n_t) * tiles_n * ns of the grid; ns * tiles never
  // exceeds max(wg_slots, tiles), so the capacity grid is bounded by that
  const int64_t grid = std::max<int64_t>((int64_t)tiles_cap, std::min<int64_t>((int64_t)tiles_cap * DW3_MAX_SPLITS, p.wg_slots));
  if (enc) {
    RK_REQUIRE(pairs, "the fused encoder backward rides on the fp16-pair dW");
    EncBwdP e = {};
    e.b = *tgt; e.row_off = enc->row_off; e.B = B; e.dZ = enc->dZ0pre; e.h = h; e.G = enc->G_en;
    e.gb = enc->gb_en; e.n_gb = enc->gb_en ? rk_cdiv(h, 64) : 0;
    const int n_enc = rk_cdiv(tgt->n_cap, 4) + e.n_gb;
    const int hv = rk_cdiv(h, 256);
    ColsumP cs = {};
    int n_cs = 0;
    if (gb_de) {            // column sums of dO ride on this launch too
      cs.X = dO; cs.rows = B; cs.counts = tgt->counts; cs.out = gb_de;
      n_cs = rk_cdiv(tgt->n_cap, 64);
      gb_de = nullptr;
    }
#define LAUNCH(BN, HV) RK_LAUNCH((dw_encbwd_kernel<BN, HV>), dim3((unsigned)grid + n_cs + n_enc), dim3(512), 0, stream, p, e, (int)grid, cs, n_cs)
    if (bn == 128) { if (hv == 1) LAUNCH(128, 1); else if (hv == 2) LAUNCH(128, 2); else LAUNCH(128, 4); }
    else { if (hv == 1) LAUNCH(256, 1); else if (hv == 2) LAUNCH(256, 2); else LAUNCH(256, 4); }
#undef LAUNCH
  } else if (pairs && red) {
    const int n_red = rkred::blocks(red->M, red->N);
    if (bn == 128)
      RK_LAUNCH((dw_reduce_kernel<128>), dim3((unsigned)grid + n_red), dim3(512), 0, stream, p, (int)grid, *red);
    else
      RK_LAUNCH((dw_reduce_kernel<256>), dim3((unsigned)grid + n_red), dim3(512), 0, stream, p, (int)grid, *red);
  } else if (pairs) {
    if (bn == 128)
      RK_LAUNCH((dw3_kernel<128, false, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
    else
      RK_LAUNCH((dw3_kernel<256, false, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
  } else if (p.plain) {
    if (bn == 128)
      RK_LAUNCH((dw3_kernel<128, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
    else
      RK_LAUNCH((dw3_kernel<256, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
  } else if (bn == 128)
    RK_LAUNCH((dw3_kernel<128>), dim3((unsigned)grid), dim3(512), 0, stream, p);
  else
    RK_LAUNCH((dw3_kernel<256>), dim3((unsigned)grid), dim3(512), 0, stream, p);
  RK_CHECK_LAUNCH("dw3");
  if (G_de) {
    const int64_t n4 = (int64_t)tgt->n_cap * h / 4;
    RK_LAUNCH(slab_sum3_kernel, dim3((unsigned)std::min<int64_t>(rk_cdiv(n4, 256), 2048)), dim3(256), 0,
              stream, p.slabs, p.slab_stride, h, tgt->counts, G_de);
    RK_CHECK_LAUNCH("slab_sum3");
  }
  if (gb_de) return rk_colsum(dO, B, tgt->n_cap, 0, tgt->counts, gb_de, stream_);
  return 0;
}

extern "C" int rk_decode_bwd_dw3(const float *dO, const float *Z, int32_t B, int32_t h,
                                 const rk_block_t *tgt, float *G_de, float *gb_de, void *workspace,
                                 const void *zt_planes, void *stream_) {
  return dw_impl(dO, Z, B, h, tgt, G_de, gb_de, workspace, zt_planes, false, nullptr, stream_);
}

// The same contraction with fp16 PAIRS (three products): see dw3_kernel.  dO's split scale comes from
// the maximum the loss kernels publish in tgt->counts[8..71] (as rk_decode_bwd_dz), Z's from
// ranges[0..63] when the planes are made here (zt_planes == NULL), else the static one the encoder
// forward wrote them with (bounded activations only: rk_dw_pairs_planes_ok).
extern "C" int rk_decode_bwd_dw2(const float *dO, const float *Z, int32_t B, int32_t h,
                                 const rk_block_t *tgt, float *G_de, float *gb_de, void *workspace,
                                 const void *zt_planes, const int32_t *ranges, void *stream_) {
  return dw_impl(dO, Z, B, h, tgt, G_de, gb_de, workspace, zt_planes, true, ranges, stream_);
}

// rk_decode_bwd_dw2 (slabs stay in the workspace: G_de == NULL semantics) and rk_ae_encode_bwd (one
// bitmap word per lane: the row window spans <= 64 words) in ONE launch -- see dw_encbwd_kernel
extern "C" int32_t rk_dw_encode_bwd_fused_ok(int32_t row_off, int32_t B) {
  return rk_tune_get(RK_TUNE_DW_ENC_FUSED) == 1 && rk_dw_pairs() && (((row_off + B + 31) >> 5) - (row_off >> 5) <= 64) ? 1 : 0;
}

extern "C" int rk_decode_bwd_dw2_encode_bwd(const float *dO, const float *Z, int32_t B, int32_t h,
                                            const rk_block_t *tgt, void *workspace, const void *zt_planes,
                                            const int32_t *ranges, int32_t row_off, const float *dZ0pre,
                                            float *G_en, float *gb_en, void *stream_) {
  RK_REQUIRE(rk_dw_encode_bwd_fused_ok(row_off, B), "outside the fused launch's domain (rk_dw_encode_bwd_fused_ok)");
  RK_REQUIRE(h > 0 && h % 4 == 0 && h <= 1024, "h must be a multiple of 4, <= 1024");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(tgt->bits_cr != nullptr && tgt->pref_rc != nullptr,
             "block was built without the transposed bitmap / prefix index");
  const EncBwdArgs enc = {row_off, dZ0pre, G_en, gb_en};
  return dw_impl(dO, Z, B, h, tgt, nullptr, nullptr, workspace, zt_planes, true, ranges, stream_, &enc);
}

// rk_decode_bwd_dw2 (slabs stay in the workspace) || rk_decode_dz_reduce in ONE launch
extern "C" int rk_decode_bwd_dw2_dz_reduce(const float *dO, const float *Z, int32_t B, int32_t h,
                                           const rk_block_t *tgt, void *workspace, const void *zt_planes,
                                           const int32_t *ranges, const float *dz_workspace,
                                           const float *Zact, int32_t act, float *dZ, void *stream_) {
  RK_REQUIRE(rk_dw_pairs() && !rk_gemm_plain_bf16(), "the fused reduce rides on the fp16-pair dW");
  RK_REQUIRE(dz_workspace != nullptr && dZ != nullptr &&
             ((((uintptr_t)dz_workspace) | ((uintptr_t)dZ)) & 15) == 0, "dz_workspace, dZ: 16-byte aligned");
  if (B == 0) return 0;
  const rkred::Args red = {dz_workspace, B, h, tgt->counts, 128, rk_cdiv(tgt->n_cap, 128), Zact, act, dZ};
  return dw_impl(dO, Z, B, h, tgt, nullptr, nullptr, workspace, zt_planes, true, ranges, stream_, nullptr, &red);
}

// ... and the decoder bias gradient gb_de[c] = sum_r dO[r][c] (multinomial loss) as a third
// workgroup range of that launch
extern "C" int rk_decode_bwd_dw2_encode_bwd_colsum(const float *dO, const float *Z, int32_t B, int32_t h,
                                                   const rk_block_t *tgt, void *workspace,
                                                   const void *zt_planes, const int32_t *ranges,
                                                   int32_t row_off, const float *dZ0pre, float *G_en,
                                                   float *gb_en, float *gb_de, void *stream_) {
  RK_REQUIRE(rk_dw_encode_bwd_fused_ok(row_off, B), "outside the fused launch's domain (rk_dw_encode_bwd_fused_ok)");
  RK_REQUIRE(h > 0 && h % 4 == 0 && h <= 1024, "h must be a multiple of 4, <= 1024");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(tgt->bits_cr != nullptr && tgt->pref_rc != nullptr,
             "block was built without the transposed bitmap / prefix index");
  const EncBwdArgs enc = {row_off, dZ0pre, G_en, gb_en};
  return dw_impl(dO, Z, B, h, tgt, nullptr, gb_de, workspace, zt_planes, true, ranges, stream_, &enc);
}

// rk_tune(RK_TUNE_DW_BF16X3, 1) keeps dW on the bf16 triples (no operand range at all) in the training step
extern "C" int32_t rk_dw_pairs(void) {
  return (rk_tune_get(RK_TUNE_DW_BF16X3) == 0 && !rk_gemm_plain_bf16()) ? 1 : 0;
}

const float *rk_dw3_slabs(const void *workspace, int32_t B, int32_t h) {
  const int64_t planes_b = (dw3_plane_bytes(B, h) + 255) & ~(int64_t)255;
  return (const float *)((const char *)workspace + planes_b);
}
