// Pre-split operand planes of the decoder contractions (decode16.hip).
//
// gemm.hip's PREC_H3 kernels split every fp32 operand into an fp16 pair  s.x = hi + lo  while they
// stage it into LDS -- every W_de row once per ROW tile (8x in the decode, 4x in dZ), Z once per
// column tile (62x): 30 VALU instructions per MFMA (profiles/r02_b_gemm_sq_counters_b500.txt).  Here
// each operand is split ONCE per step into a compact "plane image" that the contraction kernels copy
// into LDS as it is:
//
//   image of X[rows, K] (K = contraction index, padded to Kp = round_up(K, 32) with zeros):
//     row r = KT = Kp / 32 k-tiles of 128 bytes:  [ 32 x fp16 hi | 32 x fp16 lo ]
//     element (r, k): hi at  r * KT * 128 + (k >> 5) * 128 + (k & 31) * 2,  lo 64 bytes further.
//   One 128-byte line = one row's share of a 32-deep k-tile (hi and lo together), so staging a tile
//   is a plain copy of full cache lines, and in LDS (row stride 144 B, an odd number of 16-byte
//   slots) a lane's MFMA fragment (8 consecutive k of one row) is one conflict-free ds_read_b128.
//
//   Z  image  [B rows][KT(h)]        written by the encoder forward with the value in registers
//   W  image  [n_b rows][KT(h)]      the gathered decoder rows W_de[items[c]] (decode: B operand)
//   W^T image [KT(n_ld)][Hp rows]    K-TILE MAJOR: line (kt, j) = hidden unit j, the 32 compact items of
//                                     k-tile kt (dZ: B operand) at (kt * Hp + j) * 128 -- a k-tile of the
//                                     whole image is ONE contiguous block (Hp * 128 bytes), so the dZ
//                                     kernel streams and the split pass writes contiguously (row-major
//                                     [Hp][KT] put consecutive rows 195 KB apart at C5's 48.8 k items:
//                                     dZ ran 9x off its MFMA time);
//                                     items in [n_b, round_up(n_b, 32)) are written as ZEROS (the K tail)
//   scales[0] / [1]: the power-of-two split scales used for Z / W (read by the consumers).
#pragma once
#include "common.h"

namespace rkp {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LINE = 128;                 // bytes of one row's 32-deep k-tile (hi | lo)
constexpr float SCALE_W = 128.0f;         // static ranges when no bound is published (as gemm.hip)
constexpr float SCALE_Z = 32.0f;

__host__ __device__ inline int kp_of(int K) { return (K + 31) & ~31; }

// 4 consecutive-k fp32 values -> 4 fp16 "hi" + 4 fp16 "lo" of s.x  (identical to gemm.hip split4:
// the two paths produce the same pieces, so their contractions agree bit for bit)
__device__ __forceinline__ void split4(const float4 v, const float s, uint2 &hi, uint2 &lo) {
  const f32x2 a = {v.x * s, v.y * s}, b = {v.z * s, v.w * s};
  const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
  const f32x2 la = a - __builtin_convertvector(ha, f32x2), lb = b - __builtin_convertvector(hb, f32x2);
  const f16x2 qa = __builtin_convertvector(la, f16x2), qb = __builtin_convertvector(lb, f16x2);
  hi.x = __builtin_bit_cast(uint32_t, ha); hi.y = __builtin_bit_cast(uint32_t, hb);
  lo.x = __builtin_bit_cast(uint32_t, qa); lo.y = __builtin_bit_cast(uint32_t, qb);
}

// scale of an operand from its published maximum (64 slots of fp32 bit patterns, gemm.hip
// scale_from): max . s in [2^13, 2^14); all slots zero / no slots: the static default.
// Must be called by a whole wave (it shuffles).
__device__ __forceinline__ float scale_from(const uint32_t *slots, float dflt) {
  if (slots == nullptr) return dflt;
  uint32_t m = slots[threadIdx.x & 63];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if (m == 0) return dflt;
  const int e = min(max((int)(m >> 23) - 127, -100), 100);
  return __uint_as_float((uint32_t)(13 - e + 127) << 23);
}

// plain bf16 operands (RK_GEMM_PREC=bf16): hi = bf16(x), no scale, lo unused (zero)
__device__ __forceinline__ void plain4(const float4 v, uint2 &hi, uint2 &lo) {
  uint32_t m, l;
  rk_split_bf16_pair(v.x, v.y, hi.x, m, l);
  rk_split_bf16_pair(v.z, v.w, hi.y, m, l);
  lo = make_uint2(0u, 0u);
}

// store the split of 4 consecutive k (k % 4 == 0) of one image row
__device__ __forceinline__ void store_split4(char *row_base, int k, const float4 v, const float s,
                                             const bool plain = false) {
  uint2 hi, lo;
  if (plain) plain4(v, hi, lo); else split4(v, s, hi, lo);
  char *d = row_base + (k >> 5) * LINE + (k & 31) * 2;
  *reinterpret_cast<uint2 *>(d) = hi;
  *reinterpret_cast<uint2 *>(d + 64) = lo;
}

// Z^T as TWO fp16 planes hi / lo of s.x in the dW kernels' layout (dw3.hip: plane p element (k = row,
// n = col) at ((k/8)*cols_pad + n)*8 + k%8; rows >= rows and columns >= cols are zeros); s from `amax`
// (64 slots, nullable -> SCALE_Z), *scale_out <- s.  Thread i of the job grid: (chunk of 8 rows, n).
__device__ __forceinline__ void split_zt_pairs_job(const int block, const float *__restrict__ X, int rows,
                                                   int cols, int ld, int rows_pad, int cols_pad,
                                                   uint16_t *__restrict__ P, const uint32_t *amax,
                                                   float *scale_out) {
  const float s = scale_from(amax, SCALE_Z);
  if (scale_out && block == 0 && threadIdx.x == 0) *scale_out = s;
  const int64_t i = (int64_t)block * 256 + threadIdx.x;
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
  uint2 h0, l0, h1, l1;
  split4(make_float4(x[0], x[1], x[2], x[3]), s, h0, l0);
  split4(make_float4(x[4], x[5], x[6], x[7]), s, h1, l1);
  *reinterpret_cast<uint4 *>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4 *>(d + plane) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

struct SplitW {
  const float *W;            // [n_items, h] table the decoder reads
  const int32_t *items;      // compact column -> table row
  const int32_t *counts;     // [0] = n_b (device)
  const uint32_t *amax;      // 64 slots: bound of |W| (nullable)
  char *wp;                  // W image   [n rows][KT]
  char *wtp;                 // W^T image [n_ld / 32 k-tiles][Hp rows]; NULL: not made (csrc/pgemm.h reads
                             // the W image along its rows instead)
  float *scales;             // [1] <- the scale used
  int h, KT;                 // KT = kp_of(h) / 32
  int n_ld;                  // items padded (multiple of 32): row pitch of the W^T image = n_ld / 32 lines
  int plain;                 // != 0: plain bf16 images (RK_GEMM_PREC=bf16), scale 1
};

// One workgroup (NT threads, NT = 256 or 512) splits the 32 gathered rows of item tile `tile`:
// W image rows by direct 8-byte stores, the W^T image through an LDS transposition, 64 hidden units
// at a time.  smem: 2 * 32 * 66 * 2 bytes (8448).
constexpr int SPLIT_W_LDS = 2 * 32 * 66 * 2;
template <int NT>
__device__ __forceinline__ void split_w_job(const SplitW &p, const int tile, char *smem) {
  const int n_b = p.counts[0];
  const int k0 = tile * 32;
  if (k0 >= n_b) return;
  const float s = p.plain ? 1.0f : scale_from(p.amax, SCALE_W);
  if (tile == 0 && threadIdx.x == 0) p.scales[1] = s;
  const int tid = threadIdx.x;
  const int Kp = p.KT * 32;
  uint16_t *sh = reinterpret_cast<uint16_t *>(smem);          // [plane][item 32][66]
  // The table rows of FOUR 64-unit chunks are fetched at once (h = 200: all of them): chunk by chunk
  // the job was a chain of as many dependent round trips, and these workgroups, not the user rows,
  // were what the encoder-forward launch waited for.
  constexpr int PER = (32 * 16 + NT - 1) / NT;           // (item, float4) pairs per thread and chunk
  int my_c[PER];
  int64_t my_row[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int u = tid + e * NT;
    my_c[e] = k0 + (u >> 4);
    my_row[e] = (u < 32 * 16 && my_c[e] < n_b) ? (int64_t)p.items[my_c[e]] * p.h : -1;
  }
  for (int jg = 0; jg < Kp; jg += 256) {
    float4 pre[4][PER];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int j = jg + g * 64 + ((tid + e * NT) & 15) * 4;
        pre[g][e] = (my_row[e] >= 0 && j < p.h) ? *reinterpret_cast<const float4 *>(p.W + my_row[e] + j)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j0 = jg + g * 64;
      if (j0 >= Kp) break;
      // phase 1: item rows -> split -> W image + LDS (row-major)
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int u = tid + e * NT;
        if (u < 32 * 16) {
          const int it = u >> 4, q = u & 15;
          const int j = j0 + q * 4;
          const int c = my_c[e];
          uint2 hi, lo;
          if (p.plain) plain4(pre[g][e], hi, lo); else split4(pre[g][e], s, hi, lo);
          if (c < n_b && j < Kp) {
            char *d = p.wp + (int64_t)c * p.KT * LINE + (j >> 5) * LINE + (j & 31) * 2;
            *reinterpret_cast<uint2 *>(d) = hi;
            *reinterpret_cast<uint2 *>(d + 64) = lo;
          }
          if (p.wtp) {
            uint32_t *dh = reinterpret_cast<uint32_t *>(sh + it * 66 + q * 4);
            uint32_t *dl = reinterpret_cast<uint32_t *>(sh + 32 * 66 + it * 66 + q * 4);
            dh[0] = hi.x; dh[1] = hi.y;
            dl[0] = lo.x; dl[1] = lo.y;
          }
        }
      }
      if (p.wtp == nullptr) continue;              // (uniform: no transposed image, no LDS round trip)
      __syncthreads();
      // phase 2: (hidden unit j, 16-byte piece pc): 8 items of one plane -> one piece of the W^T line
      for (int u = tid; u < 64 * 8; u += NT) {
        const int jj = u >> 3, pc = u & 7;
        const int j = j0 + jj;
        if (j < Kp) {
          const uint16_t *src = sh + (pc >> 2) * 32 * 66 + ((pc & 3) * 8) * 66 + jj;
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w[e] = (uint32_t)src[(2 * e) * 66] | ((uint32_t)src[(2 * e + 1) * 66] << 16);
          char *d = p.wtp + ((int64_t)tile * Kp + j) * LINE + pc * 16;
          *reinterpret_cast<uint4 *>(d) = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace rkp

// split jobs riding on the encoder-forward launch (rk_ae_encode_fwd_at)
struct rk_enc_split_t {
  rkp::SplitW sw;       // the W split (n_split workgroups, first in the grid); sw.scales is also
  int n_split;          // where the Z image's scale goes
  char *zimg;           // nullable: Z image written by the kernel's epilogue (static scale)
  int z_kt;
  int z_ones;           // != 0 (and h % 32 != 0): image column h of every row <- 1 (rk_pg_dw_encode_bwd_ones)
};
rkp::SplitW rk_split_w_args(const float *W_de, const rk_block_t *tgt, const int32_t *ranges,
                            const rk_planes_t *pl);
