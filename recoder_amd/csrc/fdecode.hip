// The fused decode of small hidden sizes (h <= 224), round 4:  decode + loss + the dZ partial of a
// 128-item x 128-user tile with NOTHING of the tile leaving the registers in between.
//
//   * The tile is computed TRANSPOSED, O^T[item, user] = W[item, :] . Z[user, :]^T: in the 32 x 32 MFMA
//     accumulator layout a lane then holds ONE user (column) and 16 items (rows) of each block -- its
//     target bits are ONE bitmap word per block, the loss runs where the logits are (no transposition
//     through LDS), and four consecutive items of a lane + the four of its partner lane (l ^ 32) are, after
//     one v_permlane32_swap per register, exactly the B operand of  dZ^T[j, user] += W^T[j, item] .
//     dO^T[item, user]  -- the gradient tile goes from accumulators to MFMA operand without touching LDS.
//   * All KT k-tiles of the tile's 128 gathered W rows stay RESIDENT in LDS (KT x 16 KB, copied by LDS-DMA
//     in one burst at the start; a wave's own 32 users' Z lines are loaded coalesced and turned into
//     fragments through 4 KB of wave-private LDS): the dZ product reads them a second time along their rows with
//     ds_read_b64_tr_b16 -- no W^T image is made, written or fetched.
//   * dLoss/dLogits leaves as a plane image of fp16 pairs (16-byte stores straight from the fragments) cut
//     with the scale of its 32-user x 128-item granule: rk_pg_dw reads it (csrc/pgemm.hip).
// Replaces decode_planes_kernel<1, 2, EPI, 3, false, DZT> (decode16.hip: 64 x 128 tiles, fp32 dO tile and
// a W^T stage in LDS, 62 + 62 barrier pairs) where it applies: rk_fdec_ok.
// Reference: nn.py:271-280 (decoder), losses.py:43-47 / BCEWithLogits, autograd of F.linear.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "common.h"
#include "pgemm_epi.h"
#include "planes.h"

namespace {

using pg::f16x8;
using pg::f32x16;

struct FdecP {
  const char *zimg, *wimg;       // plane images (row pitch KT * 128 bytes)
  const float *scales;           // [0] Z, [1] W
  int KT;                        // live k-tiles (ceil(h / 32))
  int M;                         // users (rows of the batch)
  const int32_t *Ndev;           // live items
  int z_rows, w_rows;            // image rows (clamps)
  rk_block_t blk;
  int row_off;
  float confidence, inv_B;
  const float *bias;
  const int32_t *bidx;
  float *loss_part;
  char *dimg;                    // dO image (row m at m * ld * 4 bytes)
  int rows_img;
  float *dscale;                 // [user / 32][ds_pitch]: one scale per 32 users x 64 items
  int ds_pitch;
  float *dz_ws;                  // slabs [column tile][M][h]
  int h;
};

constexpr int WB = 128 * 128;    // bytes of one 128-row k-tile stage

// MFMA A operand of dZ^T = W^T . dO^T from a RESIDENT W stage [128 item rows][128 B] (csrc/pgemm.h "KC"
// layout, 16-byte slots swizzled by pg::kc_sw(row)): this lane's hidden unit = 16 g + (l & 15) of the stage's
// 32, 8 consecutive items starting at row0 (a multiple of 8) -- two transpose reads of 4 rows each
__device__ __forceinline__ f16x8 w_tr_frag(const char *S, const int row0, const int lane, const int plane) {
  const int t = lane & 15, g = (lane >> 4) & 1;
  const int cb = 32 * g + 8 * (t & 3) + plane * 64;       // byte of the 4-hidden chunk in an unswizzled row
  const int slot = cb >> 4, within = cb & 15;
  pg::s16x4 v[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = row0 + 4 * u + (t >> 2);
    v[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (pg::lds_s16x4 *)(S + row * 128 + ((slot ^ pg::kc_sw(row)) << 4) + within));
  }
  const pg::s16x8 x = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, x);
}

// a.upper half-wave <-> b.lower half-wave (v_permlane32_swap)
__device__ __forceinline__ void swap32(uint32_t &a, uint32_t &b) {
  const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = q[0]; b = q[1];
}

constexpr int FD_PITCH(int ktm) { return ktm * 32 + 4; }        // floats per row of the dZ^T exchange
constexpr int FD_LDS(int ktm) {                                   // W stages | exchange (aliased), Z bounce, misc
  return (ktm * WB > 4 * 32 * FD_PITCH(ktm) * 4 ? ktm * WB : 4 * 32 * FD_PITCH(ktm) * 4) + 8 * 4096 + 1024;
}

// 512 threads: wave w works on users 32 (w & 3) .. + 31 of the tile and on the item half (w >> 2) -- two
// waves per SIMD, so that one wave's loss arithmetic, fragment conversions and stores run under the other
// one's MFMAs (with one wave per SIMD the 336 MFMAs of a tile were 15% of its time).
// PLAIN (RK_GEMM_PREC=bf16): the W / Z images hold one bf16 value per element (hi halves, scale 1) -- one product on
// v_mfma_f32_32x32x16_bf16 for the decode and for dZ, the lo halves are never read; dO leaves as plain bf16 too
template <int KTM, int LOSS, bool PLAIN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void fdec_kernel(const FdecP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Wst = smem;                                   // [KTM][128 rows][128 B]
  char *Zb = smem + FD_LDS(KTM) - 1024 - 8 * 4096;    // [8 waves][32 rows][128 B]
  float *misc = reinterpret_cast<float *>(smem + FD_LDS(KTM) - 1024);   // [128] bias | [16] reductions
  const int M = p.M, N = *p.Ndev;
  const int tm = (M + 127) >> 7, tn = (N + 127) >> 7;
  int t;
  if (!pg::tile_of((int)blockIdx.x, tm * tn, t)) return;
  const int mt = t % tm, nt = t / tm;
  const int m0 = mt * 128, n0 = nt * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = pg::rfl(tid >> 6);
  const int pr = wave & 3, hf = wave >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int KT = KTM;                             // (rk_fdec_ok: exactly 2, 4 or 7 k-tiles)

  // this lane's user, its bitmap words (one per 32-item block of its half) and the tile's gathered bias
  const int m = m0 + 32 * pr + l31;
  const int mrow = p.row_off + min(m, M - 1);
  const rk_block_t &b = p.blk;
  uint32_t words[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    words[i] = b.bits_rc[(int64_t)mrow * b.ldw_rc + min((n0 >> 5) + 2 * hf + i, b.ldw_rc - 1)];
  if (tid < 128) misc[tid] = p.bias[p.bidx[min(n0 + tid, N - 1)]];

  // EVERY copy of the tile is issued before the first wait: the KT k-tiles of the W rows by LDS-DMA (they
  // stay resident for the dZ product), this lane's Z fragments -- a wave multiplies only its own 32 users,
  // so they never need LDS -- as plain 16-byte loads behind them.  One wait + one barrier, then the k-loop
  // runs without any.
  pg::Opnd ow = {p.wimg, (int64_t)KT * 128, KT, p.w_rows};
  pg::Stager<128, false, 8> sw;
  sw.init(ow, n0, 0, min(N, p.w_rows), wave, lane);
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) sw.issue(Wst + kt * WB, wave);
  asm volatile("" ::: "memory");
  // Z: lane (row slot lane >> 3, 16-byte slot lane & 7) loads one piece of 8 FULL image lines per
  // instruction (the 28 per-lane fragment loads of the first version touched 32 lines each: the copy phase
  // took 7.4 instead of 4.4 us); a k-tile's 4 pieces go through 4 KB of wave-private LDS into fragments
  pg::u32x4 zraw[KTM][4];
  {
    const int zlim = min(M, p.z_rows) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int zr = min(m0 + 32 * pr + 8 * i + (lane >> 3), zlim);
      const char *zsrc = p.zimg + (int64_t)zr * (KTM * 128) + 16 * (lane & 7);
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) zraw[kt][i] = *reinterpret_cast<const pg::u32x4 *>(zsrc + kt * 128);
    }
  }
  // the 4 KTM loads above are the only memory operations behind the DMA: it has landed when no more than
  // those are outstanding (vmcnt counts in issue order)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * KTM) : "memory");     // (lgkmcnt: the bias tile in LDS)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  pg::FragKC fr;
  fr.init(lane);
  char *zb = Zb + wave * 4096;                        // this wave's bounce stage: [32 rows][128 B], KC swizzle
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    const char *SW = Wst + kt * WB + hf * 8192;
    // (LDS runs a wave's instructions in order: no wait between the writes and the reads of the stage, nor
    // before the next k-tile's writes; the wave barriers only pin the compiler's order)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + (lane >> 3);
      *reinterpret_cast<pg::u32x4 *>(zb + r * 128 + (((lane & 7) ^ pg::kc_sw(r)) << 4)) = zraw[kt][i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f16x8 zh[2], zl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { zh[ks] = fr.load(zb, 0, ks, 0); if (!PLAIN) zl[ks] = fr.load(zb, 0, ks, 1); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 wh[2], wl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { wh[i] = fr.load(SW, i, ks, 0); if (!PLAIN) wl[i] = fr.load(SW, i, ks, 1); }
      if constexpr (PLAIN) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg::bf16x8, wh[i]),
                                                           __builtin_bit_cast(pg::bf16x8, zh[ks]), acc[i], 0, 0, 0);
        continue;
      }
      // (the order of decode16.hip per accumulator: Z lo . W hi, Z hi . W lo, hi . hi)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], zl[ks], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], zh[ks], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], zh[ks], acc[i], 0, 0, 0);
    }
  }

  // ---- loss, where the logits are: lane = user m, acc[i][r] = item n0 + 64 hf + 32 i + (r & 3) + 8 (r >> 2) + 4 lh
  const float inv = 1.0f / (p.scales[0] * p.scales[1]);
  const bool implicit = b.implicit != 0;
  const int nh = n0 + 64 * hf;
  float lsum = 0.f, gmax = 0.f;
  // (the explicit-value lookup is a uniform property of the block: two copies of the loop, one branch)
  auto loss_pass = [&](auto implicit_c) {
    constexpr bool IMPL = decltype(implicit_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t w = words[i];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = nh + 32 * i + nl;
        const bool ok = (m < M) && (n < N);
        const float o = acc[i][r] * inv + misc[64 * hf + 32 * i + nl];
        float tv = 0.f;
        if (ok && ((w >> nl) & 1u)) {
          tv = 1.0f;
          if (!IMPL) tv = b.vals[rk_entry_index(b, mrow, n, w)];
        }
        float l, g;
        if (LOSS == pg::LOSS_MSE) {
          const float wgt = (tv > 0.f) ? (1.0f + p.confidence) : 1.0f;
          const float d = o - tv;
          l = wgt * (d * d);
          g = (2.0f * d) * (wgt * p.inv_B);
        } else {  // BCE with logits
          const float ls = fminf(o, 0.f) - log1pf(expf(-fabsf(o)));
          l = (1.0f - tv) * o - ls;
          const float sg = 1.0f / (1.0f + expf(-o));
          g = (sg - tv) * p.inv_B;
        }
        if (ok) { lsum += l; gmax = fmaxf(gmax, fabsf(g)); } else g = 0.f;
        acc[i][r] = g;
      }
    }
  };
  // implicit feedback + squared error (C2): the same arithmetic in 14 instead of 25 VALU instructions per
  // element -- target and validity bits as shifted masks (one bit-field extract each), the weight as
  // fma(c, bit, 1) (= 1 or fl(1 + c) exactly), g = d * (w * 2 / B) (= (2 d) * (w / B): scaling by 2 is exact),
  // masking by AND with 0 / ~0, the tile's bias as 16-byte LDS reads
  auto loss_pass_mse_implicit = [&]() {
    const float c = p.confidence, inv_B2 = 2.0f * p.inv_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nrem = N - (nh + 32 * i);
      uint32_t vmask = nrem >= 32 ? 0xffffffffu : (nrem <= 0 ? 0u : ((1u << nrem) - 1u));
      if (m >= M) vmask = 0u;
      const uint32_t vsh = vmask >> (4 * lh), wsh = (words[i] & vmask) >> (4 * lh);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bq = *reinterpret_cast<const float4 *>(misc + 64 * hf + 32 * i + 8 * q + 4 * lh);
        const float bias4[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e, bp = e + 8 * q;            // bit of item (r & 3) + 8 (r >> 2) (+ 4 lh: shifted out)
          const float bitf = (float)((wsh >> bp) & 1u);
          const uint32_t okm = (uint32_t)(((int32_t)(vsh << (31 - bp))) >> 31);
          const float o = acc[i][r] * inv + bias4[e];
          const float d = o - bitf;
          const float w = fmaf(c, bitf, 1.0f);
          const float l = w * (d * d);
          lsum += __uint_as_float(__float_as_uint(l) & okm);
          const float g = __uint_as_float(__float_as_uint(d * (w * inv_B2)) & okm);
          gmax = fmaxf(gmax, fabsf(g));
          acc[i][r] = g;
        }
      }
    }
  };
  if (implicit && LOSS == pg::LOSS_MSE) loss_pass_mse_implicit();
  else if (implicit) loss_pass(std::true_type{});
  else loss_pass(std::false_type{});
  // the granule's (32 users x 64 items = this wave's) power-of-two scale, the tile's loss partial
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
  lsum = rk_wave_sum(lsum);
  float s_do = 1.0f;
  if (!PLAIN && gmax > 0.f) {
    const int ex = min(max((int)(__float_as_uint(gmax) >> 23) - 127, -100), 100);
    s_do = __uint_as_float((uint32_t)(13 - ex + 127) << 23);
  }
  if (lane == 0) {
    misc[128 + wave] = lsum;
    misc[136 + wave] = gmax;
    // (a half past the capacity -- items [64 ds_pitch, ...) -- has no slot: its column would be the next row's first)
    if ((m0 >> 5) + pr < ((M + 31) >> 5) && 2 * nt + hf < p.ds_pitch)
      p.dscale[(int64_t)((m0 >> 5) + pr) * p.ds_pitch + 2 * nt + hf] = s_do;
  }

  // ---- dO^T fragments (permlane32 swap), the image, and dZ^T[j, user] += W^T[j, item] . dO^T[item, user]
  const int ldi = b.counts[2];
  char *drow = p.dimg + (int64_t)m * ldi * 4 + (int64_t)(nh >> 5) * 128;
  const bool st_ok = m < p.rows_img;
  f32x16 acc2[KTM];
#pragma unroll
  for (int j = 0; j < KTM; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // this lane's items 16 s + 4 lh + {0..3} ("first") and 16 s + 8 + 4 lh + {0..3} ("second") of block i
      uint2 fh, fl, sh, sl;
      if constexpr (PLAIN) {
        rkp::plain4(make_float4(acc[i][8 * s + 0], acc[i][8 * s + 1], acc[i][8 * s + 2], acc[i][8 * s + 3]), fh, fl);
        rkp::plain4(make_float4(acc[i][8 * s + 4], acc[i][8 * s + 5], acc[i][8 * s + 6], acc[i][8 * s + 7]), sh, sl);
      } else {
      rkp::split4(make_float4(acc[i][8 * s + 0], acc[i][8 * s + 1], acc[i][8 * s + 2], acc[i][8 * s + 3]), s_do, fh, fl);
      rkp::split4(make_float4(acc[i][8 * s + 4], acc[i][8 * s + 5], acc[i][8 * s + 6], acc[i][8 * s + 7]), s_do, sh, sl);
      }
      // lanes < 32 keep `first` and take the partner's `first` (items + 4 .. + 7); lanes >= 32 take the
      // partner's `second` (items + 8 .. + 11) and keep theirs: first.upper <-> second.lower
      swap32(fh.x, sh.x); swap32(fh.y, sh.y); swap32(fl.x, sl.x); swap32(fl.y, sl.y);
      const uint4 hi4 = make_uint4(fh.x, fh.y, sh.x, sh.y), lo4 = make_uint4(fl.x, fl.y, sl.x, sl.y);
      // image: 8 consecutive items 32 i + 16 s + 8 lh .. + 7 of row m
      if (st_ok && nh + 32 * i < ldi) {
        char *d = drow + i * 128 + (16 * s + 8 * lh) * 2;
        *reinterpret_cast<uint4 *>(d) = hi4;
        *reinterpret_cast<uint4 *>(d + 64) = lo4;
      }
      const f16x8 dh = __builtin_bit_cast(f16x8, hi4), dl = __builtin_bit_cast(f16x8, lo4);
      const int row0 = 64 * hf + 32 * i + 16 * s + 8 * lh;
#pragma unroll
      for (int j = 0; j < KTM; ++j) {
        if constexpr (PLAIN) {
          const f16x8 ah = w_tr_frag(Wst + j * WB, row0, lane, 0);
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg::bf16x8, ah), __builtin_bit_cast(pg::bf16x8, dh),
                                                            acc2[j], 0, 0, 0);
        } else {
          const f16x8 ah = w_tr_frag(Wst + j * WB, row0, lane, 0), al = w_tr_frag(Wst + j * WB, row0, lane, 1);
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, dh, acc2[j], 0, 0, 0);
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, dl, acc2[j], 0, 0, 0);
          acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, dh, acc2[j], 0, 0, 0);
        }
      }
    }
  }
  // ---- the two item halves of a user group meet in LDS (over the W stages, which are done): the upper
  // half's wave leaves its partial there, the lower half's adds its own, and both then copy the 32 x h
  // block -- contiguous in the slab -- out in 16-byte pieces along the rows.
  // acc2[j][r] = hidden 32 j + (r & 3) + 8 (r >> 2) + 4 lh of user l31
  constexpr int PITCH = FD_PITCH(KTM);
  float *X = reinterpret_cast<float *>(smem) + pr * 32 * PITCH;
  const float inv2 = 1.0f / (s_do * p.scales[1]);
  __syncthreads();
  if (hf == 1) {
#pragma unroll
    for (int j = 0; j < KTM; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(X + l31 * PITCH + 32 * j + 8 * q + 4 * lh) =
            make_float4(acc2[j][4 * q] * inv2, acc2[j][4 * q + 1] * inv2, acc2[j][4 * q + 2] * inv2, acc2[j][4 * q + 3] * inv2);
  }
  __syncthreads();
  if (hf == 0) {
#pragma unroll
    for (int j = 0; j < KTM; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 *x = reinterpret_cast<float4 *>(X + l31 * PITCH + 32 * j + 8 * q + 4 * lh);
        const float4 u = *x;
        // (lower half first + upper half: one fixed order)
        *x = make_float4(acc2[j][4 * q] * inv2 + u.x, acc2[j][4 * q + 1] * inv2 + u.y, acc2[j][4 * q + 2] * inv2 + u.z,
                         acc2[j][4 * q + 3] * inv2 + u.w);
      }
  }
  __syncthreads();
  {
    const int h4 = p.h >> 2;
    const int rows = min(32, M - (m0 + 32 * pr));
    float4 *ws = reinterpret_cast<float4 *>(p.dz_ws + ((int64_t)nt * M + m0 + 32 * pr) * p.h);
    for (int idx = hf * 64 + lane; idx < rows * h4; idx += 128) {
      const int row = idx / h4, c = idx - row * h4;
      ws[idx] = *reinterpret_cast<const float4 *>(X + row * PITCH + 4 * c);
    }
  }
  if (tid == 0) {
    float ls = 0.f, gm = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { ls += misc[128 + w]; gm = fmaxf(gm, misc[136 + w]); }
    p.loss_part[t] = ls;
    atomicMax(reinterpret_cast<unsigned int *>(b.counts) + 8 + ((int)blockIdx.x & 63), __float_as_uint(gm));
  }
}

inline bool al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// the shapes the register-resident fused decode covers: MSE / logistic, 2, 4 or 7 k-tiles of 32 hidden
// units (h <= 224: 7 resident k-tiles of the W tile = 112 KB of LDS), below 1024 rows, slab workspace below 4 GB; RK_FDEC=0: off
// dZ slabs the fused decode leaves for rk_fdec_dz_reduce: one per 128-item column tile.  (A STREAMING form -- a workgroup
// walking a group of column tiles with the dZ accumulators in registers, one slab per group -- was built in round 5,
// measured slower wherever it applied and is kept as tools/probes/patches/r06_fdec_streaming_form.patch.)
int rk_fdec_slabs(int B, int n_cap) {
  (void)B;
  return rk_cdiv(n_cap, 128);
}

extern "C" int rk_fdec_dz_reduce(const float *dz_workspace, int32_t B, int32_t h, const rk_block_t *tgt,
                                 const float *Zact, int32_t act, float *dZ, void *stream_) {
  RK_REQUIRE(al16(dz_workspace) && al16(dZ), "operands must be 16-byte aligned");
  if (B == 0) return 0;
  return rk_splitk_reduce_tiles(dz_workspace, B, h, tgt->counts, rk_fdec_slabs(B, tgt->n_cap), 128, Zact, act, dZ,
                                stream_);
}

extern "C" int32_t rk_fdec_ok(int32_t B, int32_t h, int32_t n_cap, int32_t loss_kind) {
  static const int on = [] { const char *e = getenv("RK_FDEC"); return (e && atoi(e) == 0) ? 0 : 1; }();
  const int kt = rkp::kp_of(h) / 32;
  // (plain bf16 operands, RK_GEMM_PREC=bf16: the same kernel with one product -- whole single-process steps only,
  // rk_ae_train_step decides)
  return on && rk_pg_enabled() && rk_gemm_split16() &&
         (loss_kind == RK_LOSS_MSE || loss_kind == RK_LOSS_BCE) && h % 4 == 0 && h <= 224 &&
         B < 1024 &&
         (kt == 2 || kt == 4 || kt == 7) &&
         (int64_t)rk_fdec_slabs(B, n_cap) * B * h * (int64_t)sizeof(float) <= ((int64_t)4 << 30) ? 1 : 0;
}
extern "C" int64_t rk_fdec_workspace_bytes(int32_t B, int32_t h, int32_t n_cap) {
  // (slabs x rows is not monotone in the batch size: the workspace of a capacity covers every batch below it)
  int64_t rows = 0;
  for (int b = 1; b <= B; ++b) rows = std::max(rows, (int64_t)rk_fdec_slabs(b, n_cap) * b);
  return rows * h * (int64_t)sizeof(float);
}

extern "C" int rk_fdec_loss_dz(const rk_planes_t *pl, int32_t B, const rk_block_t *tgt, int32_t row_off,
                               const float *b_de, int32_t loss_kind, float confidence, float inv_B,
                               void *dO_img, int32_t rows_img, float *dO_scales, float *loss_part,
                               float *dz_workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(pl && B <= pl->B_cap && tgt->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  RK_REQUIRE(rk_fdec_ok(B, pl->h, tgt->n_cap, loss_kind), "shape / loss outside the fused decode (rk_fdec_ok)");
  RK_REQUIRE(tgt->implicit || tgt->pref_rc != nullptr, "explicit values need pref_rc");
  RK_REQUIRE(dO_img && dO_scales && al16(dO_img) && al16(dz_workspace) && rows_img >= ((B + 31) & ~31),
             "dO image: 16-byte aligned, round_up(B, 32) rows; a scale table");
  if (B == 0) return 0;
  FdecP p = {};
  p.zimg = (const char *)pl->z; p.wimg = (const char *)pl->w; p.scales = pl->scales;
  p.KT = rkp::kp_of(pl->h) / 32;
  p.M = B; p.Ndev = tgt->counts; p.z_rows = pl->B_cap; p.w_rows = pl->n_cap;
  p.blk = *tgt; p.row_off = row_off; p.confidence = confidence; p.inv_B = inv_B;
  p.bias = b_de; p.bidx = tgt->items; p.loss_part = loss_part;
  p.dimg = (char *)dO_img; p.rows_img = rows_img; p.dscale = dO_scales; p.ds_pitch = rk_cdiv(tgt->n_cap, 64);
  p.dz_ws = dz_workspace; p.h = pl->h;
  const int grid = rk_cdiv(rk_cdiv(B, 128) * rk_cdiv(tgt->n_cap, 128), 8) * 8;
  const bool plain = rk_gemm_plain_bf16() != 0;
#define GO(KTM, LOSS)                                                                                          \
  do {                                                                                                         \
    if (plain) {                                                                                               \
      auto k = fdec_kernel<KTM, LOSS, true>;                                                                   \
      static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (attr != hipSuccess) { rk_set_error("LDS attribute"); return -1; }                                    \
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), FD_LDS(KTM), stream, p);                                    \
    } else {                                                                                                   \
      auto k = fdec_kernel<KTM, LOSS>;                                                                         \
      static const hipError_t attr = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (attr != hipSuccess) { rk_set_error("LDS attribute"); return -1; }                                    \
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), FD_LDS(KTM), stream, p);                                    \
    }                                                                                                          \
  } while (0)
#define BY_KT(LOSS)                                                                                            \
  do { if (p.KT == 2) GO(2, LOSS); else if (p.KT == 4) GO(4, LOSS); else GO(7, LOSS); } while (0)
  if (loss_kind == RK_LOSS_MSE) BY_KT(pg::LOSS_MSE); else BY_KT(pg::LOSS_BCE);
#undef BY_KT
#undef GO
  RK_CHECK_LAUNCH("fdec_loss_dz");
  return 0;
}
