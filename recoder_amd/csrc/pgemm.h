// Pipelined contraction on pre-split operand planes (round 4; csrc/planes.h describes the images).
//
//   C[M, N] = sum_k A(m, k) . B(k, n)     every operand a plane image of fp16 pairs  s.x = hi + lo,
//   a.b = lo.hi + hi.lo + hi.hi accumulated in fp32 on v_mfma_f32_32x32x16_f16 (the arithmetic and the
//   per-accumulator order of decode16.hip: with equal operands the results agree bit for bit).
//
// What is new against decode16.hip's k-loop (write LDS -> barrier -> read LDS -> MFMA, serial):
//   * the global -> LDS copy is LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write
//     pass; the copy of k-tile t + 1 is issued BEFORE the MFMAs of k-tile t and lands under them -- two
//     LDS stages, ONE barrier per k-tile;
//   * 256 x 256 tiles on 8 waves (wave tile 128 x 64: 48 MFMAs per 24 ds_read_b128 and k-tile) where
//     the problem is large, 128 x 128 on 4 waves (two workgroups per CU) where it is not;
//   * an operand whose contraction index runs along the image's ROWS (W in dZ = dO . W, both operands
//     of dW = dO^T . Z) is read with ds_read_b64_tr_b16, the LDS transpose read of gfx950 -- so the
//     three contractions of a step need THREE images (Z, W[items], dO) and no transposed copy of any
//     of them (round 3 wrote W^T and Z^T images as well: 2 x 100 MB per step at C5's sizes).
//
// LDS images (lane-linear for the DMA; the bank swizzle is applied to the SOURCE address and to the
// fragment read, never to the destination):
//   K along the columns ("KC"): stage = [R rows][128 B line]; 16-byte slot s of row r sits at slot
//     s ^ kc_sw(r), a permutation of the row pair's index (r >> 1) & 7: the 16 lanes of a ds_read_b128 group hit 16
//     different slots of the 256-B bank row.  (Round 5: bit 0 of the pair index goes to bit 2 of the mask, so that the
//     rows r and r + 2 of a TRANSPOSE read of this layout -- csrc/fdecode.hip reads its resident W stage that way for
//     dZ: 4 rows x 64 bytes per half-wave -- sit in different 64-byte halves; with the plain index they shared 16 banks.)
//   K along the rows ("TR"):    stage = [32 k-rows][C / 32 lines]; byte o of k-row kr sits at
//     o ^ ((kr & 1) << 6 | (kr & 2) << 6): the 32 lanes of a ds_read_b64_tr_b16 half-wave (4 k-rows x
//     64 bytes) cover the 256-B bank row exactly once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int LINE = 128;

struct Opnd {
  const char *img;      // plane image: [rows][lines] x 128 B
  int64_t pitch;        // bytes per image row
  int lines;            // lines per image row
  int rows;             // allocated image rows (KC: clamp of the tile's rows; TR: a multiple of 32)
};

// Per-tile power-of-two scales of the A operand when it is a dO image written by a loss epilogue (the
// tile's maximum is only known to the workgroup that computed it: every (gr x gc) granule of dO carries
// its own scale, tab[(row / gr) * pitch + col / gc]).  The consumer keeps its accumulators in the scale
// of the granule it is in and multiplies them by new / old -- a power of two: exact -- when the k-loop
// crosses into another one.  mode 1: A(m, k) = dO[m][k] (dZ: rows = m, k = columns); mode 2: A(m, k) =
// dO[k][m] (dW: rows = k).
struct Rescale {
  const float *tab;
  int gr, gc, pitch;
  int mode;
};

struct Core {
  Opnd a, b;
  int M, N, K;                         // problem size (a capacity where the live size is on the device)
  const int32_t *Mdev, *Ndev, *Kdev;   // nullable: live sizes on the device
  const int32_t *a_ld_dev;             // nullable: A is a dO image whose row pitch (4 * ld bytes) and line
                                       // count (ld / 32) are device-resident (rk_block_t.counts[2])
  int splits;                          // split-K: slabs (auto_slots > 0: the most there can be)
  int auto_slots;                      // > 0: the slab count follows the LIVE tile count on the device --
                                       // min(splits, auto_slots / live tiles, k-tiles / 2), at least 1 --
  int32_t *splits_out;                 // and is published here (the consumer of the slabs reads it)
  Rescale rs;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// slot mask of row r in a KC stage (see the header comment)
__device__ __forceinline__ int kc_sw(const int r) {
  const int x = (r >> 1) & 7;
  return ((x & 1) << 2) | (x & 2) | (x >> 2);
}

// A wave-uniform word of a table an EARLIER launch wrote, through the scalar cache (s_load: lgkmcnt).  As a
// vector load it sits on the VM counter behind the LDS-DMAs of the next k-tile, and the wait for its value
// drains them: the k-loops that rescale (dW, dZ on a dO image) ran their MFMAs only AFTER the next tile had
// landed -- no overlap at all (round 5: found in the ISA, `global_load_dword; s_waitcnt vmcnt(0)` right
// behind the six global_load_lds of the next tile).
typedef const __attribute__((address_space(4))) float kconst_float;
__device__ __forceinline__ float sload(const float *tab, const int idx) {
  return *(kconst_float *)(tab + rfl(idx));
}

// ------------------------------------------------------------------ staging (global -> LDS, LDS-DMA)
// R = tile extent along the operand's non-K index (rows of a KC operand, columns of a TR operand),
// NW waves; the stage is R * 128 bytes = R / 8 wave-instructions of 1 KB, Q = R / 8 / NW per wave
template <int R, bool TR, int NW>
struct Stager {
  static constexpr int Q = R / 8 / NW;       // (NW = number of ISSUING waves)
  static_assert(R % (8 * NW) == 0, "tile extent must be a multiple of 8 * issuing waves");
  const char *src[Q];
  int64_t step;

  // x0: first row (KC) / first column (TR) of the tile; kt0: first k-tile; lim: live rows (KC clamp)
  __device__ __forceinline__ void init(const Opnd &o, const int x0, const int kt0, const int lim,
                                       const int wave, const int lane) {
    if (!TR) {
      step = LINE;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int qg = q * NW + wave;
        const int r = qg * 8 + (lane >> 3);
        const int s = (lane & 7) ^ kc_sw(r);
        const int row = min(x0 + r, lim - 1);
        src[q] = o.img + (int64_t)row * o.pitch + (int64_t)kt0 * LINE + s * 16;
      }
    } else {
      constexpr int RS = R * 4;            // bytes per k-row of the stage
      step = 32 * o.pitch;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int qg = q * NW + wave;
        const int d = qg * 1024 + lane * 16;
        const int kr = d / RS, off = d % RS;
        const int o2 = off ^ (((kr & 1) << 6) | ((kr & 2) << 6));
        const int line = min((x0 >> 5) + (o2 >> 7), o.lines - 1);
        src[q] = o.img + ((int64_t)kt0 * 32 + kr) * o.pitch + (int64_t)line * LINE + (o2 & 127);
      }
    }
  }
  __device__ __forceinline__ void issue(char *stage, const int wave) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int qg = q * NW + wave;
      __builtin_amdgcn_global_load_lds((gbl_void *)src[q], (lds_void *)(stage + qg * 1024), 16, 0, 0);
      src[q] += step;
    }
  }
};

// ---- pieces of the deep-ring k-loop (gemm_body, NS > 2) ----------------------------------------------------
// The compiler drains the VM counter (s_waitcnt vmcnt(0)) in front of the first ds_read_b64_tr_b16 INTRINSIC
// behind an LDS-DMA (it cannot tell the stage being read from the stage being filled; the plain ds_read_b128 of
// the KC operands it does tell apart) and in front of every __syncthreads(): with TR operands the "copy lands
// under the MFMAs" pipeline above never overlapped anything, and a ring deeper than two stages is impossible.
// The deep ring therefore reads its fragments with the instruction as inline asm (no memory operand for the
// pass that inserts waits to reason about), counts its own waits and crosses a raw s_barrier.
template <int OFF>
__device__ __forceinline__ s16x4 ds_tr_asm(const uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `ahead` (0 .. MAXA) later tiles' DMAs of this wave (QPW instructions each) are outstanding
template <int MAXA, int QPW>
__device__ __forceinline__ void wait_tiles(const int ahead) {
  if constexpr (MAXA == 0) wait_vm<0>();
  else {
    if (ahead >= MAXA) wait_vm<MAXA * QPW>();
    else wait_tiles<MAXA - 1, QPW>(ahead);
  }
}
__device__ __forceinline__ uint32_t lds_addr(const char *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

// ------------------------------------------------------------------ fragment reads
// one 32 x 32 x 16 MFMA operand of this lane: 8 consecutive k (8 * lh .. + 7 of k-step ks) of row /
// column (tile * 32 + l31); plane 0 = hi, 1 = lo
struct FragKC {
  int base;                 // l31 * 128
  int sw;                   // kc_sw(l31)
  int lh;
  __device__ __forceinline__ void init(const int lane) {
    const int l31 = lane & 31;
    base = l31 * LINE; sw = kc_sw(l31); lh = lane >> 5;
  }
  __device__ __forceinline__ f16x8 load(const char *S, const int tile, const int ks, const int plane) const {
    const int slot = (ks * 2 + lh + plane * 4) ^ sw;
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(S + tile * 4096 + base + slot * 16));
  }
};

template <int R>
struct FragTR {
  static constexpr int RS = R * 4;
  int row_off;              // ((8 * lh) + (t >> 2)) * RS
  int col;                  // 32 g + 8 (t & 3)
  int swz;                  // lane-constant XOR mask (bits 6, 7): k-row & 3 == t >> 2
  __device__ __forceinline__ void init(const int lane) {
    const int t = lane & 15, g = (lane >> 4) & 1, lh = lane >> 5;
    row_off = (8 * lh + (t >> 2)) * RS;
    col = 32 * g + 8 * (t & 3);
    swz = (((t >> 2) & 1) << 6) | (((t >> 3) & 1) << 7);
  }
  // deep ring: lane offset of (tile, plane) inside a stage; the two halves of k-step KS by immediate offsets
  __device__ __forceinline__ int off(const int tile, const int plane) const {
    return ((tile * LINE + plane * 64 + col) ^ swz) + row_off;
  }
  // (the two halves stay apart until the wait for them has been passed: a register copy in between would read
  // them before they are written -- nothing interlocks a VALU read of a register an LDS read is still to fill)
  template <int KS>
  static __device__ __forceinline__ void load_asm(const uint32_t a, s16x4 &v0, s16x4 &v1) {
    v0 = ds_tr_asm<KS * 16 * RS>(a);
    v1 = ds_tr_asm<KS * 16 * RS + 4 * RS>(a);
  }
  __device__ __forceinline__ f16x8 load(const char *S, const int tile, const int ks, const int plane) const {
    const int o = ((tile * LINE + plane * 64 + col) ^ swz) + row_off + ks * 16 * RS;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(S + o));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(S + o + 4 * RS));
    const s16x8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
  }
};

template <int R, bool TR> struct FragSel { typedef FragKC type; };
template <int R> struct FragSel<R, true> { typedef FragTR<R> type; };

// ------------------------------------------------------------------ the tile loop
// Geometry of one workgroup's tile, handed to the epilogue
struct Tile {
  int m0, n0;          // origin
  int M, N;            // live sizes
  int mt, nt, split;   // tile coordinates, K slab
  int tm, tn;          // tile counts
  int t;               // linear live-tile index
  int wm, wn, lane, wave;
};

// XCD-aware order (decode16.hip): workgroup L runs on XCD L % 8 and takes a contiguous chunk of the
// LIVE tile list; consecutive tiles share the B panel (mt fastest)
__device__ __forceinline__ bool tile_of(const int L, const int total, int &t) {
  const int chunk = (total + 7) >> 3;
  t = (L & 7) * chunk + (L >> 3);
  return (L >> 3) < chunk && t < total;
}

// VAR (tuning variants, bit mask): 1 = the DMAs of a stage are issued by the first half of the waves only
// (the other half -- their partners on the SIMDs -- start on the MFMAs at once); 2 = s_setprio(1) around
// the MFMA clusters
// NS: LDS stages (2: the loop above; > 2: the deep ring -- both operands TR, NS - 1 k-tiles in flight)
// PLAIN (RK_GEMM_PREC=bf16, the ring loop only): the images hold ONE bf16 value per element in their hi halves (scale 1,
// lo halves zero) -- one product on v_mfma_f32_32x32x16_bf16 instead of three on the f16 instruction, the lo planes
// are neither read from LDS nor multiplied
template <int BM, int BN, int WM, int WN, bool ATR, bool BTR, class Epi, int VAR = 0, bool RS = false, int NS = 2, bool PLAIN = false>
__device__ __forceinline__ void gemm_body(const Core &p, const typename Epi::Args &ea, const int L, char *smem) {
  static_assert(!PLAIN || (VAR & 256) != 0, "plain bf16 operands: the ring k-loop only");
  constexpr int NW = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NI = (VAR & 1) ? NW / 2 : NW;          // issuing waves
  constexpr int A_BYTES = BM * LINE, STAGE = (BM + BN) * LINE;
  const int M = p.Mdev ? *p.Mdev : p.M, N = p.Ndev ? *p.Ndev : p.N, K = p.Kdev ? *p.Kdev : p.K;
  Tile T;
  T.M = M; T.N = N;
  T.tm = (M + BM - 1) / BM; T.tn = (N + BN - 1) / BN;
  const int per_split = T.tm * T.tn;
  const int nk_all = (K + 31) >> 5;
  int splits = p.splits;
  if (p.auto_slots > 0) {
    splits = min(splits, max(1, p.auto_slots / max(1, per_split)));
    splits = min(splits, max(1, ((K + 31) >> 5) >> 1));
    if (L == 0 && threadIdx.x == 0 && p.splits_out) *p.splits_out = splits;
  }
  if (!tile_of(L, per_split * splits, T.t)) return;
  T.split = T.t / per_split;
  const int rt = T.t % per_split;
  T.mt = rt % T.tm; T.nt = rt / T.tm;
  T.m0 = T.mt * BM; T.n0 = T.nt * BN;
  const int tid = threadIdx.x;
  T.lane = tid & 63; T.wave = rfl(tid >> 6);
  T.wm = T.wave / WN; T.wn = T.wave % WN;
  // k range of this slab, in k-tiles of 32
  const int kchunk = (nk_all + splits - 1) / splits;
  const int kt0 = T.split * kchunk;
  const int nk = min(kchunk, nk_all - kt0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (probe, VAR & 128 -- a negative result kept for the record: the FIRST workgroup of every CU starts up
  // to 7/8 of a tile late, by groups of 32 CUs, so that the CUs leave lockstep and one group's epilogue
  // stores run under the others' MFMAs: 2032 vs 1977 us at C5's B = 4096 -- the epilogue is not HBM
  // contention between CUs, it is each workgroup's own store / VALU time with nothing beside it on its CU)
  if ((VAR & 128) && L < 256 && per_split * splits > 512) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long wait = (unsigned long long)(((L >> 3) & 7) * nk_all) * 40;
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }

  // scale of the dO granule each 32-row block of the wave's accumulators is in (Rescale)
  float cur[TM];
  int rs_fix[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { cur[i] = 1.0f; rs_fix[i] = 0; }
  if (RS && nk > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rb = min(T.m0 + (T.wm * TM + i) * 32, M - 1);
      rs_fix[i] = p.rs.mode == 1 ? (rb / p.rs.gr) * p.rs.pitch : rb / p.rs.gc;
      cur[i] = sload(p.rs.tab, rs_fix[i] + (p.rs.mode == 1 ? (kt0 * 32) / p.rs.gc : ((kt0 * 32) / p.rs.gr) * p.rs.pitch));
    }
  }

  if (nk > 0) {
    Opnd oa = p.a;
    if (p.a_ld_dev) { oa.lines = *p.a_ld_dev >> 5; oa.pitch = (int64_t)oa.lines * LINE; }
    Stager<BM, ATR, NI> sa;
    Stager<BN, BTR, NI> sb;
    const bool issuer = T.wave < NI;
    const int iw = issuer ? T.wave : 0;
    sa.init(oa, T.m0, kt0, ATR ? 0 : min(M, p.a.rows), iw, T.lane);
    sb.init(p.b, T.n0, kt0, BTR ? 0 : min(N, p.b.rows), iw, T.lane);
    typename FragSel<BM, ATR>::type fa;
    typename FragSel<BN, BTR>::type fb;
    fa.init(T.lane);
    fb.init(T.lane);
    if constexpr (NS > 2 || (VAR & 256)) {           // (VAR & 256: this loop at TWO stages -- one tile in flight)
      static_assert((ATR || BTR) && (VAR & ~256) == 0, "the ring loop: an operand read along its rows, no probe variants");
      constexpr int QPW = Stager<BM, ATR, NI>::Q + Stager<BN, BTR, NI>::Q;
      static_assert((NS - 2) * QPW <= 63, "vmcnt is a 6-bit counter");
      // small wave tiles hold the fragments of BOTH k-steps of a k-tile (one wait per k-tile); large ones (the
      // 128 x 64 wave tile of the 256 x 256 workgroup: 48 VGPRs per k-step) one k-step at a time
      constexpr bool BOTH = (TM + TN) * 16 <= 48;
#pragma unroll
      for (int st = 0; st < NS - 1; ++st)
        if (st < nk) {
          sa.issue(smem + st * STAGE, iw);
          sb.issue(smem + st * STAGE + A_BYTES, iw);
        }
      // lane offsets of the wave's TR fragments inside a stage (KC operands: FragKC::load on the stage pointer)
      int oa_[TM][2], ob_[TN][2];
      if constexpr (ATR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { oa_[i][0] = fa.off(T.wm * TM + i, 0); oa_[i][1] = fa.off(T.wm * TM + i, 1); }
      }
      if constexpr (BTR) {
#pragma unroll
        for (int j = 0; j < TN; ++j) { ob_[j][0] = fb.off(T.wn * TN + j, 0) + A_BYTES; ob_[j][1] = fb.off(T.wn * TN + j, 1) + A_BYTES; }
      }
      const uint32_t base = lds_addr(smem);
      int stage = 0;
      for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed: this wave's DMAs of it (at most NS - 2 later tiles may still fly), then the barrier;
        // behind the barrier every wave is also done with the stage of tile kt - 1 -- the next DMA's target
        wait_tiles<NS - 2, QPW>(nk - 1 - kt);
        asm volatile("s_barrier" ::: "memory");
        if (kt + NS - 1 < nk) {
          const int st = stage == 0 ? NS - 1 : stage - 1;
          sa.issue(smem + st * STAGE, iw);
          sb.issue(smem + st * STAGE + A_BYTES, iw);
        }
        if (RS) {
          const int k = (kt0 + kt) * 32;
          const int kv = p.rs.mode == 1 ? k / p.rs.gc : (k / p.rs.gr) * p.rs.pitch;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float sn = sload(p.rs.tab, rs_fix[i] + kv);
            if (sn != cur[i]) {
              const float f = sn / cur[i];
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
              cur[i] = sn;
            }
          }
        }
        const uint32_t sbase = base + (uint32_t)(stage * STAGE);
        const char *SA = smem + stage * STAGE + (T.wm * TM) * 4096;          // (KC operands)
        const char *SB = smem + stage * STAGE + A_BYTES + (T.wn * TN) * 4096;
        constexpr int NKS = BOTH ? 2 : 1;
#pragma unroll
        for (int k0 = 0; k0 < 2; k0 += NKS) {
          // fragment halves of the TR operands: [k-step][plane][tile][half]; whole fragments of the KC operands
          s16x4 ra[NKS][2][TM][2], rb[NKS][2][TN][2];
          f16x8 ah[NKS][TM], al[NKS][TM], bh[NKS][TN], bl[NKS][TN];
#pragma unroll
          for (int q = 0; q < NKS; ++q) {
            const int ks = k0 + q;
#pragma unroll
            for (int pl = 0; pl < (PLAIN ? 1 : 2); ++pl) {
#pragma unroll
              for (int i = 0; i < TM; ++i) {
                if constexpr (ATR) {
                  if (ks == 0) FragTR<BM>::template load_asm<0>(sbase + oa_[i][pl], ra[q][pl][i][0], ra[q][pl][i][1]);
                  else FragTR<BM>::template load_asm<1>(sbase + oa_[i][pl], ra[q][pl][i][0], ra[q][pl][i][1]);
                } else {
                  if (pl == 0) ah[q][i] = fa.load(SA, i, ks, 0); else al[q][i] = fa.load(SA, i, ks, 1);
                }
              }
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                if constexpr (BTR) {
                  if (ks == 0) FragTR<BN>::template load_asm<0>(sbase + ob_[j][pl], rb[q][pl][j][0], rb[q][pl][j][1]);
                  else FragTR<BN>::template load_asm<1>(sbase + ob_[j][pl], rb[q][pl][j][0], rb[q][pl][j][1]);
                } else {
                  if (pl == 0) bh[q][j] = fb.load(SB, j, ks, 0); else bl[q][j] = fb.load(SB, j, ks, 1);
                }
              }
            }
          }
          // (the asm reads are invisible to the compiler's LGKM bookkeeping: the wait for them is ours; its own
          // waits for the KC operands' ds_read_b128 can only come out stronger than needed.  The scale's s_load
          // above was waited for by its compare.)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define PG_CAT(X) __builtin_bit_cast(f16x8, __builtin_shufflevector(X[0], X[1], 0, 1, 2, 3, 4, 5, 6, 7))
#pragma unroll
          for (int q = 0; q < NKS; ++q) {
            if constexpr (ATR) {
#pragma unroll
              for (int i = 0; i < TM; ++i) {
                if constexpr (PLAIN) {
                  asm volatile("" : "+v"(ra[q][0][i][0]), "+v"(ra[q][0][i][1]));
                  ah[q][i] = PG_CAT(ra[q][0][i]);
                } else {
                asm volatile("" : "+v"(ra[q][0][i][0]), "+v"(ra[q][0][i][1]), "+v"(ra[q][1][i][0]), "+v"(ra[q][1][i][1]));
                ah[q][i] = PG_CAT(ra[q][0][i]); al[q][i] = PG_CAT(ra[q][1][i]);
                }
              }
            }
            if constexpr (BTR) {
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                if constexpr (PLAIN) {
                  asm volatile("" : "+v"(rb[q][0][j][0]), "+v"(rb[q][0][j][1]));
                  bh[q][j] = PG_CAT(rb[q][0][j]);
                } else {
                asm volatile("" : "+v"(rb[q][0][j][0]), "+v"(rb[q][0][j][1]), "+v"(rb[q][1][j][0]), "+v"(rb[q][1][j][1]));
                bh[q][j] = PG_CAT(rb[q][0][j]); bl[q][j] = PG_CAT(rb[q][1][j]);
                }
              }
            }
          }
#undef PG_CAT
#pragma unroll
          for (int q = 0; q < NKS; ++q) {
            if constexpr (PLAIN) {
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[q][i]),
                                                                      __builtin_bit_cast(bf16x8, bh[q][j]), acc[i][j], 0, 0, 0);
              continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][i], bh[q][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bl[q][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bh[q][j], acc[i][j], 0, 0, 0);
          }
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
      }
      __syncthreads();
    } else {
    if (issuer) {
      sa.issue(smem, iw);
      sb.issue(smem + A_BYTES, iw);
    }
#define PG_MFMA3(AH, AL, BH, BL)                                                                   \
  do {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0);      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0);      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);      \
  } while (0)
    // VAR & 32 ("ping-pong"): the two waves of a SIMD (wave w and w + NW / 2) run half a k-tile apart -- the
    // late half carries the fragments of its second k-step across the barrier and multiplies them FIRST
    // behind it, while the early half issues its DMAs and waits for its first fragments; then the roles
    // swap.  Same MFMAs, same order per accumulator: only their placement around the barrier moves.
    const bool late = (VAR & 32) && T.wave >= NW / 2;
    f16x8 pah[TM], pal[TM], pbh[TN], pbl[TN];
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed (every wave waits for its own DMAs, then the barrier) and every wave is
      // done reading the other stage (its fragment reads of iteration kt - 1 are complete)
      if (!((VAR & 64) && kt > 0)) __syncthreads();
      if ((VAR & 64) && kt > 0) {         // (probe: the MFMA sequence alone, on the first tile's fragments)
        PG_MFMA3(pah, pal, pbh, pbl);
        PG_MFMA3(pah, pal, pbh, pbl);
        continue;
      }
      if ((VAR & 32) && late && kt > 0) PG_MFMA3(pah, pal, pbh, pbl);
      if (kt + 1 < nk && issuer && !((VAR & 16) && kt >= 1) && !(VAR & 64)) {
        char *nx = smem + ((kt + 1) & 1) * STAGE;
        sa.issue(nx, iw);
        sb.issue(nx + A_BYTES, iw);
      }
      const char *SA = smem + (kt & 1) * STAGE + (T.wm * TM) * 4096 * (ATR ? 0 : 1);
      const char *SB = smem + (kt & 1) * STAGE + A_BYTES + (T.wn * TN) * 4096 * (BTR ? 0 : 1);
      const int ta = ATR ? T.wm * TM : 0, tb = BTR ? T.wn * TN : 0;
      if (VAR & 4) continue;            // (probe: DMA + barrier only)
      if (RS) {
        const int k = (kt0 + kt) * 32;
        const int kv = p.rs.mode == 1 ? k / p.rs.gc : (k / p.rs.gr) * p.rs.pitch;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float sn = sload(p.rs.tab, rs_fix[i] + kv);
          if (sn != cur[i]) {                        // (wave-uniform: a granule boundary)
            const float f = sn / cur[i];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
            cur[i] = sn;
          }
        }
      }
      {
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = fa.load(SA, ta + i, 0, 0); al[i] = fa.load(SA, ta + i, 0, 1); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = fb.load(SB, tb + j, 0, 0); bl[j] = fb.load(SB, tb + j, 0, 1); }
        if (VAR & 2) __builtin_amdgcn_s_setprio(1);
        PG_MFMA3(ah, al, bh, bl);
        if (VAR & 2) __builtin_amdgcn_s_setprio(0);
      }
      {
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = fa.load(SA, ta + i, 1, 0); al[i] = fa.load(SA, ta + i, 1, 1); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = fb.load(SB, tb + j, 1, 0); bl[j] = fb.load(SB, tb + j, 1, 1); }
        if (((VAR & 32) && late) || (VAR & 64)) {
#pragma unroll
          for (int i = 0; i < TM; ++i) { pah[i] = ah[i]; pal[i] = al[i]; }
#pragma unroll
          for (int j = 0; j < TN; ++j) { pbh[j] = bh[j]; pbl[j] = bl[j]; }
        } else {
          if (VAR & 2) __builtin_amdgcn_s_setprio(1);
          PG_MFMA3(ah, al, bh, bl);
          if (VAR & 2) __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    if ((VAR & 32) && late && !(VAR & 4)) PG_MFMA3(pah, pal, pbh, pbl);
#undef PG_MFMA3
    }
  }
  if (VAR & 8) {                        // (probe: no epilogue -- keep the accumulators alive)
    float x = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) x += acc[i][j][r];
    if (x == 1.2345e-30f) *reinterpret_cast<float *>(smem) = x;
    return;
  }
  Epi::template run<BM, BN, TM, TN>(ea, T, acc, smem, cur);
}

template <int BM, int BN, int WM, int WN, bool ATR, bool BTR, class Epi, int VAR = 0, bool RS = false, int NS = 2>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const Core p, const typename Epi::Args ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_body<BM, BN, WM, WN, ATR, BTR, Epi, VAR, RS, NS>(p, ea, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------ plain store epilogue
// C[m][n] = acc * scale (+ slab offset per K split): a lane holds 16 rows of ONE column, 32 lanes = 128
// contiguous bytes per store instruction
struct EpiStore {
  struct Args {
    float *C;
    int64_t ldc;
    int64_t slab_stride;     // floats between K slabs
    const float *scales;     // nullable: [0] * [1] = product of the operand scales
    float scale;             // used when scales == null
  };
  template <int BM, int BN, int TM, int TN>
  static __device__ __forceinline__ void run(const Args &e, const Tile &T, f32x16 (&acc)[TM][TN], char *,
                                             const float (&cur)[TM]) {
    const float inv = e.scales ? 1.0f / (e.scales[0] * e.scales[1]) : e.scale;
    float *C = e.C + (int64_t)T.split * e.slab_stride;
    const int l31 = T.lane & 31, lh = T.lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = T.n0 + (T.wn * TN + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = T.m0 + (T.wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < T.M && n < T.N) C[(int64_t)m * e.ldc + n] = acc[i][j][r] * (inv / cur[i]);
        }
      }
  }
};

// workgroups of a launch over tiles_cap tiles: with a device-chosen slab count the live workgroups are
// the first tiles(live) * splits(live) of the grid, never more than max(tiles_cap, auto_slots)
inline int grid_of(const Core &p, int tiles_cap) {
  const int wgs = p.auto_slots > 0 ? (tiles_cap > p.auto_slots ? tiles_cap : (tiles_cap * p.splits < p.auto_slots ? tiles_cap * p.splits : p.auto_slots))
                                   : tiles_cap * p.splits;
  return ((wgs + 7) / 8) * 8;
}

template <int BM, int BN, int WM, int WN, bool ATR, bool BTR, class Epi, int VAR = 0, bool RS = false, int NS = 2>
inline hipError_t launch(const Core &p, const typename Epi::Args &ea, int tiles_cap, hipStream_t s) {
  constexpr int LDS = NS * (BM + BN) * LINE;
  auto k = gemm_kernel<BM, BN, WM, WN, ATR, BTR, Epi, VAR, RS, NS>;
  static const hipError_t attr =
      hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return attr;
  const int grid = grid_of(p, tiles_cap);
  hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), LDS, s, p, ea);
  return hipGetLastError();
}

}  // namespace pg
