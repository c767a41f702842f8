// fp32 MFMA GEMMs of the hot path (gfx950: v_mfma_f32_32x32x2_f32, exact fp32
// fma chains -> parity with the reference's fp32 CPU path to rounding).
//
//   decode + loss : O[B,n_t] = Z[B,h] . W_de[T]^T + b_de[T]   (reference
//                   nn.py:271-280) with the loss and dLoss/dO fused into the
//                   epilogue (losses.py:43-47, BCEWithLogits): the logits never
//                   go to HBM for MSE/BCE, only dO does.
//   bwd dZ        : dZ[B,h]   = dO[B,n_t] . W_de[T]      (split-K over n_t)
//   bwd dW        : G[n_t,h]  = dO^T . Z
//   hidden layers : nn.Linear stack fwd/bwd (nn.py:242-249)
//
// One templated LDS-tiled kernel: block = 4 waves (WM x WN), each wave owns
// TM x TN tiles of 32x32, BK = 16.  Operands are staged global -> registers ->
// LDS (double buffered; next tile's global loads are issued before the MFMAs
// of the current one).  K-contiguous operands sit in LDS as [row][20] floats
// (80-B stride = odd multiple of 16 B -> conflict-free ds_read_b128: one read
// feeds 4 MFMA k-steps because the two lane halves take k = {0..3} / {4..7} of
// each 8-group); k-major operands sit as [16][tile] and are read with b32.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_LOSS = 1, EPI_SPLITK = 2 };

struct GemmP {
  const float *A;
  const float *Bm;
  const int32_t *bidx;      // gather rows of the B operand (item ids) or null
  int lda, ldb;
  const int32_t *lda_dev, *ldb_dev;   // leading dimension read from the device when non-null
  int M, N, K;              // host sizes (capacities when *_dev given)
  const int32_t *Mdev, *Ndev, *Kdev;
  int tiles_m;              // host: ceil(Mcap / BM)
  int kchunk;               // K range per blockIdx.y
  int a_vec, b_vec;         // 16-B loads allowed
  // store epilogue
  float *C;
  int ldc;                  // <=0 : read ld from ld_dev
  const int32_t *ld_dev;
  const float *bias;        // per output column (null = none)
  int bias_gather;          // bias index = bidx[n]
  int act;
  int accumulate;
  // loss epilogue
  rk_block_t blk;
  int row_off;
  int loss_kind;
  float confidence, inv_B;
  float *loss_part;
};

__device__ __forceinline__ float4 load4(const float *p, int valid, bool vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid >= 4 && vec) {
    r = *reinterpret_cast<const float4 *>(p);
  } else if (valid > 0) {
    r.x = p[0];
    if (valid > 1) r.y = p[1];
    if (valid > 2) r.z = p[2];
    if (valid > 3) r.w = p[3];
  }
  return r;
}

template <int WM, int WN, int TM, int TN, int AMODE, int BMODE, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, BK = 16;
  constexpr int A_SZ = (AMODE == 0) ? BM * 20 : BK * BM;
  constexpr int B_SZ = (BMODE == 0) ? BN * 20 : BK * BN;
  constexpr int A_F4 = BM * BK / 4, B_F4 = BN * BK / 4;      // float4 per tile
  constexpr int A_PT = (A_F4 + 255) / 256, B_PT = (B_F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

  const int M = p.Mdev ? *p.Mdev : p.M;
  const int N = p.Ndev ? *p.Ndev : p.N;
  const int K = p.Kdev ? *p.Kdev : p.K;
  const int lda = p.lda_dev ? *p.lda_dev : p.lda;
  const int ldb = p.ldb_dev ? *p.ldb_dev : p.ldb;
  const int mt = blockIdx.x % p.tiles_m, nt = blockIdx.x / p.tiles_m;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= M || n0 >= N) return;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(K, kbeg + p.kchunk);
  if (kbeg >= kend) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_PT], rb[B_PT];

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int idx = tid + i * 256;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < A_F4) {
        if (AMODE == 0) {
          const int row = idx >> 2, q = idx & 3;
          const int m = m0 + row, k = k0 + q * 4;
          if (m < M) ra[i] = load4(p.A + (int64_t)m * lda + k, kend - k, p.a_vec);
        } else {
          const int k = idx / (BM / 4), m4 = idx % (BM / 4);
          const int kk = k0 + k, m = m0 + m4 * 4;
          if (kk < kend) ra[i] = load4(p.A + (int64_t)kk * lda + m, M - m, p.a_vec);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int idx = tid + i * 256;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        if (BMODE == 0) {
          const int row = idx >> 2, q = idx & 3;
          const int n = n0 + row, k = k0 + q * 4;
          if (n < N) {
            const int64_t src = p.bidx ? (int64_t)p.bidx[n] : (int64_t)n;
            rb[i] = load4(p.Bm + src * ldb + k, kend - k, p.b_vec);
          }
        } else {
          const int k = idx / (BN / 4), n4 = idx % (BN / 4);
          const int kk = k0 + k, n = n0 + n4 * 4;
          if (kk < kend) {
            const int64_t src = p.bidx ? (int64_t)p.bidx[kk] : (int64_t)kk;
            rb[i] = load4(p.Bm + src * ldb + n, N - n, p.b_vec);
          }
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    float *As = smem + buf * (A_SZ + B_SZ);
    float *Bs = As + A_SZ;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int idx = tid + i * 256;
      if (idx < A_F4) {
        if (AMODE == 0) {
          const int row = idx >> 2, q = idx & 3;
          *reinterpret_cast<float4 *>(As + row * 20 + q * 4) = ra[i];
        } else {
          const int k = idx / (BM / 4), m4 = idx % (BM / 4);
          *reinterpret_cast<float4 *>(As + k * BM + m4 * 4) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int idx = tid + i * 256;
      if (idx < B_F4) {
        if (BMODE == 0) {
          const int row = idx >> 2, q = idx & 3;
          *reinterpret_cast<float4 *>(Bs + row * 20 + q * 4) = rb[i];
        } else {
          const int k = idx / (BN / 4), n4 = idx % (BN / 4);
          *reinterpret_cast<float4 *>(Bs + k * BN + n4 * 4) = rb[i];
        }
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  gload(kbeg);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
    const float *As = smem + cur * (A_SZ + B_SZ);
    const float *Bs = As + A_SZ;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + l31;
        if (AMODE == 0) {
          const float4 v = *reinterpret_cast<const float4 *>(As + row * 20 + kg * 8 + lh * 4);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[i][s] = As[(kg * 8 + lh * 4 + s) * BM + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + l31;
        if (BMODE == 0) {
          const float4 v = *reinterpret_cast<const float4 *>(Bs + col * 20 + kg * 8 + lh * 4);
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = Bs[(kg * 8 + lh * 4 + s) * BN + col];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // ------------------------------------------------------------- epilogues
  if (EPI == EPI_STORE) {
    const int ldc = p.ldc > 0 ? p.ldc : *p.ld_dev;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        float bv = 0.f;
        if (p.bias && n < N) bv = p.bias[p.bias_gather ? (p.bidx ? p.bidx[n] : n) : n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && n < N) {
            float v = acc[i][j][r] + bv;
            v = rk_act(v, p.act);
            float *dst = p.C + (int64_t)m * ldc + n;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
  } else if (EPI == EPI_SPLITK) {
    float *ws = p.C + (int64_t)blockIdx.y * p.M * p.N;   // [split][Mcap][Ncap]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && n < N) ws[(int64_t)m * p.N + n] = acc[i][j][r];
        }
      }
  } else {  // EPI_LOSS : bias + loss + dLoss/dLogits
    __shared__ float lred[4];
    const int ldc = *p.ld_dev;
    const rk_block_t &b = p.blk;
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (wn * TN + j) * 32;   // multiple of 32
        const int n = nb + l31;
        float bv = 0.f;
        if (n < N) bv = p.bias[p.bidx ? p.bidx[n] : n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && n < N) {
            const int row = p.row_off + m;
            const float o = acc[i][j][r] + bv;
            float t = 0.f;
            const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (nb >> 5)];
            if ((word >> l31) & 1u) {
              const int jj = rk_find_col(b.cols, b.indptr[row], b.indptr[row + 1], n);
              if (jj >= 0) t = b.vals[jj];
            }
            float l, g;
            if (p.loss_kind == RK_LOSS_MSE) {
              const float w = (t > 0.f) ? (1.0f + p.confidence) : 1.0f;
              const float d = o - t;
              l = w * (d * d);
              g = (2.0f * d) * (w * p.inv_B);
            } else {  // BCE with logits: (1-t)*o - logsigmoid(o)
              const float ls = fminf(o, 0.f) - log1pf(expf(-fabsf(o)));
              l = (1.0f - t) * o - ls;
              const float sg = 1.0f / (1.0f + expf(-o));
              g = (sg - t) * p.inv_B;
            }
            lsum += l;
            p.C[(int64_t)m * ldc + n] = g;
          }
        }
      }
    lsum = rk_wave_sum(lsum);
    if (lane == 0) lred[wid] = lsum;
    __syncthreads();
    if (tid == 0) p.loss_part[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
  }
}

// ws[split][M][N] -> out[M][N] (fixed split order), optional * act'(Zact)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    const float *__restrict__ ws, int M, int N, const int32_t *__restrict__ Kdev, int kchunk,
    int max_splits, const float *__restrict__ Zact, int act, float *__restrict__ out) {
  const int K = *Kdev;
  int ns = (K + kchunk - 1) / kchunk;
  if (ns > max_splits) ns = max_splits;
  const int64_t tot = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < ns; ++z) s += ws[(int64_t)z * tot + i];
    if (Zact) s *= rk_act_dy(Zact[i], act);
    out[i] = s;
  }
}

// MNLL second pass, one block per row (see rk_mnll_finish in the header)
__global__ __launch_bounds__(256) void mnll_finish_kernel(float *dO, rk_block_t b, int row_off,
                                                          float inv_B, float *loss_part) {
  __shared__ float red[4];
  __shared__ float bc[2];
  const int r = blockIdx.x, row = row_off + r;
  const int n = b.counts[0], ld = b.counts[2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float *orow = dO + (int64_t)r * ld;
  float mx = -INFINITY;
  for (int c = tid; c < n; c += 256) mx = fmaxf(mx, orow[c]);
  mx = rk_wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int c = tid; c < n; c += 256) se += expf(orow[c] - mx);
  se = rk_wave_sum(se);
  if (lane == 0) red[wid] = se;
  __syncthreads();
  const float lsum = logf((red[0] + red[1]) + (red[2] + red[3]));
  __syncthreads();
  // sparse part: loss = -sum_t t*lsm ; sum_g = sum_t (-t*inv_B)
  const int beg = b.indptr[row], end = b.indptr[row + 1];
  float lp = 0.f, sg = 0.f;
  for (int j = beg + tid; j < end; j += 256) {
    const float t = b.vals[j];
    const float lsm = (orow[b.cols[j]] - mx) - lsum;
    lp += -t * lsm;
    sg += -t * inv_B;
  }
  lp = rk_wave_sum(lp);
  sg = rk_wave_sum(sg);
  if (lane == 0) { red[wid] = lp; }
  __syncthreads();
  if (tid == 0) { loss_part[r] = (red[0] + red[1]) + (red[2] + red[3]); }
  __syncthreads();
  if (lane == 0) red[wid] = sg;
  __syncthreads();
  if (tid == 0) bc[0] = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const float sum_g = bc[0];
  // dense part: dO = g - softmax * sum_g,  g = -t*inv_B at stored positions
  for (int c = tid; c < n; c += 256) {
    const float e = expf((orow[c] - mx) - lsum);
    float g = 0.f;
    const uint32_t word = b.bits_rc[(int64_t)row * b.ldw_rc + (c >> 5)];
    if ((word >> (c & 31)) & 1u) {
      const int jj = rk_find_col(b.cols, beg, end, c);
      if (jj >= 0) g = -b.vals[jj] * inv_B;
    }
    orow[c] = g - e * sum_g;
  }
}

// one wave: lane-strided double sums + fixed shuffle tree (order-deterministic);
// re-zeroes the partials so the next rk_decode_loss finds them clean
__global__ __launch_bounds__(64) void loss_reduce_kernel(float *part, int n, float denom,
                                                         float *loss) {
  const int lane = threadIdx.x;
  double s = 0.0;
  for (int i = lane; i < n; i += 64) {
    s += (double)part[i];
    part[i] = 0.f;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) loss[0] = (float)s / denom;
}

__global__ __launch_bounds__(256) void fill_kernel(float *p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    p[i] = v;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

constexpr int DZ_SPLITS = 64;

}  // namespace

extern "C" int64_t rk_dz_workspace_bytes(int32_t B, int32_t h) {
  return (int64_t)DZ_SPLITS * B * h * sizeof(float);
}

extern "C" int32_t rk_loss_partials(int32_t B, int32_t n_cap) {
  const int tiles = rk_cdiv(B, 128) * rk_cdiv(n_cap, 128);
  return tiles > B ? tiles : B;
}

extern "C" int rk_decode_loss(const float *Z, int32_t B, int32_t h, const rk_block_t *tgt,
                              int32_t row_off, const float *W_de, const float *b_de,
                              int32_t loss_kind, float confidence, float inv_B, float *dO,
                              int32_t ld_out, float *loss_part, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h % 4 == 0, "h must be a multiple of 4");
  RK_REQUIRE(row_off >= 0 && row_off + B <= tgt->S_cap, "row slice out of range");
  if (B == 0) return 0;
  GemmP p = {};
  p.A = Z; p.lda = h;
  p.Bm = W_de; p.ldb = h; p.bidx = tgt->items;
  p.M = B; p.N = tgt->n_cap; p.K = h;
  p.Ndev = tgt->counts;          // n_t
  p.tiles_m = rk_cdiv(B, 128);
  p.kchunk = h;
  p.a_vec = aligned16(Z); p.b_vec = aligned16(W_de);
  p.C = dO; p.bias = b_de; p.bias_gather = 1; p.act = RK_ACT_NONE;
  p.blk = *tgt; p.row_off = row_off; p.loss_kind = loss_kind;
  p.confidence = confidence; p.inv_B = inv_B; p.loss_part = loss_part;
  const int tiles = p.tiles_m * rk_cdiv(tgt->n_cap, 128);
  if (loss_kind == RK_LOSS_MSE || loss_kind == RK_LOSS_BCE) {
    // loss_part must be all-zero on entry (surplus tiles never write their slot):
    // it is allocated zeroed and rk_loss_reduce re-zeroes what it consumed
    p.ld_dev = tgt->counts + 2;
    hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2, 0, 0, EPI_LOSS>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  } else {
    if (loss_kind == RK_LOSS_MNLL) { p.ldc = 0; p.ld_dev = tgt->counts + 2; }
    else { RK_REQUIRE(ld_out >= tgt->n_cap || ld_out > 0, "ld_out"); p.ldc = ld_out; }
    hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2, 0, 0, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  }
  RK_CHECK_LAUNCH("decode_loss");
  return 0;
}

extern "C" int rk_mnll_finish(float *dO, int32_t B, const rk_block_t *tgt, int32_t row_off,
                              float inv_B, float *loss_part, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return 0;
  hipLaunchKernelGGL(mnll_finish_kernel, dim3(B), dim3(256), 0, stream, dO, *tgt, row_off, inv_B,
                     loss_part);
  RK_CHECK_LAUNCH("mnll_finish");
  return 0;
}

extern "C" int rk_loss_reduce(float *loss_part, int32_t n, float denom, float *loss,
                              void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, stream, loss_part, n, denom, loss);
  RK_CHECK_LAUNCH("loss_reduce");
  return 0;
}

// dZ = dO . W_de[T]   (M = B, N = h, K = n_t) split-K, then reduce (* act')
extern "C" int rk_decode_bwd_dz(const float *dO, int32_t B, int32_t h, const rk_block_t *tgt,
                                const float *W_de, const float *Zact, int32_t act, float *dZ,
                                float *workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h % 4 == 0, "h must be a multiple of 4");
  if (B == 0) return 0;
  GemmP p = {};
  p.A = dO; p.lda_dev = tgt->counts + 2;
  p.Bm = W_de; p.ldb = h; p.bidx = tgt->items;
  p.M = B; p.N = h; p.K = tgt->n_cap; p.Kdev = tgt->counts;
  p.a_vec = aligned16(dO); p.b_vec = aligned16(W_de);
  p.C = workspace;
  int kchunk = rk_cdiv(rk_cdiv(tgt->n_cap, DZ_SPLITS), 16) * 16;
  if (kchunk < 16) kchunk = 16;
  p.kchunk = kchunk;
  const int splits = rk_cdiv(tgt->n_cap, kchunk);
  // wave tile 32 x (32*TN): pick TN by h
  const int tn = h <= 64 ? 2 : (h <= 128 ? 4 : (h <= 224 ? 7 : 8));
  p.tiles_m = rk_cdiv(B, 128);
  const int tiles = p.tiles_m * rk_cdiv(h, 32 * tn);
#define LAUNCH(TN)                                                                         \
  hipLaunchKernelGGL((gemm_kernel<4, 1, 1, TN, 0, 1, EPI_SPLITK>), dim3(tiles, splits),    \
                     dim3(256), 0, stream, p)
  if (tn == 2) LAUNCH(2); else if (tn == 4) LAUNCH(4); else if (tn == 7) LAUNCH(7); else LAUNCH(8);
#undef LAUNCH
  RK_CHECK_LAUNCH("decode_bwd_dz");
  int grid = rk_cdiv((int64_t)B * h, 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, workspace, B, h,
                     tgt->counts, kchunk, splits, Zact, act, dZ);
  RK_CHECK_LAUNCH("splitk_reduce");
  return 0;
}

// G_de[n_t,h] = dO^T . Z   (M = n_t, N = h, K = B); gb_de = colsum(dO)
extern "C" int rk_decode_bwd_dw(const float *dO, const float *Z, int32_t B, int32_t h,
                                const rk_block_t *tgt, float *G_de, float *gb_de,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h % 4 == 0, "h must be a multiple of 4");
  if (B == 0) return 0;
  GemmP p = {};
  p.A = dO; p.lda_dev = tgt->counts + 2;
  p.Bm = Z; p.ldb = h;
  p.M = tgt->n_cap; p.Mdev = tgt->counts; p.N = h; p.K = B;
  p.a_vec = aligned16(dO); p.b_vec = aligned16(Z);
  p.kchunk = B;
  p.C = G_de; p.ldc = h; p.act = RK_ACT_NONE;
  p.tiles_m = rk_cdiv(tgt->n_cap, 32);
  if (h <= 128) {
    const int tiles = p.tiles_m * rk_cdiv(h, 128);
    hipLaunchKernelGGL((gemm_kernel<1, 4, 1, 1, 1, 1, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  } else {
    const int tiles = p.tiles_m * rk_cdiv(h, 256);
    hipLaunchKernelGGL((gemm_kernel<1, 4, 1, 2, 1, 1, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  }
  RK_CHECK_LAUNCH("decode_bwd_dw");
  if (gb_de) return rk_colsum(dO, B, tgt->n_cap, 0, tgt->counts, gb_de, stream_);
  return 0;
}

// Y[B,N] = act(X[B,K] . W^T + b);  W is [N,K] (nn.Linear) or [K,N] if w_transposed
extern "C" int rk_linear_fwd(const float *X, const float *W, const float *b, int32_t B,
                             int32_t N, int32_t K, int32_t w_transposed, int32_t act, float *Y,
                             void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return 0;
  GemmP p = {};
  p.A = X; p.lda = K;
  p.Bm = W; p.ldb = w_transposed ? N : K;
  p.M = B; p.N = N; p.K = K; p.kchunk = K;
  p.a_vec = aligned16(X) && (K % 4 == 0);
  p.b_vec = aligned16(W) && (p.ldb % 4 == 0);
  p.C = Y; p.ldc = N; p.bias = b; p.act = act;
  p.tiles_m = rk_cdiv(B, 64);
  const int tiles = p.tiles_m * rk_cdiv(N, 64);
  if (!w_transposed)
    hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, 0, 0, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, 0, 1, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
  RK_CHECK_LAUNCH("linear_fwd");
  return 0;
}

extern "C" int rk_linear_bwd(float *dY, const float *Y, const float *X, const float *W, int32_t B,
                             int32_t N, int32_t K, int32_t w_transposed, int32_t act, float *dX,
                             float *dW, int32_t dw_accumulate, float *db, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return 0;
  int rc = rk_act_grad(dY, Y, (int64_t)B * N, act, stream_);
  if (rc) return rc;
  if (db) { rc = rk_colsum(dY, B, N, N, nullptr, db, stream_); if (rc) return rc; }
  const int ldw = w_transposed ? N : K;
  if (dX) {  // dX[B,K] = dY[B,N] . Weff[N,K]
    GemmP p = {};
    p.A = dY; p.lda = N;
    p.Bm = W; p.ldb = ldw;
    p.M = B; p.N = K; p.K = N; p.kchunk = N;
    p.a_vec = aligned16(dY) && (N % 4 == 0);
    p.b_vec = aligned16(W) && (ldw % 4 == 0);
    p.C = dX; p.ldc = K; p.act = RK_ACT_NONE;
    p.tiles_m = rk_cdiv(B, 64);
    const int tiles = p.tiles_m * rk_cdiv(K, 64);
    if (!w_transposed)   // W[N,K]: row = reduction index -> k-major
      hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, 0, 1, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                         stream, p);
    else                 // Wst[K,N]: row = output index, reduction contiguous
      hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, 0, 0, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                         stream, p);
    RK_CHECK_LAUNCH("linear_bwd_dx");
  }
  if (dW) {
    GemmP p = {};
    p.K = B; p.kchunk = B; p.act = RK_ACT_NONE; p.accumulate = dw_accumulate; p.C = dW;
    if (!w_transposed) {  // dW[N,K] = dY^T . X
      p.A = dY; p.lda = N; p.Bm = X; p.ldb = K; p.M = N; p.N = K; p.ldc = K;
    } else {              // dWst[K,N] = X^T . dY
      p.A = X; p.lda = K; p.Bm = dY; p.ldb = N; p.M = K; p.N = N; p.ldc = N;
    }
    p.a_vec = aligned16(p.A) && (p.lda % 4 == 0);
    p.b_vec = aligned16(p.Bm) && (p.ldb % 4 == 0);
    p.tiles_m = rk_cdiv(p.M, 64);
    const int tiles = p.tiles_m * rk_cdiv(p.N, 64);
    hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, 1, 1, EPI_STORE>), dim3(tiles, 1), dim3(256), 0,
                       stream, p);
    RK_CHECK_LAUNCH("linear_bwd_dw");
  }
  return 0;
}
