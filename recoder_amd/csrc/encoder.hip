// Encoder side of DynamicAutoencoder: CSR user rows x item-embedding rows.
//
// Forward  (reference nn.py:235-240, 269-278, K2..K6 of SURVEY 2.3):
//   Z0[r,:] = act( sum_j  (v_j / max(||v_r||,1e-12)) * keep_j/(1-p) * W_en[item_j,:] + b_en )
// The reference densifies the B x n_b block and runs a dense addmm over ~99%
// zeros; here each workgroup owns one user row, its 4 waves split the row's
// stored interactions and gather W_en rows with 16-B loads (HBM/L2-bound:
// nnz_b * h * 4 bytes of gathered rows per step), then combine through LDS.
//
// Backward (autograd of the same, model.py:397):
//   G_en[c,:] = sum_{r in column c} svals[r,c] * dZ0pre[r,:]
// one wave per sampled item column; rows are visited in ascending order through
// the transposed bitmap so the fp32 sum is order-deterministic (no atomics).
#include "common.h"
#include "encoder_bwd.h"
#include "planes.h"

namespace {

// HV = number of float4 per lane (h <= 256*HV); FW waves share one user row.  The launch lasts as
// long as its LONGEST row (a C2 batch: mean 55 stored interactions, maximum 430-900), so the row is
// cut into as many pieces as the register budget allows at two workgroups per CU
constexpr int FW = 8;
// rk_enc_probe(buffer): every row workgroup of the encoder forward records wall_clock64() at entry,
// with its first entries loaded, after the gather and at the end (tools/probes/enc_phase_probe.py)
unsigned long long *g_enc_probe = nullptr;
#define ESTAMP(k) do { if (probe && threadIdx.x == 0) probe[(size_t)(blockIdx.x - n_split) * 8 + (k)] = wall_clock64(); } while (0)
template <int HV, int UB>
__global__ __launch_bounds__(FW * 64) void ae_encode_fwd_kernel(
    rk_block_t b, int row_off, int B, const float *__restrict__ W, const float *__restrict__ bias,
    int h, const uint8_t *__restrict__ keep, float p, float scale, uint64_t seed,
    uint64_t rng_step, const int64_t *__restrict__ users, const float *__restrict__ user_norm,
    int act, float *__restrict__ Z0, uint16_t *__restrict__ planes, int64_t plane_stride,
    int cols_pad, rk_cur_t cur, rkp::SplitW sw, int n_split, char *__restrict__ zimg, int z_kt,
    int zt_pairs, unsigned long long *probe, int z_ones) {
  __shared__ float red[FW];
  constexpr int PART_B = (FW - 1) * HV * 256 * 4;
  __shared__ __attribute__((aligned(16))) char sm_raw[PART_B > rkp::SPLIT_W_LDS ? PART_B : rkp::SPLIT_W_LDS];
  // The first n_split workgroups split the decoder rows of the block's items into the fp16 plane
  // images of the decoder contractions (planes.h).  The pass touches neither the encoder table's
  // output nor anything this launch writes, and the decode that needs it is the NEXT launch: it
  // rides here off the chain (first in the grid: uniform work, never the launch's tail).
  if ((int)blockIdx.x < n_split) {
    rkp::split_w_job<FW * 64>(sw, (int)blockIdx.x, sm_raw);
    return;
  }
  if (cur.cursor) {          // graph replay (common.h): the step's RNG index and user ids
    rng_step = (uint64_t)(rk_cur_global(cur) + 1);
    if (users) users += rk_cur_local(cur) * B;       // (replayed steps are whole batches: S == B)
  }
  float (*part)[HV * 256] = reinterpret_cast<float (*)[HV * 256]>(sm_raw);
  const int r = (int)blockIdx.x - n_split;            // row within the slice
  if (r >= B) {
    // planes != null: the grid covers the rows up to the next multiple of 64; the padding rows
    // of the Z^T planes are (re)written as zeros (the dW kernel's K padding relies on it)
    if (planes) {
      for (int n = threadIdx.x; n < h; n += FW * 64) {
        const int64_t o = ((int64_t)(r >> 3) * cols_pad + n) * 8 + (r & 7);
        planes[o] = 0; planes[o + plane_stride] = 0; planes[o + 2 * plane_stride] = 0;
      }
    }
    return;
  }
  const int row = row_off + r;         // row within the block
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  ESTAMP(0);
  const int beg = b.indptr[row], end = b.indptr[row + 1];
  const int n = end - beg;
  const bool implicit = b.implicit != 0;
  const int64_t uid = users ? users[row] : (int64_t)row;

  // ---- each wave takes a contiguous 1/FW of the row's entries.  The first 64 of
  // them are fetched (value + global item id, independent loads) BEFORE the norm is
  // known, so the dependent chain is indptr -> entries -> W rows ----
  const int q = (n + FW - 1) / FW;
  const int wbeg = beg + wid * q;
  const int wend = min(end, wbeg + q);
  int item0 = 0;
  float v0 = 0.f;
  if (wbeg + lane < wend) {
    item0 = b.gcols ? b.gcols[wbeg + lane] : b.items[b.cols[wbeg + lane]];
    v0 = implicit ? 1.0f : b.vals[wbeg + lane];
  }

  // ---- L2 norm of the row (F.normalize: x / max(||x||_2, 1e-12)) ----
  float nrm;
  if (user_norm) {
    // item-parallel shards: the block holds only this rank's columns of the row; the
    // norm of the WHOLE row comes precomputed per user
    nrm = fmaxf(user_norm[uid], 1e-12f);
  } else if (implicit) {
    nrm = fmaxf(sqrtf((float)n), 1e-12f);      // n ones: the sum of squares is exactly n
  } else {
    float ss = 0.f;
    for (int j = beg + tid; j < end; j += FW * 64) {
      const float v = b.vals[j];
      ss += v * v;
    }
    ss = rk_wave_sum(ss);
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < FW; ++w) tot += red[w];
    nrm = fmaxf(sqrtf(tot), 1e-12f);
  }

  if (probe) { if (item0 == 0x7fffffff && v0 == 1e30f) return; ESTAMP(1); }     // (stamp behind the loads)
  float4 acc[HV];
#pragma unroll
  for (int k = 0; k < HV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the bias: loaded here, in front of the gather, not behind the barrier -- one dependent round
  // trip less at the end of a launch that is a chain of them: with every row cut to 8 entries the
  // launch still took 12.0 of its 15.3 us)
  float4 bb[HV];
#pragma unroll
  for (int v = 0; v < HV; ++v) {
    const int hh = (v * 64 + lane) * 4;
    bb[v] = (bias && wid == 0 && hh < h) ? *reinterpret_cast<const float4 *>(bias + hh)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (int base = wbeg; base < wend; base += 64) {
    const int j = base + lane;
    int item = item0;
    float val = v0;
    if (base != wbeg && j < wend) {
      item = b.gcols ? b.gcols[j] : b.items[b.cols[j]];
      val = implicit ? 1.0f : b.vals[j];
    }
    float s = 0.f;
    if (j < wend) {
      const float xh = val / nrm;
      bool kp = true;
      if (p > 0.f) kp = keep ? (keep[j] != 0) : rk_keep_draw(seed, rng_step, (uint64_t)uid, (uint64_t)item, p);
      s = (p > 0.f) ? (kp ? xh * scale : 0.f) : xh;
      b.svals[j] = s;
    } else {
      item = 0;
    }
    // dropped entries (s == 0: half of them at the reference's noise_prob 0.5) are not gathered at all: the
    // wave walks the set bits of the ballot of its live entries, UB at a time.  Exact: a skipped entry would
    // have added fmaf(0, w, acc) == acc (acc is never -0: it starts at +0 and x + (-x) rounds to +0).
    // The row loads of consecutive live entries are independent and stay in flight together; a pass
    // short of UB live entries pads with item 0 / s 0 (exact zeros).
    // UB row loads in flight per lane.  (16 instead of 8 does not shorten a heavy row -- 7.4 vs 7.7 us for
    // 599 entries: the row is bound by what ONE CU fetches, ~560 cache lines per us -- and costs the
    // light rows occupancy: 17.3-18.3 vs 15.8 us for the launch inside the step)
    unsigned long long live = __ballot(s != 0.f);
    while (live) {
      int it[UB];
      float sv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        // (the lane index is wave-uniform: v_readlane into a scalar register instead of a ds_bpermute
        // through the LDS crossbar; 16 VGPRs less, same speed)
        const bool has = live != 0ull;
        const int kk = has ? __builtin_ctzll(live) : 0;
        live &= live - 1ull;                          // (0 & ~0 stays 0)
        it[u] = has ? __builtin_amdgcn_readlane(item, kk) : 0;
        sv[u] = has ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), kk)) : 0.f;
      }
#pragma unroll
      for (int v = 0; v < HV; ++v) {
        const int hh = min((v * 64 + lane) * 4, h - 4);
        float4 w4[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
          w4[u] = *reinterpret_cast<const float4 *>(W + (int64_t)it[u] * h + hh);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          acc[v].x = fmaf(sv[u], w4[u].x, acc[v].x);
          acc[v].y = fmaf(sv[u], w4[u].y, acc[v].y);
          acc[v].z = fmaf(sv[u], w4[u].z, acc[v].z);
          acc[v].w = fmaf(sv[u], w4[u].w, acc[v].w);
        }
      }
    }
  }
  if (probe) { if (acc[0].x == 1.2345e30f) return; ESTAMP(2); }
  // ---- combine the 4 partial sums in fixed order, bias, activation ----
  if (wid > 0) {
#pragma unroll
    for (int v = 0; v < HV; ++v)
      *reinterpret_cast<float4 *>(&part[wid - 1][(v * 64 + lane) * 4]) = acc[v];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int hh = (v * 64 + lane) * 4;
      if (hh < h) {
        float4 a = acc[v];
        for (int w = 0; w < FW - 1; ++w) {
          const float4 o = *reinterpret_cast<const float4 *>(&part[w][hh]);
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        float4 y;
        y.x = rk_act(a.x + bb[v].x, act);
        y.y = rk_act(a.y + bb[v].y, act);
        y.z = rk_act(a.z + bb[v].z, act);
        y.w = rk_act(a.w + bb[v].w, act);
        *reinterpret_cast<float4 *>(Z0 + (int64_t)r * h + hh) = y;
        // Z as fp16 hi / lo plane image for the decode (planes.h), with the value still in registers
        // (bounded activations only: their split scale is static)
        if (zimg) {
          const float sz = sw.plain ? 1.0f : rkp::SCALE_Z;
          rkp::store_split4(zimg + (int64_t)r * z_kt * rkp::LINE, hh, y, sz, sw.plain != 0);
          if (r == 0 && tid == 0) sw.scales[0] = sz;     // (consumers read the scale from there)
          // z_ones (h % 32 != 0): the first padding column of the image row holds the constant 1 -- dW = dO^T . Z
          // then leaves the column sums of dO (the decoder bias gradient) in its output column h for free; the
          // decode multiplies the column with the W image's zero padding
          // (a launch WITHOUT z_ones writes zeros there: a step that does not use the ones column -- a ragged batch,
          // another mode of the same engine -- must not find a stale 1 of an earlier one)
          if ((h & 31) != 0 && hh == h - 4)
            rkp::store_split4(zimg + (int64_t)r * z_kt * rkp::LINE, h, make_float4(z_ones ? 1.f : 0.f, 0.f, 0.f, 0.f), sz,
                              sw.plain != 0);
        }
        if (planes) {
          // Z^T as three bf16 planes in the fragment order of the dW kernel (csrc/dw3.hip):
          // element (k = r, n) at ((k/8)*cols_pad + n)*8 + k%8 -- written here, with the value
          // still in registers, instead of by a pass of its own
          uint32_t hp[2], mp[2], lp[2];
          if (zt_pairs) {
            // fp16 pairs for rk_decode_bwd_dw2 (static scale: bounded activations): planes 0 / 1
            uint2 hh2, ll2;
            rkp::split4(y, rkp::SCALE_Z, hh2, ll2);
            hp[0] = hh2.x; hp[1] = hh2.y; mp[0] = ll2.x; mp[1] = ll2.y; lp[0] = lp[1] = 0u;
          } else {
          rk_split_bf16_pair(y.x, y.y, hp[0], mp[0], lp[0]);
          rk_split_bf16_pair(y.z, y.w, hp[1], mp[1], lp[1]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t o = ((int64_t)(r >> 3) * cols_pad + hh + e) * 8 + (r & 7);
            const int sh = (e & 1) * 16;
            planes[o] = (uint16_t)(hp[e >> 1] >> sh);
            planes[o + plane_stride] = (uint16_t)(mp[e >> 1] >> sh);
            planes[o + 2 * plane_stride] = (uint16_t)(lp[e >> 1] >> sh);
          }
        }
      }
    }
  }
  ESTAMP(3);
}

template <int HV>
__global__ __launch_bounds__(256) void ae_encode_bwd_kernel(
    rk_block_t b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb) {
  ae_encode_bwd_body<HV>(b, row_off, B, dZ, h, G, accumulate, gb, n_gb, (int)blockIdx.x);
}

// (the workgroup body lives in encoder_bwd.h: the fused dW || encoder-backward launch of dw3.hip
// runs it next to the dW tiles)
// the light-column variant (two dZ rows in flight) held to six waves per SIMD
template <int HV, int WW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void ae_encode_bwd_cols_light_kernel(
    rk_block_t b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb) {
  ae_encode_bwd_cols_body<HV, false, WW, 2>(b, row_off, B, dZ, h, G, accumulate, gb, n_gb, (int)blockIdx.x);
}

template <int HV, int WW = 1, int U = 8>
__global__ __launch_bounds__(256) void ae_encode_bwd_cols_kernel(
    rk_block_t b, int row_off, int B, const float *__restrict__ dZ, int h,
    float *__restrict__ G, int accumulate, float *__restrict__ gb, int n_gb) {
  ae_encode_bwd_cols_body<HV, false, WW, U>(b, row_off, B, dZ, h, G, accumulate, gb, n_gb, (int)blockIdx.x);
}

}  // namespace

static int encode_fwd_launch(const rk_block_t *blk, int32_t row_off, int32_t B, const float *W_en,
                             const float *b_en, int32_t h, const uint8_t *keep, float p,
                             uint64_t seed, uint64_t rng_step, const int64_t *users,
                             const float *user_norm, int32_t act, float *Z0, void *stream_,
                             void *zt_planes = nullptr, rk_cur_t cur = {nullptr, 0},
                             const rk_enc_split_t *es = nullptr, int zt_pairs = 0) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h > 0 && h % 4 == 0 && h <= 1024, "h must be a multiple of 4, <= 1024");
  RK_REQUIRE(p >= 0.f && p < 1.f, "noise_prob must be in [0,1)");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= blk->S_cap, "row slice out of range");
  RK_REQUIRE(user_norm == nullptr || users != nullptr, "user_norm is indexed by users[]");
  if (B == 0) return 0;
  // ATen dropout: noise = bernoulli(1-p) / (1-p), computed in fp32
  const float scale = 1.0f / (float)(1.0 - (double)p);
  const int hv = rk_cdiv(h, 256);
  const int rows = zt_planes ? rk_dw3_rows_pad(B) : B;
  const int cols_pad = rk_dw3_cols_pad(h);
  const int64_t plane_stride = (int64_t)rk_dw3_rows_pad(B) * cols_pad;
  rkp::SplitW sw = {};
  int n_split = 0, z_kt = 0, z_ones = 0;
  char *zimg = nullptr;
  if (es) {
    sw = es->sw; n_split = es->n_split; zimg = es->zimg; z_kt = es->z_kt;
    z_ones = (es->z_ones && zimg && h % 32 != 0) ? 1 : 0;     // (plain bf16 images too: 1.0 is exact)
  }
#define LAUNCH(HV)                                                                         \
  RK_LAUNCH((ae_encode_fwd_kernel<HV, 8>), dim3(n_split + rows), dim3(FW * 64), 0, stream, *blk, row_off, \
                     B, W_en, b_en, h, keep, p, scale, seed, rng_step, users, user_norm, act, Z0, \
                     (uint16_t *)zt_planes, plane_stride, cols_pad, cur, sw, n_split, zimg, z_kt, zt_pairs, g_enc_probe, z_ones)
  if (hv == 1) LAUNCH(1); else if (hv == 2) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
  RK_CHECK_LAUNCH("ae_encode_fwd");
  return 0;
}

extern "C" void rk_enc_probe(unsigned long long *buffer) { g_enc_probe = buffer; }

extern "C" int rk_ae_encode_fwd(const rk_block_t *blk, int32_t row_off, int32_t B,
                                const float *W_en, const float *b_en, int32_t h,
                                const uint8_t *keep, float p, uint64_t seed, uint64_t rng_step,
                                const int64_t *users, int32_t act, float *Z0, void *stream_) {
  RK_REQUIRE(b_en != nullptr, "b_en is required");
  if (const rk_replay_t *rp = rk_replay_get()) {     // replayed per-entry step: cursor-derived
    const rk_cur_t cur = {rp->cursor, rp->off};
    return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, 0, rp->users_base, nullptr,
                             act, Z0, stream_, nullptr, cur);
  }
  return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, rng_step, users, nullptr,
                           act, Z0, stream_);
}

// rk_ae_encode_fwd with the W_de[items] half of the decode's operand split (rk_split_w: pl->w, pl->wt
// and the W scale) as extra workgroups of the same launch -- what the one-call step does; here for
// the steps sequenced entry by entry (hidden stacks: the split launch in front of their decode then
// only cuts Z, rk_split_wz with W_de == NULL)
extern "C" int rk_ae_encode_fwd_split_w(const rk_block_t *blk, int32_t row_off, int32_t B,
                                        const float *W_en, const float *b_en, int32_t h,
                                        const uint8_t *keep, float p, uint64_t seed, uint64_t rng_step,
                                        const int64_t *users, int32_t act, float *Z0, const float *W_de,
                                        const int32_t *ranges, const rk_planes_t *pl, void *stream_) {
  RK_REQUIRE(b_en != nullptr, "b_en is required");
  RK_REQUIRE(pl && pl->h == h && blk->n_cap <= pl->n_cap, "planes were laid out for another shape");
  RK_REQUIRE((((uintptr_t)W_de) & 15) == 0, "W_de must be 16-byte aligned");
  rk_enc_split_t es = {};
  es.sw = rk_split_w_args(W_de, blk, ranges, pl);
  es.n_split = rk_cdiv(blk->n_cap, 32);
  es.zimg = nullptr;
  es.z_kt = rkp::kp_of(h) / 32;
  if (const rk_replay_t *rp = rk_replay_get()) {     // replayed per-entry step: cursor-derived
    const rk_cur_t cur = {rp->cursor, rp->off};
    return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, 0, rp->users_base, nullptr,
                             act, Z0, stream_, nullptr, cur, &es);
  }
  return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, rng_step, users, nullptr,
                           act, Z0, stream_, nullptr, {nullptr, 0}, &es);
}

extern "C" int rk_ae_encode_fwd_planes(const rk_block_t *blk, int32_t row_off, int32_t B,
                                       const float *W_en, const float *b_en, int32_t h,
                                       const uint8_t *keep, float p, uint64_t seed, uint64_t rng_step,
                                       const int64_t *users, int32_t act, float *Z0, void *zt_planes,
                                       void *stream_) {
  RK_REQUIRE(b_en != nullptr, "b_en is required");
  RK_REQUIRE(zt_planes != nullptr && (((uintptr_t)zt_planes) & 15) == 0, "zt_planes: 16-byte aligned");
  return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, rng_step, users, nullptr,
                           act, Z0, stream_, zt_planes);
}

// rk_ae_encode_fwd(_planes) for graph replay: the dropout RNG step is cursor[0] + off + 1, read
// on the device (zt_planes nullable)
int rk_ae_encode_fwd_at(const rk_block_t *blk, int32_t row_off, int32_t B, const float *W_en,
                        const float *b_en, int32_t h, const uint8_t *keep, float p, uint64_t seed,
                        const int64_t *cursor, int32_t cursor_off, const int64_t *users, int32_t act,
                        float *Z0, void *zt_planes, void *stream_, const rk_enc_split_t *es,
                        uint64_t rng_step) {
  const rk_cur_t cur = {cursor, cursor_off};
  // (the Z^T planes the step's dW kernel reads: fp16 pairs for rk_decode_bwd_dw2, bf16 triples else)
  return encode_fwd_launch(blk, row_off, B, W_en, b_en, h, keep, p, seed, rng_step, users, nullptr, act,
                           Z0, stream_, zt_planes, cur, es, zt_planes != nullptr && rk_dw_pairs() ? 1 : 0);
}

extern "C" int rk_ae_encode_fwd_partial(const rk_block_t *blk, int32_t row_off, int32_t B,
                                        const float *W_en, int32_t h, const uint8_t *keep, float p,
                                        uint64_t seed, uint64_t rng_step, const int64_t *users,
                                        const float *user_norm, float *Zpart, void *stream_) {
  return encode_fwd_launch(blk, row_off, B, W_en, nullptr, h, keep, p, seed, rng_step, users,
                           user_norm, RK_ACT_NONE, Zpart, stream_);
}

extern "C" int rk_ae_encode_bwd(const rk_block_t *blk, int32_t row_off, int32_t B,
                                const float *dZ0pre, int32_t h, float *G_en,
                                int32_t accumulate, float *gb_en, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RK_REQUIRE(h > 0 && h % 4 == 0 && h <= 1024, "h must be a multiple of 4, <= 1024");
  RK_REQUIRE(row_off >= 0 && B >= 0 && row_off + B <= blk->S_cap, "row slice out of range");
  RK_REQUIRE(blk->bits_cr != nullptr && blk->pref_rc != nullptr,
             "block was built without the transposed bitmap / prefix index");
  const int n_gb = gb_en ? rk_cdiv(h, 64) : 0;
  const int hv = rk_cdiv(h, 256);
  const int words = ((row_off + B + 31) >> 5) - (row_off >> 5);
  // blocks of LIGHT columns (the item set is as large as the block's entry capacity: about one stored entry
  // per sampled item -- C5's uniform catalogue): two dZ rows in flight per column instead of eight, half the
  // registers, twice the columns in flight (the launch is a chain of dependent loads per column)
  const bool light = blk->n_cap >= 4096 && (int64_t)blk->nnz_cap <= 2 * (int64_t)blk->n_cap;
  if (words <= 128 && (words <= 64 || blk->n_cap >= 4096)) {
    // the row window fits one (two: past 2048 rows) bitmap word(s) per lane: a wave per column (4 per workgroup)
    // (past 2048 rows the bias gradient is one column sum of dZ behind the launch, as for the windows below)
    const int n_gb_k = words <= 64 ? n_gb : 0;
    float *gb_k = words <= 64 ? gb_en : nullptr;
    const int grid = rk_cdiv(blk->n_cap, 4) + n_gb_k;
#define LAUNCH(HV, WW, U)                                                                                 \
  RK_LAUNCH((ae_encode_bwd_cols_kernel<HV, WW, U>), dim3(grid), dim3(256), 0, stream, *blk, row_off, B, \
            dZ0pre, h, G_en, accumulate, gb_k, n_gb_k)
#define LAUNCH_L(HV, WW)                                                                                 \
  RK_LAUNCH((ae_encode_bwd_cols_light_kernel<HV, WW>), dim3(grid), dim3(256), 0, stream, *blk, row_off, B, \
            dZ0pre, h, G_en, accumulate, gb_k, n_gb_k)
#define BY_HV(WW, U) do { if (hv == 1) LAUNCH(1, WW, U); else if (hv == 2) LAUNCH(2, WW, U); else LAUNCH(4, WW, U); } while (0)
#define BY_HV_L(WW) do { if (hv == 1) LAUNCH_L(1, WW); else if (hv == 2) LAUNCH_L(2, WW); else LAUNCH(4, WW, 2); } while (0)
    if (words <= 64) { if (light) BY_HV_L(1); else BY_HV(1, 8); }
    else { if (light) BY_HV_L(2); else BY_HV(2, 8); }
#undef BY_HV_L
#undef BY_HV
#undef LAUNCH_L
#undef LAUNCH
    RK_CHECK_LAUNCH("ae_encode_bwd");
    if (words > 64 && gb_en) return rk_colsum(dZ0pre, B, h, h, nullptr, gb_en, stream_);
    return 0;
  }
  if (blk->n_cap >= 4096) {
    // a row window of more than 128 bitmap words (more than 4096 rows) over a long item set: the wave-per-column
    // kernel once per window of <= 4064 rows (the later ones accumulate; a column without entries in a later
    // window is left alone), the bias gradient as one column sum of dZ -- the workgroup-per-column kernel
    // below took 2.68 ms for 335 k columns at B = 4096
    const int W = 4064;                            // (127 words: a window that starts mid-word still fits 128)
    for (int r0 = 0; r0 < B; r0 += W) {
      const int nb = B - r0 < W ? B - r0 : W;
      const int rc = rk_ae_encode_bwd(blk, row_off + r0, nb, dZ0pre + (int64_t)r0 * h, h, G_en,
                                      r0 == 0 ? accumulate : 1, nullptr, stream_);
      if (rc) return rc;
    }
    return gb_en ? rk_colsum(dZ0pre, B, h, h, nullptr, gb_en, stream_) : 0;
  }
  const int grid = blk->n_cap + n_gb;
#define LAUNCH(HV)                                                                           \
  RK_LAUNCH(ae_encode_bwd_kernel<HV>, dim3(grid), dim3(256), 0, stream, *blk, row_off, \
                     B, dZ0pre, h, G_en, accumulate, gb_en, n_gb)
  if (hv == 1) LAUNCH(1); else if (hv == 2) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
  RK_CHECK_LAUNCH("ae_encode_bwd");
  return 0;
}
