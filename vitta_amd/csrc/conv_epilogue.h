// Epilogue of one finished 64 x 64 output tile of the implicit-GEMM convolution, shared by the persistent stream-K kernel
// (conv_sk.hip) and the tile-per-workgroup pointwise kernel (conv_pw.hip).  Accumulator layout of a wave (2 x 2 waves per
// tile): register v = pixel row 8 (v / 4) + 4 lk + (v % 4), channel column li.  Eval-mode BatchNorm (+ residual) (+ ReLU)
// (+ hooked moments) (+ raw copy), or the BatchNorm (+ ReLU mask) backward with the statistics-loss injection and
// d gamma / d beta (models/tanet_models/temporal_module.py:85-106; utils/norm_stats_utils.py:238-253).
#pragma once
#include "conv_common.h"

namespace vitta_conv {

// VITTA_CONV_POOL: relu(z) of ONE 32-pixel block of one channel summed per frame.  A block of 32 consecutive pixels touches at most
// two frames (Hy * Wy >= 32); the two lane halves (lk = 0 / 1: the same channel, interleaved pixel quads) meet by one shuffle, then
// one atomic per (block, channel, frame) into pool[frame][channel] -- the 32 lanes of a half-wave add into contiguous bytes.  The sums
// are 64-bit FIXED-POINT (32 fractional bits of the mean): integer addition is associative, so the pooled means -- forward
// activations of the TAM -- do not depend on the order in which the workgroups arrive (float atomics made two runs of the same step
// differ by sign flips of the L1 alignment; same cost: tools/ubench/atomic_line_probe.hip).
// One contribution to a fixed-point pooled mean.  Range / non-finite semantics (round 5): the reference's adaptive_avg_pool2d
// (temporal_module.py:53) hands NaN / Inf on to the TAM; an integer sum cannot, and |mean| >= 2^31 would wrap silently.  A
// contribution that is not finite, or whose magnitude reaches 2^22 (a 32-pixel block's share of a frame mean: activations of ~1e8),
// POISONS the sum instead: the word is exchanged for INT64_MIN, which later additions (each < 2^54) cannot move out of the poisoned
// band |v| >= 2^61; the readers (tam_branch.hip: pooled_tc) decode that band as NaN.  Legitimate sums stay below 2^29 in magnitude
// (at most 128 blocks per frame).  Deviation, stated: Inf arrives as NaN, and finite means beyond ~5e8 do too.
__device__ __forceinline__ void pool_add(unsigned long long* word, float contribution) {
  if (fabsf(contribution) < 4194304.f) atomicAdd(word, (unsigned long long)__float2ll_rn(contribution * 4294967296.f));
  else atomicExch(word, 0x8000000000000000ull);
}

struct PoolSums {
  int mblk, mB, fA;
  float s0 = 0.f, s1 = 0.f;
  __device__ __forceinline__ PoolSums(const ConvK& a, int mblk_, bool on) : mblk(mblk_), mB(0), fA(0) {
    if (on) {  // (one integer division per block: only where the launch pools)
      const int hw = a.d.Hy * a.d.Wy;
      fA = mblk / hw;
      mB = (fA + 1) * hw;
    }
  }
  __device__ __forceinline__ void add(int m, float z) {
    const float r = z < 0.f ? 0.f : z;  // relu that keeps NaN (fmaxf would return 0 for it: the reference's relu hands NaN on)
    if (m < mB) s0 += r;
    else s1 += r;
  }
  __device__ __forceinline__ void flush(const ConvK& a, int k, int lk) {
    const vitta_conv_desc& d = a.d;
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (lk == 0 && mblk < a.Mtot) {
      unsigned long long* pool = reinterpret_cast<unsigned long long*>(d.pool);
      pool_add(pool + (int64_t)fA * d.K + k, s0 * d.pool_scale);
      if (mB < mblk + 32 && mB < a.Mtot) pool_add(pool + (int64_t)(fA + 1) * d.K + k, s1 * d.pool_scale);
    }
  }
};

struct TileEpilogue {
  const ConvK& a;
  const vitta_conv_desc& d;
  float* red;  // LDS [2][32][2]: per-channel sums of the upper pixel half of a tile
  int wm, wn, li, lk;
  bool BWD;
  int BM = 64;  // pixel rows of the workgroup's tile: wave row wm covers [wm * BM / 2, (wm + 1) * BM / 2) in 32-row blocks
  int BN = 64;  // output channels of the tile: column block wn covers [32 wn, 32 wn + 32)
  int oa, ob;   // scattered output (ostride 2): row / column offset of this tile's parity class
  // tile origin handed over by a kernel that has it already (conv_b3.hip: no divisions by nNt in the epilogue)
  int tm0 = 0, tk0 = 0;
  bool has_tile = false;
  __device__ __forceinline__ void set_tile(int m0, int k0) {
    tm0 = m0;
    tk0 = k0;
    has_tile = true;
  }
  // constants of the current tile's channel of this lane: loaded at the tile's start, used at its end
  float c_gam = 1.f, c_bet = 0.f, c_mean = 0.f, c_var = 1.f, c_sh = 0.f, c_a = 0.f, c_b = 0.f, c_mu = 0.f, c_gs = 0.f;

  __device__ __forceinline__ TileEpilogue(const ConvK& a_, float* red_, int wm_, int wn_, int li_, int lk_, int bm_ = 64, int bn_ = 64)
      : a(a_), d(a_.d), red(red_), wm(wm_), wn(wn_), li(li_), lk(lk_), BWD(a_.d.flags & VITTA_CONV_BWD_BN), BM(bm_), BN(bn_), oa(a_.d.oa), ob(a_.d.ob) {}

  // The nine per-channel constants of the tile's columns staged in LDS at the tile's start (cst[9][BN], filled by
  // stage_consts with one lane per column) instead of held in registers through the K walk.
  static __device__ __forceinline__ void stage_consts(const ConvK& a, int L, int bn, float* cst, int col) {
    const vitta_conv_desc& d = a.d;
    const int k = (L % a.nNt) * bn + col;
    float v[9] = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // gam bet mean var sh a b mu gs
    if (d.flags & VITTA_CONV_BWD_BN) {
      v[0] = d.bwd_bn[0][k];
      v[1] = d.bwd_bn[1][k];
      v[2] = d.bwd_bn[2][k];
      v[3] = d.bwd_bn[3][k];
      if (d.inj_mu) {
        v[8] = d.inj_gscale ? d.inj_gscale[0] : 1.f;
        v[5] = d.inj_a[k];
        v[6] = d.inj_b[k];
        v[7] = d.inj_mu[k];
      }
    } else {
      if (d.epi_bn[0]) {
        v[0] = d.epi_bn[0][k];
        v[1] = d.epi_bn[1][k];
        v[2] = d.epi_bn[2][k];
        v[3] = d.epi_bn[3][k];
      }
      if (d.st_shift) v[4] = d.st_shift[k];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) cst[i * bn + col] = v[i];
  }
  __device__ __forceinline__ void consts_from_lds(const float* cst) {
    const int col = wn * 32 + li;
    c_gam = cst[col];
    c_bet = cst[BN + col];
    c_mean = cst[2 * BN + col];
    c_var = cst[3 * BN + col];
    c_sh = cst[4 * BN + col];
    c_a = cst[5 * BN + col];
    c_b = cst[6 * BN + col];
    c_mu = cst[7 * BN + col];
    c_gs = cst[8 * BN + col];
  }

  __device__ __forceinline__ void load_consts(int L) {
    const int k = (has_tile ? tk0 : (L % a.nNt) * BN) + wn * 32 + li;
    if (BWD) {
      c_gam = d.bwd_bn[0][k];
      c_bet = d.bwd_bn[1][k];
      c_mean = d.bwd_bn[2][k];
      c_var = d.bwd_bn[3][k];
      if (d.inj_mu) {
        c_gs = d.inj_gscale ? d.inj_gscale[0] : 1.f;
        c_a = d.inj_a[k];
        c_b = d.inj_b[k];
        c_mu = d.inj_mu[k];
      }
    } else {
      if (d.epi_bn[0]) {
        c_gam = d.epi_bn[0][k];
        c_bet = d.epi_bn[1][k];
        c_mean = d.epi_bn[2][k];
        c_var = d.epi_bn[3][k];
      }
      if (d.st_shift) c_sh = d.st_shift[k];
    }
  }

  // ---- epilogue of one finished tile (register v: pixel row 8 (v / 4) + 4 lk + (v % 4), channel column li) ------------
  // p0..p3: the four 16-byte pieces tile_prefetch() requested for this lane at the tile's start (used when has_pre)
  template <bool PRE = false>
  __device__ __forceinline__ void run(int L, const f32x16& acc, float4 p0 = float4{}, float4 p1 = float4{}, float4 p2 = float4{},
                                      float4 p3 = float4{}) {
    float r1 = 0.f, r2 = 0.f;
    body<PRE>(L, 0, acc, r1, r2, p0, p1, p2, p3);
    finish(L, r1, r2);
  }

  // one 32 x 32 accumulator block (block xb of this wave's rows); r1 / r2 collect the lane's per-channel sums
  // POOLT: the kernel instantiation that carries the VITTA_CONV_POOL sums (conv_b3.hip compiles one for the launches that pool:
  // with the code behind a run-time flag in every instantiation all 156 launches of a step ran 2 % slower, measured)
  template <bool PRE = false, bool POOLT = false>
  __device__ __forceinline__ void body(int L, int xb, const f32x16& acc, float& r1, float& r2, float4 p0 = float4{}, float4 p1 = float4{},
                                       float4 p2 = float4{}, float4 p3 = float4{}) {
    const int flags = d.flags;
    const int m0 = has_tile ? tm0 : (L / a.nNt) * BM, k0 = has_tile ? tk0 : (L % a.nNt) * BN;
    const bool STATS = (flags & VITTA_CONV_STATS) && d.st_s1;
    const bool RAWST = flags & VITTA_CONV_STATS_RAW;
    const bool APPLY = flags & VITTA_CONV_EPI_APPLY;
    const bool RELU = flags & VITTA_CONV_EPI_RELU;
    const bool RES = (flags & VITTA_CONV_RES) && d.res;
    const bool RESH = (flags & VITTA_CONV_RES_HALF) && d.res;
    const bool BRELU = flags & VITTA_CONV_BWD_RELU;
    const bool IRAW = (flags & VITTA_CONV_INJ_RAW) && d.inj_mu;
    const bool POOL = POOLT && (flags & VITTA_CONV_POOL) && d.pool;
    const int HWy = d.Hy * d.Wy;
    const int k = k0 + wn * 32 + li;
    PoolSums pool(a, m0 + wm * (BM >> 1) + 32 * xb, POOL);
    float es = 1.f, et = 0.f, sh = c_sh, bsc = 0.f, bt = 0.f, brm = 0.f, brs = 0.f, ia = 0.f, ib = 0.f;
    if (BWD) {
      brs = rsqrtf(c_var + d.bwd_eps);
      bsc = c_gam * brs;
      bt = c_bet - c_mean * bsc;
      brm = c_mean;
      ia = c_gs * c_a;
      ib = c_gs * c_b;
      sh = c_mu;
    } else if (d.epi_bn[0]) {
      es = c_gam * rsqrtf(c_var + d.epi_eps);
      et = c_bet - c_mean * es;
    }
    const int64_t yrow = (int64_t)k * a.yP, brow = (int64_t)k * a.yP;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int m = m0 + wm * (BM >> 1) + 32 * xb + 8 * qd + 4 * lk;
      if (m >= a.Mtot) continue;
      const bool counted = STATS;
      float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
      if (a.contig) {
        float* yp = d.y + yrow + m;
        if (RES && BWD) {
          const float4 r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (RESH) {
          const int Hh = (d.Hy + 1) >> 1, Wh = (d.Wy + 1) >> 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = m + e;
            const int n = p / HWy, r = p - n * HWy, h = r / d.Wy, w = r - h * d.Wy;
            if (!((h | w) & 1)) v[e] += d.res[(int64_t)k * a.rP + (int64_t)n * Hh * Wh + (h >> 1) * Wh + (w >> 1)];
          }
        }
        if (BWD) {
          float4 xr;
          if constexpr (PRE) xr = qd == 0 ? p0 : qd == 1 ? p1 : qd == 2 ? p2 : p3;
          else xr = *reinterpret_cast<const float4*>(d.bwd_x + brow + m);
          const float xv[4] = {xr.x, xr.y, xr.z, xr.w};
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (BRELU && d.bwd_mask) {
            const float4 mr = *reinterpret_cast<const float4*>(d.bwd_mask + brow + m);
            mk[0] = mr.x > 0.f; mk[1] = mr.y > 0.f; mk[2] = mr.z > 0.f; mk[3] = mr.w > 0.f;
          }
          float o[4], gm[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(xv[e], bsc, bt);
            const float mm = (BRELU && !d.bwd_mask) ? (z > 0.f ? 1.f : 0.f) : mk[e];
            gm[e] = v[e] * mm;
            // (VITTA_CONV_INJ_RAW: the hooked feature is the raw input of the BatchNorm -- its statistics-loss gradient joins dx)
            const float dz = IRAW ? gm[e] : gm[e] + fmaf(ib, z - sh, ia);
            r1 += dz * (xv[e] - brm) * brs;
            r2 += dz;
            o[e] = IRAW ? fmaf(dz, bsc, fmaf(ib, xv[e] - sh, ia)) : dz * bsc;
          }
          *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
          if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(gm[0], gm[1], gm[2], gm[3]);
        } else {
          if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(v[0], v[1], v[2], v[3]);
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(v[e], es, et);
            if (counted) {
              const float dd = (RAWST ? v[e] : z) - sh;
              r1 += dd;
              r2 = fmaf(dd, dd, r2);
            }
            if (POOL) pool.add(m + e, z);
            o[e] = APPLY ? z : v[e];
          }
          if (RES) {
            float4 r;
            if constexpr (PRE) r = qd == 0 ? p0 : qd == 1 ? p1 : qd == 2 ? p2 : p3;
            else r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
          }
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? 0.f : o[e];  // (keeps NaN, as torch's relu)
          }
          *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
        }
      } else {
        // scattered destination (data gradient of a stride-2 convolution, one parity class per launch): plain values
        const int hwg = d.Hg * d.Wg;
        const bool fast = a.hot.nwg > 0;  // (conv_b3.hip: hot.hw / hot.w divide by Hs * Ws == Hg * Wg and Ws == Wg there)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = m + e;
          const int n = fast ? fdiv(p, a.hot.d_hw) : p / hwg, r = p - n * hwg, gi = fast ? fdiv(r, a.hot.d_w) : r / d.Wg, gj = r - gi * d.Wg;
          const int h = gi * d.ostride + oa, w = gj * d.ostride + ob;
          if (h < d.Hy && w < d.Wy) d.y[yrow + (int64_t)n * HWy + h * d.Wy + w] = v[e];
        }
      }
    }
    if (POOL) pool.flush(a, k, lk);
  }

  // per-channel sums of the tile -> statistics / d gamma, d beta
  __device__ __forceinline__ void finish(int L, float r1, float r2) {
    const bool STATS = (d.flags & VITTA_CONV_STATS) && d.st_s1;
    const int k = (L % a.nNt) * BN + wn * 32 + li;
    if (STATS || BWD) {
      // per-channel sums: lanes of a wave, then the two waves that share the channels (pixel halves wm = 0 / 1) through
      // LDS, then ONE atomic per (tile, channel) -- same-address atomics serialise at ~8 ns each, and a 64-channel layer
      // has 784 tiles adding into the same 64 + 64 words
      r1 += __shfl_xor(r1, 32, 64);
      r2 += __shfl_xor(r2, 32, 64);
      if (wm == 1 && lk == 0) {
        red[(wn * 32 + li) * 2] = r1;
        red[(wn * 32 + li) * 2 + 1] = r2;
      }
      __syncthreads();
      if (wm == 0 && lk == 0) {
        r1 += red[(wn * 32 + li) * 2];
        r2 += red[(wn * 32 + li) * 2 + 1];
        if (BWD) {
          if (d.dgamma) atomicAdd(d.dgamma + k, r1);
          if (d.dbeta) atomicAdd(d.dbeta + k, r2);
        } else {
          atomicAdd(d.st_s1 + k, r1);
          atomicAdd(d.st_s2 + k, r2);
        }
      }
      __syncthreads();  // `red` is free again before the next tile's epilogue
    }
  }
};

// The epilogue's first input stream of this lane -- the residual of a forward launch, or the BatchNorm-backward input of a
// data-gradient launch -- requested at the tile's START so that its latency hides under the K walk (conv_pw.hip, launches
// the host marked with ConvK::pw_prefetch: contiguous output, BWD_BN or RES without RES_HALF; the stream-K kernel cannot
// afford the sixteen registers at three workgroups per CU).  Rows past Mtot read the last valid quad (never used).
// (Written without conditionals around the loads: the compiler turns `c ? *p : zero` into a select of ADDRESSES with the
// zero parked in scratch memory.)
__device__ __forceinline__ void tile_prefetch_at(const ConvK& a, int m0, int k0, int wm, int wn, int li, int lk, float4& p0, float4& p1,
                                                 float4& p2, float4& p3, int bm = 64, int xb = 0) {
  const vitta_conv_desc& d = a.d;
  const bool bwd = d.flags & VITTA_CONV_BWD_BN;
  const float* row = (bwd ? d.bwd_x : d.res) + (int64_t)(k0 + wn * 32 + li) * (bwd ? a.yP : a.rP);
  const int m = m0 + wm * (bm >> 1) + 32 * xb + 4 * lk, last = a.Mtot - 4;
  p0 = *reinterpret_cast<const float4*>(row + min(m, last));
  p1 = *reinterpret_cast<const float4*>(row + min(m + 8, last));
  p2 = *reinterpret_cast<const float4*>(row + min(m + 16, last));
  p3 = *reinterpret_cast<const float4*>(row + min(m + 24, last));
}
__device__ __forceinline__ void tile_prefetch(const ConvK& a, int L, int wm, int wn, int li, int lk, float4& p0, float4& p1, float4& p2,
                                              float4& p3, int bm = 64, int xb = 0) {
  tile_prefetch_at(a, (L / a.nNt) * bm, (L % a.nNt) * 64, wm, wn, li, lk, p0, p1, p2, p3, bm, xb);
}

}  // namespace vitta_conv
