// Row-wise element passes of a TemporalBottleneck's TAM on channel-major planes (tensor[c][f * HW + hw]), as device functions:
// tam_cm.hip launches them on their own, tam_branch.hip runs them as the CONSUMER workgroups of a launch whose first workgroups are
// the branch kernels (round 5: one launch per TAM forward, one per TAM backward -- see "merged launches" there).
//
// Reference: models/tanet_models/temporal_module.py:43-65 (TAM.forward), :85-106 (TemporalBottleneck.forward).
#pragma once
#include "conv_common.h"

namespace tamrows {

using namespace vitta;

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, VITTA_WAVE);
  return v;
}

struct Row {
  int c, f, n, t;
  bool ok;
};

// rows are enumerated (c, n, t) with t fastest == memory order; blk = the workgroup's index among the row workgroups
template <int LPR>
__device__ __forceinline__ Row row_of(int blk, int C, int N, int T, int* sub) {
  constexpr int RPB = VITTA_BLOCK / LPR;
  const int F = N * T;
  *sub = threadIdx.x % LPR;
  const int64_t row = (int64_t)blk * RPB + threadIdx.x / LPR;
  Row r;
  r.ok = row < (int64_t)C * F;
  const int64_t rr = r.ok ? row : 0;
  r.c = (int)(rr / F);
  r.f = (int)(rr - (int64_t)r.c * F);
  r.n = r.f / T;
  r.t = r.f - r.n * T;
  return r;
}

struct BN {
  const float *g, *b, *m, *v;
  float eps;
};

__device__ __forceinline__ void bn_coef(const BN& bn, int c, float& s, float& t) {
  s = bn.g[c] * rsqrtf(bn.v[c] + bn.eps);
  t = bn.b[c] - bn.m[c] * s;
}

__device__ __forceinline__ float act(float x, float s, float t) { return fmaxf(fmaf(x, s, t), 0.f); }

constexpr int EWU = 4;  // 16-byte pieces per lane and stream that the element-wise row kernels keep in flight

// ---- hand-over between the workgroups of ONE launch ---------------------------------------------------------------------------
// Producers come FIRST in the grid, consumers behind them: workgroups are dispatched in id order, so whatever a consumer waits for is
// resident or finished when the consumer starts -- no co-residency requirement on the consumers, no cycle.  Words (zero at rest):
// done = producers that have finished (their write-through stores drained), left = consumers that are past their wait; the last
// consumer to leave zeroes both.  Data crosses with write-through stores and sc1 loads (no agent-scope release: that would write back
// every dirty line of the XCD's L2, the convolution outputs of the step).
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ void wt_store4(float* base, int64_t idx, float4 v) {  // idx in floats, < 2^29
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vitta_conv::u32x4, v), rs, (int)(idx * 4), 0, AUX_SC1);
}
__device__ __forceinline__ void wt_store1(float* base, int64_t idx, float v) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)(idx * 4), 0, AUX_SC1);
}
__device__ __forceinline__ float4 coh_load4(const float* base, int64_t idx) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * 4), 0, AUX_SC1));
}
__device__ __forceinline__ float coh_load1(const float* base, int64_t idx) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(idx * 4), 0, AUX_SC1));
}

// A hand-over point = one arrival counter + NFLAG flag words, each in its own 128-byte line.  The LAST producer to arrive (the counter
// tells it) raises every flag; a consumer polls the flag its workgroup id selects.  (Round 5, first form: every consumer polled the
// counter itself -- ~500 resident workgroups reading ONE word with device-scope loads serialise at its memory channel, and the
// producers' arrivals queue behind them: the merged launches took 2-3x the sum of their parts.)
constexpr int NFLAG = 64, FLAG_STRIDE = 32;  // words between flags
struct Handover {
  unsigned* cnt;    // arrivals (zero at rest)
  unsigned* flags;  // NFLAG x FLAG_STRIDE words (zero at rest)
};

// every store of the workgroup is at the coherence point, then one arrival; the last of `nprod` raises the flags
__device__ __forceinline__ void signal_done(const Handover h, unsigned nprod) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(h.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nprod - 1) {
#pragma unroll 8
      for (int i = 0; i < NFLAG; ++i) __hip_atomic_store(h.flags + i * FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// wait until the producers' flags are up.  The barrier does not drain the wave's outstanding loads (what the caller requested before
// the wait stays in flight).  Bounded (~2 s): if the bound is ever reached the launch ends with wrong numbers, not a hung device.
__device__ __forceinline__ void wait_done(const Handover h, int id) {
  if (threadIdx.x == 0) {
    const unsigned* f = h.flags + (id & (NFLAG - 1)) * FLAG_STRIDE;
    for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++spin)
      __builtin_amdgcn_s_sleep(4);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// the consumer is past every wait; the last of `ncons` consumers puts the hand-over points (and the leave word) to rest
__device__ __forceinline__ void leave(const Handover* h, int nh, unsigned* left, unsigned ncons) {
  if (threadIdx.x == 0) {
    const unsigned l = __hip_atomic_fetch_add(left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (l == ncons - 1) {
      for (int k = 0; k < nh; ++k) {
        __hip_atomic_store(h[k].cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 8
        for (int i = 0; i < NFLAG; ++i) __hip_atomic_store(h[k].flags + i * FLAG_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// the words of a stream's meeting buffer (32 KiB): [0, 1 KiB) counters -- the branch kernels' pairs, and from word FUSE_CNT_OFF the
// merged launches' arrival / leave counters --, from byte 4096 three flag areas of NFLAG lines (forward; backward: aggregation, branches)
constexpr int FUSE_CNT_OFF = 64;
constexpr int FLAG_AREA_OFF = 1024, FLAG_AREA_WORDS = NFLAG * FLAG_STRIDE;  // words
constexpr size_t SYNC_BYTES = 4 * (size_t)(FLAG_AREA_OFF + 3 * FLAG_AREA_WORDS);
__device__ __forceinline__ Handover handover(unsigned* sync, int k) {
  return Handover{sync + FUSE_CNT_OFF + k, sync + FLAG_AREA_OFF + k * FLAG_AREA_WORDS};
}

// ---- forward: out[c][n,t,:] = sum_j K[n,c,j] * gate[n,c,t+j-1] * a[c][n,t+j-1,:],  a = relu(bn1(x1)) applied while loading -------
struct AggFwd {
  const float* x;
  BN bn;
  const float* gate;   // [N, C, T]
  const float* kern;   // [N * C, 3]
  int C, N, T, HW;
  float* out;
};

// FUSED: gate / kern are written by the producer workgroups of THIS launch: the first batch of x is requested, then the workgroup waits
// at the hand-over point and reads gate / kern coherently
template <int LPR, bool FUSED>
__device__ __forceinline__ void agg_fwd_rows(const AggFwd& g, int blk, const Handover hd = Handover{nullptr, nullptr}) {
  int sub;
  const Row r = row_of<LPR>(blk, g.C, g.N, g.T, &sub);
  const int C = g.C, T = g.T, HW = g.HW, t = r.t;
  float s, sh;
  bn_coef(g.bn, r.c, s, sh);
  const int64_t nc = (int64_t)r.n * C + r.c;
  const int64_t off = ((int64_t)r.c * g.N * T + r.f) * HW;
  const float* xc = g.x + off;
  const float* xp = t > 0 ? xc - HW : xc;
  const float* xn = t + 1 < T ? xc + HW : xc;
  float* o = g.out + off;
  const bool vec = (HW & 3) == 0;
  const float4 *p4 = reinterpret_cast<const float4*>(xp), *c4 = reinterpret_cast<const float4*>(xc), *n4 = reinterpret_cast<const float4*>(xn);
  const int q4 = HW >> 2;
  float4 a[EWU], b[EWU], c[EWU];
  auto request = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < EWU; ++u) {
      const int i = min(i0 + u * LPR, q4 - 1);
      a[u] = p4[i]; b[u] = c4[i]; c[u] = n4[i];
    }
  };
  if (vec) request(sub);  // (rows past the end read row 0: harmless)
  float k0, k1, k2, gp, gc, gn;
  if constexpr (FUSED) {
    asm volatile("" ::: "memory");  // the requests above are issued in front of the wait
    wait_done(hd, blk);
    k0 = coh_load1(g.kern, nc * 3); k1 = coh_load1(g.kern, nc * 3 + 1); k2 = coh_load1(g.kern, nc * 3 + 2);
    gp = coh_load1(g.gate, nc * T + max(t - 1, 0)); gc = coh_load1(g.gate, nc * T + t); gn = coh_load1(g.gate, nc * T + min(t + 1, T - 1));
  } else {
    const float* gt = g.gate + nc * T;
    const float* k = g.kern + nc * 3;
    k0 = k[0]; k1 = k[1]; k2 = k[2];
    gp = gt[max(t - 1, 0)]; gc = gt[t]; gn = gt[min(t + 1, T - 1)];
  }
  if (!r.ok) return;
  const float w0 = t > 0 ? k0 * gp : 0.f;
  const float w1 = k1 * gc;
  const float w2 = t + 1 < T ? k2 * gn : 0.f;
  if (vec) {
    float4* o4 = reinterpret_cast<float4*>(o);
    // (EWU pieces per lane and stream in flight together -- the one-piece loop was a chain of HW / (4 LPR) dependent round trips per
    // lane, 12 at 56 x 56, which is what these launches cost: 2 TB/s at layer 1)
    for (int i0 = sub; i0 < q4;) {
#pragma unroll
      for (int u = 0; u < EWU; ++u) {
        float4 q;
        q.x = fmaf(w2, act(c[u].x, s, sh), fmaf(w1, act(b[u].x, s, sh), w0 * act(a[u].x, s, sh)));
        q.y = fmaf(w2, act(c[u].y, s, sh), fmaf(w1, act(b[u].y, s, sh), w0 * act(a[u].y, s, sh)));
        q.z = fmaf(w2, act(c[u].z, s, sh), fmaf(w1, act(b[u].z, s, sh), w0 * act(a[u].z, s, sh)));
        q.w = fmaf(w2, act(c[u].w, s, sh), fmaf(w1, act(b[u].w, s, sh), w0 * act(a[u].w, s, sh)));
        if (i0 + u * LPR < q4) o4[i0 + u * LPR] = q;
      }
      i0 += LPR * EWU;
      if (i0 < q4) request(i0);
    }
  } else {
    for (int i = sub; i < HW; i += LPR)
      o[i] = fmaf(w2, act(xn[i], s, sh), fmaf(w1, act(xc[i], s, sh), w0 * act(xp[i], s, sh)));
  }
}

// ---- backward of the aggregation: d a[c][n,t',:] = gate[t'] * (K0 gout[t'+1] + K1 gout[t'] + K2 gout[t'-1]); D[t', j] = <gout[t'-j+1], a[t']>
// -> d gate, d K.  The workgroup holds whole (c, n) groups of T rows (rows per workgroup % T == 0): d gate / d K of its groups are
// finished here from LDS.  WT: every result leaves with write-through stores (consumers in the same launch).
struct AggBwd {
  const float* x;
  BN bn;
  const float* gate;
  const float* kern;
  const float* gout;
  int C, N, T, HW;
  int64_t xld;
  float* ga;
  float* ggate;
  float* gkern;
};

template <int LPR, bool WT>
__device__ __forceinline__ void agg_bwd_rows_fin(const AggBwd& g, int blk, float* sd /* [RPB * 3] LDS */) {
  constexpr int RPB = VITTA_BLOCK / LPR;
  const int C = g.C, N = g.N, T = g.T, HW = g.HW;
  int sub;
  const Row r = row_of<LPR>(blk, C, N, T, &sub);
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (r.ok) {
    float s, sh;
    bn_coef(g.bn, r.c, s, sh);
    const int64_t nc = (int64_t)r.n * C + r.c;
    const float* k = g.kern + nc * 3;
    const int t = r.t;
    const float gt = g.gate[nc * T + t];
    const bool hn = t + 1 < T, hp = t > 0;
    const float v0 = hn ? gt * k[0] : 0.f;  // multiplies gout[t+1]
    const float v1 = gt * k[1];
    const float v2 = hp ? gt * k[2] : 0.f;  // multiplies gout[t-1]
    const int64_t off = ((int64_t)r.c * N * T + r.f) * HW;
    const float* xc = g.x + (int64_t)r.c * g.xld + (int64_t)r.f * HW;
    const float* gc = g.gout + off;
    const float* gn = hn ? gc + HW : gc;
    const float* gp = hp ? gc - HW : gc;
    float* o = g.ga + off;
    if ((HW & 3) == 0) {
      const float4 *x4 = reinterpret_cast<const float4*>(xc), *c4 = reinterpret_cast<const float4*>(gc),
                   *n4 = reinterpret_cast<const float4*>(gn), *p4 = reinterpret_cast<const float4*>(gp);
      float4* o4 = reinterpret_cast<float4*>(o);
      const int q4 = HW >> 2;
      for (int i0 = sub; i0 < q4; i0 += LPR * EWU) {  // (batches of EWU pieces per stream in flight, as agg_fwd_rows; same order of sums)
        float4 xr_[EWU], a_[EWU], b_[EWU], c_[EWU];
#pragma unroll
        for (int u = 0; u < EWU; ++u) {
          const int i = min(i0 + u * LPR, q4 - 1);
          xr_[u] = x4[i]; a_[u] = n4[i]; b_[u] = c4[i]; c_[u] = p4[i];
        }
#pragma unroll
        for (int u = 0; u < EWU; ++u) {
          const bool on = i0 + u * LPR < q4;
          const float4 xr = xr_[u], a = a_[u], b = b_[u], c = c_[u];
          const float4 xv = make_float4(act(xr.x, s, sh), act(xr.y, s, sh), act(xr.z, s, sh), act(xr.w, s, sh));
          float4 q;
          q.x = fmaf(v2, c.x, fmaf(v1, b.x, v0 * a.x));
          q.y = fmaf(v2, c.y, fmaf(v1, b.y, v0 * a.y));
          q.z = fmaf(v2, c.z, fmaf(v1, b.z, v0 * a.z));
          q.w = fmaf(v2, c.w, fmaf(v1, b.w, v0 * a.w));
          if (on) {
            if (WT) wt_store4(g.ga, off + 4 * (int64_t)(i0 + u * LPR), q);
            else o4[i0 + u * LPR] = q;
            d0 += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
            d1 += b.x * xv.x + b.y * xv.y + b.z * xv.z + b.w * xv.w;
            d2 += c.x * xv.x + c.y * xv.y + c.z * xv.z + c.w * xv.w;
          }
        }
      }
    } else {
      for (int i = sub; i < HW; i += LPR) {
        const float xv = act(xc[i], s, sh), a = gn[i], b = gc[i], c = gp[i];
        const float q = fmaf(v2, c, fmaf(v1, b, v0 * a));
        if (WT) wt_store1(g.ga, off + i, q);
        else o[i] = q;
        d0 = fmaf(a, xv, d0);
        d1 = fmaf(b, xv, d1);
        d2 = fmaf(c, xv, d2);
      }
    }
    if (!hn) d0 = 0.f;
    if (!hp) d2 = 0.f;
  }
  d0 = group_sum<LPR>(d0);
  d1 = group_sum<LPR>(d1);
  d2 = group_sum<LPR>(d2);
  const int rl = threadIdx.x / LPR;
  if (sub == 0) {
    sd[rl * 3] = r.ok ? d0 : 0.f;
    sd[rl * 3 + 1] = r.ok ? d1 : 0.f;
    sd[rl * 3 + 2] = r.ok ? d2 : 0.f;
  }
  __syncthreads();
  const int gi = threadIdx.x;  // one lane per (c, n) group of the workgroup
  const int64_t row0 = (int64_t)blk * RPB + (int64_t)gi * T;
  if (gi < RPB / T && row0 < (int64_t)C * N * T) {
    const int F = N * T;
    const int c = (int)(row0 / F), n = (int)((row0 - (int64_t)c * F) / T);
    const int64_t i = (int64_t)n * C + c;
    const float k0 = g.kern[i * 3], k1 = g.kern[i * 3 + 1], k2 = g.kern[i * 3 + 2];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float* d = sd + (gi * T + t) * 3;
      const float gt = g.gate[i * T + t];
      const float gg = k0 * d[0] + k1 * d[1] + k2 * d[2];
      if (WT) wt_store1(g.ggate, i * T + t, gg);
      else g.ggate[i * T + t] = gg;
      g0 = fmaf(gt, d[0], g0);
      g1 = fmaf(gt, d[1], g1);
      g2 = fmaf(gt, d[2], g2);
    }
    if (WT) {
      wt_store1(g.gkern, i * 3, g0); wt_store1(g.gkern, i * 3 + 1, g1); wt_store1(g.gkern, i * 3 + 2, g2);
    } else {
      g.gkern[i * 3] = g0; g.gkern[i * 3 + 1] = g1; g.gkern[i * 3 + 2] = g2;
    }
  }
}

// ---- BatchNorm (+ReLU) backward, channel-major planes -------------------------------------------------------------------------
// workgroup = (pixel chunk, channel): per-channel constants are workgroup-uniform, d gamma / d beta leave with one atomic pair per
// workgroup.   dz = g * mask + gscale (a_c + b_c (z - mu_c)),  d gamma += sum dz x_hat, d beta += sum dz,  dx = dz * s_c.
constexpr int BB_UNROLL = 4;  // float4 per lane
struct BnBwd {
  const float* g;      // gradient arriving at the (activated) BN output [C][P]
  const float* g2;     // optional second gradient, added
  const float* x;      // raw convolution output [C][P]
  const float* mask;   // optional: tensor whose sign is the ReLU mask (else z > 0)
  const float* rowadd; // optional [N][C][T]: added to g per (n, c, t) row, scaled by rowadd_scale (TAM pooling gradient)
  float rowadd_scale;
  BN bn;
  const float *mu, *ca, *cb, *gs;
  float* dx;           // [C][P]  dz * s
  float* gm;           // optional [C][P]  (g + g2 + rowadd) * mask
  float *dgamma, *dbeta;
  int C, N, T, HW, relu;
  int64_t xld;         // pixels between channel rows of x / mask (P unless they hold more frames)
  vitta_conv::FastDiv d_hw, d_t;  // host-made reciprocals of HW and T (the frame of a pixel, the clip of a frame)
};

// G2 / MASK: the optional streams exist; ROWADD 0: none, 1: HW % 4 == 0 (a 16-byte piece lies in one frame: one row value per piece),
// 2: any HW.  Compile-time, and every load of the lane's BB_UNROLL pieces is issued before the first use: with run-time flags and a
// `break` in the piece loop the loads of a piece waited for the previous piece's stores -- four dependent round trips per lane, and with
// the pooling gradient eight integer divisions per piece in front of a dependent gather: 2 TB/s on the 56 x 56 layers (round 5).
// FUSED (merged TAM backward): g is written by the aggregation workgroups of this launch (hand-over point ha) and rowadd by the
// branch workgroups behind them (hb): x is requested, then g after the first wait, then the row values after the second.
template <bool G2, bool MASK, int ROWADD, bool FUSED>
__device__ __forceinline__ void bn_bwd_body(const BnBwd& a, int bx, int c, float (*red)[VITTA_BLOCK / VITTA_WAVE],
                                            const Handover ha = Handover{nullptr, nullptr}, const Handover hb = Handover{nullptr, nullptr}, int id = 0) {
  const int64_t P = (int64_t)a.N * a.T * a.HW;
  const int64_t base = (int64_t)c * P, xbase = (int64_t)c * a.xld;
  const float rstd = rsqrtf(a.bn.v[c] + a.bn.eps);
  const float s = a.bn.g[c] * rstd, t = a.bn.b[c] - a.bn.m[c] * s, rm = a.bn.m[c];
  float ia = 0.f, ib = 0.f, mu = 0.f;
  if (a.mu) {
    const float gsc = a.gs ? a.gs[0] : 1.f;
    ia = gsc * a.ca[c];
    ib = gsc * a.cb[c];
    mu = a.mu[c];
  }
  const bool relu = a.relu & 1, raw = (a.relu & 2) && a.mu;
  float sg = 0.f, sb = 0.f;
  const int64_t p0 = ((int64_t)bx * VITTA_BLOCK * BB_UNROLL + threadIdx.x) * 4;
  const int64_t plast = P - 4;
  float4 gv[BB_UNROLL], xv[BB_UNROLL], hv[BB_UNROLL], mv[BB_UNROLL];
  float ra[BB_UNROLL][4];
  int ri[BB_UNROLL][4];
#pragma unroll
  for (int u = 0; u < BB_UNROLL; ++u) {
    const int64_t p = min(p0 + (int64_t)u * VITTA_BLOCK * 4, plast);
    xv[u] = *reinterpret_cast<const float4*>(a.x + xbase + p);
    if (G2) hv[u] = *reinterpret_cast<const float4*>(a.g2 + base + p);
    if (MASK) mv[u] = *reinterpret_cast<const float4*>(a.mask + xbase + p);
    if (ROWADD == 1) {
      const int f = vitta_conv::fdiv((int)p, a.d_hw), n = vitta_conv::fdiv(f, a.d_t), tt = f - n * a.T;
      ri[u][0] = (n * a.C + c) * a.T + tt;
    } else if (ROWADD == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = vitta_conv::fdiv((int)p + e, a.d_hw), n = vitta_conv::fdiv(f, a.d_t), tt = f - n * a.T;
        ri[u][e] = (n * a.C + c) * a.T + tt;
      }
    }
  }
  if constexpr (FUSED) {
    asm volatile("" ::: "memory");
    wait_done(ha, id);
#pragma unroll
    for (int u = 0; u < BB_UNROLL; ++u) gv[u] = coh_load4(a.g, base + min(p0 + (int64_t)u * VITTA_BLOCK * 4, plast));
    asm volatile("" ::: "memory");
    wait_done(hb, id);
#pragma unroll
    for (int u = 0; u < BB_UNROLL; ++u) {
      if (ROWADD == 1) ra[u][0] = coh_load1(a.rowadd, ri[u][0]);
      else if (ROWADD == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[u][e] = coh_load1(a.rowadd, ri[u][e]);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < BB_UNROLL; ++u) {
      gv[u] = *reinterpret_cast<const float4*>(a.g + base + min(p0 + (int64_t)u * VITTA_BLOCK * 4, plast));
      if (ROWADD == 1) ra[u][0] = a.rowadd[ri[u][0]];
      else if (ROWADD == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[u][e] = a.rowadd[ri[u][e]];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < BB_UNROLL; ++u) {
    const int64_t p = p0 + (int64_t)u * VITTA_BLOCK * 4;
    const bool on = p < P;
    float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
    const float xr[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
    if (G2) { g[0] += hv[u].x; g[1] += hv[u].y; g[2] += hv[u].z; g[3] += hv[u].w; }
    if (ROWADD == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += a.rowadd_scale * ra[u][0];
    } else if (ROWADD == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += a.rowadd_scale * ra[u][e];
    }
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (MASK) {
      if (relu) { mk[0] = mv[u].x > 0.f; mk[1] = mv[u].y > 0.f; mk[2] = mv[u].z > 0.f; mk[3] = mv[u].w > 0.f; }
    }
    float o[4], gmv[4];
    float sgu = 0.f, sbu = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = fmaf(xr[e], s, t);
      const float m = (relu && !MASK) ? (z > 0.f ? 1.f : 0.f) : mk[e];
      gmv[e] = g[e] * m;
      // statistics-loss gradient of the hooked feature: of z (added before the affine map is differentiated) or -- before_norm
      // hooks, utils/norm_stats_utils.py:185 -- of the RAW input x (added to dx as it is; d gamma / d beta do not see it)
      const float dz = raw ? gmv[e] : gmv[e] + fmaf(ib, z - mu, ia);
      sgu += dz * (xr[e] - rm) * rstd;
      sbu += dz;
      o[e] = raw ? fmaf(dz, s, fmaf(ib, xr[e] - mu, ia)) : dz * s;
    }
    if (on) {
      sg += sgu;
      sb += sbu;
      *reinterpret_cast<float4*>(a.dx + base + p) = make_float4(o[0], o[1], o[2], o[3]);
      if (a.gm) *reinterpret_cast<float4*>(a.gm + base + p) = make_float4(gmv[0], gmv[1], gmv[2], gmv[3]);
    }
  }
  sg = wave_sum(sg);
  sb = wave_sum(sb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[0][wave] = sg;
    red[1][wave] = sb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float g0 = 0.f, b0 = 0.f;
#pragma unroll
    for (int wv = 0; wv < VITTA_BLOCK / VITTA_WAVE; ++wv) {
      g0 += red[0][wv];
      b0 += red[1][wv];
    }
    if (a.dgamma) atomicAdd(a.dgamma + c, g0);
    if (a.dbeta) atomicAdd(a.dbeta + c, b0);
  }
}

}  // namespace tamrows
