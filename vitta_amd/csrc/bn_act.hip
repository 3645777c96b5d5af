// Fused eval-mode BatchNorm2d (+ residual add) (+ ReLU) with the ViTTA statistics riding on the same pass
// ("the hooked-layer reduction rides for free on the BN pass").
//
// During adaptation every BatchNorm runs in eval() (corpus/basics.py:606-611): y = x*scale_c + shift_c with
// scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.  Unfused (reference and
// r1a/r1b profile) one bottleneck conv output crosses HBM as
//   BN fwd 8 B + ReLU 8 B (+ residual add 12 B) + moments 4 B            forward
//   ReLU bwd 12 B + stat-loss injection 12 B + BN bwd 12 B                backward     (per element)
// Here: forward = read x (+ residual), write z, moments of y from registers (y itself is never stored);
// backward = read x, gz (+ z for the residual form), write gx (+ g_residual), with the ReLU mask, the
// injected statistics gradient a_c + b_c (y - mu_c) and the d gamma / d beta reductions in the same pass.
// Same "flat chunk" mapping as moments.hip: a lane's four slots keep their channel for all frames.
#include "common.h"

using namespace vitta;

namespace {

// streaming load: every operand of these passes is read exactly once per pass and not again before the far end of the
// step (the conv output x only in the backward), so the loads carry the non-temporal hint -- what has to stay in the
// Infinity Cache is what the pass WRITES (the next convolution reads z / gx right away).  VITTA_BN_NT=0 at compile
// time restores plain loads.
#ifndef VITTA_BN_NT
#define VITTA_BN_NT 1
#endif
__device__ __forceinline__ float4 ldnt(const float4* p) {
#if VITTA_BN_NT
  float4 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z);
  v.w = __builtin_nontemporal_load(&p->w);
  return v;
#else
  return *p;
#endif
}

// one wave per channel touched by the chunk: sum the per-slot values staged in LDS
__device__ __forceinline__ void segmented_sum2(const float* lds_a, const float* lds_b, int64_t base, int64_t plane,
                                               int64_t HW, float* out2 /* [slots][2] of this (split, chunk) */,
                                               float* acc_a /* or NULL: per-channel totals, added atomically */,
                                               float* acc_b) {
  const int64_t end = base + VITTA_CHUNK < plane ? base + VITTA_CHUNK : plane;
  const int64_t c_lo = base / HW, c_hi = (end - 1) / HW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t c = c_lo + wave; c <= c_hi; c += VITTA_BLOCK / VITTA_WAVE) {
    const int64_t lo = (c * HW > base ? c * HW : base) - base;
    const int64_t hi = ((c + 1) * HW < end ? (c + 1) * HW : end) - base;
    float a = 0.f, b = 0.f;
    for (int64_t o = lo + lane; o < hi; o += VITTA_WAVE) {
      a += lds_a[o];
      b += lds_b[o];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
      if (acc_a) {
        atomicAdd(acc_a + c, a);
        atomicAdd(acc_b + c, b);
      } else {
        out2[2 * (c - c_lo)] = a;
        out2[2 * (c - c_lo) + 1] = b;
      }
    }
  }
}

__device__ __forceinline__ void segmented_moments(float cnt, const float* lds_mean, const float* lds_m2, int64_t base,
                                                  int64_t plane, int64_t HW, float* out3) {
  const int64_t end = base + VITTA_CHUNK < plane ? base + VITTA_CHUNK : plane;
  const int64_t c_lo = base / HW, c_hi = (end - 1) / HW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t c = c_lo + wave; c <= c_hi; c += VITTA_BLOCK / VITTA_WAVE) {
    const int64_t lo = (c * HW > base ? c * HW : base) - base;
    const int64_t hi = ((c + 1) * HW < end ? (c + 1) * HW : end) - base;
    Moments acc{0.f, 0.f, 0.f};
    for (int64_t o = lo + lane; o < hi; o += VITTA_WAVE) acc = merge(acc, Moments{cnt, lds_mean[o], lds_m2[o]});
    acc = wave_merge(acc);
    if (lane == 0) {
      float* t = out3 + 3 * (c - c_lo);
      t[0] = acc.n; t[1] = acc.mean; t[2] = acc.m2;
    }
  }
}

struct BnGeom {
  int64_t outer, plane, HW;
  int nsplit, nchunks, slots;
};

// ------------------------------------------------------------------------------------------------
// forward: z = act(x*scale + shift (+ res)); optional (n, mean, M2) triples of y per (split, chunk, channel)
// ------------------------------------------------------------------------------------------------
template <bool RELU, bool RES, bool STATS>
__global__ __launch_bounds__(VITTA_BLOCK) void bn_act_fwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ res, float* __restrict__ z,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ rmean,
                                                                 const float* __restrict__ rvar, float eps, BnGeom g,
                                                                 float* __restrict__ triples) {
  __shared__ float lds_a[STATS ? VITTA_CHUNK : 1];
  __shared__ float lds_b[STATS ? VITTA_CHUNK : 1];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * VITTA_CHUNK;
  const int64_t per = (g.outer + g.nsplit - 1) / g.nsplit;
  const int64_t n0 = (int64_t)blockIdx.y * per;
  const int64_t n1 = n0 + per < g.outer ? n0 + per : g.outer;
  const int64_t j = base + 4 * (int64_t)tid;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, y0[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < g.plane && n1 > n0) {
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = (j + k) / g.HW;
      sc[k] = weight[c] * rsqrtf(rvar[c] + eps);
      sh[k] = bias[c] - rmean[c] * sc[k];
    }
    const int64_t stride4 = g.plane >> 2;
    const float4* px = reinterpret_cast<const float4*>(x + j);
    const float4* pr = RES ? reinterpret_cast<const float4*>(res + j) : nullptr;
    float4* pz = reinterpret_cast<float4*>(z + j);
    if (STATS) {
      const float4 f = px[n0 * stride4];  // (re-read below: plain load)
      y0[0] = fmaf(f.x, sc[0], sh[0]); y0[1] = fmaf(f.y, sc[1], sh[1]);
      y0[2] = fmaf(f.z, sc[2], sh[2]); y0[3] = fmaf(f.w, sc[3], sh[3]);
    }
#pragma unroll 4
    for (int64_t n = n0; n < n1; ++n) {
      const float4 v = ldnt(px + n * stride4);
      float y[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3])};
      if (STATS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = y[k] - y0[k];
          s[k] += d;
          q[k] = fmaf(d, d, q[k]);
        }
      }
      if (RES) {
        const float4 r = ldnt(pr + n * stride4);
        y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
      }
      if (RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = fmaxf(y[k], 0.f);
      }
      pz[n * stride4] = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
  if (STATS) {
    const float cnt = n1 > n0 ? (float)(n1 - n0) : 0.f;
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lds_a[4 * tid + k] = y0[k] + s[k] * inv;
      lds_b[4 * tid + k] = fmaxf(q[k] - s[k] * s[k] * inv, 0.f);
    }
    __syncthreads();
    float* out3 = triples + 3 * (((int64_t)blockIdx.y * g.nchunks + blockIdx.x) * g.slots);
    segmented_moments(cnt, lds_a, lds_b, base, g.plane, g.HW, out3);
  }
}

// ------------------------------------------------------------------------------------------------
// backward
//   gy = gz * [z > 0] (+ gscale * (a_c + b_c (y - mu_c)))      (mask only with RELU; injection only with INJ)
//   gx = gy * scale_c ; g_res = gz * [z > 0] ; dgamma_c = sum gy * xhat ; dbeta_c = sum gy
// ------------------------------------------------------------------------------------------------
template <bool RELU, bool RES, bool INJ>
__global__ __launch_bounds__(VITTA_BLOCK) void bn_act_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ gz,
    const float* __restrict__ gz2 /* or NULL: a second upstream gradient of z, summed on the fly */,
    float* __restrict__ gx,
    float* __restrict__ gres, const float* __restrict__ weight, const float* __restrict__ bias,
    const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, const float* __restrict__ mu,
    const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ gscale, BnGeom g,
    float* __restrict__ partial /* [nsplit][nchunks][slots][2] */, float* __restrict__ acc_gamma,
    float* __restrict__ acc_beta) {
  __shared__ float lds_a[VITTA_CHUNK];
  __shared__ float lds_b[VITTA_CHUNK];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * VITTA_CHUNK;
  const int64_t per = (g.outer + g.nsplit - 1) / g.nsplit;
  const int64_t n0 = (int64_t)blockIdx.y * per;
  const int64_t n1 = n0 + per < g.outer ? n0 + per : g.outer;
  const int64_t j = base + 4 * (int64_t)tid;
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < g.plane && n1 > n0) {
    float sc[4], sh[4], rm[4], is[4], ia[4], ib[4];
    const float gs = (INJ && gscale) ? *gscale : 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = (j + k) / g.HW;
      rm[k] = rmean[c];
      is[k] = rsqrtf(rvar[c] + eps);
      sc[k] = weight[c] * is[k];
      sh[k] = bias[c] - rm[k] * sc[k];
      if (INJ) {
        ib[k] = gs * cb[c];
        ia[k] = gs * ca[c] - ib[k] * mu[c];  // a + b (y - mu) = (a - b mu) + b y
      }
    }
    const int64_t stride4 = g.plane >> 2;
    const float4* px = reinterpret_cast<const float4*>(x + j);
    const float4* pz = (RELU && RES) ? reinterpret_cast<const float4*>(z + j) : nullptr;
    const float4* pg = reinterpret_cast<const float4*>(gz + j);
    const float4* pg2 = gz2 ? reinterpret_cast<const float4*>(gz2 + j) : nullptr;
    float4* pgx = gx ? reinterpret_cast<float4*>(gx + j) : nullptr;  // null: the input needs no gradient (stem, frozen conv)
    float4* pgr = RES ? reinterpret_cast<float4*>(gres + j) : nullptr;
#pragma unroll 4
    for (int64_t n = n0; n < n1; ++n) {
      const float4 v4 = ldnt(px + n * stride4);
      const float4 g4 = ldnt(pg + n * stride4);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      float gg[4] = {g4.x, g4.y, g4.z, g4.w};
      if (pg2) {
        const float4 h4 = ldnt(pg2 + n * stride4);
        gg[0] += h4.x; gg[1] += h4.y; gg[2] += h4.z; gg[3] += h4.w;
      }
      float zz[4] = {1.f, 1.f, 1.f, 1.f};
      if (RELU && RES) {
        const float4 z4 = ldnt(pz + n * stride4);
        zz[0] = z4.x; zz[1] = z4.y; zz[2] = z4.z; zz[3] = z4.w;
      }
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y = fmaf(v[k], sc[k], sh[k]);
        if (RELU) {
          const bool pos = RES ? (zz[k] > 0.f) : (y > 0.f);
          gg[k] = pos ? gg[k] : 0.f;
        }
        float gy = gg[k];
        if (INJ) gy += fmaf(ib[k], y, ia[k]);
        o[k] = gy * sc[k];
        dg[k] = fmaf(gy, (v[k] - rm[k]) * is[k], dg[k]);
        db[k] += gy;
      }
      if (pgx) pgx[n * stride4] = make_float4(o[0], o[1], o[2], o[3]);
      if (RES) pgr[n * stride4] = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lds_a[4 * tid + k] = dg[k];
    lds_b[4 * tid + k] = db[k];
  }
  __syncthreads();
  float* out2 = acc_gamma ? nullptr : partial + 2 * (((int64_t)blockIdx.y * g.nchunks + blockIdx.x) * g.slots);
  segmented_sum2(lds_a, lds_b, base, g.plane, g.HW, out2, acc_gamma, acc_beta);
}

// sum the (split, chunk) partials of every channel -> dgamma[c], dbeta[c]
__global__ __launch_bounds__(VITTA_BLOCK) void bn_affine_grad_kernel(const float* __restrict__ partial, BnGeom g, int C,
                                                                     float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * VITTA_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t k0 = ((int64_t)c * g.HW) / VITTA_CHUNK;
  const int64_t k1 = (((int64_t)c + 1) * g.HW - 1) / VITTA_CHUNK;
  double a = 0.0, b = 0.0;
  for (int sp = 0; sp < g.nsplit; ++sp)
    for (int64_t k = k0; k <= k1; ++k) {
      const int64_t slot = c - (k * VITTA_CHUNK) / g.HW;
      const float* t = partial + 2 * (((int64_t)sp * g.nchunks + k) * g.slots + slot);
      a += (double)t[0];
      b += (double)t[1];
    }
  dgamma[c] = (float)a;
  dbeta[c] = (float)b;
}

inline int make_geom(int64_t outer, int32_t C, int64_t HW, int nsplit, BnGeom* g) {
  if (outer <= 0 || C <= 0 || HW <= 0 || nsplit <= 0 || nsplit > outer || nsplit > 65535) return VITTA_ERR_INVALID_ARG;
  g->outer = outer;
  g->HW = HW;
  g->plane = (int64_t)C * HW;
  if (g->plane % 4) return VITTA_ERR_UNSUPPORTED;  // the fused form needs the 16-byte path
  g->nsplit = nsplit;
  g->nchunks = (int)((g->plane + VITTA_CHUNK - 1) / VITTA_CHUNK);
  g->slots = (int)((VITTA_CHUNK + HW - 2) / HW + 1);
  return VITTA_OK;
}

inline bool unaligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                      const void* e = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(e)) & 15u) != 0;
}

}  // namespace

#define BN_FWD_CASE(R, S, T)                                                                                          \
  VITTA_LAUNCH((bn_act_fwd_kernel<R, S, T>), grid, dim3(VITTA_BLOCK), 0, st, d_x, d_res, d_z, d_weight, d_bias,    \
               d_rmean, d_rvar, eps, g, d_triples)
#define BN_BWD_CASE(R, S, I)                                                                                          \
  VITTA_LAUNCH((bn_act_bwd_kernel<R, S, I>), grid, dim3(VITTA_BLOCK), 0, st, d_x, d_z, d_gz, d_gz2, d_gx, d_gres, d_weight, \
               d_bias, d_rmean, d_rvar, eps, d_mu, d_coef_a, d_coef_b, d_gscale, g, d_partial, acc_g, acc_b)

extern "C" {

size_t vitta_bn_act_partial_floats(int64_t outer, int32_t C, int64_t HW, int32_t nsplit) {
  BnGeom g;
  if (make_geom(outer, C, HW, nsplit, &g) != VITTA_OK) return 0;
  return (size_t)3 * g.nsplit * g.nchunks * g.slots;  // enough for the forward triples; the backward uses 2/3 of it
}

int vitta_bn_act_fwd_f32(const float* d_x, const float* d_res, float* d_z, const float* d_weight, const float* d_bias,
                         const float* d_rmean, const float* d_rvar, float eps, int64_t outer, int32_t C, int64_t HW,
                         int32_t nsplit, int32_t relu, float* d_triples, void* stream) {
  BnGeom g;
  const int rc = make_geom(outer, C, HW, nsplit, &g);
  if (rc != VITTA_OK) return rc;
  if (!d_x || !d_z || !d_weight || !d_bias || !d_rmean || !d_rvar) return VITTA_ERR_INVALID_ARG;
  if (unaligned(d_x, d_res, d_z)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(g.nchunks, g.nsplit);
  const bool R = relu != 0, S = d_res != nullptr, T = d_triples != nullptr;
  if (R && S && T) BN_FWD_CASE(true, true, true);
  else if (R && S) BN_FWD_CASE(true, true, false);
  else if (R && T) BN_FWD_CASE(true, false, true);
  else if (R) BN_FWD_CASE(true, false, false);
  else if (S && T) BN_FWD_CASE(false, true, true);
  else if (S) BN_FWD_CASE(false, true, false);
  else if (T) BN_FWD_CASE(false, false, true);
  else BN_FWD_CASE(false, false, false);
  return VITTA_OK;
}

int vitta_bn_act_bwd_f32(const float* d_x, const float* d_z, const float* d_gz, const float* d_gz2, float* d_gx, float* d_gres,
                         const float* d_weight, const float* d_bias, const float* d_rmean, const float* d_rvar,
                         float eps, const float* d_mu, const float* d_coef_a, const float* d_coef_b, const float* d_gscale,
                         int64_t outer, int32_t C, int64_t HW, int32_t nsplit, int32_t relu, float* d_partial,
                         float* d_dgamma, float* d_dbeta, int32_t accumulate, void* stream) {
  BnGeom g;
  const int rc = make_geom(outer, C, HW, nsplit, &g);
  if (rc != VITTA_OK) return rc;
  if (!d_x || !d_gz || !d_weight || !d_bias || !d_rmean || !d_rvar || !d_dgamma || !d_dbeta)
    return VITTA_ERR_INVALID_ARG;  // d_gx may be NULL: the BN input needs no gradient
  if (!accumulate && !d_partial) return VITTA_ERR_INVALID_ARG;
  // accumulate: every workgroup adds its per-channel sums straight into the live gradient (fp32 atomics, no partial
  // buffer, no second launch); otherwise partials + a deterministic fp64 finalize
  float* acc_g = accumulate ? d_dgamma : nullptr;
  float* acc_b = accumulate ? d_dbeta : nullptr;
  const bool R = relu != 0, S = d_gres != nullptr, I = d_mu != nullptr;
  if (I && (!d_coef_a || !d_coef_b)) return VITTA_ERR_INVALID_ARG;
  if (R && S && !d_z) return VITTA_ERR_INVALID_ARG;
  if (unaligned(d_x, d_z, d_gz, d_gx, d_gres) || unaligned(d_gz2)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(g.nchunks, g.nsplit);
  if (R && S && I) BN_BWD_CASE(true, true, true);
  else if (R && S) BN_BWD_CASE(true, true, false);
  else if (R && I) BN_BWD_CASE(true, false, true);
  else if (R) BN_BWD_CASE(true, false, false);
  else if (S && I) BN_BWD_CASE(false, true, true);
  else if (S) BN_BWD_CASE(false, true, false);
  else if (I) BN_BWD_CASE(false, false, true);
  else BN_BWD_CASE(false, false, false);
  if (!accumulate)
    VITTA_LAUNCH(bn_affine_grad_kernel, dim3((C + VITTA_BLOCK - 1) / VITTA_BLOCK), dim3(VITTA_BLOCK), 0, st, d_partial, g,
                 (int)C, d_dgamma, d_dbeta);
  return VITTA_OK;
}

/* geometry of layer `layer` of a plan: out[0..4] = nsplit, nchunks, slots, ws_off (triples), vec */
int vitta_plan_layer_geometry(const vitta_plan* plan, int layer, int64_t* out5) {
  if (!plan || !out5 || layer < 0 || layer >= plan->n_layers) return VITTA_ERR_INVALID_ARG;
  const LayerInfo& L = plan->h_info[layer];
  out5[0] = L.nsplit; out5[1] = L.nchunks; out5[2] = L.slots; out5[3] = L.ws_off; out5[4] = L.vec;
  return VITTA_OK;
}

}  // extern "C"
