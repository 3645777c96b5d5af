// TAM tail and the element-wise backward pieces of a TemporalBottleneck on CHANNEL-MAJOR PLANES (the layout of the
// hand-written convolutions, conv.hip): tensor[c][f * HW + hw], f = n * T + t over all frames of the clip.
//
// Reference: models/tanet_models/temporal_module.py:43-65 (TAM.forward) and :85-106 (TemporalBottleneck.forward).  The
// TAM sits behind conv1 -> bn1 -> relu; conv1's kernel writes the RAW convolution output x1, and every kernel here
// applies a = relu(bn1(x1)) while loading (eval-mode BatchNorm2d: a = max(0, x1 * s_c + t_c)), so the activated
// tensor is never materialised:
//   pooled[n,c,t] = mean_hw a[c][n,t,:]
//   out[c][n,t,:] = sum_j K[n,c,j] * gate[n,c,t+j-1] * a[c][n,t+j-1,:]        (zero padding in t)
// backward, given gout = d out:
//   d a[c][n,t',:] = gate[t'] * (K0 gout[t'+1] + K1 gout[t'] + K2 gout[t'-1])   (+ d pooled[n,c,t'] / HW, added by
//   D[t', j] = <gout[t'-j+1], a[t']>  -> d gate, d K (tam_finish)                  vitta_bn_bwd_cm_f32's row_add)
// and vitta_bn_bwd_cm_f32 is the BatchNorm(+ReLU) backward of any layer of the block in this layout:
//   dz = g * mask + gscale (a_c + b_c (z - mu_c)),  d gamma += sum dz x_hat, d beta += sum dz,  dx = dz * s_c.
#include "tam_rows.h"

using namespace vitta;
using namespace tamrows;

namespace {

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void pool_kernel(const float* __restrict__ x, BN bn, int C, int N, int T, int HW,
                                                           float* __restrict__ pool) {
  int sub;
  const Row r = row_of<LPR>(blockIdx.x, C, N, T, &sub);
  float acc = 0.f;
  if (r.ok) {
    float s, t;
    bn_coef(bn, r.c, s, t);
    const float* row = x + ((int64_t)r.c * N * T + r.f) * HW;
    if ((HW & 3) == 0) {
      const float4* r4 = reinterpret_cast<const float4*>(row);
      for (int i = sub; i < (HW >> 2); i += LPR) {
        const float4 v = r4[i];
        acc += (act(v.x, s, t) + act(v.y, s, t)) + (act(v.z, s, t) + act(v.w, s, t));
      }
    } else {
      for (int i = sub; i < HW; i += LPR) acc += act(row[i], s, t);
    }
  }
  acc = group_sum<LPR>(acc);
  if (r.ok && sub == 0) pool[((int64_t)r.n * C + r.c) * T + r.t] = acc / (float)HW;
}

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void agg_fwd_kernel(const AggFwd g) { agg_fwd_rows<LPR, false>(g, blockIdx.x); }

// FIN: the block holds whole (c, n) groups of T rows (rows per block % T == 0): d gate / d K of its groups are finished here
// from LDS (the arithmetic of finish_kernel, same order) instead of by a second launch
template <int LPR, bool FIN>
__global__ __launch_bounds__(VITTA_BLOCK) void agg_bwd_kernel(const float* __restrict__ x, BN bn, const float* __restrict__ gate,
                                                              const float* __restrict__ kern, const float* __restrict__ gout,
                                                              int C, int N, int T, int HW, int64_t xld, float* __restrict__ ga,
                                                              float* __restrict__ dots, float* __restrict__ ggate,
                                                              float* __restrict__ gkern) {
  constexpr int RPB = VITTA_BLOCK / LPR;
  __shared__ float sd[FIN ? RPB * 3 : 1];
  int sub;
  const Row r = row_of<LPR>(blockIdx.x, C, N, T, &sub);
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (r.ok) {
    float s, sh;
    bn_coef(bn, r.c, s, sh);
    const int64_t nc = (int64_t)r.n * C + r.c;
    const float* k = kern + nc * 3;
    const int t = r.t;
    const float gt = gate[nc * T + t];
    const bool hn = t + 1 < T, hp = t > 0;
    const float v0 = hn ? gt * k[0] : 0.f;  // multiplies gout[t+1]
    const float v1 = gt * k[1];
    const float v2 = hp ? gt * k[2] : 0.f;  // multiplies gout[t-1]
    const int64_t off = ((int64_t)r.c * N * T + r.f) * HW;
    const float* xc = x + (int64_t)r.c * xld + (int64_t)r.f * HW;
    const float* gc = gout + off;
    const float* gn = hn ? gc + HW : gc;
    const float* gp = hp ? gc - HW : gc;
    float* o = ga + off;
    if ((HW & 3) == 0) {
      const float4 *x4 = reinterpret_cast<const float4*>(xc), *c4 = reinterpret_cast<const float4*>(gc),
                   *n4 = reinterpret_cast<const float4*>(gn), *p4 = reinterpret_cast<const float4*>(gp);
      float4* o4 = reinterpret_cast<float4*>(o);
      const int q4 = HW >> 2;
      for (int i0 = sub; i0 < q4; i0 += LPR * EWU) {  // (batches of EWU pieces per stream in flight, as agg_fwd_kernel; same order of sums)
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
            o4[i0 + u * LPR] = q;
            d0 += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
            d1 += b.x * xv.x + b.y * xv.y + b.z * xv.z + b.w * xv.w;
            d2 += c.x * xv.x + c.y * xv.y + c.z * xv.z + c.w * xv.w;
          }
        }
      }
    } else {
      for (int i = sub; i < HW; i += LPR) {
        const float xv = act(xc[i], s, sh), a = gn[i], b = gc[i], c = gp[i];
        o[i] = fmaf(v2, c, fmaf(v1, b, v0 * a));
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
  if constexpr (FIN) {
    const int rl = threadIdx.x / LPR;
    if (sub == 0) {
      sd[rl * 3] = r.ok ? d0 : 0.f;
      sd[rl * 3 + 1] = r.ok ? d1 : 0.f;
      sd[rl * 3 + 2] = r.ok ? d2 : 0.f;
    }
    __syncthreads();
    const int g = threadIdx.x;  // one lane per (c, n) group of the block
    const int64_t row0 = (int64_t)blockIdx.x * RPB + (int64_t)g * T;
    if (g < RPB / T && row0 < (int64_t)C * N * T) {
      const int F = N * T;
      const int c = (int)(row0 / F), n = (int)((row0 - (int64_t)c * F) / T);
      const int64_t i = (int64_t)n * C + c;
      const float k0 = kern[i * 3], k1 = kern[i * 3 + 1], k2 = kern[i * 3 + 2];
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      for (int t = 0; t < T; ++t) {
        const float* d = sd + (g * T + t) * 3;
        const float gt = gate[i * T + t];
        ggate[i * T + t] = k0 * d[0] + k1 * d[1] + k2 * d[2];
        g0 = fmaf(gt, d[0], g0);
        g1 = fmaf(gt, d[1], g1);
        g2 = fmaf(gt, d[2], g2);
      }
      gkern[i * 3] = g0;
      gkern[i * 3 + 1] = g1;
      gkern[i * 3 + 2] = g2;
    }
  } else {
    if (r.ok && sub == 0) {
      float* d = dots + (((int64_t)r.n * C + r.c) * T + r.t) * 3;
      d[0] = d0;
      d[1] = d1;
      d[2] = d2;
    }
  }
}

// one lane per (n, c): ggate[t] = sum_j K[j] D[t,j] ; gK[j] = sum_t gate[t] D[t,j]
__global__ __launch_bounds__(VITTA_BLOCK) void finish_kernel(const float* __restrict__ gate, const float* __restrict__ kern,
                                                             const float* __restrict__ dots, int64_t NC, int T,
                                                             float* __restrict__ ggate, float* __restrict__ gkern) {
  const int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x;
  if (i >= NC) return;
  const float k0 = kern[i * 3], k1 = kern[i * 3 + 1], k2 = kern[i * 3 + 2];
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  for (int t = 0; t < T; ++t) {
    const float* d = dots + (i * T + t) * 3;
    const float gt = gate[i * T + t];
    ggate[i * T + t] = k0 * d[0] + k1 * d[1] + k2 * d[2];
    g0 = fmaf(gt, d[0], g0);
    g1 = fmaf(gt, d[1], g1);
    g2 = fmaf(gt, d[2], g2);
  }
  gkern[i * 3] = g0;
  gkern[i * 3 + 1] = g1;
  gkern[i * 3 + 2] = g2;
}

// ---- BatchNorm (+ReLU) backward, channel-major planes: tam_rows.h ---------------------------------------------------------------
template <bool G2, bool MASK, int ROWADD>
__global__ __launch_bounds__(VITTA_BLOCK) void bn_bwd_kernel(const BnBwd a) {
  __shared__ float red[2][VITTA_BLOCK / VITTA_WAVE];
  bn_bwd_body<G2, MASK, ROWADD, false>(a, blockIdx.x, blockIdx.y, red);
}

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void agg_bwd_fin_kernel(const AggBwd g) {
  __shared__ float sd[VITTA_BLOCK / LPR * 3];
  agg_bwd_rows_fin<LPR, false>(g, blockIdx.x, sd);
}

// ---- head: global average pooling and its backward ----------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void avgpool_kernel(const float* __restrict__ x, int C, int F, int HW,
                                                              float* __restrict__ feat) {
  constexpr int RPB = VITTA_BLOCK / LPR;
  const int sub = threadIdx.x % LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;  // (c, f)
  const bool ok = row < (int64_t)C * F;
  float acc = 0.f;
  if (ok) {
    const float* r = x + row * HW;
    for (int i = sub; i < HW; i += LPR) acc += r[i];
  }
  acc = group_sum<LPR>(acc);
  if (ok && sub == 0) {
    const int c = (int)(row / F), f = (int)(row - (int64_t)c * F);
    feat[(int64_t)f * C + c] = acc / (float)HW;
  }
}

__global__ __launch_bounds__(VITTA_BLOCK) void avgpool_bwd_kernel(const float* __restrict__ gfeat, int C, int F, int HW,
                                                                  float* __restrict__ gx) {
  const int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x;
  if (i >= (int64_t)C * F * HW) return;
  const int64_t row = i / HW;
  const int c = (int)(row / F), f = (int)(row - (int64_t)c * F);
  gx[i] = gfeat[(int64_t)f * C + c] / (float)HW;
}

inline unsigned row_grid(int64_t rows, int lpr) { return (unsigned)((rows + VITTA_BLOCK / lpr - 1) / (VITTA_BLOCK / lpr)); }
inline bool bad(int C, int N, int T, int HW) { return C <= 0 || N <= 0 || T <= 0 || HW <= 0; }
inline bool bn_ok(const float* const* bn) { return bn && bn[0] && bn[1] && bn[2] && bn[3]; }

}  // namespace

#define CM_DISPATCH(KERNEL, rows, HW, st, ...)                                                       \
  do {                                                                                               \
    if ((HW) > 256) VITTA_LAUNCH(KERNEL<64>, dim3(row_grid(rows, 64)), dim3(VITTA_BLOCK), 0, st, __VA_ARGS__); \
    else VITTA_LAUNCH(KERNEL<16>, dim3(row_grid(rows, 16)), dim3(VITTA_BLOCK), 0, st, __VA_ARGS__);   \
  } while (0)

extern "C" {

int vitta_tam_pool_cm_f32(const float* d_x, const float* const* h_bn, float eps, int32_t C, int32_t N, int32_t T, int32_t HW,
                          float* d_pool, void* stream) {
  if (!d_x || !bn_ok(h_bn) || !d_pool || bad(C, N, T, HW)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BN bn{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  CM_DISPATCH(pool_kernel, (int64_t)C * N * T, HW, st, d_x, bn, (int)C, (int)N, (int)T, (int)HW, d_pool);
  return VITTA_OK;
}

int vitta_tam_agg_fwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, const float* d_gate, const float* d_kern,
                             int32_t C, int32_t N, int32_t T, int32_t HW, float* d_out, void* stream) {
  if (!d_x || !bn_ok(h_bn) || !d_gate || !d_kern || !d_out || bad(C, N, T, HW)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BN bn{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  const AggFwd g{d_x, bn, d_gate, d_kern, (int)C, (int)N, (int)T, (int)HW, d_out};
  CM_DISPATCH(agg_fwd_kernel, (int64_t)C * N * T, HW, st, g);
  return VITTA_OK;
}

int vitta_tam_agg_bwd_cm_ld_f32(const float* d_x, int64_t x_ld, const float* const* h_bn, float eps, const float* d_gate,
                                const float* d_kern, const float* d_gout, int32_t C, int32_t N, int32_t T, int32_t HW, float* d_ga,
                                float* d_ggate, float* d_gkern, void* stream) {
  if (!d_x || !bn_ok(h_bn) || !d_gate || !d_kern || !d_gout || !d_ga || !d_ggate || !d_gkern || bad(C, N, T, HW))
    return VITTA_ERR_INVALID_ARG;
  const int64_t xld = x_ld ? x_ld : (int64_t)N * T * HW;
  if (xld < (int64_t)N * T * HW || ((HW & 3) == 0 && (xld & 3))) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BN bn{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  float* dots = d_ggate + (int64_t)N * C * T;  // the caller gives d_ggate room for N*C*T*4 floats
  const int64_t rows = (int64_t)C * N * T;
  const AggBwd gb{d_x, bn, d_gate, d_kern, d_gout, (int)C, (int)N, (int)T, (int)HW, xld, d_ga, d_ggate, d_gkern};
  // whole (c, n) groups per workgroup -> d gate / d K finished in the same launch (T = 8: 32 or 16 lanes per row)
  if (HW > 256 && (VITTA_BLOCK / 32) % T == 0) {
    VITTA_LAUNCH((agg_bwd_fin_kernel<32>), dim3(row_grid(rows, 32)), dim3(VITTA_BLOCK), 0, st, gb);
    return VITTA_OK;
  }
  if (HW <= 256 && (VITTA_BLOCK / 16) % T == 0) {
    VITTA_LAUNCH((agg_bwd_fin_kernel<16>), dim3(row_grid(rows, 16)), dim3(VITTA_BLOCK), 0, st, gb);
    return VITTA_OK;
  }
  if (HW > 256)
    VITTA_LAUNCH((agg_bwd_kernel<64, false>), dim3(row_grid(rows, 64)), dim3(VITTA_BLOCK), 0, st, d_x, bn, d_gate, d_kern, d_gout, (int)C,
                 (int)N, (int)T, (int)HW, xld, d_ga, dots, d_ggate, d_gkern);
  else
    VITTA_LAUNCH((agg_bwd_kernel<16, false>), dim3(row_grid(rows, 16)), dim3(VITTA_BLOCK), 0, st, d_x, bn, d_gate, d_kern, d_gout, (int)C,
                 (int)N, (int)T, (int)HW, xld, d_ga, dots, d_ggate, d_gkern);
  const int64_t NC = (int64_t)N * C;
  VITTA_LAUNCH(finish_kernel, dim3((unsigned)((NC + VITTA_BLOCK - 1) / VITTA_BLOCK)), dim3(VITTA_BLOCK), 0, st, d_gate,
               d_kern, dots, NC, (int)T, d_ggate, d_gkern);
  return VITTA_OK;
}

int vitta_tam_agg_bwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, const float* d_gate, const float* d_kern,
                             const float* d_gout, int32_t C, int32_t N, int32_t T, int32_t HW, float* d_ga, float* d_ggate,
                             float* d_gkern, void* stream) {
  return vitta_tam_agg_bwd_cm_ld_f32(d_x, 0, h_bn, eps, d_gate, d_kern, d_gout, C, N, T, HW, d_ga, d_ggate, d_gkern, stream);
}

int vitta_bn_bwd_cm_ld_f32(const float* d_g, const float* d_g2, const float* d_x, const float* d_mask, int64_t x_ld,
                           const float* d_rowadd, float rowadd_scale, const float* const* h_bn, float eps, const float* d_mu,
                           const float* d_coef_a, const float* d_coef_b, const float* d_gscale, int32_t relu, float* d_dx,
                           float* d_gm, float* d_dgamma, float* d_dbeta, int32_t C, int32_t N, int32_t T, int32_t HW,
                           void* stream) {
  if (!d_g || !d_x || !bn_ok(h_bn) || !d_dx || bad(C, N, T, HW) || C > 65535) return VITTA_ERR_INVALID_ARG;
  const int64_t P = (int64_t)N * T * HW;
  if (P % 4) return VITTA_ERR_UNSUPPORTED;
  if (x_ld && (x_ld < P || x_ld % 4)) return VITTA_ERR_INVALID_ARG;
  if (d_mu && (!d_coef_a || !d_coef_b)) return VITTA_ERR_INVALID_ARG;
  BnBwd a;
  a.g = d_g; a.g2 = d_g2; a.x = d_x; a.mask = d_mask; a.rowadd = d_rowadd; a.rowadd_scale = rowadd_scale;
  a.bn = BN{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  a.mu = d_mu; a.ca = d_coef_a; a.cb = d_coef_b; a.gs = d_gscale;
  a.dx = d_dx; a.gm = d_gm; a.dgamma = d_dgamma; a.dbeta = d_dbeta;
  a.C = C; a.N = N; a.T = T; a.HW = HW; a.relu = relu;
  a.xld = x_ld ? x_ld : P;
  if (P >= (1ll << 31) - 4) return VITTA_ERR_UNSUPPORTED;
  a.d_hw = vitta_conv::make_fastdiv(HW);
  a.d_t = vitta_conv::make_fastdiv(T);
  const int64_t per = (int64_t)VITTA_BLOCK * BB_UNROLL * 4;
  const dim3 grid((unsigned)((P + per - 1) / per), (unsigned)C), block(VITTA_BLOCK);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ra = !d_rowadd ? 0 : (HW % 4 == 0 ? 1 : 2);
#define BN_BWD_GO(G2, MK)                                                             \
  do {                                                                                \
    if (ra == 0) VITTA_LAUNCH((bn_bwd_kernel<G2, MK, 0>), grid, block, 0, st, a);      \
    else if (ra == 1) VITTA_LAUNCH((bn_bwd_kernel<G2, MK, 1>), grid, block, 0, st, a); \
    else VITTA_LAUNCH((bn_bwd_kernel<G2, MK, 2>), grid, block, 0, st, a);              \
  } while (0)
  if (d_g2 && d_mask) BN_BWD_GO(true, true);
  else if (d_g2) BN_BWD_GO(true, false);
  else if (d_mask) BN_BWD_GO(false, true);
  else BN_BWD_GO(false, false);
#undef BN_BWD_GO
  return VITTA_OK;
}

int vitta_bn_bwd_cm_f32(const float* d_g, const float* d_g2, const float* d_x, const float* d_mask, const float* d_rowadd,
                        float rowadd_scale, const float* const* h_bn, float eps, const float* d_mu, const float* d_coef_a,
                        const float* d_coef_b, const float* d_gscale, int32_t relu, float* d_dx, float* d_gm, float* d_dgamma,
                        float* d_dbeta, int32_t C, int32_t N, int32_t T, int32_t HW, void* stream) {
  return vitta_bn_bwd_cm_ld_f32(d_g, d_g2, d_x, d_mask, 0, d_rowadd, rowadd_scale, h_bn, eps, d_mu, d_coef_a, d_coef_b, d_gscale, relu,
                                d_dx, d_gm, d_dgamma, d_dbeta, C, N, T, HW, stream);
}

int vitta_avgpool_cm_f32(const float* d_x, int32_t C, int32_t F, int32_t HW, float* d_feat, void* stream) {
  if (!d_x || !d_feat || C <= 0 || F <= 0 || HW <= 0) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CM_DISPATCH(avgpool_kernel, (int64_t)C * F, HW, st, d_x, (int)C, (int)F, (int)HW, d_feat);
  return VITTA_OK;
}

int vitta_avgpool_cm_bwd_f32(const float* d_gfeat, int32_t C, int32_t F, int32_t HW, float* d_gx, void* stream) {
  if (!d_gfeat || !d_gx || C <= 0 || F <= 0 || HW <= 0) return VITTA_ERR_INVALID_ARG;
  const int64_t n = (int64_t)C * F * HW;
  VITTA_LAUNCH(avgpool_bwd_kernel, dim3((unsigned)((n + VITTA_BLOCK - 1) / VITTA_BLOCK)), dim3(VITTA_BLOCK), 0,
               static_cast<hipStream_t>(stream), d_gfeat, (int)C, (int)F, (int)HW, d_gx);
  return VITTA_OK;
}

}  // extern "C"
