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
#include "conv_common.h"

using namespace vitta;

namespace {

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

// rows are enumerated (c, n, t) with t fastest == memory order
template <int LPR, int BLOCK = VITTA_BLOCK>
__device__ __forceinline__ Row row_of(int C, int N, int T, int* sub) {
  constexpr int RPB = BLOCK / LPR;
  const int F = N * T;
  *sub = threadIdx.x % LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
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

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void pool_kernel(const float* __restrict__ x, BN bn, int C, int N, int T, int HW,
                                                           float* __restrict__ pool) {
  int sub;
  const Row r = row_of<LPR>(C, N, T, &sub);
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

constexpr int EWU = 4;  // 16-byte pieces per lane and stream that the element-wise row kernels keep in flight

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void agg_fwd_kernel(const float* __restrict__ x, BN bn, const float* __restrict__ gate,
                                                              const float* __restrict__ kern, int C, int N, int T, int HW,
                                                              float* __restrict__ out) {
  int sub;
  const Row r = row_of<LPR>(C, N, T, &sub);
  if (!r.ok) return;
  float s, sh;
  bn_coef(bn, r.c, s, sh);
  const int64_t nc = (int64_t)r.n * C + r.c;
  const float* g = gate + nc * T;
  const float* k = kern + nc * 3;
  const int t = r.t;
  const float w0 = t > 0 ? k[0] * g[t - 1] : 0.f;
  const float w1 = k[1] * g[t];
  const float w2 = t + 1 < T ? k[2] * g[t + 1] : 0.f;
  const int64_t off = ((int64_t)r.c * N * T + r.f) * HW;
  const float* xc = x + off;
  const float* xp = t > 0 ? xc - HW : xc;
  const float* xn = t + 1 < T ? xc + HW : xc;
  float* o = out + off;
  if ((HW & 3) == 0) {
    const float4 *p4 = reinterpret_cast<const float4*>(xp), *c4 = reinterpret_cast<const float4*>(xc),
                 *n4 = reinterpret_cast<const float4*>(xn);
    float4* o4 = reinterpret_cast<float4*>(o);
    // (round 5: EWU pieces per lane and stream in flight together -- the one-piece loop was a chain of HW / (4 LPR) dependent round
    // trips per lane, 12 at 56 x 56, which is what these launches cost: 2 TB/s at layer 1)
    const int q4 = HW >> 2;
    for (int i0 = sub; i0 < q4; i0 += LPR * EWU) {
      float4 a[EWU], b[EWU], c[EWU];
#pragma unroll
      for (int u = 0; u < EWU; ++u) {
        const int i = min(i0 + u * LPR, q4 - 1);
        a[u] = p4[i]; b[u] = c4[i]; c[u] = n4[i];
      }
#pragma unroll
      for (int u = 0; u < EWU; ++u) {
        float4 q;
        q.x = fmaf(w2, act(c[u].x, s, sh), fmaf(w1, act(b[u].x, s, sh), w0 * act(a[u].x, s, sh)));
        q.y = fmaf(w2, act(c[u].y, s, sh), fmaf(w1, act(b[u].y, s, sh), w0 * act(a[u].y, s, sh)));
        q.z = fmaf(w2, act(c[u].z, s, sh), fmaf(w1, act(b[u].z, s, sh), w0 * act(a[u].z, s, sh)));
        q.w = fmaf(w2, act(c[u].w, s, sh), fmaf(w1, act(b[u].w, s, sh), w0 * act(a[u].w, s, sh)));
        if (i0 + u * LPR < q4) o4[i0 + u * LPR] = q;
      }
    }
  } else {
    for (int i = sub; i < HW; i += LPR)
      o[i] = fmaf(w2, act(xn[i], s, sh), fmaf(w1, act(xc[i], s, sh), w0 * act(xp[i], s, sh)));
  }
}

// FIN: the block holds whole (c, n) groups of T rows (rows per block % T == 0): d gate / d K of its groups are finished here
// from LDS (the arithmetic of finish_kernel, same order) instead of by a second launch
// BLOCK: threads per workgroup -- 512 at 56 x 56 (64 lanes per row and still whole (c, n) groups of T = 8 rows per workgroup: 128
// workgroups of EIGHT waves; with 32 lanes per row and four waves the launch was 128 workgroups on 256 CUs, 14 us for 38 MB)
template <int LPR, bool FIN, int BLOCK = VITTA_BLOCK>
__global__ __launch_bounds__(BLOCK) void agg_bwd_kernel(const float* __restrict__ x, BN bn, const float* __restrict__ gate,
                                                              const float* __restrict__ kern, const float* __restrict__ gout,
                                                              int C, int N, int T, int HW, int64_t xld, float* __restrict__ ga,
                                                              float* __restrict__ dots, float* __restrict__ ggate,
                                                              float* __restrict__ gkern) {
  constexpr int RPB = BLOCK / LPR;
  __shared__ float sd[FIN ? RPB * 3 : 1];
  int sub;
  const Row r = row_of<LPR, BLOCK>(C, N, T, &sub);
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

// ---- BatchNorm (+ReLU) backward, channel-major planes ----------------------------------------------------------------
// workgroup = (pixel chunk, channel): per-channel constants are workgroup-uniform, d gamma / d beta leave with one atomic
// pair per workgroup.
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
template <bool G2, bool MASK, int ROWADD>
__global__ __launch_bounds__(VITTA_BLOCK) void bn_bwd_kernel(const BnBwd a) {
  __shared__ float red[2][VITTA_BLOCK / VITTA_WAVE];
  const int c = blockIdx.y;
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
  const int64_t p0 = ((int64_t)blockIdx.x * VITTA_BLOCK * BB_UNROLL + threadIdx.x) * 4;
  const int64_t plast = P - 4;
  float4 gv[BB_UNROLL], xv[BB_UNROLL], hv[BB_UNROLL], mv[BB_UNROLL];
  float ra[BB_UNROLL][4];
#pragma unroll
  for (int u = 0; u < BB_UNROLL; ++u) {
    const int64_t p = min(p0 + (int64_t)u * VITTA_BLOCK * 4, plast);
    gv[u] = *reinterpret_cast<const float4*>(a.g + base + p);
    xv[u] = *reinterpret_cast<const float4*>(a.x + xbase + p);
    if (G2) hv[u] = *reinterpret_cast<const float4*>(a.g2 + base + p);
    if (MASK) mv[u] = *reinterpret_cast<const float4*>(a.mask + xbase + p);
    if (ROWADD == 1) {
      const int f = vitta_conv::fdiv((int)p, a.d_hw), n = vitta_conv::fdiv(f, a.d_t), tt = f - n * a.T;
      ra[u][0] = a.rowadd[((int64_t)n * a.C + c) * a.T + tt];
    } else if (ROWADD == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = vitta_conv::fdiv((int)p + e, a.d_hw), n = vitta_conv::fdiv(f, a.d_t), tt = f - n * a.T;
        ra[u][e] = a.rowadd[((int64_t)n * a.C + c) * a.T + tt];
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
    for (int w = 0; w < VITTA_BLOCK / VITTA_WAVE; ++w) {
      g0 += red[0][w];
      b0 += red[1][w];
    }
    if (a.dgamma) atomicAdd(a.dgamma + c, g0);
    if (a.dbeta) atomicAdd(a.dbeta + c, b0);
  }
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
  CM_DISPATCH(agg_fwd_kernel, (int64_t)C * N * T, HW, st, d_x, bn, d_gate, d_kern, (int)C, (int)N, (int)T, (int)HW, d_out);
  return VITTA_OK;
}

int vitta_tam_agg_bwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, const float* d_gate, const float* d_kern,
                             const float* d_gout, int32_t C, int32_t N, int32_t T, int32_t HW, float* d_ga, float* d_ggate,
                             float* d_gkern, void* stream) {
  if (!d_x || !bn_ok(h_bn) || !d_gate || !d_kern || !d_gout || !d_ga || !d_ggate || !d_gkern || bad(C, N, T, HW))
    return VITTA_ERR_INVALID_ARG;
  const int64_t xld = (int64_t)N * T * HW;  // pixels per channel row
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BN bn{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  float* dots = d_ggate + (int64_t)N * C * T;  // the caller gives d_ggate room for N*C*T*4 floats
  const int64_t rows = (int64_t)C * N * T;
  // whole (c, n) groups per workgroup -> d gate / d K finished in the same launch (T = 8: 32 or 16 lanes per row)
  if (HW > 256 && (512 / 64) % T == 0 && (HW & 3) == 0) {
    VITTA_LAUNCH((agg_bwd_kernel<64, true, 512>), dim3((unsigned)((rows + 7) / 8)), dim3(512), 0, st, d_x, bn, d_gate, d_kern, d_gout, (int)C,
                 (int)N, (int)T, (int)HW, xld, d_ga, dots, d_ggate, d_gkern);
    return VITTA_OK;
  }
  if (HW > 256 && (VITTA_BLOCK / 32) % T == 0) {
    VITTA_LAUNCH((agg_bwd_kernel<32, true>), dim3(row_grid(rows, 32)), dim3(VITTA_BLOCK), 0, st, d_x, bn, d_gate, d_kern, d_gout, (int)C,
                 (int)N, (int)T, (int)HW, xld, d_ga, dots, d_ggate, d_gkern);
    return VITTA_OK;
  }
  if (HW <= 256 && (VITTA_BLOCK / 16) % T == 0) {
    VITTA_LAUNCH((agg_bwd_kernel<16, true>), dim3(row_grid(rows, 16)), dim3(VITTA_BLOCK), 0, st, d_x, bn, d_gate, d_kern, d_gout, (int)C,
                 (int)N, (int)T, (int)HW, xld, d_ga, dots, d_ggate, d_gkern);
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

int vitta_bn_bwd_cm_f32(const float* d_g, const float* d_g2, const float* d_x, const float* d_mask, const float* d_rowadd,
                        float rowadd_scale, const float* const* h_bn, float eps, const float* d_mu, const float* d_coef_a,
                        const float* d_coef_b, const float* d_gscale, int32_t relu, float* d_dx, float* d_gm, float* d_dgamma,
                        float* d_dbeta, int32_t C, int32_t N, int32_t T, int32_t HW, void* stream) {
  if (!d_g || !d_x || !bn_ok(h_bn) || !d_dx || bad(C, N, T, HW) || C > 65535) return VITTA_ERR_INVALID_ARG;
  const int64_t P = (int64_t)N * T * HW;
  if (P % 4) return VITTA_ERR_UNSUPPORTED;
  if (d_mu && (!d_coef_a || !d_coef_b)) return VITTA_ERR_INVALID_ARG;
  BnBwd a;
  a.g = d_g; a.g2 = d_g2; a.x = d_x; a.mask = d_mask; a.rowadd = d_rowadd; a.rowadd_scale = rowadd_scale;
  a.bn = BN{h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps};
  a.mu = d_mu; a.ca = d_coef_a; a.cb = d_coef_b; a.gs = d_gscale;
  a.dx = d_dx; a.gm = d_gm; a.dgamma = d_dgamma; a.dbeta = d_dbeta;
  a.C = C; a.N = N; a.T = T; a.HW = HW; a.relu = relu;
  a.xld = P;
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
