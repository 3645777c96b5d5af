// ResNet stem tail: eval-mode BatchNorm -> ReLU -> MaxPool2d(3, stride 2, pad 1) in one pass over the 7x7 convolution's
// output (torchvision ResNet.forward: bn1, relu, maxpool; the largest activation of the network: 16 x 64 x 112 x 112).
//
//   forward : pooled[n,c,ph,pw] = max over the 3x3 window of relu(x * s_c + t_c)     (zero padding == -inf padding after ReLU)
//   backward: ONLY the affine gradients (update_only_bn_affine: the stem convolution is frozen and its input is the
//             video, so no gradient flows below the BN).  The window maximum is recomputed per pooled element, which turns
//             MaxPool's scatter into a plain reduction:  d beta_c = sum g * [ymax > 0],  d gamma_c = sum g * xhat(argmax).
// Unfused this is BN+ReLU pass (102 MB) + max-pool (64 MB) forward and max-pool backward (atomics, 92 us) + BN backward
// (154 MB) per adaptation step; fused 64 MB forward and 64 MB backward.
#include <algorithm>

#include "common.h"

using namespace vitta;

namespace {

struct StemGeom {
  int64_t N; int C, H, W, PH, PW;
  int cm;  // 1: the pooled tensor (forward output / backward upstream gradient) is CHANNEL-MAJOR [C][N * PH * PW] (tiled kernels)
};

// y values of the window of pooled pixel (ph, pw); returns the maximum, *arg = its x value (pre-BN) for the backward
__device__ __forceinline__ float window_max(const float* __restrict__ xp, int H, int W, int ph, int pw, float sc, float sh,
                                            float* arg_x) {
  float best = 0.f, bx = 0.f;  // relu(.) >= 0 and padding contributes 0: starting at 0 is exact
  bool found = false;
  const int h0 = 2 * ph - 1, w0 = 2 * pw - 1;
#pragma unroll
  for (int dh = 0; dh < 3; ++dh) {
    const int h = h0 + dh;
    if (h < 0 || h >= H) continue;
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int w = w0 + dw;
      if (w < 0 || w >= W) continue;
      const float xv = xp[(int64_t)h * W + w];
      const float y = fmaxf(fmaf(xv, sc, sh), 0.f);
      if (!found || y > best) {  // first maximum in scan order, like aten::max_pool2d_with_indices
        best = y; bx = xv; found = true;
      }
    }
  }
  *arg_x = bx;
  return best;
}

__global__ __launch_bounds__(VITTA_BLOCK) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ rmean,
                                                               const float* __restrict__ rvar, float eps, StemGeom g,
                                                               float* __restrict__ out) {
  const int64_t plane_out = (int64_t)g.PH * g.PW;
  const int64_t nc = blockIdx.y;  // (n, c) plane
  const int c = (int)(nc % g.C);
  const float sc = gamma[c] * rsqrtf(rvar[c] + eps);
  const float sh = beta[c] - rmean[c] * sc;
  const float* xp = x + nc * (int64_t)g.H * g.W;
  float* op = out + nc * plane_out;
  for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < plane_out; i += (int64_t)gridDim.x * VITTA_BLOCK) {
    float ax;
    op[i] = window_max(xp, g.H, g.W, (int)(i / g.PW), (int)(i % g.PW), sc, sh, &ax);
  }
}

__global__ __launch_bounds__(VITTA_BLOCK) void stem_bwd_affine_kernel(const float* __restrict__ x, const float* __restrict__ gpool,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                                      float eps, StemGeom g, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta) {
  __shared__ float red[2][VITTA_BLOCK / VITTA_WAVE];
  const int64_t plane_out = (int64_t)g.PH * g.PW;
  const int64_t nc = blockIdx.y;
  const int c = (int)(nc % g.C);
  const float is = rsqrtf(rvar[c] + eps);
  const float sc = gamma[c] * is;
  const float rm = rmean[c];
  const float sh = beta[c] - rm * sc;
  const float* xp = x + nc * (int64_t)g.H * g.W;
  const float* gp = gpool + nc * plane_out;
  float a = 0.f, b = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < plane_out; i += (int64_t)gridDim.x * VITTA_BLOCK) {
    float ax;
    const float ymax = window_max(xp, g.H, g.W, (int)(i / g.PW), (int)(i % g.PW), sc, sh, &ax);
    const float gy = ymax > 0.f ? gp[i] : 0.f;  // ReLU mask of the routed gradient
    a = fmaf(gy, (ax - rm) * is, a);
    b += gy;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int w = 0; w < VITTA_BLOCK / VITTA_WAVE; ++w) { ta += red[0][w]; tb += red[1][w]; }
    atomicAdd(dgamma + c, ta);
    atomicAdd(dbeta + c, tb);
  }
}

// LDS-tiled variants (W % 4 == 0, W <= 256): a workgroup owns SR pooled rows of one (n, c) plane; the 2 SR + 1 input rows
// they need are brought into LDS with 16-byte loads (each input row is read once from memory instead of ~2.25 times
// through stride-2 scalar loads), already normalised and rectified.
constexpr int SR = 8;
constexpr int SW_MAX = 256;

template <bool BWD>
__global__ __launch_bounds__(VITTA_BLOCK) void stem_tiled_kernel(const float* __restrict__ x, const float* __restrict__ gpool,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                                 float eps, StemGeom g, float* __restrict__ out,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ dy) {
  __shared__ __attribute__((aligned(16))) float tile[(2 * SR + 1) * SW_MAX];
  __shared__ float red[2][VITTA_BLOCK / VITTA_WAVE];
  const int64_t nc = blockIdx.y;
  const int c = (int)(nc % g.C);
  const float is = rsqrtf(rvar[c] + eps);
  const float sc = gamma[c] * is;
  const float rm = rmean[c];
  const float sh = beta[c] - rm * sc;
  const int W4 = g.W >> 2;
  const float* xp = x + nc * (int64_t)g.H * g.W;
  float a = 0.f, b = 0.f;
  // forward: one strip per workgroup; backward: the workgroup walks every strip of its plane, so that a plane ends in
  // ONE pair of atomics (7168 workgroups x 2 atomics on 128 addresses cost as much as the 64 MB pass itself)
  for (int strip = BWD ? 0 : (int)blockIdx.x; strip * SR < g.PH; strip += BWD ? 1 : g.PH) {
  const int ph0 = strip * SR;
  const int h_lo = 2 * ph0 - 1;                 // first input row of the strip (may be -1: padding)
  if (BWD && strip > 0) __syncthreads();        // the previous strip's readers are done with the tile
  // stage y = relu(bn(x)) (forward) or raw x (backward: xhat of the arg-max is needed) ; rows outside the image = 0 / -inf
  for (int i = threadIdx.x; i < (2 * SR + 1) * W4; i += VITTA_BLOCK) {
    const int r = i / W4, c4 = i % W4, h = h_lo + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool in = h >= 0 && h < g.H;
    if (in) v = *reinterpret_cast<const float4*>(xp + (int64_t)h * g.W + 4 * c4);
    if (!BWD) {
      v.x = in ? fmaxf(fmaf(v.x, sc, sh), 0.f) : 0.f; v.y = in ? fmaxf(fmaf(v.y, sc, sh), 0.f) : 0.f;
      v.z = in ? fmaxf(fmaf(v.z, sc, sh), 0.f) : 0.f; v.w = in ? fmaxf(fmaf(v.w, sc, sh), 0.f) : 0.f;
    }
    *reinterpret_cast<float4*>(tile + r * g.W + 4 * c4) = v;
  }
  __syncthreads();
  const int rows = min(SR, g.PH - ph0);
  for (int i = threadIdx.x; i < rows * g.PW; i += VITTA_BLOCK) {
    const int pr = i / g.PW, pw = i % g.PW;
    const int w0 = 2 * pw - 1;
    float best = 0.f, bx = 0.f;
    int bpos = 0;
    bool found = false;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int h = h_lo + 2 * pr + dh;
      if (h < 0 || h >= g.H) continue;
      const float* row = tile + (2 * pr + dh) * g.W;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int w = w0 + dw;
        if (w < 0 || w >= g.W) continue;
        const float xv = row[w];
        const float y = BWD ? fmaxf(fmaf(xv, sc, sh), 0.f) : xv;
        if (!found || y > best) { best = y; bx = xv; bpos = h * g.W + w; found = true; }
      }
    }
    const int64_t o = (g.cm ? ((int64_t)c * g.N + nc / g.C) : nc) * (int64_t)g.PH * g.PW + (int64_t)(ph0 + pr) * g.PW + pw;
    if (!BWD) {
      out[o] = best;
    } else {
      const float gy = best > 0.f ? gpool[o] : 0.f;
      a = fmaf(gy, (bx - rm) * is, a);
      b += gy;
      // gradient w.r.t. the convolution output: max-pool routes it to the window's (first) arg-max, ReLU passes it where
      // the activation is positive, BatchNorm scales it; windows overlap, hence the atomic (dy is zeroed by the caller)
      if (dy && gy != 0.f) atomicAdd(dy + nc * (int64_t)g.H * g.W + bpos, gy * sc);
    }
  }
  }  // strips
  if (BWD) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ta = 0.f, tb = 0.f;
      for (int w = 0; w < VITTA_BLOCK / VITTA_WAVE; ++w) { ta += red[0][w]; tb += red[1][w]; }
      atomicAdd(dgamma + c, ta);
      atomicAdd(dbeta + c, tb);
    }
  }
}

inline bool tiled_ok(const StemGeom& g, const void* x) {
  return (g.W % 4) == 0 && g.W <= SW_MAX && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
}

inline int geom(int64_t N, int C, int H, int W, StemGeom* g) {
  if (N <= 0 || C <= 0 || H < 2 || W < 2) return VITTA_ERR_INVALID_ARG;
  g->N = N; g->C = C; g->H = H; g->W = W; g->cm = 0;
  g->PH = (H + 2 - 3) / 2 + 1;
  g->PW = (W + 2 - 3) / 2 + 1;
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_stem_bn_relu_pool_fwd_f32(const float* d_x, const float* const* h_bn, float eps, int64_t N, int32_t C, int32_t H,
                                    int32_t W, float* d_out, void* stream) {
  StemGeom g;
  if (!d_x || !h_bn || !h_bn[0] || !h_bn[1] || !h_bn[2] || !h_bn[3] || !d_out) return VITTA_ERR_INVALID_ARG;
  const int rc = geom(N, C, H, W, &g);
  if (rc != VITTA_OK) return rc;
  if (N * C > 65535) return VITTA_ERR_UNSUPPORTED;
  if (tiled_ok(g, d_x)) {
    VITTA_LAUNCH(stem_tiled_kernel<false>, dim3((g.PH + SR - 1) / SR, (unsigned)(N * C)), dim3(VITTA_BLOCK), 0,
                 static_cast<hipStream_t>(stream), d_x, nullptr, h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps, g, d_out, nullptr,
                 nullptr, nullptr);
    return VITTA_OK;
  }
  const int64_t po = (int64_t)g.PH * g.PW;
  const dim3 grid((unsigned)std::min<int64_t>((po + VITTA_BLOCK - 1) / VITTA_BLOCK, 64), (unsigned)(N * C));
  VITTA_LAUNCH(stem_fwd_kernel, grid, dim3(VITTA_BLOCK), 0, static_cast<hipStream_t>(stream), d_x, h_bn[0], h_bn[1], h_bn[2],
               h_bn[3], eps, g, d_out);
  return VITTA_OK;
}

int vitta_stem_bn_relu_pool_bwd_affine_f32(const float* d_x, const float* d_gpool, const float* const* h_bn, float eps,
                                           int64_t N, int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta,
                                           void* stream) {
  return vitta_stem_bn_relu_pool_bwd_f32(d_x, d_gpool, h_bn, eps, N, C, H, W, d_dgamma, d_dbeta, nullptr, stream);
}

int vitta_stem_bn_relu_pool_bwd_f32(const float* d_x, const float* d_gpool, const float* const* h_bn, float eps, int64_t N,
                                    int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta, float* d_dy, void* stream) {
  StemGeom g;
  if (!d_x || !d_gpool || !h_bn || !h_bn[0] || !h_bn[1] || !h_bn[2] || !h_bn[3] || !d_dgamma || !d_dbeta)
    return VITTA_ERR_INVALID_ARG;
  const int rc = geom(N, C, H, W, &g);
  if (rc != VITTA_OK) return rc;
  if (N * C > 65535) return VITTA_ERR_UNSUPPORTED;
  if (tiled_ok(g, d_x)) {
    VITTA_LAUNCH(stem_tiled_kernel<true>, dim3(1, (unsigned)(N * C)), dim3(VITTA_BLOCK), 0,
                 static_cast<hipStream_t>(stream), d_x, d_gpool, h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps, g, nullptr, d_dgamma,
                 d_dbeta, d_dy);
    return VITTA_OK;
  }
  if (d_dy) return VITTA_ERR_UNSUPPORTED;  // the convolution-output gradient exists on the tiled path only
  // one workgroup per (n, c) plane, or two for large planes: every workgroup ends in two atomics on its channel, and
  // 13 312 workgroups on 128 addresses measured 88 us for this 64 MB pass
  const int64_t po = (int64_t)g.PH * g.PW;
  const dim3 grid(po > 8192 ? 2u : 1u, (unsigned)(N * C));
  VITTA_LAUNCH(stem_bwd_affine_kernel, grid, dim3(VITTA_BLOCK), 0, static_cast<hipStream_t>(stream), d_x, d_gpool, h_bn[0],
               h_bn[1], h_bn[2], h_bn[3], eps, g, d_dgamma, d_dbeta);
  return VITTA_OK;
}

// The same two passes with the POOLED tensor in channel-major planes [C][N * PH * PW] -- the layout of the trunk's convolutions
// (conv.hip) -- so that no transposing copy sits between the stem and layer1 (forward) / in front of the stem's backward.
int vitta_stem_bn_relu_pool_fwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, int64_t N, int32_t C, int32_t H,
                                       int32_t W, float* d_out_cm, void* stream) {
  StemGeom g;
  if (!d_x || !h_bn || !h_bn[0] || !h_bn[1] || !h_bn[2] || !h_bn[3] || !d_out_cm) return VITTA_ERR_INVALID_ARG;
  const int rc = geom(N, C, H, W, &g);
  if (rc != VITTA_OK) return rc;
  if (N * C > 65535 || !tiled_ok(g, d_x)) return VITTA_ERR_UNSUPPORTED;
  g.cm = 1;
  VITTA_LAUNCH(stem_tiled_kernel<false>, dim3((g.PH + SR - 1) / SR, (unsigned)(N * C)), dim3(VITTA_BLOCK), 0,
               static_cast<hipStream_t>(stream), d_x, nullptr, h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps, g, d_out_cm, nullptr, nullptr,
               nullptr);
  return VITTA_OK;
}

int vitta_stem_bn_relu_pool_bwd_cm_f32(const float* d_x, const float* d_gpool_cm, const float* const* h_bn, float eps, int64_t N,
                                       int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta, float* d_dy, void* stream) {
  StemGeom g;
  if (!d_x || !d_gpool_cm || !h_bn || !h_bn[0] || !h_bn[1] || !h_bn[2] || !h_bn[3] || !d_dgamma || !d_dbeta)
    return VITTA_ERR_INVALID_ARG;
  const int rc = geom(N, C, H, W, &g);
  if (rc != VITTA_OK) return rc;
  if (N * C > 65535 || !tiled_ok(g, d_x)) return VITTA_ERR_UNSUPPORTED;
  g.cm = 1;
  VITTA_LAUNCH(stem_tiled_kernel<true>, dim3(1, (unsigned)(N * C)), dim3(VITTA_BLOCK), 0, static_cast<hipStream_t>(stream), d_x,
               d_gpool_cm, h_bn[0], h_bn[1], h_bn[2], h_bn[3], eps, g, nullptr, d_dgamma, d_dbeta, d_dy);
  return VITTA_OK;
}

}  // extern "C"
