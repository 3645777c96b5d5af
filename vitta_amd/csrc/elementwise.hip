// Residual update with per-sample stochastic depth, one pass:  out = x + s_b * branch
// (Video Swin, swin_transformer.py:268-275: `x = shortcut + self.drop_path(x)`; `x = x + self.drop_path(mlp(...))`;
// timm DropPath: branch * bernoulli(keep)/keep per sample).  As torch ops this is bernoulli + div + mul + add = four
// launches and 2.5x the traffic per site, 48 sites per forward.  s_b is a device array (one value per sample), so the
// pass sits in a captured graph; s == NULL: plain residual add.
#include <algorithm>

#include "common.h"

using namespace vitta;

namespace {

template <bool ADD>
__global__ __launch_bounds__(VITTA_BLOCK) void scale_add_kernel(const float* __restrict__ x, const float* __restrict__ br,
                                                                const float* __restrict__ scale, int64_t per4,
                                                                float* __restrict__ out) {
  const int64_t b = blockIdx.y;
  const float s = scale ? scale[b] : 1.f;
  const float4* pb = reinterpret_cast<const float4*>(br) + b * per4;
  const float4* px = ADD ? reinterpret_cast<const float4*>(x) + b * per4 : nullptr;
  float4* po = reinterpret_cast<float4*>(out) + b * per4;
  const int64_t stride = (int64_t)gridDim.x * VITTA_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < per4; i += stride) {
    const float4 v = pb[i];
    float4 o;
    if (ADD) {
      const float4 a = px[i];
      o = make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
    } else {
      o = make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
    }
    po[i] = o;
  }
}

// PatchMerging's gather (swin_transformer.py:281-286): merged[p][i][j][k C + c] = x[p][2 i + (k & 1)][2 j + (k >> 1)][c], k = 0..3
// (x0 | x1 | x2 | x3 = (even, even) | (odd, even) | (even, odd) | (odd, odd) rows / columns), p = the leading (batch, frame) index.
// FWD: x -> merged; else merged -> x (the gradient: every element exactly once).  One lane per 16 bytes of the MERGED tensor: a row of it
// is four runs of C contiguous floats of x.  torch.cat of the four strided slices is four copy kernels, the gradient four more.
template <bool FWD>
__global__ __launch_bounds__(VITTA_BLOCK) void patch_gather_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n4,
                                                                   int H2, int W2, int C4 /* C / 4 */) {
  const int64_t stride = (int64_t)gridDim.x * VITTA_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int k = (int)(r & 3);
    r >>= 2;
    const int j = (int)(r % W2);
    r /= W2;
    const int ii = (int)(r % H2);
    const int64_t p = r / H2;
    const int64_t xi = (((p * (2 * H2) + 2 * ii + (k & 1)) * (2 * W2)) + 2 * j + (k >> 1)) * C4 + c;
    if (FWD) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[xi];
    else reinterpret_cast<float4*>(dst)[xi] = reinterpret_cast<const float4*>(src)[i];
  }
}

inline bool bad(const void* a, const void* b, const void* c) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15u) != 0;
}

}  // namespace

extern "C" {

int vitta_scale_add_f32(const float* d_x, const float* d_branch, const float* d_scale, int64_t samples,
                        int64_t per_sample, float* d_out, void* stream) {
  if (!d_branch || !d_out || samples <= 0 || samples > 65535 || per_sample <= 0) return VITTA_ERR_INVALID_ARG;
  if (per_sample % 4) return VITTA_ERR_UNSUPPORTED;
  if (bad(d_x, d_branch, d_out)) return VITTA_ERR_INVALID_ARG;
  const int64_t per4 = per_sample / 4;
  const unsigned gx = (unsigned)std::min<int64_t>((per4 + VITTA_BLOCK - 1) / VITTA_BLOCK, 4096);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d_x)
    VITTA_LAUNCH(scale_add_kernel<true>, dim3(gx, (unsigned)samples), dim3(VITTA_BLOCK), 0, st, d_x, d_branch, d_scale, per4,
                 d_out);
  else
    VITTA_LAUNCH(scale_add_kernel<false>, dim3(gx, (unsigned)samples), dim3(VITTA_BLOCK), 0, st, d_x, d_branch, d_scale, per4,
                 d_out);
  return VITTA_OK;
}

int vitta_patch_gather_f32(const float* d_src, float* d_dst, int64_t planes, int32_t H2, int32_t W2, int32_t C, int32_t inverse,
                           void* stream) {
  if (!d_src || !d_dst || planes <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return VITTA_ERR_INVALID_ARG;
  if (C % 4) return VITTA_ERR_UNSUPPORTED;
  if (bad(d_src, d_dst, nullptr)) return VITTA_ERR_INVALID_ARG;
  const int64_t n4 = planes * H2 * W2 * C;  // float4 units of the merged tensor: planes x H2 x W2 x 4 C / 4
  const unsigned gx = (unsigned)std::min<int64_t>((n4 + VITTA_BLOCK - 1) / VITTA_BLOCK, 16384);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (inverse) VITTA_LAUNCH(patch_gather_kernel<false>, dim3(gx), dim3(VITTA_BLOCK), 0, st, d_src, d_dst, n4, (int)H2, (int)W2, (int)(C / 4));
  else VITTA_LAUNCH(patch_gather_kernel<true>, dim3(gx), dim3(VITTA_BLOCK), 0, st, d_src, d_dst, n4, (int)H2, (int)W2, (int)(C / 4));
  return VITTA_OK;
}

}  // extern "C"
