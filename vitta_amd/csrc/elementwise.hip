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

}  // extern "C"
