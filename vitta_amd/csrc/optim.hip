// Optimizer updates on the flat parameter arena (vitta_amd/tta.py FlatArena): ONE launch per step.
//
// The reference steps torch.optim.Adam(lr, betas=(0.9, 0.999), weight_decay=0) over the BN/LN affine tensors, or
// torch.optim.SGD(lr, momentum, weight_decay) over every parameter (corpus/basics.py:547-560).  Through torch's
// foreach / capturable implementations that is 25 launches (Adam) or 4 (SGD) of 4-8 us on a 40 k .. 88 M element
// buffer -- 0.2 ms of an 8 ms step (r1i profile).  Both updates are element-wise: one kernel, 16 B/element in, 12 out.
#include "common.h"

using namespace vitta;

namespace {

// torch.optim.Adam, single-tensor formulation (torch/optim/adam.py _single_tensor_adam, amsgrad off, maximize off):
//   t = step + 1; g' = g + wd * p; m = lerp(m, g', 1 - b1); v = b2 * v + (1 - b2) g'^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(VITTA_BLOCK) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v,
                                                                float* step, unsigned* ticket, float lr, float b1,
                                                                float b2, float eps, float wd, int64_t n) {
  const double t = (double)*step + 1.0;
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  const float step_size = lr / bc1;
  const float w1 = 1.f - b1, w2 = 1.f - b2;
  const int64_t i0 = ((int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x) * 4;
  if (i0 + 4 <= n) {
    float4 pp = *reinterpret_cast<float4*>(p + i0);
    const float4 gg = *reinterpret_cast<const float4*>(g + i0);
    float4 mm = *reinterpret_cast<float4*>(m + i0);
    float4 vv = *reinterpret_cast<float4*>(v + i0);
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = wd != 0.f ? fmaf(wd, pa[k], ga[k]) : ga[k];
      ma[k] = ma[k] + w1 * (gr - ma[k]);
      va[k] = va[k] * b2 + w2 * gr * gr;
      const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
      pa[k] = pa[k] - step_size * (ma[k] / denom);
    }
    *reinterpret_cast<float4*>(p + i0) = pp;
    *reinterpret_cast<float4*>(m + i0) = mm;
    *reinterpret_cast<float4*>(v + i0) = vv;
  } else {
    for (int64_t i = i0; i < n; ++i) {
      const float gr = wd != 0.f ? fmaf(wd, p[i], g[i]) : g[i];
      m[i] = m[i] + w1 * (gr - m[i]);
      v[i] = v[i] * b2 + w2 * gr * gr;
      p[i] = p[i] - step_size * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
    }
  }
  // step += 1 once every workgroup has read it: the last one to arrive (*ticket: arrival counter of the caller's, zero at rest)
  // writes it.  (Rounds 1-4: a second one-thread launch; round 5: the counter sat in step[1].)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *step = (float)t;
    }
  }
}

// torch.optim.SGD (dampening 0, no nesterov): d = g + wd p; buf = mu buf + d (buf starts at 0, so the first step
// gives buf = d like torch's clone); p -= lr buf.  mu == 0: p -= lr d, buf untouched (may be NULL).
__global__ __launch_bounds__(VITTA_BLOCK) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                               float* __restrict__ buf, float lr, float mu, float wd,
                                                               int64_t n) {
  const int64_t i0 = ((int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x) * 4;
  if (i0 >= n) return;
  if (i0 + 4 <= n) {
    float4 pp = *reinterpret_cast<float4*>(p + i0);
    const float4 gg = *reinterpret_cast<const float4*>(g + i0);
    float4 bb = buf ? *reinterpret_cast<float4*>(buf + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* pa = &pp.x; const float* ga = &gg.x; float* ba = &bb.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = wd != 0.f ? ga[k] + wd * pa[k] : ga[k];
      if (buf) {
        ba[k] = ba[k] * mu + d;
        d = ba[k];
      }
      pa[k] = pa[k] - lr * d;
    }
    *reinterpret_cast<float4*>(p + i0) = pp;
    if (buf) *reinterpret_cast<float4*>(buf + i0) = bb;
  } else {
    for (int64_t i = i0; i < n; ++i) {
      float d = wd != 0.f ? g[i] + wd * p[i] : g[i];
      if (buf) {
        buf[i] = buf[i] * mu + d;
        d = buf[i];
      }
      p[i] = p[i] - lr * d;
    }
  }
}

inline bool misaligned(const void* a, const void* b, const void* c, const void* d) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15u) != 0;
}

}  // namespace

extern "C" {

int vitta_adam_step_f32(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_step,
                        uint32_t* d_ticket, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t n, void* stream) {
  if (!d_param || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_step || !d_ticket || n <= 0) return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_param, d_grad, d_exp_avg, d_exp_avg_sq)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t grid = (n + 4 * VITTA_BLOCK - 1) / (4 * VITTA_BLOCK);
  VITTA_LAUNCH(adam_step_kernel, dim3((unsigned)grid), dim3(VITTA_BLOCK), 0, st, d_param, d_grad, d_exp_avg, d_exp_avg_sq,
               d_step, d_ticket, lr, beta1, beta2, eps, weight_decay, n);
  return VITTA_OK;
}

int vitta_sgd_step_f32(float* d_param, const float* d_grad, float* d_momentum_buf, float lr, float momentum,
                       float weight_decay, int64_t n, void* stream) {
  if (!d_param || !d_grad || n <= 0 || (momentum != 0.f && !d_momentum_buf)) return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_param, d_grad, d_momentum_buf, nullptr)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t grid = (n + 4 * VITTA_BLOCK - 1) / (4 * VITTA_BLOCK);
  VITTA_LAUNCH(sgd_step_kernel, dim3((unsigned)grid), dim3(VITTA_BLOCK), 0, st, d_param, d_grad,
               momentum != 0.f ? d_momentum_buf : nullptr, lr, momentum, weight_decay, n);
  return VITTA_OK;
}

}  // extern "C"
