// EMA update + statistics-alignment loss (A3/A4), its backward w.r.t. the hooked
// feature (A6) and the prediction-consistency loss (A5).
//
// Reference semantics reproduced exactly:
//   MovingAverageTensor.update   utils/utils_.py:204-211   avg <- m*val + (1-m)*avg.detach(), avg0 = 0
//   compute_regularization       utils/norm_stats_utils.py:531-542
//   compute_kld                  utils/norm_stats_utils.py:8-16
//   compute_pred_consis          utils/pred_consistency_utils.py:15-31
// Because only the current batch term of the EMA carries gradient (factor m), the
// whole stat-loss backward collapses to two per-channel coefficients:
//   dL/dx[i,c] = a_c + b_c (x[i,c] - mu_c),  a_c = (dL/dmean_c)/n,  b_c = 2 (dL/dvar_c)/n.
#include "common.h"

using namespace vitta;

namespace {

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// One workgroup per hooked layer: finish the moments of its channels, update both EMAs, emit the backward coefficients and the
// layer's loss (fixed-order tree sum of the channel terms -> r_feature[l]); the LAST workgroup to arrive adds the layers in layer
// order (= the reference's `loss_reg += hook.r_feature`) and, if asked, zeroes one more device word (the engine's gradient
// scale, which a step resets here).  (Rounds 1-4: three dependent launches -- channels, layers, total -- ~15 us of the chain.)
__global__ __launch_bounds__(VITTA_BLOCK) void stat_align_kernel(
    const LayerInfo* __restrict__ linfo, int n_layers,
    const float* __restrict__ shift, const float* __restrict__ cnt, const float* __restrict__ s1,
    const float* __restrict__ s2, float* __restrict__ ema_mean, float* __restrict__ ema_var,
    const float* __restrict__ src_mean, const float* __restrict__ src_var, float momentum, int reg_type,
    float* __restrict__ mu_out, float* __restrict__ coef_a, float* __restrict__ coef_b, float* layer_loss,
    float* __restrict__ total, unsigned* ticket, float* __restrict__ zero_word) {
  __shared__ float red[VITTA_BLOCK / VITTA_WAVE];
  __shared__ int last_flag;
  const int l = blockIdx.x;
  const LayerInfo L = linfo[l];
  const float C = (float)L.C;
  const double n = (double)cnt[l];
  const float inv_n = n > 0.0 ? (float)(1.0 / n) : 0.f;
  float acc = 0.f;
  constexpr int U = 4;  // channels per thread whose operands are in flight together (a thread walks c = tid, tid + 256, ...)
  for (int c0 = threadIdx.x; c0 < L.C; c0 += U * VITTA_BLOCK) {
    float a1[U], a2[U], ak[U], aem[U], aev[U], asm_[U], asv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * VITTA_BLOCK;
      const int64_t g = L.chan_off + (c < L.C ? c : c0);
      a1[u] = s1[g]; a2[u] = s2[g]; ak[u] = shift ? shift[g] : 0.f;
      aem[u] = ema_mean[g]; aev[u] = ema_var[g]; asm_[u] = src_mean[g]; asv[u] = src_var[g];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * VITTA_BLOCK;
      if (c >= L.C) continue;
      const int64_t g = L.chan_off + c;
      const double m1 = n > 0.0 ? (double)a1[u] / n : 0.0;
      const double m2 = n > 0.0 ? (double)a2[u] / n : 0.0;
      const float mu = (float)((double)ak[u] + m1);
      double vd = m2 - m1 * m1;
      const float var = (float)(vd > 0.0 ? vd : 0.0);

      const float em = momentum * mu + (1.f - momentum) * aem[u];
      const float ev = momentum * var + (1.f - momentum) * aev[u];
      ema_mean[g] = em;
      ema_var[g] = ev;

      const float sm = asm_[u], sv = asv[u];
      float t, gm, gv;  // loss term, dL/d(ema_mean), dL/d(ema_var)
      if (reg_type == VITTA_REG_L1) {
        t = (fabsf(sv - ev) + fabsf(sm - em)) / C;
        gm = sgn(em - sm) / C;
        gv = sgn(ev - sv) / C;
      } else if (reg_type == VITTA_REG_MSE) {
        const float dm = em - sm, dv = ev - sv;
        t = (dv * dv + dm * dm) / C;
        gm = 2.f * dm / C;
        gv = 2.f * dv / C;
      } else {  // KLD: 0.5 log(ev/sv) + (sv + (sm-em)^2) / (2 ev) - 0.5, summed over channels
        const float dm = sm - em;
        t = 0.5f * logf(ev / sv) + (sv + dm * dm) / (2.f * ev) - 0.5f;
        gm = -dm / ev;
        gv = 0.5f / ev - (sv + dm * dm) / (2.f * ev * ev);
      }
      acc += t;
      mu_out[g] = mu;
      coef_a[g] = momentum * gm * inv_n;
      coef_b[g] = 2.f * momentum * gv * inv_n;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < VITTA_BLOCK / VITTA_WAVE; ++w) t += red[w];
    // the layer's loss reaches the coherence point before the arrival (write-through store, drained); the last arriver reads the
    // layers with device-scope loads
    __hip_atomic_store(layer_loss + l, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == (unsigned)n_layers - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = last ? 1 : 0;
  }
  __syncthreads();
  if (!last_flag || threadIdx.x >= VITTA_WAVE) return;
  // the last workgroup's first wave: one load per lane, then the sum in layer order (= `loss_reg += hook.r_feature`)
  float tot = 0.f;
  for (int base = 0; base < n_layers; base += VITTA_WAVE) {
    const int i = base + (int)threadIdx.x;
    const float v = i < n_layers ? __hip_atomic_load(layer_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    const int m = min(VITTA_WAVE, n_layers - base);
    for (int j = 0; j < m; ++j) tot += __shfl(v, j, VITTA_WAVE);
  }
  if (threadIdx.x == 0) {
    *total = tot;
    if (zero_word) *zero_word = 0.f;
  }
}

// ----------------------------------------------------------------------------
// backward injection, NCHW: same flat-chunk walk as the moments kernel; the four
// slots of a lane keep their channel (hence their coefficients) for all frames.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(VITTA_BLOCK) void align_bwd_nchw_kernel(
    const float* __restrict__ x, const float* gout, float* gin, int64_t outer,
    int64_t plane, int64_t HW, int nsplit, const float* __restrict__ mu, const float* __restrict__ ca,
    const float* __restrict__ cb, const float* __restrict__ gscale, int vec) {
  const int64_t base = (int64_t)blockIdx.x * VITTA_CHUNK;
  const int64_t per = (outer + nsplit - 1) / nsplit;
  const int64_t n0 = (int64_t)blockIdx.y * per;
  const int64_t n1 = n0 + per < outer ? n0 + per : outer;
  const float gs = gscale ? *gscale : 1.f;
  const int tid = threadIdx.x;
  if (vec == 4) {
    const int64_t j = base + 4 * (int64_t)tid;
    if (j >= plane) return;
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = (j + k) / HW;
      const float bb = gs * cb[c];
      a[k] = gs * ca[c] - bb * mu[c];  // a + b (x - mu) = (a - b mu) + b x
      b[k] = bb;
    }
    const int64_t stride4 = plane >> 2;
    const float4* px = reinterpret_cast<const float4*>(x + j);
    const float4* pg = gout ? reinterpret_cast<const float4*>(gout + j) : nullptr;
    float4* po = reinterpret_cast<float4*>(gin + j);
#pragma unroll 4
    for (int64_t n = n0; n < n1; ++n) {
      const float4 v = px[n * stride4];
      float4 g = pg ? pg[n * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
      g.x += fmaf(b[0], v.x, a[0]);
      g.y += fmaf(b[1], v.y, a[1]);
      g.z += fmaf(b[2], v.z, a[2]);
      g.w += fmaf(b[3], v.w, a[3]);
      po[n * stride4] = g;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = base + tid + (int64_t)VITTA_BLOCK * k;
      if (j >= plane) continue;
      const int64_t c = j / HW;
      const float bb = gs * cb[c];
      const float aa = gs * ca[c] - bb * mu[c];
      for (int64_t n = n0; n < n1; ++n) {
        const int64_t o = n * plane + j;
        gin[o] = (gout ? gout[o] : 0.f) + fmaf(bb, x[o], aa);
      }
    }
  }
}

// NHWC: grid-stride over float4 quads of the [rows, C] matrix.
__global__ __launch_bounds__(VITTA_BLOCK) void align_bwd_nhwc_kernel(
    const float* __restrict__ x, const float* gout, float* gin, int64_t rows,
    int C, const float* __restrict__ mu, const float* __restrict__ ca, const float* __restrict__ cb,
    const float* __restrict__ gscale, int vec) {
  const float gs = gscale ? *gscale : 1.f;
  const int64_t total = rows * C;
  if (vec == 4) {
    const int64_t nq = total >> 2;
    const int qpr = C >> 2;  // quads per row
    for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < nq;
         i += (int64_t)gridDim.x * VITTA_BLOCK) {
      const int c0 = (int)(i % qpr) * 4;
      const float4 m = *reinterpret_cast<const float4*>(mu + c0);
      const float4 a = *reinterpret_cast<const float4*>(ca + c0);
      const float4 b = *reinterpret_cast<const float4*>(cb + c0);
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 g = gout ? reinterpret_cast<const float4*>(gout)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      g.x += gs * fmaf(b.x, v.x - m.x, a.x);
      g.y += gs * fmaf(b.y, v.y - m.y, a.y);
      g.z += gs * fmaf(b.z, v.z - m.z, a.z);
      g.w += gs * fmaf(b.w, v.w - m.w, a.w);
      reinterpret_cast<float4*>(gin)[i] = g;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * VITTA_BLOCK) {
      const int c = (int)(i % C);
      gin[i] = (gout ? gout[i] : 0.f) + gs * fmaf(cb[c], x[i] - mu[c], ca[c]);
    }
  }
}

// ----------------------------------------------------------------------------
// prediction consistency: one workgroup per video, V*K softmax probabilities in LDS.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(VITTA_BLOCK) void pred_consis_kernel(const float* __restrict__ logits, int V,
                                                                  int K, float* __restrict__ loss_part,
                                                                  float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // p[V*K], gp[V*K], red[8]
  float* p = sm;
  float* gp = sm + (size_t)V * K;
  float* red = gp + (size_t)V * K;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* z = logits + (size_t)b * V * K;
  constexpr int NW = VITTA_BLOCK / VITTA_WAVE;

  // softmax per view: wave w handles views w, w+4, ...
  for (int v = wave; v < V; v += NW) {
    float mx = -INFINITY;
    for (int k = lane; k < K; k += VITTA_WAVE) mx = fmaxf(mx, z[v * K + k]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int k = lane; k < K; k += VITTA_WAVE) {
      const float e = expf(z[v * K + k] - mx);
      p[v * K + k] = e;
      s += e;
    }
    s = wave_sum(s);
    for (int k = lane; k < K; k += VITTA_WAVE) p[v * K + k] = p[v * K + k] / s;
  }
  __syncthreads();
  // per class: mean over views, L1 terms and dL/dp
  float acc = 0.f;
  const float invV = 1.f / (float)V;
  for (int k = tid; k < K; k += VITTA_BLOCK) {
    float pb = 0.f;
    for (int v = 0; v < V; ++v) pb += p[v * K + k];
    pb *= invV;
    float ssum = 0.f;
    for (int v = 0; v < V; ++v) {
      const float d = p[v * K + k] - pb;
      acc += fabsf(d);
      ssum += sgn(d);
    }
    for (int v = 0; v < V; ++v) gp[v * K + k] = invV * (sgn(p[v * K + k] - pb) - invV * ssum);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += red[w];
    loss_part[b] = t * invV;
  }
  if (!grad) return;
  // through the softmax: dz_j = p_j (g_j - sum_k g_k p_k)
  for (int v = wave; v < V; v += NW) {
    float dot = 0.f;
    for (int k = lane; k < K; k += VITTA_WAVE) dot = fmaf(gp[v * K + k], p[v * K + k], dot);
    dot = wave_sum(dot);
    for (int k = lane; k < K; k += VITTA_WAVE)
      grad[(size_t)b * V * K + v * K + k] = p[v * K + k] * (gp[v * K + k] - dot);
  }
}

__global__ void pred_consis_total_kernel(const float* __restrict__ part, int B, float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += part[b];
    *loss = t;
  }
}

}  // namespace

extern "C" {

int vitta_stat_align_fwd_f32(const vitta_plan* p, const float* d_shift, const float* d_cnt,
                             const float* d_s1, const float* d_s2, float* d_ema_mean, float* d_ema_var,
                             const float* d_src_mean, const float* d_src_var, float momentum, int reg_type,
                             float* d_layer_loss, float* d_total_loss, float* d_mu, float* d_coef_a,
                             float* d_coef_b, float* d_zero_word, void* stream) {
  if (!p || !d_cnt || !d_s1 || !d_s2 || !d_ema_mean || !d_ema_var || !d_src_mean || !d_src_var ||
      !d_layer_loss || !d_total_loss || !d_mu || !d_coef_a || !d_coef_b)
    return VITTA_ERR_INVALID_ARG;
  if (reg_type < VITTA_REG_L1 || reg_type > VITTA_REG_KLD || !p->d_info || !p->d_ticket) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  VITTA_LAUNCH(stat_align_kernel, dim3(p->n_layers), dim3(VITTA_BLOCK), 0, st, p->d_info, p->n_layers, d_shift, d_cnt, d_s1, d_s2,
               d_ema_mean, d_ema_var, d_src_mean, d_src_var, momentum, reg_type, d_mu, d_coef_a, d_coef_b, d_layer_loss,
               d_total_loss, p->d_ticket, d_zero_word);
  return VITTA_OK;
}

int vitta_stat_align_bwd_f32(const float* d_x, const float* d_gout, float* d_gin, int64_t outer, int32_t C,
                             int64_t inner, int32_t layout, const float* d_mu, const float* d_coef_a,
                             const float* d_coef_b, const float* d_gscale, void* stream) {
  if (!d_x || !d_gin || !d_mu || !d_coef_a || !d_coef_b || outer <= 0 || C <= 0 || inner <= 0)
    return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool aligned = !((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_gin) |
                          reinterpret_cast<uintptr_t>(d_gout)) & 15u);
  if (layout == VITTA_LAYOUT_NCHW) {
    const int64_t plane = (int64_t)C * inner;
    const int vec = (plane % 4 == 0 && aligned) ? 4 : 1;
    const int64_t nchunks = (plane + VITTA_CHUNK - 1) / VITTA_CHUNK;
    int64_t nsplit = (4096 + nchunks - 1) / nchunks;  // aim at ~4k workgroups
    if (nsplit > outer / 4) nsplit = outer / 4;       // keep >= 4 frames per lane
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    VITTA_LAUNCH(align_bwd_nchw_kernel, dim3((unsigned)nchunks, (unsigned)nsplit), dim3(VITTA_BLOCK), 0,
                       st, d_x, d_gout, d_gin, outer, plane, inner, (int)nsplit, d_mu, d_coef_a, d_coef_b,
                       d_gscale, vec);
  } else if (layout == VITTA_LAYOUT_NHWC) {
    if (inner != 1) return VITTA_ERR_INVALID_ARG;
    const bool caligned = !((reinterpret_cast<uintptr_t>(d_mu) | reinterpret_cast<uintptr_t>(d_coef_a) |
                             reinterpret_cast<uintptr_t>(d_coef_b)) & 15u);
    const int vec = (C % 4 == 0 && aligned && caligned) ? 4 : 1;
    const int64_t work = (outer * C) / vec;
    int64_t grid = (work + VITTA_BLOCK - 1) / VITTA_BLOCK;
    if (grid > 8192) grid = 8192;
    VITTA_LAUNCH(align_bwd_nhwc_kernel, dim3((unsigned)grid), dim3(VITTA_BLOCK), 0, st, d_x, d_gout,
                       d_gin, outer, (int)C, d_mu, d_coef_a, d_coef_b, d_gscale, vec);
  } else {
    return VITTA_ERR_INVALID_ARG;
  }
  return VITTA_OK;
}

int vitta_pred_consis_f32(const float* d_logits, int32_t B, int32_t V, int32_t K, float* d_loss,
                          float* d_grad, void* stream) {
  if (!d_logits || !d_loss || B <= 0 || V <= 0 || K <= 0) return VITTA_ERR_INVALID_ARG;
  const size_t lds = sizeof(float) * (2 * (size_t)V * K + 8);
  if (lds > 48 * 1024) return VITTA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // d_loss has room for 1 + B floats: [0] = total, [1..B] = per-video partial sums (see header)
  float* part = d_loss + 1;
  VITTA_LAUNCH(pred_consis_kernel, dim3(B), dim3(VITTA_BLOCK), lds, st, d_logits, (int)V, (int)K, part,
                     d_grad);
  VITTA_LAUNCH(pred_consis_total_kernel, dim3(1), dim3(VITTA_WAVE), 0, st, part, (int)B, d_loss);
  return VITTA_OK;
}

}  // extern "C"
