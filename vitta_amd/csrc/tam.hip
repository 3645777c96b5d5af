// TAM (temporal adaptive module) memory-bound tail, fused.
//
// Reference (models/tanet_models/temporal_module.py:43-65) on x [N*T, C, H, W]:
//   new_x = x.view(N,T,C,H,W).permute(0,2,1,3,4).contiguous()          copy
//   pooled = adaptive_avg_pool2d(new_x.view(N*C,T,H,W), 1)              read
//   new_x = new_x * local_activation                                    read+write
//   out = conv2d(new_x.view(1,N*C,T,HW), kernel, pad (1,0), groups N*C) read+write
//   out = out.view(N,C,T,H,W).permute(0,2,1,3,4).contiguous()           copy
// Here x stays in its [N*T, C, HW] layout: a row is the HW contiguous floats of one
// (clip n, frame t, channel c).  LPR lanes own a row; the temporal taps touch rows
// t-1, t, t+1 of the same (n, c), which sit C*HW floats apart.
//   out[n,t,c,:] = sum_j K[n,c,j] * gate[n,c,t+j-1] * x[n,t+j-1,c,:]   (cross-correlation, zero pad)
#include "common.h"

using namespace vitta;

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, VITTA_WAVE);
  return v;
}

struct RowId {
  int n, t, c;
  bool ok;
};

// rows are enumerated (n, t, c) with c fastest == memory order of x
template <int LPR>
__device__ __forceinline__ RowId row_of(int64_t nrows, int T, int C, int* sub) {
  constexpr int RPB = VITTA_BLOCK / LPR;  // rows per workgroup
  const int r_in = threadIdx.x / LPR;
  *sub = threadIdx.x % LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + r_in;
  RowId id;
  id.ok = row < nrows;
  const int64_t rr = id.ok ? row : 0;
  id.c = (int)(rr % C);
  id.t = (int)((rr / C) % T);
  id.n = (int)(rr / ((int64_t)C * T));
  return id;
}

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void tam_pool_kernel(const float* __restrict__ x, int N, int T, int C,
                                                               int HW, float* __restrict__ pool) {
  int sub;
  const RowId id = row_of<LPR>((int64_t)N * T * C, T, C, &sub);
  float acc = 0.f;
  if (id.ok) {
    const float* row = x + (((int64_t)id.n * T + id.t) * C + id.c) * HW;
    if ((HW & 3) == 0) {
      const float4* r4 = reinterpret_cast<const float4*>(row);
      for (int i = sub; i < (HW >> 2); i += LPR) {
        const float4 v = r4[i];
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int i = sub; i < HW; i += LPR) acc += row[i];
    }
  }
  acc = group_sum<LPR>(acc);
  if (id.ok && sub == 0) pool[((int64_t)id.n * C + id.c) * T + id.t] = acc / (float)HW;
}

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void tam_agg_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ gate,
                                                                  const float* __restrict__ kern, int N, int T,
                                                                  int C, int HW, float* __restrict__ out) {
  int sub;
  const RowId id = row_of<LPR>((int64_t)N * T * C, T, C, &sub);
  if (!id.ok) return;
  const int64_t nc = (int64_t)id.n * C + id.c;
  const float* g = gate + nc * T;
  const float* k = kern + nc * 3;
  const int t = id.t;
  // tap weights w_j = K[j] * gate[t+j-1]; 0 outside [0, T)
  const float w0 = t > 0 ? k[0] * g[t - 1] : 0.f;
  const float w1 = k[1] * g[t];
  const float w2 = t + 1 < T ? k[2] * g[t + 1] : 0.f;
  const int64_t tstride = (int64_t)C * HW;
  const float* xc = x + (((int64_t)id.n * T + t) * C + id.c) * HW;
  const float* xp = t > 0 ? xc - tstride : xc;  // never dereferenced with weight != 0 out of range
  const float* xn = t + 1 < T ? xc + tstride : xc;
  float* o = out + (((int64_t)id.n * T + t) * C + id.c) * HW;
  if ((HW & 3) == 0) {
    const float4 *p4 = reinterpret_cast<const float4*>(xp), *c4 = reinterpret_cast<const float4*>(xc),
                 *n4 = reinterpret_cast<const float4*>(xn);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int i = sub; i < (HW >> 2); i += LPR) {
      const float4 a = p4[i], b = c4[i], c = n4[i];
      float4 r;
      r.x = fmaf(w2, c.x, fmaf(w1, b.x, w0 * a.x));
      r.y = fmaf(w2, c.y, fmaf(w1, b.y, w0 * a.y));
      r.z = fmaf(w2, c.z, fmaf(w1, b.z, w0 * a.z));
      r.w = fmaf(w2, c.w, fmaf(w1, b.w, w0 * a.w));
      o4[i] = r;
    }
  } else {
    for (int i = sub; i < HW; i += LPR) o[i] = fmaf(w2, xn[i], fmaf(w1, xc[i], w0 * xp[i]));
  }
}

// backward, row (n, t', c):
//   gx[t'] = gate[t'] * (K0 gout[t'+1] + K1 gout[t'] + K2 gout[t'-1])
//   D[t', j] = <gout[t'-j+1], x[t']>   (j = 0,1,2)
template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void tam_agg_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ gate,
                                                                  const float* __restrict__ kern,
                                                                  const float* __restrict__ gout, int N, int T,
                                                                  int C, int HW, float* __restrict__ gx,
                                                                  float* __restrict__ dots) {
  int sub;
  const RowId id = row_of<LPR>((int64_t)N * T * C, T, C, &sub);
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (id.ok) {
    const int64_t nc = (int64_t)id.n * C + id.c;
    const float* k = kern + nc * 3;
    const int t = id.t;
    const float gt = gate[nc * T + t];
    const bool hn = t + 1 < T, hp = t > 0;
    const float v0 = hn ? gt * k[0] : 0.f;  // multiplies gout[t+1]
    const float v1 = gt * k[1];
    const float v2 = hp ? gt * k[2] : 0.f;  // multiplies gout[t-1]
    const int64_t tstride = (int64_t)C * HW;
    const int64_t off = (((int64_t)id.n * T + t) * C + id.c) * HW;
    const float* xc = x + off;
    const float* gc = gout + off;
    const float* gn = hn ? gc + tstride : gc;
    const float* gp = hp ? gc - tstride : gc;
    float* o = gx + off;
    if ((HW & 3) == 0) {
      const float4 *x4 = reinterpret_cast<const float4*>(xc), *c4 = reinterpret_cast<const float4*>(gc),
                   *n4 = reinterpret_cast<const float4*>(gn), *p4 = reinterpret_cast<const float4*>(gp);
      float4* o4 = reinterpret_cast<float4*>(o);
      for (int i = sub; i < (HW >> 2); i += LPR) {
        const float4 xv = x4[i], a = n4[i], b = c4[i], c = p4[i];
        float4 r;
        r.x = fmaf(v2, c.x, fmaf(v1, b.x, v0 * a.x));
        r.y = fmaf(v2, c.y, fmaf(v1, b.y, v0 * a.y));
        r.z = fmaf(v2, c.z, fmaf(v1, b.z, v0 * a.z));
        r.w = fmaf(v2, c.w, fmaf(v1, b.w, v0 * a.w));
        o4[i] = r;
        d0 += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
        d1 += b.x * xv.x + b.y * xv.y + b.z * xv.z + b.w * xv.w;
        d2 += c.x * xv.x + c.y * xv.y + c.z * xv.z + c.w * xv.w;
      }
    } else {
      for (int i = sub; i < HW; i += LPR) {
        const float xv = xc[i], a = gn[i], b = gc[i], c = gp[i];
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
  if (id.ok && sub == 0) {
    float* d = dots + (((int64_t)id.n * C + id.c) * T + id.t) * 3;
    d[0] = d0; d[1] = d1; d[2] = d2;
  }
}

// one lane per (n, c): ggate[t] = sum_j K[j] D[t,j] ; gK[j] = sum_t gate[t] D[t,j]
__global__ __launch_bounds__(VITTA_BLOCK) void tam_finish_bwd_kernel(const float* __restrict__ gate,
                                                                     const float* __restrict__ kern,
                                                                     const float* __restrict__ dots, int64_t NC,
                                                                     int T, float* __restrict__ ggate,
                                                                     float* __restrict__ gkern) {
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
  gkern[i * 3] = g0; gkern[i * 3 + 1] = g1; gkern[i * 3 + 2] = g2;
}

template <int LPR>
__global__ __launch_bounds__(VITTA_BLOCK) void tam_pool_bwd_kernel(const float* __restrict__ gpool, int N, int T,
                                                                   int C, int HW, float* __restrict__ gx) {
  int sub;
  const RowId id = row_of<LPR>((int64_t)N * T * C, T, C, &sub);
  if (!id.ok) return;
  const float g = gpool[((int64_t)id.n * C + id.c) * T + id.t] / (float)HW;
  float* o = gx + (((int64_t)id.n * T + id.t) * C + id.c) * HW;
  if ((HW & 3) == 0) {
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int i = sub; i < (HW >> 2); i += LPR) {
      float4 v = o4[i];
      v.x += g; v.y += g; v.z += g; v.w += g;
      o4[i] = v;
    }
  } else {
    for (int i = sub; i < HW; i += LPR) o[i] += g;
  }
}

inline bool bad_dims(int N, int T, int C, int HW) { return N <= 0 || T <= 0 || C <= 0 || HW <= 0; }
inline unsigned row_grid(int64_t rows, int lpr) {
  const int rpb = VITTA_BLOCK / lpr;
  return (unsigned)((rows + rpb - 1) / rpb);
}

}  // namespace

#define TAM_DISPATCH(KERNEL, rows, HW, st, ...)                                                          \
  do {                                                                                                   \
    if ((HW) > 256)                                                                                      \
      VITTA_LAUNCH(KERNEL<64>, dim3(row_grid(rows, 64)), dim3(VITTA_BLOCK), 0, st, __VA_ARGS__);   \
    else                                                                                                 \
      VITTA_LAUNCH(KERNEL<16>, dim3(row_grid(rows, 16)), dim3(VITTA_BLOCK), 0, st, __VA_ARGS__);   \
  } while (0)

extern "C" {

int vitta_tam_pool_f32(const float* d_x, int32_t N, int32_t T, int32_t C, int32_t HW, float* d_pool,
                       void* stream) {
  if (!d_x || !d_pool || bad_dims(N, T, C, HW)) return VITTA_ERR_INVALID_ARG;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(d_x) & 15u)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T * C;
  TAM_DISPATCH(tam_pool_kernel, rows, HW, st, d_x, (int)N, (int)T, (int)C, (int)HW, d_pool);
  return VITTA_OK;
}

int vitta_tam_agg_fwd_f32(const float* d_x, const float* d_gate, const float* d_kern, int32_t N, int32_t T,
                          int32_t C, int32_t HW, float* d_out, void* stream) {
  if (!d_x || !d_gate || !d_kern || !d_out || bad_dims(N, T, C, HW)) return VITTA_ERR_INVALID_ARG;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_out)) & 15u))
    return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T * C;
  TAM_DISPATCH(tam_agg_fwd_kernel, rows, HW, st, d_x, d_gate, d_kern, (int)N, (int)T, (int)C, (int)HW, d_out);
  return VITTA_OK;
}

int vitta_tam_agg_bwd_f32(const float* d_x, const float* d_gate, const float* d_kern, const float* d_gout,
                          int32_t N, int32_t T, int32_t C, int32_t HW, float* d_gx, float* d_ggate,
                          float* d_gkern, void* stream) {
  if (!d_x || !d_gate || !d_kern || !d_gout || !d_gx || !d_ggate || !d_gkern || bad_dims(N, T, C, HW))
    return VITTA_ERR_INVALID_ARG;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_gout) |
                         reinterpret_cast<uintptr_t>(d_gx)) & 15u))
    return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T * C;
  // the [N,C,T,3] dot products are staged in d_ggate's neighbour: the caller gives d_ggate room for
  // N*C*T*4 floats ([N,C,T] result followed by the [N,C,T,3] scratch)
  float* dots = d_ggate + (int64_t)N * C * T;
  TAM_DISPATCH(tam_agg_bwd_kernel, rows, HW, st, d_x, d_gate, d_kern, d_gout, (int)N, (int)T, (int)C, (int)HW,
               d_gx, dots);
  const int64_t NC = (int64_t)N * C;
  VITTA_LAUNCH(tam_finish_bwd_kernel, dim3((unsigned)((NC + VITTA_BLOCK - 1) / VITTA_BLOCK)),
                     dim3(VITTA_BLOCK), 0, st, d_gate, d_kern, dots, NC, (int)T, d_ggate, d_gkern);
  return VITTA_OK;
}

int vitta_tam_pool_bwd_f32(const float* d_gpool, int32_t N, int32_t T, int32_t C, int32_t HW,
                           float* d_gx_accum, void* stream) {
  if (!d_gpool || !d_gx_accum || bad_dims(N, T, C, HW)) return VITTA_ERR_INVALID_ARG;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(d_gx_accum) & 15u)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T * C;
  TAM_DISPATCH(tam_pool_bwd_kernel, rows, HW, st, d_gpool, (int)N, (int)T, (int)C, (int)HW, d_gx_accum);
  return VITTA_OK;
}

}  // extern "C"
