// LayerNorm over the channel axis of channels-last activations [rows, C], fused with what surrounds it in a Video Swin
// block (models/videoswintransformer_models/swin_transformer.py:245-275):
//
//   forward   x' = x + s_b * branch         (residual + per-sample stochastic depth; optional)
//             y  = (x' - mean) * rstd * gamma + beta
//             + per-channel shifted moments of y for a hooked layer: sum (y - k_c), sum (y - k_c)^2   (ViTTA statistics:
//               "the hooked-layer reduction rides on the norm pass", like bn_act.hip does for BatchNorm)
//   backward  g  = g_y + gscale * (a_c + b_c (y - mu_c))                 (stat-loss injection, hooked layers)
//             dx = rstd * (g gamma - mean_c(g gamma) - xhat * mean_c(g gamma xhat)) + g_x'   (+ the residual path's gradient)
//             d branch = s_b * dx ; d gamma = sum g xhat ; d beta = sum g
//
// As torch ops one such site is mul + add + layer_norm forward, and add + 3 layer-norm-backward kernels + the injection
// pass + mul backward: 20 B/element forward and 40 B/element backward.  Here: 16 and 16-20.
//
// Mapping: one wave per row, 256-thread workgroups, ROWS rows per workgroup.  Lane l owns the float4 (or float2 for
// C = 128) vectors l, l + 64, ... of the row: every access is a fully coalesced 16-byte load/store, the row statistics
// are two butterfly reductions, per-channel sums stay in the lane's registers across its rows, are merged across
// the four waves through LDS and leave the workgroup as one partial row [2][C]; vitta_colsum2_f32 adds the partial
// rows (row-parallel, one atomic per column and 32 partial rows).
#include "common.h"

using namespace vitta;

namespace {

// rows per workgroup: 4 (one per wave) .. 32, chosen so that even the 3136-row stage-3 tensors spread over >= 784
// workgroups (one wave per row is latency-bound: the row count per wave, not the bytes, sets the time)
inline int ln_rows_per_wg(int64_t rows) {
  const int64_t r = (rows / 1024) & ~int64_t(3);
  return (int)(r < 4 ? 4 : (r > 32 ? 32 : r));
}

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using T = float4; };
template <>
struct VecT<2> { using T = float2; };

template <int VEC>
__device__ __forceinline__ void ldv(float* dst, const float* src) {
  if constexpr (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4*>(src);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  } else {
    const float2 v = *reinterpret_cast<const float2*>(src);
    dst[0] = v.x; dst[1] = v.y;
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* dst, const float* src) {
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
  else *reinterpret_cast<float2*>(dst) = make_float2(src[0], src[1]);
}

// element i .. i + VEC - 1 of a row held as fp32 or as bfloat16 (the 2-byte activations of the bf16 recipe: a LayerNorm output that
// only feeds a dense product, a branch that a dense product wrote, a gradient a dense product returns)
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <int VEC>
__device__ __forceinline__ void ldv_any(float* dst, const void* base, int64_t idx, bool bf16) {
  if (!bf16) {
    ldv<VEC>(dst, static_cast<const float*>(base) + idx);
  } else if constexpr (VEC == 4) {
    const ushort4 v = *reinterpret_cast<const ushort4*>(static_cast<const unsigned short*>(base) + idx);
    dst[0] = bf2f(v.x); dst[1] = bf2f(v.y); dst[2] = bf2f(v.z); dst[3] = bf2f(v.w);
  } else {
    const ushort2 v = *reinterpret_cast<const ushort2*>(static_cast<const unsigned short*>(base) + idx);
    dst[0] = bf2f(v.x); dst[1] = bf2f(v.y);
  }
}
template <int VEC>
__device__ __forceinline__ void stv_any(void* base, int64_t idx, const float* src, bool bf16) {
  if (!bf16) {
    stv<VEC>(static_cast<float*>(base) + idx, src);
  } else if constexpr (VEC == 4) {
    *reinterpret_cast<ushort4*>(static_cast<unsigned short*>(base) + idx) = make_ushort4(f2bf(src[0]), f2bf(src[1]), f2bf(src[2]), f2bf(src[3]));
  } else {
    *reinterpret_cast<ushort2*>(static_cast<unsigned short*>(base) + idx) = make_ushort2(f2bf(src[0]), f2bf(src[1]));
  }
}

struct LnFwdArgs {
  const float* x; const void* branch; const float* scale;  // branch / scale may be null
  const float* gamma; const float* beta; float eps;
  float* xnew;   // x + s*branch (written only with a branch)
  void* y; float* mean; float* rstd;
  const float* shift;  // hooked: k_c (source mean); null -> no statistics
  float* partial;      // [gridDim.x][2][C]
  int64_t rows, rows_per_sample;
  int C, rpw;
  int branch16, y16;   // branch / y are bfloat16
};

template <int NV, int VEC, bool RES, bool STATS>
__global__ __launch_bounds__(VITTA_BLOCK) void ln_fwd_kernel(LnFwdArgs a) {
  constexpr int E = NV * VEC;  // elements of a row per lane
  extern __shared__ __attribute__((aligned(16))) float lds[];  // gamma | beta | shift (| 4 x 2 x C merge area)
  const int C = a.C, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* g_l = lds; float* b_l = lds + C; float* k_l = lds + 2 * C; float* m_l = lds + 3 * C;
  for (int i = tid; i < C; i += VITTA_BLOCK) {
    g_l[i] = a.gamma[i];
    b_l[i] = a.beta[i];
    if (STATS) k_l[i] = a.shift[i];
  }
  __syncthreads();
  float s1[E], s2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * a.rpw;
  const float inv_c = 1.f / (float)C;
  for (int i = wave; i < a.rpw; i += VITTA_BLOCK / VITTA_WAVE) {
    const int64_t r = r0 + i;
    if (r >= a.rows) break;
    const float* xr = a.x + r * C;
    float v[E];
#pragma unroll
    for (int j = 0; j < NV; ++j) ldv<VEC>(v + j * VEC, xr + (lane + 64 * j) * VEC);
    if (RES) {
      const float s = a.scale ? a.scale[r / a.rows_per_sample] : 1.f;
      float* xo = a.xnew + r * C;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float b[VEC];
        ldv_any<VEC>(b, a.branch, r * C + (lane + 64 * j) * VEC, a.branch16);
#pragma unroll
        for (int u = 0; u < VEC; ++u) v[j * VEC + u] = fmaf(s, b[u], v[j * VEC + u]);
        stv<VEC>(xo + (lane + 64 * j) * VEC, v + j * VEC);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) sum += v[e];
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float d = v[e] - mean;
      sq = fmaf(d, d, sq);
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_c + a.eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float o[VEC];
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const int c = (lane + 64 * j) * VEC + u;
        o[u] = fmaf((v[j * VEC + u] - mean) * rstd, g_l[c], b_l[c]);
        if (STATS) {
          const float d = o[u] - k_l[c];
          s1[j * VEC + u] += d;
          s2[j * VEC + u] = fmaf(d, d, s2[j * VEC + u]);
        }
      }
      stv_any<VEC>(a.y, r * C + (lane + 64 * j) * VEC, o, a.y16);
    }
    if (lane == 0) {
      a.mean[r] = mean;
      a.rstd[r] = rstd;
    }
  }
  if (STATS) {
    // merge the four waves: m_l[wave][2][C]
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const int c = (lane + 64 * j) * VEC + u;
        m_l[(wave * 2) * C + c] = s1[j * VEC + u];
        m_l[(wave * 2 + 1) * C + c] = s2[j * VEC + u];
      }
    __syncthreads();
    float* out = a.partial + (int64_t)blockIdx.x * 2 * C;
    for (int i = tid; i < 2 * C; i += VITTA_BLOCK)
      out[i] = m_l[i] + m_l[2 * C + i] + m_l[4 * C + i] + m_l[6 * C + i];
  }
}

struct LnBwdArgs {
  const void* gy; const float* gxnew;  // gxnew: gradient arriving at x' from the residual path, or null
  const float* x;                       // the normalised tensor (x' of the forward)
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  const float* scale;                   // per-sample branch scale or null
  const float* mu; const float* ca; const float* cb; const float* gscale;  // injection (null: none)
  float* gx; void* gbranch;             // gbranch written only with a scale (or as the bf16 gradient of a bf16 branch)
  float* partial;                       // [gridDim.x][2][C]: d gamma | d beta
  int64_t rows, rows_per_sample;
  int C, rpw;
  int gy16, gbranch16;                  // gy / gbranch are bfloat16
};

template <int NV, int VEC, bool INJ>
__global__ __launch_bounds__(VITTA_BLOCK) void ln_bwd_kernel(LnBwdArgs a) {
  constexpr int E = NV * VEC;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // gamma | (beta | ia | ib) | merge area
  const int C = a.C, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* g_l = lds; float* b_l = lds + C; float* ia_l = lds + 2 * C; float* ib_l = lds + 3 * C; float* m_l = lds + 4 * C;
  const float gs = (INJ && a.gscale) ? *a.gscale : 1.f;
  for (int i = tid; i < C; i += VITTA_BLOCK) {
    g_l[i] = a.gamma[i];
    if (INJ) {
      b_l[i] = a.beta[i];
      ib_l[i] = gs * a.cb[i];
      ia_l[i] = gs * a.ca[i] - ib_l[i] * a.mu[i];  // a + b (y - mu) = (a - b mu) + b y
    }
  }
  __syncthreads();
  float dg[E], db[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * a.rpw;
  const float inv_c = 1.f / (float)C;
  for (int i = wave; i < a.rpw; i += VITTA_BLOCK / VITTA_WAVE) {
    const int64_t r = r0 + i;
    if (r >= a.rows) break;
    const float mean = a.mean[r], rstd = a.rstd[r];
    float xh[E], g[E];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      ldv<VEC>(xh + j * VEC, a.x + r * C + (lane + 64 * j) * VEC);
      ldv_any<VEC>(g + j * VEC, a.gy, r * C + (lane + 64 * j) * VEC, a.gy16);
    }
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const int e = j * VEC + u, c = (lane + 64 * j) * VEC + u;
        xh[e] = (xh[e] - mean) * rstd;
        if (INJ) g[e] += fmaf(ib_l[c], fmaf(xh[e], g_l[c], b_l[c]), ia_l[c]);
        dg[e] = fmaf(g[e], xh[e], dg[e]);
        db[e] += g[e];
        g[e] *= g_l[c];  // g gamma
        t1 += g[e];
        t2 = fmaf(g[e], xh[e], t2);
      }
    t1 = wave_sum(t1) * inv_c;
    t2 = wave_sum(t2) * inv_c;
    const float s = a.scale ? a.scale[r / a.rows_per_sample] : 1.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float o[VEC], ob[VEC], gr[VEC];
      if (a.gxnew) ldv<VEC>(gr, a.gxnew + r * C + (lane + 64 * j) * VEC);
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const int e = j * VEC + u;
        o[u] = rstd * (g[e] - t1 - xh[e] * t2);
        if (a.gxnew) o[u] += gr[u];
        ob[u] = s * o[u];
      }
      stv<VEC>(a.gx + r * C + (lane + 64 * j) * VEC, o);
      if (a.gbranch) stv_any<VEC>(a.gbranch, r * C + (lane + 64 * j) * VEC, ob, a.gbranch16);
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      const int c = (lane + 64 * j) * VEC + u;
      m_l[(wave * 2) * C + c] = dg[j * VEC + u];
      m_l[(wave * 2 + 1) * C + c] = db[j * VEC + u];
    }
  __syncthreads();
  float* out = a.partial + (int64_t)blockIdx.x * 2 * C;
  for (int i = tid; i < 2 * C; i += VITTA_BLOCK) out[i] = m_l[i] + m_l[2 * C + i] + m_l[4 * C + i] + m_l[6 * C + i];
}

// [out_a | out_b][c] += sum_b partial[b][c], c < 2C (out_a takes columns < C, out_b the rest).  Workgroup (x, y) sums
// 64 columns over CS_ROWS partial rows (4 row groups, LDS merge) and adds its result with ONE fp32 atomic per column:
// parallel over rows as well as columns (a 1568 x 256 partial matrix on 4 workgroups took 96 us).  The outputs must
// hold what the sum is to be added to (zero for a fresh result).  cnt (optional) receives cnt_value.
constexpr int CS_ROWS = 32;
__global__ __launch_bounds__(VITTA_BLOCK) void colsum2_kernel(const float* __restrict__ partial, int64_t nb, int C,
                                                              float* __restrict__ out_a, float* __restrict__ out_b,
                                                              float* __restrict__ cnt, float cnt_value) {
  __shared__ float red[4][64];
  const int n = 2 * C;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.y * CS_ROWS, b1 = b0 + CS_ROWS < nb ? b0 + CS_ROWS : nb;
  float s = 0.f;
  if (col < n)
    for (int64_t b = b0 + grp; b < b1; b += 4) s += partial[b * n + col];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < n) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    atomicAdd(col < C ? out_a + col : out_b + (col - C), t);
  }
  if (cnt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *cnt = cnt_value;
}

// The same sums for up to CS_BATCH partial matrices in ONE launch (the column sums of a pass's LayerNorm sites, deferred to
// the point where their results are first read: the statistics alignment after the forward, the exchange / optimizer after the
// backward -- ~100 five-microsecond launches per Video Swin step leave the dependent chain).  Workgroup w belongs to the item
// whose range [first[i], first[i + 1]) holds it; inside the item it is the (x, y) workgroup of colsum2_kernel.
constexpr int CS_BATCH = 32;
struct ColsumBatch {
  vitta_colsum_item it[CS_BATCH];
  int first[CS_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(VITTA_BLOCK) void colsum2_multi_kernel(const ColsumBatch b) {
  __shared__ float red[4][64];
  const int w = blockIdx.x;
  int i = 0;
  while (i + 1 < b.n && w >= b.first[i + 1]) ++i;
  const vitta_colsum_item& it = b.it[i];
  const int C = it.C, n = 2 * C, gx = (n + 63) / 64;
  const int local = w - b.first[i], bx = local % gx, by = local / gx;
  const int col = bx * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const int64_t nb = it.n_partials;
  const int64_t b0 = (int64_t)by * CS_ROWS, b1 = b0 + CS_ROWS < nb ? b0 + CS_ROWS : nb;
  float s = 0.f;
  if (col < n)
    for (int64_t r = b0 + grp; r < b1; r += 4) s += it.d_partial[r * n + col];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < n) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    atomicAdd(col < C ? it.d_out_a + col : it.d_out_b + (col - C), t);
  }
  if (it.d_cnt && local == 0 && threadIdx.x == 0) *it.d_cnt = it.cnt_value;
}

inline bool ln_shape(int C, int* nv, int* vec) {
  if (C == 128) { *nv = 1; *vec = 2; return true; }
  if (C % 256 == 0 && C >= 256 && C <= 2048 && ((C / 256) & (C / 256 - 1)) == 0) { *nv = C / 256; *vec = 4; return true; }
  return false;
}

inline bool mis16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

}  // namespace

#define LN_DISPATCH(NVV, VECV, ...)                \
  do {                                             \
    if (VECV == 2) { constexpr int NV = 1, VEC = 2; __VA_ARGS__; }          \
    else if (NVV == 1) { constexpr int NV = 1, VEC = 4; __VA_ARGS__; }      \
    else if (NVV == 2) { constexpr int NV = 2, VEC = 4; __VA_ARGS__; }      \
    else if (NVV == 4) { constexpr int NV = 4, VEC = 4; __VA_ARGS__; }      \
    else { constexpr int NV = 8, VEC = 4; __VA_ARGS__; }                    \
  } while (0)

// dynamic LDS beyond the 64 KB default needs the attribute (C = 2048: 90 KB)
#define LN_LAUNCH(KERNEL)                                                                                             \
  do {                                                                                                                \
    if (lds > 48 * 1024 &&                                                                                            \
        hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != \
            hipSuccess)                                                                                               \
      return VITTA_ERR_LAUNCH;                                                                                        \
    VITTA_LAUNCH(KERNEL, grid, dim3(VITTA_BLOCK), lds, st, a);                                                        \
  } while (0)

extern "C" {

int vitta_ln_supported(int32_t C) {
  int nv, vec;
  return ln_shape(C, &nv, &vec) ? 1 : 0;
}

int64_t vitta_ln_num_partials(int64_t rows) {
  if (rows <= 0) return 0;
  const int rpw = ln_rows_per_wg(rows);
  return (rows + rpw - 1) / rpw;
}

int vitta_ln_fwd_f32(const float* d_x, const float* d_branch, const float* d_scale, int64_t rows, int64_t rows_per_sample,
                     int32_t C, const float* d_gamma, const float* d_beta, float eps, float* d_xnew, float* d_y,
                     float* d_mean, float* d_rstd, const float* d_shift, float* d_partial, void* stream) {
  return vitta_ln_fwd_mixed(d_x, d_branch, d_scale, rows, rows_per_sample, C, d_gamma, d_beta, eps, d_xnew, d_y, d_mean, d_rstd,
                            d_shift, d_partial, 0, stream);
}

int vitta_ln_fwd_mixed(const float* d_x, const void* d_branch, const float* d_scale, int64_t rows, int64_t rows_per_sample,
                       int32_t C, const float* d_gamma, const float* d_beta, float eps, float* d_xnew, void* d_y,
                       float* d_mean, float* d_rstd, const float* d_shift, float* d_partial, int32_t flags, void* stream) {
  int nv, vec;
  if (!ln_shape(C, &nv, &vec)) return VITTA_ERR_UNSUPPORTED;
  if (!d_x || !d_gamma || !d_beta || !d_y || !d_mean || !d_rstd || rows <= 0 || rows_per_sample <= 0)
    return VITTA_ERR_INVALID_ARG;
  if ((d_branch != nullptr) != (d_xnew != nullptr) || (d_shift != nullptr) != (d_partial != nullptr))
    return VITTA_ERR_INVALID_ARG;
  if (mis16(d_x) || mis16(d_branch) || mis16(d_xnew) || mis16(d_y)) return VITTA_ERR_INVALID_ARG;
  const LnFwdArgs a{d_x, d_branch, d_scale, d_gamma, d_beta, eps, d_xnew, d_y, d_mean, d_rstd, d_shift, d_partial,
                    rows, rows_per_sample, C, ln_rows_per_wg(rows), (flags & VITTA_LN_BRANCH_BF16) ? 1 : 0, (flags & VITTA_LN_Y_BF16) ? 1 : 0};
  const dim3 grid((unsigned)vitta_ln_num_partials(rows));
  const size_t lds = sizeof(float) * (size_t)C * (3 + (d_shift ? 8 : 0));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool R = d_branch != nullptr, S = d_shift != nullptr;
  LN_DISPATCH(nv, vec, {
    if (R && S) LN_LAUNCH((ln_fwd_kernel<NV, VEC, true, true>));
    else if (R) LN_LAUNCH((ln_fwd_kernel<NV, VEC, true, false>));
    else if (S) LN_LAUNCH((ln_fwd_kernel<NV, VEC, false, true>));
    else LN_LAUNCH((ln_fwd_kernel<NV, VEC, false, false>));
  });
  return VITTA_OK;
}

int vitta_ln_bwd_f32(const float* d_gy, const float* d_gxnew, const float* d_x, const float* d_mean, const float* d_rstd,
                     const float* d_gamma, const float* d_beta, const float* d_scale, const float* d_mu,
                     const float* d_coef_a, const float* d_coef_b, const float* d_gscale, int64_t rows,
                     int64_t rows_per_sample, int32_t C, float* d_gx, float* d_gbranch, float* d_partial, void* stream) {
  return vitta_ln_bwd_mixed(d_gy, d_gxnew, d_x, d_mean, d_rstd, d_gamma, d_beta, d_scale, d_mu, d_coef_a, d_coef_b, d_gscale, rows,
                            rows_per_sample, C, d_gx, d_gbranch, d_partial, 0, stream);
}

int vitta_ln_bwd_mixed(const void* d_gy, const float* d_gxnew, const float* d_x, const float* d_mean, const float* d_rstd,
                       const float* d_gamma, const float* d_beta, const float* d_scale, const float* d_mu,
                       const float* d_coef_a, const float* d_coef_b, const float* d_gscale, int64_t rows,
                       int64_t rows_per_sample, int32_t C, float* d_gx, void* d_gbranch, float* d_partial, int32_t flags,
                       void* stream) {
  int nv, vec;
  if (!ln_shape(C, &nv, &vec)) return VITTA_ERR_UNSUPPORTED;
  if (!d_gy || !d_x || !d_mean || !d_rstd || !d_gamma || !d_gx || !d_partial || rows <= 0 || rows_per_sample <= 0)
    return VITTA_ERR_INVALID_ARG;
  const bool I = d_mu != nullptr;
  if (I && (!d_coef_a || !d_coef_b || !d_beta)) return VITTA_ERR_INVALID_ARG;
  if (mis16(d_gy) || mis16(d_gxnew) || mis16(d_x) || mis16(d_gx) || mis16(d_gbranch)) return VITTA_ERR_INVALID_ARG;
  const LnBwdArgs a{d_gy, d_gxnew, d_x, d_mean, d_rstd, d_gamma, d_beta, d_scale, d_mu, d_coef_a, d_coef_b, d_gscale,
                    d_gx, d_gbranch, d_partial, rows, rows_per_sample, C, ln_rows_per_wg(rows), (flags & VITTA_LN_GY_BF16) ? 1 : 0,
                    (flags & VITTA_LN_GBRANCH_BF16) ? 1 : 0};
  const dim3 grid((unsigned)vitta_ln_num_partials(rows));
  const size_t lds = sizeof(float) * (size_t)C * 12;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LN_DISPATCH(nv, vec, {
    if (I) LN_LAUNCH((ln_bwd_kernel<NV, VEC, true>));
    else LN_LAUNCH((ln_bwd_kernel<NV, VEC, false>));
  });
  return VITTA_OK;
}

int vitta_colsum2_f32(const float* d_partial, int64_t n_partials, int32_t C, float* d_out_a, float* d_out_b, float* d_cnt,
                      float cnt_value, void* stream) {
  if (!d_partial || !d_out_a || !d_out_b || n_partials <= 0 || C <= 0) return VITTA_ERR_INVALID_ARG;
  const dim3 grid((2 * C + 63) / 64, (unsigned)((n_partials + CS_ROWS - 1) / CS_ROWS));
  VITTA_LAUNCH(colsum2_kernel, grid, dim3(VITTA_BLOCK), 0, static_cast<hipStream_t>(stream), d_partial, n_partials, (int)C,
               d_out_a, d_out_b, d_cnt, cnt_value);
  return VITTA_OK;
}

int vitta_colsum2_multi_f32(const vitta_colsum_item* h_items, int32_t n_items, void* stream) {
  if (!h_items || n_items <= 0) return VITTA_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n_items; ++i) {
    const vitta_colsum_item& it = h_items[i];
    if (!it.d_partial || !it.d_out_a || !it.d_out_b || it.n_partials <= 0 || it.C <= 0) return VITTA_ERR_INVALID_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int32_t i0 = 0; i0 < n_items; i0 += CS_BATCH) {
    ColsumBatch b;
    b.n = n_items - i0 < CS_BATCH ? n_items - i0 : CS_BATCH;
    int64_t total = 0;
    for (int j = 0; j < b.n; ++j) {
      b.it[j] = h_items[i0 + j];
      b.first[j] = (int)total;
      total += (int64_t)((2 * b.it[j].C + 63) / 64) * ((b.it[j].n_partials + CS_ROWS - 1) / CS_ROWS);
      if (total >= (1ll << 31)) return VITTA_ERR_UNSUPPORTED;
    }
    for (int j = b.n; j <= CS_BATCH; ++j) b.first[j] = (int)total;
    VITTA_LAUNCH(colsum2_multi_kernel, dim3((unsigned)total), dim3(VITTA_BLOCK), 0, st, b);
  }
  return VITTA_OK;
}

}  // extern "C"
