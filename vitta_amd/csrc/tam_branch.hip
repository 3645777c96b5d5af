// TAM global (G) and local (L) branches fused: pooled [N, C, T] -> adaptive kernel [N*C, 3] and gate [N, C, T].
//
// Reference (models/tanet_models/temporal_module.py:27-41, 53-55), every BatchNorm1d in eval():
//   G: Linear(T, 2T, no bias) -> BatchNorm1d(2T) -> ReLU -> Linear(2T, 3, no bias) -> Softmax       per (n, c) row
//   L: Conv1d(C, C/4, k3, pad 1, no bias) -> BatchNorm1d(C/4) -> ReLU -> Conv1d(C/4, C, k1) -> Sigmoid   per clip n
// As torch modules this is ~14 launches forward and ~25 backward of 3-5 us kernels on KB-sized tensors, 16 TAMs
// per pass: half of all launches of a TTA step (r1e profile).  Here: one launch forward, one backward; one
// workgroup per clip keeps every intermediate in LDS.  The arithmetic is tiny (<= 3 MFLOP per clip).
#include "common.h"

using namespace vitta;

namespace {

constexpr int TB_THREADS = 1024;
constexpr int T_MAX = 16;  // clip length (n_segment) supported by the register arrays

struct BnEval {  // eval-mode BatchNorm1d parameters
  const float* w; const float* b; const float* rm; const float* rv; float eps;
};

struct TamBranchArgs {
  const float* pooled;   // [N, C, T]
  const float* wg1;      // G.0.weight [2T, T]
  BnEval bng;            // G.1
  const float* wg3;      // G.3.weight [3, 2T]
  const float* w0;       // L.0.weight [C/4, C, 3]
  BnEval bnl;            // L.1
  const float* w3;       // L.3.weight [C, C/4]  (k = 1)
  int N, C, T;
};

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// LDS carve: pl [C][T+2] (zero padded in t), hl [C/4][T] (post-ReLU), dz [C][T] / misc
struct TbCarve {
  float* pl; float* hl; float* aux; float* aux2;
};
__device__ __forceinline__ TbCarve tb_carve(float* smem, int C, int T) {
  TbCarve c;
  c.pl = smem;
  c.hl = c.pl + C * (T + 2);
  c.aux = c.hl + (C / 4) * T;
  c.aux2 = c.aux + C * T;
  return c;
}

__device__ __forceinline__ void load_pooled(const TamBranchArgs& a, int n, float* pl) {
  const int C = a.C, T = a.T, TP = T + 2;
  for (int i = threadIdx.x; i < C * TP; i += TB_THREADS) {
    const int c = i / TP, t = i % TP - 1;
    pl[i] = (t >= 0 && t < T) ? a.pooled[((int64_t)n * C + c) * T + t] : 0.f;
  }
}

// G branch for one (n, c) row: u_pre (pre-BN), kern[3]
__device__ __forceinline__ void g_forward(const TamBranchArgs& a, const float* prow /* pl + c*(T+2) + 1 */, float* u_pre,
                                          float* u, float* kern3) {
  const int T = a.T, M = 2 * a.T;
  float v[3] = {0.f, 0.f, 0.f};
  for (int m = 0; m < M; ++m) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(a.wg1[m * T + t], prow[t], acc);
    u_pre[m] = acc;
    const float s = a.bng.w[m] * rsqrtf(a.bng.rv[m] + a.bng.eps);
    const float y = fmaxf(fmaf(acc - a.bng.rm[m], s, a.bng.b[m]), 0.f);
    u[m] = y;
    v[0] = fmaf(a.wg3[m], y, v[0]);
    v[1] = fmaf(a.wg3[M + m], y, v[1]);
    v[2] = fmaf(a.wg3[2 * M + m], y, v[2]);
  }
  const float mx = fmaxf(v[0], fmaxf(v[1], v[2]));
  const float e0 = __expf(v[0] - mx), e1 = __expf(v[1] - mx), e2 = __expf(v[2] - mx);
  const float inv = 1.f / (e0 + e1 + e2);
  kern3[0] = e0 * inv; kern3[1] = e1 * inv; kern3[2] = e2 * inv;
}

// L conv1 for item (o, t): pre-BN value
__device__ __forceinline__ float l_conv1(const TamBranchArgs& a, const float* pl, int o, int t) {
  const int C = a.C, TP = a.T + 2;
  const float* w = a.w0 + (int64_t)o * C * 3;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* p = pl + c * TP + t;  // p[0..2] = pooled[c][t-1..t+1]
    acc = fmaf(w[3 * c], p[0], acc);
    acc = fmaf(w[3 * c + 1], p[1], acc);
    acc = fmaf(w[3 * c + 2], p[2], acc);
  }
  return acc;
}

__global__ __launch_bounds__(TB_THREADS) void tam_branch_fwd_kernel(TamBranchArgs a, float* __restrict__ kern,
                                                                    float* __restrict__ gate,
                                                                    float* __restrict__ h_pre /* [N, C/4, T] */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, C = a.C, T = a.T, O = C / 4, TP = T + 2;
  const TbCarve cv = tb_carve(smem, C, T);
  load_pooled(a, n, cv.pl);
  __syncthreads();
  // G: one lane per channel
  for (int c = threadIdx.x; c < C; c += TB_THREADS) {
    float u_pre[2 * T_MAX], u[2 * T_MAX], k3[3];
    g_forward(a, cv.pl + c * TP + 1, u_pre, u, k3);
    float* o = kern + ((int64_t)n * C + c) * 3;
    o[0] = k3[0]; o[1] = k3[1]; o[2] = k3[2];
  }
  // L conv1 + BN + ReLU: one lane per (o, t)
  for (int i = threadIdx.x; i < O * T; i += TB_THREADS) {
    const int o = i / T, t = i % T;
    const float pre = l_conv1(a, cv.pl, o, t);
    h_pre[((int64_t)n * O + o) * T + t] = pre;
    const float s = a.bnl.w[o] * rsqrtf(a.bnl.rv[o] + a.bnl.eps);
    cv.hl[i] = fmaxf(fmaf(pre - a.bnl.rm[o], s, a.bnl.b[o]), 0.f);
  }
  __syncthreads();
  // L conv2 (k = 1) + sigmoid: one lane per (c, t)
  for (int i = threadIdx.x; i < C * T; i += TB_THREADS) {
    const int c = i / T, t = i % T;
    const float* w = a.w3 + (int64_t)c * O;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(w[o], cv.hl[o * T + t], acc);
    gate[((int64_t)n * C + c) * T + t] = sigmoidf(acc);
  }
}

struct TamBranchGrads {
  float* gpooled;  // [N, C, T]
  // eval-BN affine gradients (always produced; zero-initialised by the caller, accumulated with atomics)
  float* dbng_w; float* dbng_b;   // [2T]
  float* dbnl_w; float* dbnl_b;   // [C/4]
  // weight gradients: NULL when the weights are frozen (update_only_bn_affine)
  float* dwg1; float* dwg3; float* dw0; float* dw3;
};

__global__ __launch_bounds__(TB_THREADS) void tam_branch_bwd_kernel(TamBranchArgs a, const float* __restrict__ kern,
                                                                    const float* __restrict__ gate,
                                                                    const float* __restrict__ h_pre,
                                                                    const float* __restrict__ gkern,
                                                                    const float* __restrict__ ggate, TamBranchGrads g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, C = a.C, T = a.T, O = C / 4, TP = T + 2, M = 2 * T;
  const TbCarve cv = tb_carve(smem, C, T);
  float* dz = cv.aux;     // [C][T]  d(pre-sigmoid)
  float* dpre = cv.aux2;  // [O][T+2] d(conv1 output), zero padded in t for the transposed conv
  float* gacc = dpre + O * TP;  // [4*M + 3*M + M*T] G-branch block accumulators: dbng_w, dbng_b | dwg3 | dwg1
  load_pooled(a, n, cv.pl);
  for (int i = threadIdx.x; i < C * T; i += TB_THREADS) {
    const float gt = gate[((int64_t)n * C) * T + i];
    dz[i] = ggate[((int64_t)n * C) * T + i] * gt * (1.f - gt);
  }
  for (int i = threadIdx.x; i < O * T; i += TB_THREADS) {
    const int o = i / T;
    const float s = a.bnl.w[o] * rsqrtf(a.bnl.rv[o] + a.bnl.eps);
    cv.hl[i] = fmaxf(fmaf(h_pre[((int64_t)n * O) * T + i] - a.bnl.rm[o], s, a.bnl.b[o]), 0.f);
  }
  for (int i = threadIdx.x; i < O * TP; i += TB_THREADS) dpre[i] = 0.f;
  for (int i = threadIdx.x; i < 2 * M + 3 * M + M * T; i += TB_THREADS) gacc[i] = 0.f;
  __syncthreads();

  // ---- L, stage 2: dh[o,t] = sum_c W3[c,o] dz[c,t]; through ReLU and eval BN ----
  for (int i = threadIdx.x; i < O * T; i += TB_THREADS) {
    const int o = i / T, t = i % T;
    float dh = 0.f;
    for (int c = 0; c < C; ++c) dh = fmaf(a.w3[(int64_t)c * O + o], dz[c * T + t], dh);
    const float hv = cv.hl[i];
    const float is = rsqrtf(a.bnl.rv[o] + a.bnl.eps);
    const float gy = hv > 0.f ? dh : 0.f;
    dpre[o * TP + t + 1] = gy * a.bnl.w[o] * is;
    const float xhat = (h_pre[((int64_t)n * O) * T + i] - a.bnl.rm[o]) * is;
    atomicAdd(g.dbnl_w + o, gy * xhat);
    atomicAdd(g.dbnl_b + o, gy);
  }
  if (g.dw3) {
    for (int i = threadIdx.x; i < C * O; i += TB_THREADS) {
      const int c = i / O, o = i % O;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(dz[c * T + t], cv.hl[o * T + t], acc);
      atomicAdd(g.dw3 + i, acc);
    }
  }
  __syncthreads();

  // ---- L, stage 1: transposed conv for d pooled, and dW0 ----
  float* gp = g.gpooled + ((int64_t)n * C) * T;
  for (int i = threadIdx.x; i < C * T; i += TB_THREADS) {
    const int c = i / T, t = i % T;
    // pooled[c][t] feeds conv output t' = t - j + 1 through tap j: d = sum_o sum_j W0[o,c,j] dpre[o][t - j + 1]
    float acc = 0.f;
    for (int o = 0; o < O; ++o) {
      const float* w = a.w0 + ((int64_t)o * C + c) * 3;
      const float* d = dpre + o * TP + t;  // d[2 - j] = dpre[o][t - j + 1] (padded index +1)
      acc = fmaf(w[0], d[2], acc);
      acc = fmaf(w[1], d[1], acc);
      acc = fmaf(w[2], d[0], acc);
    }
    gp[i] = acc;  // the G-branch contribution is added below (same lane owns the same (c, t))
  }
  if (g.dw0) {
    for (int i = threadIdx.x; i < O * C * 3; i += TB_THREADS) {
      const int j = i % 3, c = (i / 3) % C, o = i / (3 * C);
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(dpre[o * TP + t + 1], cv.pl[c * TP + t + j], acc);
      atomicAdd(g.dw0 + i, acc);
    }
  }
  __syncthreads();

  // ---- G: one lane per channel ----
  for (int c = threadIdx.x; c < C; c += TB_THREADS) {
    const float* prow = cv.pl + c * TP + 1;
    float u_pre[2 * T_MAX], u[2 * T_MAX], k3[3];
    g_forward(a, prow, u_pre, u, k3);
    const float* gk = gkern + ((int64_t)n * C + c) * 3;
    const float* ks = kern + ((int64_t)n * C + c) * 3;
    const float dot = gk[0] * ks[0] + gk[1] * ks[1] + gk[2] * ks[2];
    const float dv[3] = {ks[0] * (gk[0] - dot), ks[1] * (gk[1] - dot), ks[2] * (gk[2] - dot)};
    float dp[T_MAX];
    for (int t = 0; t < T; ++t) dp[t] = 0.f;
    for (int m = 0; m < M; ++m) {
      const float du = a.wg3[m] * dv[0] + a.wg3[M + m] * dv[1] + a.wg3[2 * M + m] * dv[2];
      const float is = rsqrtf(a.bng.rv[m] + a.bng.eps);
      const float gy = u[m] > 0.f ? du : 0.f;
      const float dpre_g = gy * a.bng.w[m] * is;
      atomicAdd(gacc + m, gy * (u_pre[m] - a.bng.rm[m]) * is);  // LDS accumulators
      atomicAdd(gacc + M + m, gy);
      if (g.dwg3) {
        atomicAdd(gacc + 2 * M + m, dv[0] * u[m]);
        atomicAdd(gacc + 3 * M + m, dv[1] * u[m]);
        atomicAdd(gacc + 4 * M + m, dv[2] * u[m]);
      }
      for (int t = 0; t < T; ++t) {
        dp[t] = fmaf(a.wg1[m * T + t], dpre_g, dp[t]);
        if (g.dwg1) atomicAdd(gacc + 5 * M + m * T + t, dpre_g * prow[t]);
      }
    }
    for (int t = 0; t < T; ++t) gp[c * T + t] += dp[t];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M; i += TB_THREADS) {
    atomicAdd(g.dbng_w + i, gacc[i]);
    atomicAdd(g.dbng_b + i, gacc[M + i]);
  }
  if (g.dwg3)
    for (int i = threadIdx.x; i < 3 * M; i += TB_THREADS) atomicAdd(g.dwg3 + i, gacc[2 * M + i]);
  if (g.dwg1)
    for (int i = threadIdx.x; i < M * T; i += TB_THREADS) atomicAdd(g.dwg1 + i, gacc[5 * M + i]);
}

inline size_t tb_lds_bytes(int C, int T, bool bwd) {
  size_t f = (size_t)C * (T + 2) + (size_t)(C / 4) * T;
  if (bwd) f += (size_t)C * T + (size_t)(C / 4) * (T + 2) + (size_t)(5 * 2 * T + 2 * T * T);
  return sizeof(float) * f + 64;
}

inline bool tb_bad(const TamBranchArgs& a) {
  return !a.pooled || !a.wg1 || !a.wg3 || !a.w0 || !a.w3 || !a.bng.w || !a.bng.b || !a.bng.rm || !a.bng.rv || !a.bnl.w ||
         !a.bnl.b || !a.bnl.rm || !a.bnl.rv || a.N <= 0 || a.C <= 0 || a.T <= 0;
}

}  // namespace

extern "C" {

int vitta_tam_branch_supported(int32_t C, int32_t T) {
  return (T >= 1 && T <= T_MAX && C >= 4 && C % 4 == 0 && tb_lds_bytes(C, T, true) <= 160 * 1024) ? 1 : 0;
}

int vitta_tam_branch_fwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                             float* d_hpre, void* stream) {
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre) return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T};
  if (tb_bad(a)) return VITTA_ERR_INVALID_ARG;
  const size_t lds = tb_lds_bytes(C, T, false);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(tam_branch_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_fwd_kernel, dim3(N), dim3(TB_THREADS), lds, static_cast<hipStream_t>(stream), a, d_kern, d_gate,
               d_hpre);
  return VITTA_OK;
}

int vitta_tam_branch_bwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                             const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                             float* d_gpooled, float* const* h_dbn /* {dG.w, dG.b, dL.w, dL.b} zeroed */,
                             float* const* h_dw /* {dwg1, dwg3, dw0, dw3} zeroed, or NULL entries */, void* stream) {
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre || !d_gkern || !d_ggate || !d_gpooled || !h_dbn)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T};
  if (tb_bad(a) || !h_dbn[0] || !h_dbn[1] || !h_dbn[2] || !h_dbn[3]) return VITTA_ERR_INVALID_ARG;
  TamBranchGrads g{d_gpooled, h_dbn[0], h_dbn[1], h_dbn[2], h_dbn[3], h_dw ? h_dw[0] : nullptr, h_dw ? h_dw[1] : nullptr,
                   h_dw ? h_dw[2] : nullptr, h_dw ? h_dw[3] : nullptr};
  const size_t lds = tb_lds_bytes(C, T, true);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(tam_branch_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_bwd_kernel, dim3(N), dim3(TB_THREADS), lds, static_cast<hipStream_t>(stream), a, d_kern, d_gate,
               d_hpre, d_gkern, d_ggate, g);
  return VITTA_OK;
}

}  // extern "C"
