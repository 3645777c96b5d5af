// TAM global (G) and local (L) branches fused: pooled [N, C, T] -> adaptive kernel [N*C, 3] and gate [N, C, T].
//
// Reference (models/tanet_models/temporal_module.py:27-41, 53-55), every BatchNorm1d in eval():
//   G: Linear(T, 2T, no bias) -> BatchNorm1d(2T) -> ReLU -> Linear(2T, 3, no bias) -> Softmax       per (n, c) row
//   L: Conv1d(C, C/4, k3, pad 1, no bias) -> BatchNorm1d(C/4) -> ReLU -> Conv1d(C/4, C, k1) -> Sigmoid   per clip n
// As torch modules this is ~14 launches forward and ~25 backward of 3-5 us kernels on KB-sized tensors, 16 TAMs
// per pass: half of all launches of a TTA step (r1e profile).  Here: two launches forward, two backward, each
// spread over N x (C/32 .. C/4/8) workgroups with the clip's intermediates in LDS (<= 3 MFLOP per clip).
#include "common.h"

using namespace vitta;

namespace {

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

// ------------------------------------------------------------------------------------------------
// The work of one clip is spread over several workgroups (a clip is only N = B*V = 2 workgroups otherwise):
//   forward  F1 grid (N, O/OB): h_pre / h for OB conv1 output channels + the G branch of a slice of channels
//            F2 grid (N, C/CB): gate for CB channels
//   backward B1 grid (N, O/OB): d(conv1 output) for OB channels (+ dW3 slice, BN1d(L) affine grads)
//            B2 grid (N, C/CB): d pooled for CB channels (L transposed conv + G branch) (+ dW0 slice, G grads)
// ------------------------------------------------------------------------------------------------
constexpr int OB = 8;    // conv1 output channels per workgroup
constexpr int CB = 32;   // channels per workgroup in the per-channel stages
constexpr int TBW = 256; // threads of the split kernels

// copy `n` contiguous floats global -> LDS with every lane keeping four 4-byte loads in flight (the weights of a
// TAM are read once per workgroup: a load -> fma loop over them is pure latency, ~60 us per launch in the r1g profile)
__device__ __forceinline__ void stage_linear(float* __restrict__ dst, const float* __restrict__ src, int n) {
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * TBW) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TBW;
      v[u] = i < n ? src[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TBW;
      if (i < n) dst[i] = v[u];
    }
  }
}

__device__ __forceinline__ void load_pooled_t(const TamBranchArgs& a, int n, float* pl, int nthreads) {
  const int C = a.C, T = a.T, TP = T + 2;
  for (int i = threadIdx.x; i < C * TP; i += nthreads) {
    const int c = i / TP, t = i % TP - 1;
    pl[i] = (t >= 0 && t < T) ? a.pooled[((int64_t)n * C + c) * T + t] : 0.f;
  }
}

__global__ __launch_bounds__(TBW) void tam_branch_f1_kernel(TamBranchArgs a, float* __restrict__ kern,
                                                            float* __restrict__ h_pre, float* __restrict__ h_act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y, ntiles = gridDim.y;
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2;
  float* pl = smem;                 // [C][T+2]
  float* red = pl + C * TP;         // [OB*T][CS] partial sums
  float* wl = red + TBW;            // [OB][C*3] conv1 weights of this tile
  load_pooled_t(a, n, pl, TBW);
  {
    const int o0 = tile * OB, rows = min(OB, O - o0);
    stage_linear(wl, a.w0 + (int64_t)o0 * C * 3, rows * C * 3);
  }
  __syncthreads();
  // G branch: this workgroup's slice of channels
  const int cper = (C + ntiles - 1) / ntiles;
  for (int c = tile * cper + threadIdx.x; c < min(C, (tile + 1) * cper); c += TBW) {
    float u_pre[2 * T_MAX], u[2 * T_MAX], k3[3];
    g_forward(a, pl + c * TP + 1, u_pre, u, k3);
    float* o = kern + ((int64_t)n * C + c) * 3;
    o[0] = k3[0]; o[1] = k3[1]; o[2] = k3[2];
  }
  // conv1 for OB output channels: item (o_local, t), CS lanes split the C reduction
  const int items = OB * T;
  const int CS = TBW / items > 0 ? TBW / items : 1;
  const int item = threadIdx.x / CS, cs = threadIdx.x % CS;
  float acc = 0.f;
  const int o = tile * OB + item / T, t = item % T;
  if (item < items && o < O) {
    const float* w = wl + (item / T) * C * 3;
    for (int c = cs; c < C; c += CS) {
      const float* p = pl + c * TP + t;
      acc = fmaf(w[3 * c], p[0], acc);
      acc = fmaf(w[3 * c + 1], p[1], acc);
      acc = fmaf(w[3 * c + 2], p[2], acc);
    }
  }
  if (threadIdx.x < items * CS) red[threadIdx.x] = acc;
  __syncthreads();
  if (cs == 0 && item < items && o < O) {
    float pre = 0.f;
    for (int k = 0; k < CS; ++k) pre += red[item * CS + k];
    const float sc = a.bnl.w[o] * rsqrtf(a.bnl.rv[o] + a.bnl.eps);
    const int64_t idx = ((int64_t)n * O + o) * T + t;
    h_pre[idx] = pre;
    h_act[idx] = fmaxf(fmaf(pre - a.bnl.rm[o], sc, a.bnl.b[o]), 0.f);
  }
}

__global__ __launch_bounds__(TBW) void tam_branch_f2_kernel(TamBranchArgs a, const float* __restrict__ h_act,
                                                            float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int C = a.C, T = a.T, O = C / 4;
  float* hl = smem;          // [O][T]
  float* wl = hl + O * T;    // [CB][O] conv2 weights of this tile
  stage_linear(hl, h_act + (int64_t)n * O * T, O * T);
  stage_linear(wl, a.w3 + (int64_t)c0 * O, min(CB, C - c0) * O);
  __syncthreads();
  for (int i = threadIdx.x; i < CB * T; i += TBW) {
    const int c = c0 + i / T, t = i % T;
    if (c >= C) continue;
    const float* w = wl + (i / T) * O;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(w[o], hl[o * T + t], acc);
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

// B1: d(conv1 output) for OB channels
__global__ __launch_bounds__(TBW) void tam_branch_b1_kernel(TamBranchArgs a, const float* __restrict__ gate,
                                                            const float* __restrict__ h_pre,
                                                            const float* __restrict__ h_act,
                                                            const float* __restrict__ ggate, float* __restrict__ dpre_g,
                                                            TamBranchGrads g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y;
  const int C = a.C, T = a.T, O = C / 4;
  float* dz = smem;            // [C][T]
  float* red = dz + C * T;     // [OB*T][CS]
  float* wl = red + TBW;       // [C][OB] conv2 weights W3[c, o0 .. o0+OB)
  for (int i = threadIdx.x; i < C * T; i += TBW) {
    const float gt = gate[(int64_t)n * C * T + i];
    dz[i] = ggate[(int64_t)n * C * T + i] * gt * (1.f - gt);
  }
#pragma unroll 4
  for (int i = threadIdx.x; i < C * OB; i += TBW) {
    const int c = i / OB, oo = tile * OB + i % OB;
    wl[i] = oo < O ? a.w3[(int64_t)c * O + oo] : 0.f;
  }
  __syncthreads();
  const int items = OB * T;
  const int CS = TBW / items > 0 ? TBW / items : 1;
  const int item = threadIdx.x / CS, cs = threadIdx.x % CS;
  const int o = tile * OB + item / T, t = item % T;
  float acc = 0.f;
  if (item < items && o < O)
    for (int c = cs; c < C; c += CS) acc = fmaf(wl[c * OB + item / T], dz[c * T + t], acc);
  if (threadIdx.x < items * CS) red[threadIdx.x] = acc;
  __syncthreads();
  if (cs == 0 && item < items && o < O) {
    float dh = 0.f;
    for (int k = 0; k < CS; ++k) dh += red[item * CS + k];
    const int64_t idx = ((int64_t)n * O + o) * T + t;
    const float is = rsqrtf(a.bnl.rv[o] + a.bnl.eps);
    const float gy = h_act[idx] > 0.f ? dh : 0.f;
    dpre_g[idx] = gy * a.bnl.w[o] * is;
    atomicAdd(g.dbnl_w + o, gy * (h_pre[idx] - a.bnl.rm[o]) * is);
    atomicAdd(g.dbnl_b + o, gy);
  }
  if (g.dw3) {  // dW3[c, o] += sum_t dz[c,t] h[o,t] for this tile's o
    for (int i = threadIdx.x; i < C * OB; i += TBW) {
      const int c = i / OB, oo = tile * OB + i % OB;
      if (oo >= O) continue;
      float s = 0.f;
      for (int tt = 0; tt < T; ++tt) s = fmaf(dz[c * T + tt], h_act[((int64_t)n * O + oo) * T + tt], s);
      atomicAdd(g.dw3 + (int64_t)c * O + oo, s);
    }
  }
}

// B2: d pooled for CB channels (+ G branch)
__global__ __launch_bounds__(TBW) void tam_branch_b2_kernel(TamBranchArgs a, const float* __restrict__ kern,
                                                            const float* __restrict__ gkern,
                                                            const float* __restrict__ dpre_g, TamBranchGrads g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2, M = 2 * T;
  float* dpre = smem;              // [O][T+2], zero padded in t
  float* pl = dpre + O * TP;       // [CB][T+2] pooled of this tile, zero padded
  float* gp = pl + CB * TP;        // [CB][T] result staging
  float* gacc = gp + CB * T;       // [5M + M*T] block accumulators of the G-branch parameter gradients
  float* wl = gacc + 5 * M + M * T;  // [O][CB*3] conv1 weights W0[o, c0 .. c0+CB, :]
  {
    const int cw = min(CB, C - c0) * 3;  // floats per o-row of this tile (contiguous in W0)
#pragma unroll 4
    for (int i = threadIdx.x; i < O * CB * 3; i += TBW) {
      const int o = i / (CB * 3), r = i % (CB * 3);
      wl[i] = r < cw ? a.w0[((int64_t)o * C + c0) * 3 + r] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < O * TP; i += TBW) {
    const int o = i / TP, t = i % TP - 1;
    dpre[i] = (t >= 0 && t < T) ? dpre_g[((int64_t)n * O + o) * T + t] : 0.f;
  }
  for (int i = threadIdx.x; i < CB * TP; i += TBW) {
    const int c = c0 + i / TP, t = i % TP - 1;
    pl[i] = (c < C && t >= 0 && t < T) ? a.pooled[((int64_t)n * C + c) * T + t] : 0.f;
  }
  for (int i = threadIdx.x; i < 5 * M + M * T; i += TBW) gacc[i] = 0.f;
  __syncthreads();
  // L: transposed conv, one lane per (c, t)
  for (int i = threadIdx.x; i < CB * T; i += TBW) {
    const int cl = i / T, c = c0 + cl, t = i % T;
    float acc = 0.f;
    if (c < C) {
      for (int o = 0; o < O; ++o) {
        const float* w = wl + (o * CB + cl) * 3;
        const float* d = dpre + o * TP + t;
        acc = fmaf(w[0], d[2], acc);
        acc = fmaf(w[1], d[1], acc);
        acc = fmaf(w[2], d[0], acc);
      }
    }
    gp[i] = acc;
  }
  if (g.dw0) {  // dW0[o, c, j] += sum_t dpre[o,t] pooled[c, t+j-1] for this tile's c
    for (int i = threadIdx.x; i < O * CB * 3; i += TBW) {
      const int j = i % 3, cl = (i / 3) % CB, o = i / (3 * CB);
      if (c0 + cl >= C) continue;
      float s = 0.f;
      for (int t = 0; t < T; ++t) s = fmaf(dpre[o * TP + t + 1], pl[cl * TP + t + j], s);
      atomicAdd(g.dw0 + ((int64_t)o * C + c0 + cl) * 3 + j, s);
    }
  }
  __syncthreads();
  // G: one lane per channel of the tile
  if (threadIdx.x < CB && c0 + threadIdx.x < C) {
    const int cl = threadIdx.x, c = c0 + cl;
    const float* prow = pl + cl * TP + 1;
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
      const float dpg = gy * a.bng.w[m] * is;
      atomicAdd(gacc + m, gy * (u_pre[m] - a.bng.rm[m]) * is);
      atomicAdd(gacc + M + m, gy);
      if (g.dwg3) {
        atomicAdd(gacc + 2 * M + m, dv[0] * u[m]);
        atomicAdd(gacc + 3 * M + m, dv[1] * u[m]);
        atomicAdd(gacc + 4 * M + m, dv[2] * u[m]);
      }
      for (int t = 0; t < T; ++t) {
        dp[t] = fmaf(a.wg1[m * T + t], dpg, dp[t]);
        if (g.dwg1) atomicAdd(gacc + 5 * M + m * T + t, dpg * prow[t]);
      }
    }
    for (int t = 0; t < T; ++t) gp[cl * T + t] += dp[t];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CB * T; i += TBW) {
    const int c = c0 + i / T;
    if (c < C) g.gpooled[((int64_t)n * C + c) * T + i % T] = gp[i];
  }
  for (int i = threadIdx.x; i < M; i += TBW) {
    atomicAdd(g.dbng_w + i, gacc[i]);
    atomicAdd(g.dbng_b + i, gacc[M + i]);
  }
  if (g.dwg3)
    for (int i = threadIdx.x; i < 3 * M; i += TBW) atomicAdd(g.dwg3 + i, gacc[2 * M + i]);
  if (g.dwg1)
    for (int i = threadIdx.x; i < M * T; i += TBW) atomicAdd(g.dwg1 + i, gacc[5 * M + i]);
}

inline size_t f1_lds(int C, int T) { return sizeof(float) * ((size_t)C * (T + 2) + TBW + (size_t)OB * C * 3) + 64; }
inline size_t f2_lds(int C, int T) { return sizeof(float) * ((size_t)(C / 4) * T + (size_t)CB * (C / 4)) + 64; }
inline size_t b1_lds(int C, int T) { return sizeof(float) * ((size_t)C * T + TBW + (size_t)C * OB) + 64; }
inline size_t b2_lds(int C, int T) {
  return sizeof(float) * ((size_t)(C / 4) * (T + 2) + CB * (T + 2) + CB * T + 5 * 2 * T + 2 * T * T +
                          (size_t)(C / 4) * CB * 3) + 64;
}

template <typename K>
inline bool set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
         hipSuccess;
}

inline bool tb_bad(const TamBranchArgs& a) {
  return !a.pooled || !a.wg1 || !a.wg3 || !a.w0 || !a.w3 || !a.bng.w || !a.bng.b || !a.bng.rm || !a.bng.rv || !a.bnl.w ||
         !a.bnl.b || !a.bnl.rm || !a.bnl.rv || a.N <= 0 || a.C <= 0 || a.T <= 0;
}

}  // namespace

extern "C" {

int vitta_tam_branch_supported(int32_t C, int32_t T) {
  return (T >= 1 && T <= T_MAX && OB * T <= TBW && C >= 4 && C % 4 == 0 && f1_lds(C, T) <= 160 * 1024 && f2_lds(C, T) <= 160 * 1024 &&
          b1_lds(C, T) <= 160 * 1024 && b2_lds(C, T) <= 160 * 1024) ? 1 : 0;
}

int vitta_tam_branch_fwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                             float* d_hpre, void* stream) {
  // d_hpre: 2 * N * (C/4) * T floats: conv1 output before BN, then after BN + ReLU
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre) return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T};
  if (tb_bad(a)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int O = C / 4;
  float* d_hact = d_hpre + (int64_t)N * O * T;
  if (!set_lds(tam_branch_f1_kernel, f1_lds(C, T)) || !set_lds(tam_branch_f2_kernel, f2_lds(C, T))) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_f1_kernel, dim3(N, (O + OB - 1) / OB), dim3(TBW), f1_lds(C, T), st, a, d_kern, d_hpre, d_hact);
  VITTA_LAUNCH(tam_branch_f2_kernel, dim3(N, (C + CB - 1) / CB), dim3(TBW), f2_lds(C, T), st, a, d_hact, d_gate);
  return VITTA_OK;
}

int vitta_tam_branch_bwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                             const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                             float* d_gpooled, float* const* h_dbn /* {dG.w, dG.b, dL.w, dL.b} zeroed */,
                             float* const* h_dw /* {dwg1, dwg3, dw0, dw3} zeroed, or NULL entries */, void* stream) {
  // d_gpooled doubles as scratch: it must have room for N*C*T + N*(C/4)*T floats (result, then d conv1-output)
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre || !d_gkern || !d_ggate || !d_gpooled || !h_dbn)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T};
  if (tb_bad(a) || !h_dbn[0] || !h_dbn[1] || !h_dbn[2] || !h_dbn[3]) return VITTA_ERR_INVALID_ARG;
  TamBranchGrads g{d_gpooled, h_dbn[0], h_dbn[1], h_dbn[2], h_dbn[3], h_dw ? h_dw[0] : nullptr, h_dw ? h_dw[1] : nullptr,
                   h_dw ? h_dw[2] : nullptr, h_dw ? h_dw[3] : nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int O = C / 4;
  const float* d_hact = d_hpre + (int64_t)N * O * T;
  float* d_dpre = d_gpooled + (int64_t)N * C * T;
  if (!set_lds(tam_branch_b1_kernel, b1_lds(C, T)) || !set_lds(tam_branch_b2_kernel, b2_lds(C, T))) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_b1_kernel, dim3(N, (O + OB - 1) / OB), dim3(TBW), b1_lds(C, T), st, a, d_gate, d_hpre, d_hact,
               d_ggate, d_dpre, g);
  VITTA_LAUNCH(tam_branch_b2_kernel, dim3(N, (C + CB - 1) / CB), dim3(TBW), b2_lds(C, T), st, a, d_kern, d_gkern, d_dpre, g);
  return VITTA_OK;
}

}  // extern "C"
