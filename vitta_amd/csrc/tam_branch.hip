// TAM global (G) and local (L) branches fused: pooled [N, C, T] -> adaptive kernel [N*C, 3] and gate [N, C, T].
//
// Reference (models/tanet_models/temporal_module.py:27-41, 53-55), every BatchNorm1d in eval():
//   G: Linear(T, 2T, no bias) -> BatchNorm1d(2T) -> ReLU -> Linear(2T, 3, no bias) -> Softmax       per (n, c) row
//   L: Conv1d(C, C/4, k3, pad 1, no bias) -> BatchNorm1d(C/4) -> ReLU -> Conv1d(C/4, C, k1) -> Sigmoid   per clip n
// As torch modules this is ~14 launches forward and ~25 backward of 3-5 us kernels on KB-sized tensors, 16 TAMs
// per pass: half of all launches of a TTA step (r1e profile).  Here: two launches forward, two backward, each
// spread over N x (C/32 .. C/4/8) workgroups with the clip's intermediates in LDS (<= 3 MFLOP per clip).
#include <cstdlib>
#include <initializer_list>
#include <utility>
#include <map>
#include <mutex>
#include <utility>
#include "conv_common.h"

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
  int pooled_tc;         // 1: pooled is int64 fixed point [N][T][C] (frame-major, what a convolution's VITTA_CONV_POOL epilogue accumulates)
  vitta_conv::FastDiv d_c2, d_c2b, d_t4;  // host-made reciprocals (fast fused kernels): C / 2, CBB / 2, T / 4
};

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }  // (fmaxf(NaN, 0) is 0; torch's relu hands NaN on)

// 64-bit fixed point (32 fractional bits: integer word, fraction word) -> float, exactly (integer part + fraction); a word in the
// poisoned band |v| >= 2^61 -- a non-finite or out-of-range activation met the sum (conv_epilogue.h: pool_add) -- is NaN, which the
// branches hand on as the reference's pooling would
__device__ __forceinline__ float fixed_to_float(int hi, unsigned lo) {
  const float v = (float)hi + (float)lo * 2.3283064365386963e-10f;
  return ((unsigned)(hi + 0x20000000) >= 0x40000000u) ? __uint_as_float(0x7fc00000u) : v;
}

// ------------------------------------------------------------------------------------------------
// The work of one clip is spread over several workgroups (a clip is only N = B*V = 2 workgroups otherwise):
//   forward  F1 grid (N, O/OBF): h_pre / h for OBF conv1 output channels + the G branch of a slice of channels
//            F2 grid (N, C/CB): gate for CB channels
//   backward B1 grid (N, O/OBB): d(conv1 output) for OBB channels (+ dW3 slice, BN1d(L) affine grads)
//            B2 grid (N, C/CBB): d pooled for CBB channels (L transposed conv + G branch) (+ dW0 slice, G grads)
// These launches are 4..32 workgroups of a few microseconds: what they cost is LATENCY.  Every operand is
// therefore brought into LDS with wide loads that are all in flight together (a load -> fma loop over global
// weights measured 30-60 us per launch in the r1g profile), and nothing in the arithmetic loops touches global.
// ------------------------------------------------------------------------------------------------
constexpr int TBW = 256; // threads of the split kernels
constexpr int OBF = 2;   // conv1 output channels per workgroup, forward (F1): N x C/8 workgroups
constexpr int OBB = 4;   // conv1 output channels per workgroup, backward (B1): one 16-byte load per weight row
constexpr int CB = 32;   // channels per workgroup in F2
constexpr int CBB = 16;  // channels per workgroup in B2
constexpr int GL = 16;   // lanes that share one (n, c) row of the G branch
constexpr int GM = (2 * T_MAX + GL - 1) / GL;  // hidden units of G per lane
constexpr int SU = 8;    // 16-byte loads every lane keeps in flight while staging
static_assert(CBB * GL == TBW, "B2 maps one G row per GL lanes");
static_assert(VITTA_WAVE % GL == 0, "the rows of a wave are whole");

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// n contiguous floats global -> LDS; 16-byte loads, four per lane in flight, when both sides allow it
__device__ __forceinline__ void stage_linear(float* __restrict__ dst, const float* __restrict__ src, int n) {
  if (aligned16(src) && aligned16(dst)) {
    const int n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i0 = threadIdx.x; i0 < n4; i0 += SU * TBW) {
      float4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = s4[min(i0 + u * TBW, n4 - 1)];  // clamped: loads stay unconditional
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * TBW;
        if (i < n4) d4[i] = v[u];
      }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += TBW) dst[i] = src[i];
    return;
  }
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * TBW) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + u * TBW, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TBW;
      if (i < n) dst[i] = v[u];
    }
  }
}

// `rows` rows of `len` floats (len % 4 == 0), source rows `stride` floats apart, packed densely into LDS
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, int rows, int len,
                                           int64_t stride) {
  if (aligned16(src) && aligned16(dst) && (stride & 3) == 0 && (len & 3) == 0) {
    const int l4 = len >> 2, n4 = rows * l4;
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i0 = threadIdx.x; i0 < n4; i0 += SU * TBW) {
      float4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(i0 + u * TBW, n4 - 1);
        v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(i / l4) * stride + 4 * (i % l4));
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * TBW;
        if (i < n4) d4[i] = v[u];
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < rows * len; i += TBW) dst[i] = src[(int64_t)(i / len) * stride + i % len];
}

// pooled [C][T] of clip n -> LDS [C][T+2], zero padded in t.  T % 4 == 0 (the shipped 8 / 16 frames): the rows are read
// as 16-byte vectors, SU of them in flight per lane (the scalar loop was 20 dependent L2 round trips per lane at C = 512:
// most of F1's 9-11 us)
__device__ __forceinline__ void load_pooled_t(const TamBranchArgs& a, int n, int c0, int nc, float* pl) {
  const int C = a.C, T = a.T, TP = T + 2;
  const int ncv = min(nc, C - c0);  // valid channels of the tile
  if (a.pooled_tc) {
    // frame-major int64 fixed-point source [N][T][C] (32 fractional bits): 16-byte vectors = two channels, converted exactly
    // (integer part + fraction) and transposed into the [c][T + 2] rows.  C % 4 == 0, c0 % 2 == 0.
    const long long* srct = reinterpret_cast<const long long*>(a.pooled) + (int64_t)n * T * C + c0;
    const int c2n = (ncv + 1) >> 1, n2 = c2n * T;
    for (int i0 = threadIdx.x; i0 < n2; i0 += SU * TBW) {
      longlong2 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(i0 + u * TBW, n2 - 1);
        v[u] = *reinterpret_cast<const longlong2*>(srct + (int64_t)(i / c2n) * C + 2 * (i % c2n));
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * TBW;
        if (i < n2) {
          const int t = i / c2n, c = 2 * (i % c2n);
          const long long e[2] = {v[u].x, v[u].y};
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (c + j < ncv)
              pl[(c + j) * TP + 1 + t] = fixed_to_float((int)(e[j] >> 32), (unsigned)(e[j] & 0xffffffffll));
        }
      }
    }
    for (int c = threadIdx.x; c < nc; c += TBW) {  // the two pad columns; whole rows of channels past C
      pl[c * TP] = 0.f;
      pl[c * TP + T + 1] = 0.f;
      if (c >= ncv)
        for (int t = 0; t < T; ++t) pl[c * TP + 1 + t] = 0.f;
    }
    return;
  }
  const float* src = a.pooled + ((int64_t)n * C + c0) * T;
  if ((T & 3) == 0 && aligned16(src)) {
    const int t4 = T >> 2, n4 = ncv * t4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int i0 = threadIdx.x; i0 < n4; i0 += SU * TBW) {
      float4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = s4[min(i0 + u * TBW, n4 - 1)];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = i0 + u * TBW;
        if (i < n4) {
          float* d = pl + (i / t4) * TP + 1 + 4 * (i % t4);
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    }
    for (int c = threadIdx.x; c < nc; c += TBW) {  // the two pad columns; whole rows of channels past C
      pl[c * TP] = 0.f;
      pl[c * TP + T + 1] = 0.f;
      if (c >= ncv)
        for (int t = 0; t < T; ++t) pl[c * TP + 1 + t] = 0.f;
    }
    return;
  }
  for (int i = threadIdx.x; i < nc * TP; i += TBW) {
    const int c = c0 + i / TP, t = i % TP - 1;
    pl[i] = (c < C && t >= 0 && t < T) ? a.pooled[((int64_t)n * C + c) * T + t] : 0.f;
  }
}

// G-branch parameters in LDS: W1 [2T][T] | W3 [3][2T] | running mean | scale = gamma*rsqrt(var+eps) | beta |
// rsqrt(var+eps) | gamma
struct GLds {
  const float* wg1; const float* wg3; const float* rm; const float* s; const float* b; const float* is; const float* w;
};
__device__ __forceinline__ GLds stage_g(const TamBranchArgs& a, float* gl) {
  const int T = a.T, M = 2 * T;
  float* wg1 = gl; float* wg3 = wg1 + M * T; float* rm = wg3 + 3 * M; float* sc = rm + M; float* b = sc + M;
  float* is = b + M; float* w = is + M;
  for (int i = threadIdx.x; i < M * T; i += TBW) wg1[i] = a.wg1[i];
  for (int i = threadIdx.x; i < 3 * M; i += TBW) wg3[i] = a.wg3[i];
  for (int m = threadIdx.x; m < M; m += TBW) {
    const float r = rsqrtf(a.bng.rv[m] + a.bng.eps), g = a.bng.w[m];
    rm[m] = a.bng.rm[m]; sc[m] = g * r; b[m] = a.bng.b[m]; is[m] = r; w[m] = g;
  }
  return GLds{wg1, wg3, rm, sc, b, is, w};
}

__device__ __forceinline__ float group_sum(float v) {  // over the GL lanes of one row
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) v += __shfl_xor(v, m, VITTA_WAVE);
  return v;
}
__device__ __forceinline__ float rows_sum(float v) {  // over the rows of a wave, same sub-lane
#pragma unroll
  for (int m = GL; m < VITTA_WAVE; m <<= 1) v += __shfl_xor(v, m, VITTA_WAVE);
  return v;
}

// G branch of one (n, c) row on GL lanes: lane `sub` owns hidden units m = sub, sub+GL, ...  Every lane of the
// wave must call this (shuffles); `active` masks rows past the end.
__device__ __forceinline__ void g_forward(const GLds& p, int T, int sub, bool active, const float* prow, float* u_pre,
                                          float* u, float* k3) {
  const int M = 2 * T;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
  for (int j = 0; j < GM; ++j) {
    const int m = sub + j * GL;
    float acc = 0.f, y = 0.f;
    if (active && m < M) {
      for (int t = 0; t < T; ++t) acc = fmaf(p.wg1[m * T + t], prow[t], acc);
      y = relu_keep_nan(fmaf(acc - p.rm[m], p.s[m], p.b[m]));
      v0 = fmaf(p.wg3[m], y, v0);
      v1 = fmaf(p.wg3[M + m], y, v1);
      v2 = fmaf(p.wg3[2 * M + m], y, v2);
    }
    u_pre[j] = acc;
    u[j] = y;
  }
  v0 = group_sum(v0); v1 = group_sum(v1); v2 = group_sum(v2);
  const float mx = fmaxf(v0, fmaxf(v1, v2));
  const float e0 = __expf(v0 - mx), e1 = __expf(v1 - mx), e2 = __expf(v2 - mx);
  const float inv = 1.f / (e0 + e1 + e2);
  k3[0] = e0 * inv; k3[1] = e1 * inv; k3[2] = e2 * inv;
}

// write-through store / coherent load of a tensor another workgroup of the SAME launch consumes (fused kernels below)
__device__ __forceinline__ void store_wt(float* base, int64_t idx, float v) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)(idx * 4), 0, 16);
}
__device__ __forceinline__ float load_wt(const float* base, int64_t idx) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(idx * 4), 0, 16));
}

// All workgroups of clip n meet (fused kernels): arrive on cnt[0], leave on cnt[1]; the last one to leave zeroes both, so
// the pair is at rest (zero) when the launch ends.  Data handed across the meeting point travels with write-through stores
// and coherent loads (store_wt / load_wt): no agent-scope release, which would write back every dirty line of the XCD's
// L2 (the convolution outputs of the step).  Every workgroup of the launch is resident (<= 2 x 128 workgroups).
__device__ __forceinline__ void clip_meet(unsigned* cnt, unsigned nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < (1 << 21) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg; ++spin)  // (bounded, see meet())
      __builtin_amdgcn_s_sleep(2);
    const unsigned left = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == nwg - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

struct BnItem {  // eval-BatchNorm parameters of one conv1 output channel (+ its saved activations in the backward)
  float w = 0.f, rv = 1.f, rm = 0.f, b = 0.f, hact = 0.f, hpre = 0.f;
};

// conv1 of the L branch for OBF output channels + the G branch of a slice of channels, operands in LDS.  PRE: the item's
// BatchNorm parameters were requested by the caller together with everything else the workgroup stages
template <bool WT, bool PRE>
__device__ __forceinline__ void f1_compute(const TamBranchArgs& a, float* __restrict__ kern, float* __restrict__ h_pre,
                                           float* __restrict__ h_act, int n, int tile, int ntiles, const float* wl, const float* pl,
                                           float* red, const GLds& gp, BnItem pre) {
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2;
  // G branch: this workgroup's slice of channels, GL lanes per channel
  const int cper = (C + ntiles - 1) / ntiles, cend = min(C, (tile + 1) * cper);
  const int sub = threadIdx.x % GL;
  for (int g0 = tile * cper; g0 < cend; g0 += TBW / GL) {
    const int c = g0 + threadIdx.x / GL;
    const bool active = c < cend;
    float u_pre[GM], u[GM], k3[3];
    g_forward(gp, T, sub, active, pl + (active ? c : 0) * TP + 1, u_pre, u, k3);
    if (active && sub < 3) kern[((int64_t)n * C + c) * 3 + sub] = sub == 0 ? k3[0] : (sub == 1 ? k3[1] : k3[2]);
  }
  // conv1 for OBF output channels: item (o_local, t), CS lanes split the C reduction
  const int items = OBF * T;
  const int CS = TBW / items > 0 ? TBW / items : 1;
  const int item = threadIdx.x / CS, cs = threadIdx.x % CS;
  float acc = 0.f;
  const int o = tile * OBF + item / T, t = item % T;
  // eval-BN parameters of this item's channel: issued now, consumed after the reduction (a dependent chain of global
  // loads at the very end of a 7 us kernel is a fifth of its run time)
  float bw = pre.w, brv = pre.rv, brm = pre.rm, bb = pre.b;
  if (!PRE && cs == 0 && item < items && o < O) { bw = a.bnl.w[o]; brv = a.bnl.rv[o]; brm = a.bnl.rm[o]; bb = a.bnl.b[o]; }
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
    float pre_ = 0.f;
    for (int k = 0; k < CS; ++k) pre_ += red[item * CS + k];
    const float sc = bw * rsqrtf(brv + a.bnl.eps);
    const int64_t idx = ((int64_t)n * O + o) * T + t;
    h_pre[idx] = pre_;
    const float hv = relu_keep_nan(fmaf(pre_ - brm, sc, bb));
    if (WT) store_wt(h_act, idx, hv);
    else h_act[idx] = hv;
  }
}

template <bool WT>
__device__ __forceinline__ void f1_body(const TamBranchArgs& a, float* __restrict__ kern, float* __restrict__ h_pre,
                                        float* __restrict__ h_act, int n, int tile, int ntiles, float* smem) {
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2;
  float* wl = smem;                      // [OBF][C*3] conv1 weights of this tile
  float* pl = wl + OBF * C * 3;           // [C][T+2]
  float* red = pl + C * TP;              // [OBF*T][CS] partial sums
  float* gl = red + TBW;                 // G parameters
  {
    const int o0 = tile * OBF, rows = min(OBF, O - o0);
    stage_linear(wl, a.w0 + (int64_t)o0 * C * 3, rows * C * 3);
  }
  load_pooled_t(a, n, 0, C, pl);
  const GLds gp = stage_g(a, gl);
  __syncthreads();
  f1_compute<WT, false>(a, kern, h_pre, h_act, n, tile, ntiles, wl, pl, red, gp, BnItem{});
}

__global__ __launch_bounds__(TBW) void tam_branch_f1_kernel(TamBranchArgs a, float* __restrict__ kern,
                                                            float* __restrict__ h_pre, float* __restrict__ h_act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f1_body<false>(a, kern, h_pre, h_act, blockIdx.x, blockIdx.y, gridDim.y, smem);
}

// gate for CB channels of clip n: operands in LDS (wl [CB][O] conv2 weights of the tile, hl [O][T])
__device__ __forceinline__ void f2_compute(const TamBranchArgs& a, float* __restrict__ gate, int n, int c0, const float* wl, const float* hl) {
  const int C = a.C, T = a.T, O = C / 4;
  for (int i = threadIdx.x; i < CB * T; i += TBW) {
    const int c = c0 + i / T, t = i % T;
    if (c >= C) continue;
    const float* w = wl + (i / T) * O;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(w[o], hl[o * T + t], acc);
    gate[((int64_t)n * C + c) * T + t] = sigmoidf(acc);
  }
}

// PHASE 0: whole body; 1: everything that does not need h_act (weight staging); 2: the rest
template <int PHASE>
__device__ __forceinline__ void f2_body(const TamBranchArgs& a, const float* __restrict__ h_act, float* __restrict__ gate,
                                        int n, int c0, float* smem) {
  const int C = a.C, T = a.T, O = C / 4;
  float* wl = smem;            // [CB][O] conv2 weights of this tile
  float* hl = wl + CB * O;     // [O][T]
  if (PHASE != 2) stage_linear(wl, a.w3 + (int64_t)c0 * O, min(CB, C - c0) * O);
  if (PHASE == 1) return;
  if (PHASE == 2) {
    for (int i = threadIdx.x; i < O * T; i += TBW) hl[i] = load_wt(h_act, (int64_t)n * O * T + i);
  } else {
    stage_linear(hl, h_act + (int64_t)n * O * T, O * T);
  }
  __syncthreads();
  f2_compute(a, gate, n, c0, wl, hl);
}

__global__ __launch_bounds__(TBW) void tam_branch_f2_kernel(TamBranchArgs a, const float* __restrict__ h_act,
                                                            float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f2_body<0>(a, h_act, gate, blockIdx.x, blockIdx.y * CB, smem);
}

// F1 and F2 in ONE launch: grid (N, max(tiles of F1, tiles of F2)); F2's weights are staged while the clip's F1 tiles finish
__global__ __launch_bounds__(TBW) void tam_branch_fwd_fused_kernel(TamBranchArgs a, float* __restrict__ kern,
                                                                   float* __restrict__ h_pre, float* __restrict__ h_act,
                                                                   float* __restrict__ gate, unsigned* sync, int nt1, int nt2,
                                                                   int lds1_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y;
  if (tile < nt1) f1_body<true>(a, kern, h_pre, h_act, n, tile, nt1, smem);
  if (tile < nt2) f2_body<1>(a, h_act, gate, n, tile * CB, smem + lds1_floats);
  clip_meet(sync + 2 * n, gridDim.y);
  if (tile < nt2) f2_body<2>(a, h_act, gate, n, tile * CB, smem + lds1_floats);
}

struct TamBranchGrads {
  float* gpooled;  // [N, C, T]
  // eval-BN affine gradients: ACCUMULATED with atomics (the caller hands zeroed buffers or live .grad views)
  float* dbng_w; float* dbng_b;   // [2T]
  float* dbnl_w; float* dbnl_b;   // [C/4]
  // weight gradients, accumulated likewise: NULL when the weights are frozen (update_only_bn_affine)
  float* dwg1; float* dwg3; float* dw0; float* dw3;
};

// B1: d(conv1 output) for OBB channels.  Operands in LDS: dz [C][T] holds the gate, gg [C][T] its upstream gradient, wl [C][OBB]
// the conv2 weights W3[c, o0 .. o0 + OBB).  PRE: the item's BatchNorm parameters / saved activations came with the staging
template <bool WT, bool PRE>
__device__ __forceinline__ void b1_compute(const TamBranchArgs& a, const float* __restrict__ h_pre, const float* __restrict__ h_act,
                                           float* __restrict__ dpre_g, const TamBranchGrads& g, int n, int tile, float* dz,
                                           const float* gg, const float* wl, float* red, BnItem pre) {
  const int C = a.C, T = a.T, O = C / 4;
  for (int i = threadIdx.x; i < C * T; i += TBW) {
    const float gt = dz[i];
    dz[i] = gg[i] * gt * (1.f - gt);
  }
  __syncthreads();
  const int items = OBB * T;
  const int CS = TBW / items > 0 ? TBW / items : 1;
  const int item = threadIdx.x / CS, cs = threadIdx.x % CS;
  const int o = tile * OBB + item / T, t = item % T;
  // operands of the item's epilogue, issued before the reduction (see F1)
  float bw = pre.w, brv = pre.rv, brm = pre.rm, hact = pre.hact, hpre = pre.hpre;
  if (!PRE && cs == 0 && item < items && o < O) {
    const int64_t idx0 = ((int64_t)n * O + o) * T + t;
    bw = a.bnl.w[o]; brv = a.bnl.rv[o]; brm = a.bnl.rm[o];
    hact = h_act[idx0]; hpre = h_pre[idx0];
  }
  float acc = 0.f;
  if (item < items && o < O)
    for (int c = cs; c < C; c += CS) acc = fmaf(wl[c * OBB + item / T], dz[c * T + t], acc);
  if (threadIdx.x < items * CS) red[threadIdx.x] = acc;
  __syncthreads();
  if (cs == 0 && item < items && o < O) {
    float dh = 0.f;
    for (int k = 0; k < CS; ++k) dh += red[item * CS + k];
    const int64_t idx = ((int64_t)n * O + o) * T + t;
    const float is = rsqrtf(brv + a.bnl.eps);
    const float gy = hact > 0.f ? dh : 0.f;
    if (WT) store_wt(dpre_g, idx, gy * bw * is);
    else dpre_g[idx] = gy * bw * is;
    atomicAdd(g.dbnl_w + o, gy * (hpre - brm) * is);
    atomicAdd(g.dbnl_b + o, gy);
  }
  if (g.dw3) {  // dW3[c, o] += sum_t dz[c,t] h[o,t] for this tile's o
    for (int i = threadIdx.x; i < C * OBB; i += TBW) {
      const int c = i / OBB, oo = tile * OBB + i % OBB;
      if (oo >= O) continue;
      float s = 0.f;
      for (int tt = 0; tt < T; ++tt) s = fmaf(dz[c * T + tt], h_act[((int64_t)n * O + oo) * T + tt], s);
      atomicAdd(g.dw3 + (int64_t)c * O + oo, s);
    }
  }
}

template <bool WT>
__device__ __forceinline__ void b1_body(const TamBranchArgs& a, const float* __restrict__ gate, const float* __restrict__ h_pre,
                                        const float* __restrict__ h_act, const float* __restrict__ ggate,
                                        float* __restrict__ dpre_g, const TamBranchGrads& g, int n, int tile, float* smem) {
  const int C = a.C, T = a.T, O = C / 4;
  float* dz = smem;            // [C][T]   (staged as gate, then overwritten by d(pre-sigmoid))
  float* gg = dz + C * T;      // [C][T]   upstream gradient of the gate
  float* wl = gg + C * T;      // [C][OBB] conv2 weights W3[c, o0 .. o0+OBB)
  float* red = wl + C * OBB;    // [OBB*T][CS]
  stage_linear(dz, gate + (int64_t)n * C * T, C * T);
  stage_linear(gg, ggate + (int64_t)n * C * T, C * T);
  if (tile * OBB + OBB <= O) {
    stage_rows(wl, a.w3 + tile * OBB, C, OBB, O);
  } else {
    for (int i = threadIdx.x; i < C * OBB; i += TBW) {
      const int c = i / OBB, oo = tile * OBB + i % OBB;
      wl[i] = oo < O ? a.w3[(int64_t)c * O + oo] : 0.f;
    }
  }
  __syncthreads();
  b1_compute<WT, false>(a, h_pre, h_act, dpre_g, g, n, tile, dz, gg, wl, red, BnItem{});
}

__global__ __launch_bounds__(TBW) void tam_branch_b1_kernel(TamBranchArgs a, const float* __restrict__ gate,
                                                            const float* __restrict__ h_pre,
                                                            const float* __restrict__ h_act,
                                                            const float* __restrict__ ggate, float* __restrict__ dpre_g,
                                                            TamBranchGrads g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  b1_body<false>(a, gate, h_pre, h_act, ggate, dpre_g, g, blockIdx.x, blockIdx.y, smem);
}

// ---- B2 in three pieces (operands in LDS) ----------------------------------------------------------------------------
// L: transposed conv, item (c, t); the lanes of a pair split the o reduction (even / odd o).  wl [O][CBB*3], dpre [O][T+2]
__device__ __forceinline__ void b2_l_compute(const TamBranchArgs& a, const TamBranchGrads& g, int c0, const float* wl, const float* dpre,
                                             const float* pl, float* gp) {
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2;
  for (int i0 = 0; i0 < CBB * T; i0 += TBW / 2) {
    const int i = i0 + threadIdx.x / 2, half = threadIdx.x & 1;
    const int cl = i / T, c = c0 + cl, t = i % T;
    float acc = 0.f;
    if (i < CBB * T && c < C) {
      for (int o = half; o < O; o += 2) {
        const float* w = wl + (o * CBB + cl) * 3;
        const float* d = dpre + o * TP + t;
        acc = fmaf(w[0], d[2], acc);
        acc = fmaf(w[1], d[1], acc);
        acc = fmaf(w[2], d[0], acc);
      }
    }
    acc += __shfl_xor(acc, 1, VITTA_WAVE);
    if (half == 0 && i < CBB * T) gp[i] += acc;
  }
  if (g.dw0) {  // dW0[o, c, j] += sum_t dpre[o,t] pooled[c, t+j-1] for this tile's c
    for (int i = threadIdx.x; i < O * CBB * 3; i += TBW) {
      const int j = i % 3, cl = (i / 3) % CBB, o = i / (3 * CBB);
      if (c0 + cl >= C) continue;
      float s = 0.f;
      for (int t = 0; t < T; ++t) s = fmaf(dpre[o * TP + t + 1], pl[cl * TP + t + j], s);
      atomicAdd(g.dw0 + ((int64_t)o * C + c0 + cl) * 3 + j, s);
    }
  }
}

// G: GL lanes per channel of the tile (all 256 lanes busy), hidden unit m on lane m % GL.  PRE: the row's upstream gradient of
// the adaptive kernel (three floats) came with the staging
template <bool PRE>
__device__ __forceinline__ void b2_g_compute(const TamBranchArgs& a, const TamBranchGrads& g, const GLds& p, const float* __restrict__ gkern,
                                             int n, int c0, const float* pl, float* gp, float* gacc, float gk0, float gk1, float gk2) {
  const int C = a.C, T = a.T, TP = T + 2, M = 2 * T;
  const int cl = threadIdx.x / GL, sub = threadIdx.x % GL, c = c0 + cl;
  const bool active = c < C;
  const float* prow = pl + cl * TP + 1;
  float u_pre[GM], u[GM], k3[3];
  g_forward(p, T, sub, active, prow, u_pre, u, k3);
  float dv0 = 0.f, dv1 = 0.f, dv2 = 0.f;
  if (active) {
    if (!PRE) {
      const float* gk = gkern + ((int64_t)n * C + c) * 3;
      gk0 = gk[0]; gk1 = gk[1]; gk2 = gk[2];
    }
    const float dot = gk0 * k3[0] + gk1 * k3[1] + gk2 * k3[2];
    dv0 = k3[0] * (gk0 - dot); dv1 = k3[1] * (gk1 - dot); dv2 = k3[2] * (gk2 - dot);
  }
  float dp[T_MAX];
#pragma unroll
  for (int t = 0; t < T_MAX; ++t) dp[t] = 0.f;
  const bool lead = (threadIdx.x & (VITTA_WAVE - 1)) < GL;  // the lanes of a wave that publish its row sums
#pragma unroll
  for (int j = 0; j < GM; ++j) {
    const int m = sub + j * GL;   // uniform across the rows of a wave for a given sub-lane
    const bool on = active && m < M;
    const int ms = m < M ? m : 0;
    const float du = p.wg3[ms] * dv0 + p.wg3[M + ms] * dv1 + p.wg3[2 * M + ms] * dv2;
    const float gy = (on && u[j] > 0.f) ? du : 0.f;
    const float dpg = gy * p.w[ms] * p.is[ms];
    const float gw = rows_sum(gy * (u_pre[j] - p.rm[ms]) * p.is[ms]);
    const float gb = rows_sum(gy);
    if (lead && m < M) {
      atomicAdd(gacc + m, gw);
      atomicAdd(gacc + M + m, gb);
    }
    if (g.dwg3) {
      const float uj = on ? u[j] : 0.f;
      const float a0 = rows_sum(dv0 * uj), a1 = rows_sum(dv1 * uj), a2 = rows_sum(dv2 * uj);
      if (lead && m < M) {
        atomicAdd(gacc + 2 * M + m, a0);
        atomicAdd(gacc + 3 * M + m, a1);
        atomicAdd(gacc + 4 * M + m, a2);
      }
    }
#pragma unroll
    for (int t = 0; t < T_MAX; ++t) {
      if (t < T) {
        dp[t] = fmaf(p.wg1[ms * T + t], dpg, dp[t]);
        if (g.dwg1) {
          const float w1 = rows_sum(dpg * prow[t]);
          if (lead && m < M) atomicAdd(gacc + 5 * M + m * T + t, w1);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T_MAX; ++t) {
    if (t < T) {
      const float sdp = group_sum(dp[t]);
      if (active && sub == (t % GL)) gp[cl * T + t] += sdp;
    }
  }
}

__device__ __forceinline__ void b2_finish(const TamBranchArgs& a, const TamBranchGrads& g, int n, int c0, const float* gp, const float* gacc) {
  const int C = a.C, T = a.T, M = 2 * T;
  for (int i = threadIdx.x; i < CBB * T; i += TBW) {
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

// B2: d pooled for CBB channels (+ G branch).  PHASE 0: whole body; 1: everything that does not need d(conv1 output)
// (staging, the G branch); 2: the L branch and the results
template <int PHASE>
__device__ __forceinline__ void b2_body(const TamBranchArgs& a, const float* __restrict__ kern, const float* __restrict__ gkern,
                                        const float* __restrict__ dpre_g, const TamBranchGrads& g, int n, int c0, float* smem) {
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2, M = 2 * T;
  float* wl = smem;                  // [O][CBB*3] conv1 weights W0[o, c0 .. c0+CBB, :]
  float* dpre = wl + O * CBB * 3;     // [O][T+2], zero padded in t
  float* pl = dpre + O * TP;         // [CBB][T+2] pooled of this tile, zero padded
  float* gp = pl + CBB * TP;          // [CBB][T] result staging
  float* gacc = gp + CBB * T;         // [5M + M*T] block accumulators of the G-branch parameter gradients
  float* gl = gacc + 5 * M + M * T;  // G parameters
  GLds p;
  if (PHASE != 2) {
  if (c0 + CBB <= C) {
    stage_rows(wl, a.w0 + (int64_t)c0 * 3, O, CBB * 3, (int64_t)C * 3);
  } else {
    const int cw = (C - c0) * 3;  // floats per o-row of this tile (contiguous in W0)
    for (int i = threadIdx.x; i < O * CBB * 3; i += TBW) {
      const int o = i / (CBB * 3), r = i % (CBB * 3);
      wl[i] = r < cw ? a.w0[((int64_t)o * C + c0) * 3 + r] : 0.f;
    }
  }
  load_pooled_t(a, n, c0, CBB, pl);
  for (int i = threadIdx.x; i < 5 * M + M * T; i += TBW) gacc[i] = 0.f;
  for (int i = threadIdx.x; i < CBB * T; i += TBW) gp[i] = 0.f;
  p = stage_g(a, gl);
  }
  if (PHASE == 0) {  // d(conv1 output) [O][T] of the clip -> [O][T+2]: the same padded staging as the pooled rows
    TamBranchArgs tmp = a;
    tmp.pooled = dpre_g;
    tmp.pooled_tc = 0;
    tmp.C = O;
    load_pooled_t(tmp, n, 0, O, dpre);
  }
  if (PHASE == 2) {  // written by the B1 tiles of this launch: coherent loads
    for (int i = threadIdx.x; i < O * TP; i += TBW) {
      const int t = i % TP - 1;
      dpre[i] = (t >= 0 && t < T) ? load_wt(dpre_g, ((int64_t)n * O + i / TP) * T + t) : 0.f;
    }
  }
  __syncthreads();
  if (PHASE != 1) b2_l_compute(a, g, c0, wl, dpre, pl, gp);
  if (PHASE == 0) __syncthreads();
  if (PHASE != 2) b2_g_compute<false>(a, g, p, gkern, n, c0, pl, gp, gacc, 0.f, 0.f, 0.f);
  __syncthreads();
  if (PHASE == 1) return;
  b2_finish(a, g, n, c0, gp, gacc);
}

__global__ __launch_bounds__(TBW) void tam_branch_b2_kernel(TamBranchArgs a, const float* __restrict__ kern,
                                                            const float* __restrict__ gkern,
                                                            const float* __restrict__ dpre_g, TamBranchGrads g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  b2_body<0>(a, kern, gkern, dpre_g, g, blockIdx.x, blockIdx.y * CBB, smem);
}

// B1 and B2 in ONE launch: grid (N, max(tiles)); B2's staging and its whole G branch run while the clip's B1 tiles finish
__global__ __launch_bounds__(TBW) void tam_branch_bwd_fused_kernel(TamBranchArgs a, const float* __restrict__ kern,
                                                                   const float* __restrict__ gate,
                                                                   const float* __restrict__ h_pre,
                                                                   const float* __restrict__ h_act,
                                                                   const float* __restrict__ gkern,
                                                                   const float* __restrict__ ggate, float* __restrict__ dpre_g,
                                                                   TamBranchGrads g, unsigned* sync, int nt1, int nt2,
                                                                   int lds1_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y;
  if (tile < nt1) b1_body<true>(a, gate, h_pre, h_act, ggate, dpre_g, g, n, tile, smem);
  if (tile < nt2) b2_body<1>(a, kern, gkern, dpre_g, g, n, tile * CBB, smem + lds1_floats);
  clip_meet(sync + 2 * n, gridDim.y);
  if (tile < nt2) b2_body<2>(a, kern, gkern, dpre_g, g, n, tile * CBB, smem + lds1_floats);
}

// ====================================================================================================================
// Round 5: the fused launches with every operand of a workgroup requested in ONE batch.
// The launches above are 2 .. 128 workgroups of microseconds on the critical chain of a step (16 forward + 16 backward per
// adaptation pass: 0.63 ms of a 5.3 ms chain, profiles/r5a_timeline.csv); what they cost is the NUMBER OF DEPENDENT GLOBAL ROUND
// TRIPS, ~1 us each under load.  The bodies above stage region after region (each loop waits for its loads before it stores to
// LDS: 5-6 round trips before F1 computes, 8 before the backward's meeting point, one more for the conv2 weights in front of the
// meeting point, 4-5 for the scalar re-reads behind it) and the meeting costs three (store drain, arrive / poll, leave).  Here:
// issue() puts every load of the workgroup in flight, commit() stores them; the hand-over behind the meeting point is one batch
// of 16-byte coherent loads; the leave ticket is consumed at the end of the kernel instead of in front of phase 2.  Same LDS
// layouts and the SAME compute functions as the two-launch kernels (bit-identical results: test_tam_branch_single_launch_forms...).
// Shapes: C % 64 == 0, C <= 512, T % 4 == 0, C * T <= 4096 (fast_ok); anything else takes the kernels above.
// ====================================================================================================================
constexpr int AUX_SC1 = 16;

// (The per-piece code is expanded by FOLD EXPRESSIONS over an index pack, not by `#pragma unroll` loops: with a loop the register
// array is indexed dynamically until the unroller has run, and -- measured on this compiler -- it then stays in scratch memory
// whenever anything with side effects sits between issue() and commit(): 272 bytes per lane of scratch stores and reloads.)
template <int K, int UPR>  // UPR: 16-byte pieces per source row (0: one contiguous run)
struct Stage16 {
  float4 v[K];
  static __device__ __forceinline__ void rc(int u, int& row, int& col) {
    if (UPR == 0) { row = 0; col = u; }
    else { row = u / UPR; col = u - row * UPR; }
  }
  static __device__ __forceinline__ float4 ld(const float* __restrict__ src, int units, int64_t sstride, int k) {
    const int u = min((int)threadIdx.x + k * TBW, units - 1);
    int row, col;
    rc(u, row, col);
    return *reinterpret_cast<const float4*>(src + row * sstride + 4 * col);
  }
  // (every store is UNCONDITIONAL -- a piece past the region's end goes to the 64 spare bytes `trash` at the end of the
  // workgroup's LDS: behind a condition the compiler sinks the load into the conditional block, next to its only use, and the
  // batch becomes a chain of load / wait / store again)
  static __device__ __forceinline__ void st(float* __restrict__ dst, int units, int dstride, float* trash, int k, float4 val) {
    const int u = threadIdx.x + k * TBW;
    int row, col;
    rc(u, row, col);
    float* d = u < units ? dst + row * dstride + 4 * col : trash;
    *reinterpret_cast<float4*>(d) = val;
  }
  template <int... I>
  __device__ __forceinline__ void issue_(const float* __restrict__ src, int units, int64_t sstride, std::integer_sequence<int, I...>) {
    ((v[I] = ld(src, units, sstride, I)), ...);
  }
  template <int... I>
  __device__ __forceinline__ void commit_(float* __restrict__ dst, int units, int dstride, float* trash, std::integer_sequence<int, I...>) const {
    (st(dst, units, dstride, trash, I, v[I]), ...);
  }
  __device__ __forceinline__ void issue(const float* __restrict__ src, int units, int64_t sstride) {
    issue_(src, units, sstride, std::make_integer_sequence<int, K>{});
  }
  __device__ __forceinline__ void commit(float* __restrict__ dst, int units, int dstride, float* trash) const {
    commit_(dst, units, dstride, trash, std::make_integer_sequence<int, K>{});
  }
};

// pooled rows of clip n, channels [c0, c0 + nc) -> LDS [nc][T + 2] zero padded in t (what load_pooled_t builds)
template <int K, bool TC>
struct StagePool {
  float4 v[K];  // 16 bytes: two int64 fixed-point sums (frame-major source) or four floats
  static __device__ __forceinline__ float4 ld(const TamBranchArgs& a, int n, int c0, int nc, const vitta_conv::FastDiv dc2, int k) {
    const int C = a.C, T = a.T;
    if (TC) {
      const long long* srct = reinterpret_cast<const long long*>(a.pooled) + (int64_t)n * T * C + c0;
      const int c2n = nc >> 1, n2 = c2n * T;
      const int i = min((int)threadIdx.x + k * TBW, n2 - 1);
      const int t = vitta_conv::fdiv(i, dc2), c2 = i - t * c2n;
      return *reinterpret_cast<const float4*>(srct + (int64_t)t * C + 2 * c2);
    }
    const float4* s4 = reinterpret_cast<const float4*>(a.pooled + ((int64_t)n * C + c0) * T);
    return s4[min((int)threadIdx.x + k * TBW, nc * (T >> 2) - 1)];
  }
  static __device__ __forceinline__ void st(const TamBranchArgs& a, float* __restrict__ pl, int nc, const vitta_conv::FastDiv dc2, float* trash,
                                            int k, float4 val) {
    const int T = a.T, TP = T + 2;
    const int i = threadIdx.x + k * TBW;
    if (TC) {
      const int c2n = nc >> 1, n2 = c2n * T;
      const int t = vitta_conv::fdiv(i, dc2), c = 2 * (i - t * c2n);
      const unsigned w[4] = {__float_as_uint(val.x), __float_as_uint(val.y), __float_as_uint(val.z), __float_as_uint(val.w)};
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // (integer part + fraction: the exact conversion of load_pooled_t)
        float* d = i < n2 ? pl + (c + j) * TP + 1 + t : trash + j;
        *d = fixed_to_float((int)w[2 * j + 1], w[2 * j]);
      }
    } else {
      const int t4 = T >> 2, n4 = nc * t4;
      const int c = vitta_conv::fdiv(i, a.d_t4), tq = i - c * t4;
      float* d = i < n4 ? pl + c * TP + 1 + 4 * tq : trash;
      d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
  }
  template <int... I>
  __device__ __forceinline__ void issue_(const TamBranchArgs& a, int n, int c0, int nc, const vitta_conv::FastDiv dc2, std::integer_sequence<int, I...>) {
    ((v[I] = ld(a, n, c0, nc, dc2, I)), ...);
  }
  template <int... I>
  __device__ __forceinline__ void commit_(const TamBranchArgs& a, float* __restrict__ pl, int nc, const vitta_conv::FastDiv dc2, float* trash,
                                          std::integer_sequence<int, I...>) const {
    (st(a, pl, nc, dc2, trash, I, v[I]), ...);
  }
  __device__ __forceinline__ void issue(const TamBranchArgs& a, int n, int c0, int nc, const vitta_conv::FastDiv dc2) {
    issue_(a, n, c0, nc, dc2, std::make_integer_sequence<int, K>{});
  }
  __device__ __forceinline__ void commit(const TamBranchArgs& a, float* __restrict__ pl, int nc, const vitta_conv::FastDiv dc2, float* trash) const {
    commit_(a, pl, nc, dc2, trash, std::make_integer_sequence<int, K>{});
    const int T = a.T, TP = T + 2;
    for (int c = threadIdx.x; c < nc; c += TBW) {
      pl[c * TP] = 0.f;
      pl[c * TP + T + 1] = 0.f;
    }
  }
};

struct GRegs {  // the G branch's parameters on their way to LDS (stage_g's layout): M * T <= 512, 3 M <= 96, M <= 32
  float w1a, w1b, w3, rv, wg, rm, b;
  __device__ __forceinline__ void issue(const TamBranchArgs& a) {
    const int T = a.T, M = 2 * T;
    w1a = a.wg1[min((int)threadIdx.x, M * T - 1)];
    w1b = a.wg1[min((int)threadIdx.x + TBW, M * T - 1)];
    w3 = a.wg3[min((int)threadIdx.x, 3 * M - 1)];
    const int m = min((int)threadIdx.x, M - 1);
    rv = a.bng.rv[m]; wg = a.bng.w[m]; rm = a.bng.rm[m]; b = a.bng.b[m];
  }
  __device__ __forceinline__ GLds commit(const TamBranchArgs& a, float* gl, float* trash) const {
    const int T = a.T, M = 2 * T;
    float* wg1 = gl; float* wg3 = wg1 + M * T; float* rm_ = wg3 + 3 * M; float* sc = rm_ + M; float* b_ = sc + M;
    float* is = b_ + M; float* w = is + M;
    const int tx = threadIdx.x;
    *(tx < M * T ? wg1 + tx : trash) = w1a;
    *(tx + TBW < M * T ? wg1 + tx + TBW : trash) = w1b;
    *(tx < 3 * M ? wg3 + tx : trash) = w3;
    const bool on = tx < M;
    const float r = rsqrtf(rv + a.bng.eps);
    *(on ? rm_ + tx : trash) = rm;
    *(on ? sc + tx : trash + 1) = wg * r;
    *(on ? b_ + tx : trash + 2) = b;
    *(on ? is + tx : trash + 3) = r;
    *(on ? w + tx : trash + 4) = wg;
    return GLds{wg1, wg3, rm_, sc, b_, is, w};
  }
};

// barrier for LDS hand-over between the waves of a workgroup WITHOUT draining vector-memory operations (a __syncthreads()
// waits for every outstanding global access of the wave -- among them the leave ticket below)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// All workgroups of clip n meet, with NO ticket that returns a value (a returning atomic is a round trip in front of phase 2; the
// compiler's wave-level atomic optimisation makes it un-deferrable) and no reset: the arrival counter only ever grows, and a launch
// waits for base + workgroups, `base` being a second word = the arrivals of all EARLIER launches, which tile 0 advances once everybody
// has arrived -- i.e. after every workgroup of this launch has read it (at its start) and before the next launch on the stream can
// (launches of different grids share the pair).  Words: fast[2 n] = arrivals, fast[2 n + 1] = base, in the second half of the
// stream's meeting buffer (the kernels above keep their zero-at-rest pair in the first half); at rest both are equal.  Wrap-around
// safe (differences).  The caller's write-through stores are drained before lane 0 arrives.
constexpr int FAST_SYNC_OFF = 128;
__device__ __forceinline__ void meet(unsigned* fast, unsigned nwg, unsigned base, bool bump) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(fast, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = base + nwg;
    // (bounded: ~2 s of polling.  Every workgroup of the launch is resident -- fused_capacity -- so the bound is never reached in a
    // correct run; if it ever is, the launch finishes with wrong numbers instead of hanging the device)
    for (int spin = 0; spin < (1 << 21) && (int)(__hip_atomic_load(fast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0; ++spin)
      __builtin_amdgcn_s_sleep(1);
    if (bump) __hip_atomic_store(fast + 1, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  lds_barrier();
}

struct StageCoherent2 {
  float4 v0, v1;
  __device__ __forceinline__ void issue(const float* src, int units) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, units * 16, 0x00020000);
    // (a piece past the end reads zero: no clamp)
    v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16, 0, AUX_SC1));
    v1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((int)threadIdx.x + TBW) * 16, 0, AUX_SC1));
  }
};

// (every workgroup of these launches has an F1 / B1 tile -- fast_ok's shapes give nt1 >= nt2 --, so the loads are issued
// unconditionally, straight-line: a conditional issue makes the compiler copy the loaded registers at the join, i.e. wait for them)
template <bool TC>
__global__ __launch_bounds__(TBW) __attribute__((amdgpu_waves_per_eu(1, 2))) void tam_branch_fwd_fast_kernel(TamBranchArgs a, float* __restrict__ kern, float* __restrict__ h_pre,
                                                                  float* __restrict__ h_act, float* __restrict__ gate, unsigned* sync,
                                                                  int nt1, int nt2, int lds1_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y;
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2;
  const bool do2 = tile < nt2;
  float* wl1 = smem;                 // F1 (f1_body's layout): [OBF][C*3] | pl [C][T+2] | red | G parameters
  float* pl = wl1 + OBF * C * 3;
  float* red = pl + C * TP;
  float* gl = red + TBW;
  float* wl2 = smem + lds1_floats;   // F2: [CB][O] | hl [O][T]
  float* hl = wl2 + CB * O;
  Stage16<4, 0> s_w0, s_w3;
  StagePool<8, TC> s_pool;
  GRegs s_g;
  BnItem bn;
  const int items = OBF * T, CS = TBW / items > 0 ? TBW / items : 1, item = threadIdx.x / CS;
  const int o0 = tile * OBF, c0 = do2 ? tile * CB : 0, ob = min(o0 + item / T, O - 1);
  const int u_w0 = OBF * C * 3 / 4, u_w3 = CB * O / 4;
  // ---- every global operand of this workgroup, in flight together ----
  s_w0.issue(a.w0 + (int64_t)o0 * C * 3, u_w0, 0);
  s_pool.issue(a, n, 0, C, a.d_c2);
  s_g.issue(a);
  bn.w = a.bnl.w[ob]; bn.rv = a.bnl.rv[ob]; bn.rm = a.bnl.rm[ob]; bn.b = a.bnl.b[ob];
  s_w3.issue(a.w3 + (int64_t)c0 * O, u_w3, 0);  // (a workgroup without an F2 tile reads tile 0's and drops it)
  unsigned* const fast = sync + FAST_SYNC_OFF + 2 * n;
  const unsigned gen = __hip_atomic_load(fast + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");  // (every load above is issued before the first LDS store below)
  float* const trash = hl + O * T;  // the 64 spare bytes behind the last region (f2_lds)
  s_w0.commit(wl1, u_w0, 0, trash);
  s_pool.commit(a, pl, C, a.d_c2, trash);
  const GLds gp = s_g.commit(a, gl, trash);
  s_w3.commit(wl2, do2 ? u_w3 : 0, 0, trash);
  red[threadIdx.x] = bn.w + bn.rv + bn.rm + bn.b;  // (an unconditional use: the four loads stay up here, see Stage16::commit)
  __syncthreads();
  f1_compute<true, true>(a, kern, h_pre, h_act, n, tile, nt1, wl1, pl, red, gp, bn);
  meet(fast, gridDim.y, gen, tile == 0);
  {
    StageCoherent2 s_h;
    const int u_h = O * T / 4, u0 = threadIdx.x, u1 = threadIdx.x + TBW;
    s_h.issue(h_act + (int64_t)n * O * T, do2 ? u_h : 0);
    *reinterpret_cast<float4*>((do2 && u0 < u_h) ? hl + 4 * u0 : trash) = s_h.v0;
    *reinterpret_cast<float4*>((do2 && u1 < u_h) ? hl + 4 * u1 : trash) = s_h.v1;
  }
  lds_barrier();
  if (do2) f2_compute(a, gate, n, c0, wl2, hl);
}

template <bool TC>
__global__ __launch_bounds__(TBW) __attribute__((amdgpu_waves_per_eu(1, 2))) void tam_branch_bwd_fast_kernel(TamBranchArgs a, const float* __restrict__ kern,
                                                                  const float* __restrict__ gate, const float* __restrict__ h_pre,
                                                                  const float* __restrict__ h_act, const float* __restrict__ gkern,
                                                                  const float* __restrict__ ggate, float* __restrict__ dpre_g,
                                                                  TamBranchGrads g, unsigned* sync, int nt1, int nt2, int lds1_floats) {
  // (fast_ok's shapes: nt1 == nt2 == C / 16, every workgroup has a B1 and a B2 tile)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tile = blockIdx.y;
  const int C = a.C, T = a.T, O = C / 4, TP = T + 2, M = 2 * T;
  float* dz = smem;                  // B1 (b1_body's layout): dz [C][T] | gg [C][T] | wl [C][OBB] | red
  float* gg = dz + C * T;
  float* wl1 = gg + C * T;
  float* red = wl1 + C * OBB;
  float* wl2 = smem + lds1_floats;   // B2 (b2_body's layout)
  float* dpre = wl2 + O * CBB * 3;
  float* pl = dpre + O * TP;
  float* gpo = pl + CBB * TP;
  float* gacc = gpo + CBB * T;
  float* gl = gacc + 5 * M + M * T;
  Stage16<4, 0> s_gate, s_gg;
  Stage16<2, 1> s_w3;
  Stage16<6, CBB * 3 / 4> s_w0;
  StagePool<1, TC> s_pool;
  GRegs s_g;
  BnItem bn;
  const int items = OBB * T, CS = TBW / items > 0 ? TBW / items : 1, item = threadIdx.x / CS;
  const int ob = min(tile * OBB + item / T, O - 1), t = item % T, c0 = tile * CBB;
  const int u_ct = C * T / 4;
  // ---- every global operand of this workgroup, in flight together ----
  s_gate.issue(gate + (int64_t)n * C * T, u_ct, 0);
  s_gg.issue(ggate + (int64_t)n * C * T, u_ct, 0);
  s_w3.issue(a.w3 + tile * OBB, C, O);
  {
    const int64_t idx0 = ((int64_t)n * O + ob) * T + t;
    bn.w = a.bnl.w[ob]; bn.rv = a.bnl.rv[ob]; bn.rm = a.bnl.rm[ob];
    bn.hact = h_act[idx0]; bn.hpre = h_pre[idx0];
  }
  s_w0.issue(a.w0 + (int64_t)c0 * 3, O * (CBB * 3 / 4), (int64_t)C * 3);
  s_pool.issue(a, n, c0, CBB, a.d_c2b);
  s_g.issue(a);
  const float* gk = gkern + ((int64_t)n * C + c0 + threadIdx.x / GL) * 3;
  const float gk0 = gk[0], gk1 = gk[1], gk2 = gk[2];
  unsigned* const fast = sync + FAST_SYNC_OFF + 2 * n;
  const unsigned gen = __hip_atomic_load(fast + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");  // (every load above is issued before the first LDS store below)
  float* const trash = gl + 2 * T * T + 16 * T;  // the 64 spare bytes behind the G parameters (b2_lds, g_floats)
  s_gate.commit(dz, u_ct, 0, trash);
  s_gg.commit(gg, u_ct, 0, trash);
  s_w3.commit(wl1, C, OBB, trash);
  s_w0.commit(wl2, O * (CBB * 3 / 4), CBB * 3, trash);
  s_pool.commit(a, pl, CBB, a.d_c2b, trash);
  const GLds gp = s_g.commit(a, gl, trash);
  red[threadIdx.x] = bn.w + bn.rv + bn.rm + bn.hact + bn.hpre + gk0 + gk1 + gk2;  // (an unconditional use: these loads stay up here)
  for (int i = threadIdx.x; i < 5 * M + M * T; i += TBW) gacc[i] = 0.f;
  for (int i = threadIdx.x; i < CBB * T; i += TBW) gpo[i] = 0.f;
  __syncthreads();
  b1_compute<true, true>(a, h_pre, h_act, dpre_g, g, n, tile, dz, gg, wl1, red, bn);
  b2_g_compute<true>(a, g, gp, gkern, n, c0, pl, gpo, gacc, gk0, gk1, gk2);
  meet(fast, gridDim.y, gen, tile == 0);
  {
    StageCoherent2 s_d;
    const int u_d = O * T / 4, t4 = T >> 2;
    s_d.issue(dpre_g + (int64_t)n * O * T, u_d);
    for (int i = threadIdx.x; i < O; i += TBW) {  // the two pad columns of [O][T + 2]
      dpre[i * TP] = 0.f;
      dpre[i * TP + T + 1] = 0.f;
    }
    {
      const int u = threadIdx.x;
      const int oo = vitta_conv::fdiv(u, a.d_t4), tq = u - oo * t4;
      float* d = u < u_d ? dpre + oo * TP + 1 + 4 * tq : trash;
      d[0] = s_d.v0.x; d[1] = s_d.v0.y; d[2] = s_d.v0.z; d[3] = s_d.v0.w;
    }
    {
      const int u = threadIdx.x + TBW;
      const int oo = vitta_conv::fdiv(u, a.d_t4), tq = u - oo * t4;
      float* d = u < u_d ? dpre + oo * TP + 1 + 4 * tq : trash;
      d[0] = s_d.v1.x; d[1] = s_d.v1.y; d[2] = s_d.v1.z; d[3] = s_d.v1.w;
    }
  }
  lds_barrier();
  b2_l_compute(a, g, c0, wl2, dpre, pl, gpo);
  lds_barrier();
  b2_finish(a, g, n, c0, gpo, gacc);
}

// The L branch's two weight gradients as their own launch (round 5: SGD over all parameters -- inside the fused backward they are
// C x C/4 x 4 global atomics per clip on the MAIN chain of the step, 37 us per block instead of 17-21; nothing in the backward waits for
// them, so the trunk issues this launch on its weight-gradient helper stream):
//   dW3[c, o]    += sum_n sum_t dz[n, c, t] h[n, o, t],                dz = d gate * gate * (1 - gate)
//   dW0[o, c, j] += sum_n sum_t dpre[n, o, t] pooled[n, c, t + j - 1]  (zero padding in t)
// one thread per output element, both clips summed in registers: single writer, plain read-modify-write, deterministic.
__global__ __launch_bounds__(TBW) void tam_branch_wgrad_kernel(TamBranchArgs a, const float* __restrict__ gate, const float* __restrict__ ggate,
                                                               const float* __restrict__ h_act, const float* __restrict__ dpre,
                                                               float* __restrict__ dw0, float* __restrict__ dw3) {
  const int C = a.C, T = a.T, O = C / 4;
  const int64_t n3 = (int64_t)C * O, n0 = (int64_t)O * C * 3;
  const int64_t idx = (int64_t)blockIdx.x * TBW + threadIdx.x;
  if (idx < n3 && dw3) {
    const int c = (int)(idx / O), o = (int)(idx - (int64_t)c * O);
    float s = 0.f;
    for (int n = 0; n < a.N; ++n) {
      const float* gt = gate + ((int64_t)n * C + c) * T;
      const float* gg = ggate + ((int64_t)n * C + c) * T;
      const float* hh = h_act + ((int64_t)n * O + o) * T;
      for (int t = 0; t < T; ++t) {
        const float g1 = gt[t];
        s = fmaf(gg[t] * g1 * (1.f - g1), hh[t], s);
      }
    }
    dw3[idx] += s;
  } else if (idx >= n3 && idx < n3 + n0 && dw0) {
    const int64_t e = idx - n3;
    const int j = (int)(e % 3), c = (int)((e / 3) % C), o = (int)(e / (3 * (int64_t)C));
    float s = 0.f;
    for (int n = 0; n < a.N; ++n) {
      const float* dp = dpre + ((int64_t)n * O + o) * T;
      for (int t = 0; t < T; ++t) {
        const int tp = t + j - 1;
        if (tp < 0 || tp >= T) continue;
        float pv;
        if (a.pooled_tc) {
          const long long q = reinterpret_cast<const long long*>(a.pooled)[((int64_t)n * T + tp) * C + c];
          pv = fixed_to_float((int)(q >> 32), (unsigned)(q & 0xffffffffll));
        } else {
          pv = a.pooled[((int64_t)n * C + c) * T + tp];
        }
        s = fmaf(dp[t], pv, s);
      }
    }
    dw0[e] += s;
  }
}

inline bool fast_ok(int C, int T) { return C % 64 == 0 && C <= 512 && T % 4 == 0 && T <= T_MAX && C * T <= 4096; }

inline size_t g_floats(int T) { return (size_t)2 * T * T + 3 * 2 * T + 5 * 2 * T; }
inline size_t f1_lds(int C, int T) {
  return sizeof(float) * ((size_t)C * (T + 2) + TBW + (size_t)OBF * C * 3 + g_floats(T)) + 64;
}
inline size_t f2_lds(int C, int T) { return sizeof(float) * ((size_t)(C / 4) * T + (size_t)CB * (C / 4)) + 64; }
inline size_t b1_lds(int C, int T) { return sizeof(float) * ((size_t)2 * C * T + TBW + (size_t)C * OBB) + 64; }
inline size_t b2_lds(int C, int T) {
  return sizeof(float) * ((size_t)(C / 4) * (T + 2) + CBB * (T + 2) + CBB * T + 5 * 2 * T + 2 * T * T +
                          (size_t)(C / 4) * CBB * 3 + g_floats(T)) + 64;
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

namespace {

// The fused launches spin in clip_meet until every workgroup of a clip has arrived: the WHOLE grid has to be resident.
// Capacity = what the occupancy query admits per CU for this kernel and LDS size, times the CUs, HALVED: the overlapped
// schedule may run the evaluation pass's fused launch beside the adaptation pass's on a second stream.
template <typename Kern>
int64_t fused_capacity(Kern kernel, size_t lds) {
  // one table per kernel instantiation, keyed by (device, LDS size), under a lock: the entry points are called from the main
  // thread and from autograd's worker thread, and the answer differs between devices with different CU counts.
  // ASSUMPTION the halving encodes: at most TWO fused launches of a kernel run side by side (the adaptation stream and the
  // evaluation stream of tta.ViTTAAdapter.step); a caller that runs the trunk on more streams at once must take the two-launch
  // form (vitta_tam_branch_{fwd,bwd}_f32), which has no device-side meeting point.
  static std::mutex mu;
  static std::map<std::pair<std::pair<int, size_t>, const void*>, int64_t> table;  // (kernels of one signature share this instantiation)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(std::make_pair(dev, lds), reinterpret_cast<const void*>(kernel));
  const auto hit = table.find(key);
  if (hit != table.end()) return hit->second;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0 ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TBW, lds) != hipSuccess || per_cu <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  const int64_t cap = (int64_t)per_cu * cus / 2;
  table[key] = cap;
  return cap;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// the one-batch kernels take the launch when the shape fits their batches and every 16-byte access is aligned
inline bool use_fast(TamBranchArgs& a, std::initializer_list<const void*> ptrs) {
  if (!fast_ok(a.C, a.T) || std::getenv("VITTA_TAM_FAST_OFF")) return false;
  for (const void* p : ptrs)
    if (!al16(p)) return false;
  a.d_c2 = vitta_conv::make_fastdiv(a.C / 2);
  a.d_c2b = vitta_conv::make_fastdiv(CBB / 2);
  a.d_t4 = vitta_conv::make_fastdiv(a.T / 4);
  return true;
}

int fused_geometry(int32_t N, int32_t C, int32_t T, bool bwd, int& nt1, int& nt2, size_t& l1, size_t& lds) {
  if (!vitta_tam_branch_supported(C, T) || N < 1 || N > 32) return 0;
  const int O = C / 4;
  nt1 = bwd ? (O + OBB - 1) / OBB : (O + OBF - 1) / OBF;
  nt2 = bwd ? (C + CBB - 1) / CBB : (C + CB - 1) / CB;
  l1 = ((bwd ? b1_lds(C, T) : f1_lds(C, T)) + 15) / 16 * 16;
  lds = l1 + (bwd ? b2_lds(C, T) : f2_lds(C, T));
  return lds <= 160 * 1024 ? 1 : 0;
}

}  // namespace

extern "C" {

int vitta_tam_branch_supported(int32_t C, int32_t T) {
  return (T >= 1 && T <= T_MAX && OBB * T <= TBW && C >= 4 && C % 4 == 0 && f1_lds(C, T) <= 160 * 1024 && f2_lds(C, T) <= 160 * 1024 &&
          b1_lds(C, T) <= 160 * 1024 && b2_lds(C, T) <= 160 * 1024) ? 1 : 0;
}

int vitta_tam_branch_fwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                             float* d_hpre, int32_t pooled_tc, void* stream) {
  // d_hpre: 2 * N * (C/4) * T floats: conv1 output before BN, then after BN + ReLU
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre) return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T, pooled_tc ? 1 : 0};
  if (tb_bad(a)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int O = C / 4;
  float* d_hact = d_hpre + (int64_t)N * O * T;
  if (!set_lds(tam_branch_f1_kernel, f1_lds(C, T)) || !set_lds(tam_branch_f2_kernel, f2_lds(C, T))) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_f1_kernel, dim3(N, (O + OBF - 1) / OBF), dim3(TBW), f1_lds(C, T), st, a, d_kern, d_hpre, d_hact);
  VITTA_LAUNCH(tam_branch_f2_kernel, dim3(N, (C + CB - 1) / CB), dim3(TBW), f2_lds(C, T), st, a, d_hact, d_gate);
  return VITTA_OK;
}

int vitta_tam_branch_fused_supported(int32_t N, int32_t C, int32_t T) {
  int nt1, nt2;
  size_t l1, lds;
  for (int bwd = 0; bwd < 2; ++bwd) {
    if (!fused_geometry(N, C, T, bwd, nt1, nt2, l1, lds)) return 0;
    const bool ok = bwd ? (set_lds(tam_branch_bwd_fused_kernel, lds) && set_lds(tam_branch_bwd_fast_kernel<true>, lds) && set_lds(tam_branch_bwd_fast_kernel<false>, lds))
                        : (set_lds(tam_branch_fwd_fused_kernel, lds) && set_lds(tam_branch_fwd_fast_kernel<true>, lds) && set_lds(tam_branch_fwd_fast_kernel<false>, lds));
    if (!ok) return 0;
    int64_t cap = bwd ? fused_capacity(tam_branch_bwd_fused_kernel, lds) : fused_capacity(tam_branch_fwd_fused_kernel, lds);
    for (int tc = 0; tc < 2; ++tc) {
      const int64_t cap2 = bwd ? (tc ? fused_capacity(tam_branch_bwd_fast_kernel<true>, lds) : fused_capacity(tam_branch_bwd_fast_kernel<false>, lds))
                               : (tc ? fused_capacity(tam_branch_fwd_fast_kernel<true>, lds) : fused_capacity(tam_branch_fwd_fast_kernel<false>, lds));
      cap = cap2 < cap ? cap2 : cap;
    }
    if ((int64_t)N * (nt1 > nt2 ? nt1 : nt2) > cap) return 0;
  }
  return 1;
}

int vitta_tam_branch_fwd_fused_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                                   const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                                   const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                                   float* d_hpre, void* d_sync, int32_t pooled_tc, void* stream) {
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre || !d_sync) return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T) || N > 32) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T, pooled_tc ? 1 : 0};
  if (tb_bad(a)) return VITTA_ERR_INVALID_ARG;
  const int O = C / 4;
  int nt1, nt2;
  size_t l1, lds;
  if (!fused_geometry(N, C, T, false, nt1, nt2, l1, lds)) return VITTA_ERR_UNSUPPORTED;
  float* d_hact = d_hpre + (int64_t)N * O * T;
  if (use_fast(a, {d_pooled, d_w0, d_w3, d_hact})) {
    auto kfn = a.pooled_tc ? tam_branch_fwd_fast_kernel<true> : tam_branch_fwd_fast_kernel<false>;
    if (!set_lds(kfn, lds)) return VITTA_ERR_LAUNCH;
    if ((int64_t)N * (nt1 > nt2 ? nt1 : nt2) > fused_capacity(kfn, lds)) return VITTA_ERR_UNSUPPORTED;
    VITTA_LAUNCH(kfn, dim3(N, nt1 > nt2 ? nt1 : nt2), dim3(TBW), lds, static_cast<hipStream_t>(stream), a, d_kern,
                 d_hpre, d_hact, d_gate, static_cast<unsigned*>(d_sync), nt1, nt2, (int)(l1 / 4));
    return VITTA_OK;
  }
  if (!set_lds(tam_branch_fwd_fused_kernel, lds)) return VITTA_ERR_LAUNCH;
  if ((int64_t)N * (nt1 > nt2 ? nt1 : nt2) > fused_capacity(tam_branch_fwd_fused_kernel, lds)) return VITTA_ERR_UNSUPPORTED;  // every workgroup resident
  VITTA_LAUNCH(tam_branch_fwd_fused_kernel, dim3(N, nt1 > nt2 ? nt1 : nt2), dim3(TBW), lds, static_cast<hipStream_t>(stream), a, d_kern,
               d_hpre, d_hact, d_gate, static_cast<unsigned*>(d_sync), nt1, nt2, (int)(l1 / 4));
  return VITTA_OK;
}

int vitta_tam_branch_bwd_fused_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                                   const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                                   const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                                   const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                                   float* d_gpooled, float* const* h_dbn, float* const* h_dw, void* d_sync, int32_t pooled_tc, void* stream) {
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre || !d_gkern || !d_ggate || !d_gpooled || !h_dbn || !d_sync)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T) || N > 32) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T, pooled_tc ? 1 : 0};
  if (tb_bad(a) || !h_dbn[0] || !h_dbn[1] || !h_dbn[2] || !h_dbn[3]) return VITTA_ERR_INVALID_ARG;
  TamBranchGrads g{d_gpooled, h_dbn[0], h_dbn[1], h_dbn[2], h_dbn[3], h_dw ? h_dw[0] : nullptr, h_dw ? h_dw[1] : nullptr,
                   h_dw ? h_dw[2] : nullptr, h_dw ? h_dw[3] : nullptr};
  const int O = C / 4;
  int nt1, nt2;
  size_t l1, lds;
  if (!fused_geometry(N, C, T, true, nt1, nt2, l1, lds)) return VITTA_ERR_UNSUPPORTED;
  const float* d_hact = d_hpre + (int64_t)N * O * T;
  float* d_dpre = d_gpooled + (int64_t)N * C * T;
  if (use_fast(a, {d_pooled, d_w0, d_w3, d_gate, d_ggate, d_dpre})) {
    auto kfn = a.pooled_tc ? tam_branch_bwd_fast_kernel<true> : tam_branch_bwd_fast_kernel<false>;
    if (!set_lds(kfn, lds)) return VITTA_ERR_LAUNCH;
    if ((int64_t)N * (nt1 > nt2 ? nt1 : nt2) > fused_capacity(kfn, lds)) return VITTA_ERR_UNSUPPORTED;
    VITTA_LAUNCH(kfn, dim3(N, nt1 > nt2 ? nt1 : nt2), dim3(TBW), lds, static_cast<hipStream_t>(stream), a, d_kern,
                 d_gate, d_hpre, d_hact, d_gkern, d_ggate, d_dpre, g, static_cast<unsigned*>(d_sync), nt1, nt2, (int)(l1 / 4));
    return VITTA_OK;
  }
  if (!set_lds(tam_branch_bwd_fused_kernel, lds)) return VITTA_ERR_LAUNCH;
  if ((int64_t)N * (nt1 > nt2 ? nt1 : nt2) > fused_capacity(tam_branch_bwd_fused_kernel, lds)) return VITTA_ERR_UNSUPPORTED;
  VITTA_LAUNCH(tam_branch_bwd_fused_kernel, dim3(N, nt1 > nt2 ? nt1 : nt2), dim3(TBW), lds, static_cast<hipStream_t>(stream), a, d_kern,
               d_gate, d_hpre, d_hact, d_gkern, d_ggate, d_dpre, g, static_cast<unsigned*>(d_sync), nt1, nt2, (int)(l1 / 4));
  return VITTA_OK;
}

int vitta_tam_branch_bwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                             const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                             float* d_gpooled, float* const* h_dbn /* {dG.w, dG.b, dL.w, dL.b} accumulated into */,
                             float* const* h_dw /* {dwg1, dwg3, dw0, dw3} accumulated into, or NULL entries */, int32_t pooled_tc, void* stream) {
  // d_gpooled doubles as scratch: it must have room for N*C*T + N*(C/4)*T floats (result, then d conv1-output)
  if (!h_bn_g || !h_bn_l || !d_kern || !d_gate || !d_hpre || !d_gkern || !d_ggate || !d_gpooled || !h_dbn)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_tam_branch_supported(C, T)) return VITTA_ERR_UNSUPPORTED;
  TamBranchArgs a{d_pooled, d_wg1, BnEval{h_bn_g[0], h_bn_g[1], h_bn_g[2], h_bn_g[3], eps_g}, d_wg3, d_w0,
                  BnEval{h_bn_l[0], h_bn_l[1], h_bn_l[2], h_bn_l[3], eps_l}, d_w3, N, C, T, pooled_tc ? 1 : 0};
  if (tb_bad(a) || !h_dbn[0] || !h_dbn[1] || !h_dbn[2] || !h_dbn[3]) return VITTA_ERR_INVALID_ARG;
  TamBranchGrads g{d_gpooled, h_dbn[0], h_dbn[1], h_dbn[2], h_dbn[3], h_dw ? h_dw[0] : nullptr, h_dw ? h_dw[1] : nullptr,
                   h_dw ? h_dw[2] : nullptr, h_dw ? h_dw[3] : nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int O = C / 4;
  const float* d_hact = d_hpre + (int64_t)N * O * T;
  float* d_dpre = d_gpooled + (int64_t)N * C * T;
  if (!set_lds(tam_branch_b1_kernel, b1_lds(C, T)) || !set_lds(tam_branch_b2_kernel, b2_lds(C, T))) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tam_branch_b1_kernel, dim3(N, (O + OBB - 1) / OBB), dim3(TBW), b1_lds(C, T), st, a, d_gate, d_hpre, d_hact,
               d_ggate, d_dpre, g);
  VITTA_LAUNCH(tam_branch_b2_kernel, dim3(N, (C + CBB - 1) / CBB), dim3(TBW), b2_lds(C, T), st, a, d_kern, d_gkern, d_dpre, g);
  return VITTA_OK;
}

int vitta_tam_branch_wgrad_f32(const float* d_pooled, int32_t pooled_tc, const float* d_gate, const float* d_ggate, const float* d_hact,
                               const float* d_dpre, int32_t N, int32_t C, int32_t T, float* d_dw0, float* d_dw3, void* stream) {
  if (!d_pooled || !d_gate || !d_ggate || !d_hact || !d_dpre || N <= 0 || C <= 0 || T <= 0 || C % 4) return VITTA_ERR_INVALID_ARG;
  if (!d_dw0 && !d_dw3) return VITTA_OK;
  TamBranchArgs a{};
  a.pooled = d_pooled; a.N = N; a.C = C; a.T = T; a.pooled_tc = pooled_tc ? 1 : 0;
  const int64_t total = (int64_t)C * (C / 4) * 4;
  VITTA_LAUNCH(tam_branch_wgrad_kernel, dim3((unsigned)((total + TBW - 1) / TBW)), dim3(TBW), 0, static_cast<hipStream_t>(stream), a, d_gate,
               d_ggate, d_hact, d_dpre, d_dw0, d_dw3);
  return VITTA_OK;
}

}  // extern "C"
