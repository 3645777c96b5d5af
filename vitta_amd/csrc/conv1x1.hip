// 1x1 convolution + eval-mode BatchNorm (+ residual) (+ ReLU) as ONE fp32 MFMA GEMM with the normalisation in the
// epilogue (the pointwise convolutions of the ResNet-50 bottlenecks, models/tanet_models/tanet.py via
// torchvision's Bottleneck: conv1 -> bn1 -> relu and conv3 -> bn3 -> (+identity) -> relu).
//
//   z[n, k, p] = act( s_k * sum_c W[k, c] x[n, c, p] + t_k (+ res[n, k, p]) ),  s = gamma/sqrt(var+eps), t = beta - mean*s
//
// NCHW makes this a plain row-major GEMM per frame: A = W [K x C], B = x[n] [C x HW] (pixels contiguous), D = z[n].
// The library path writes the conv output and a second pass reads it back to normalise: at layer1/2 of TANet these
// GEMMs sit at the roofline knee (25 flop/byte), so the extra 8 B/element pass costs as much as the GEMM itself.
//
// Tiling (wave = 64): workgroup = 4 waves, tile BM x BN of D; v_mfma_f32_32x32x2_f32 (exact fp32), each wave owns a
// (BM/WM) x (BN/WN) sub-tile as MT x NT accumulators of 16 registers; K is walked in slabs of 16 through
// double-buffered LDS (weights stored transposed, row stride = 32 mod 64 banks -> the two k-halves of an operand read
// hit disjoint banks); the next slab's global loads are issued before the current slab's MFMAs.  The 32x32 accumulator
// layout puts 32 CONSECUTIVE PIXELS of one output channel on the 32 lanes of a half-wave: every epilogue access
// (residual read, output write) is a full 128-byte line of the NCHW plane.
#include "common.h"

using namespace vitta;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int CT = 256;  // threads

template <int BM, int BN, int WM, int WN, bool RELU, bool RES>
__global__ __launch_bounds__(CT) void conv1x1_bn_act_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                            float eps, const float* __restrict__ res, float* __restrict__ z,
                                                            int C, int K, int HW) {
  constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  constexpr int MT = TM / 32, NT = TN / 32;  // MFMA tiles per wave
  constexpr int SA = BM + 32, SB = BN + 32;  // LDS row strides (floats)
  constexpr int A4 = BM * BK / 4 / CT;       // float4 loads of the weight slab per thread
  constexpr int B4 = BN * BK / 4 / CT;       // float4 loads of the activation slab per thread
  static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0 && A4 >= 1 && B4 >= 1, "tile configuration");
  __shared__ __attribute__((aligned(16))) float As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * SB];
  __shared__ float sc_lds[BM], sh_lds[BM];  // BatchNorm scale / shift of this tile's output channels

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * BM, p0 = blockIdx.x * BN;
  const int64_t n = blockIdx.z;
  const float* __restrict__ xn = x + n * (int64_t)C * HW;

  float4 ra[A4], rb[B4];
  auto load_slab = [&](int c0) {
#pragma unroll
    for (int u = 0; u < A4; ++u) {
      const int i = tid + u * CT;          // (row m, k-quad)
      const int m = i >> 2, c4 = i & 3;
      ra[u] = (m0 + m < K) ? *reinterpret_cast<const float4*>(w + (int64_t)(m0 + m) * C + c0 + 4 * c4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < B4; ++u) {
      const int i = tid + u * CT;          // (k row, pixel quad)
      const int kr = i / (BN / 4), p4 = i % (BN / 4);
      rb[u] = (p0 + 4 * p4 < HW) ? *reinterpret_cast<const float4*>(xn + (int64_t)(c0 + kr) * HW + p0 + 4 * p4)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int u = 0; u < A4; ++u) {
      const int i = tid + u * CT;
      const int m = i >> 2, c4 = i & 3;
      float* a = As[buf] + (4 * c4) * SA + m;
      a[0] = ra[u].x; a[SA] = ra[u].y; a[2 * SA] = ra[u].z; a[3 * SA] = ra[u].w;
    }
#pragma unroll
    for (int u = 0; u < B4; ++u) {
      const int i = tid + u * CT;
      const int kr = i / (BN / 4), p4 = i % (BN / 4);
      *reinterpret_cast<float4*>(Bs[buf] + kr * SB + 4 * p4) = rb[u];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  load_slab(0);
  if (tid < BM) {
    const int m = min(m0 + tid, K - 1);
    const float sc = gamma[m] * rsqrtf(rvar[m] + eps);
    sc_lds[tid] = sc;
    sh_lds[tid] = beta[m] - rmean[m] * sc;
  }
  store_slab(0);
  __syncthreads();
  const int nslab = C / BK;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) load_slab((s + 1) * BK);
    const float* as = As[buf] + wm * TM + li;
    const float* bs = Bs[buf] + wn * TN + li;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float af[MT], bf[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) af[a] = as[(2 * ks + lk) * SA + 32 * a];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = bs[(2 * ks + lk) * SB + 32 * b];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (s + 1 < nslab) {
      store_slab(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue: register v of tile (a, b) is D[m = 32a + 8*(v/4) + 4*lk + (v%4)][p = 32b + li]
  const int64_t zn = n * (int64_t)K * HW;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    float rr[16][NT];
    if (RES) {  // all residual loads of the tile row in flight before the first use
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + wm * TM + 32 * a + 8 * (v >> 2) + 4 * lk + (v & 3);
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int p = p0 + wn * TN + 32 * b + li;
          rr[v][b] = (m < K && p < HW) ? res[zn + (int64_t)m * HW + p] : 0.f;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int ml = wm * TM + 32 * a + 8 * (v >> 2) + 4 * lk + (v & 3);
      const int m = m0 + ml;
      const float sc = sc_lds[ml], sh = sh_lds[ml];
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int p = p0 + wn * TN + 32 * b + li;
        if (m < K && p < HW) {
          float y = fmaf(acc[a][b][v], sc, sh);
          if (RES) y += rr[v][b];
          if (RELU) y = fmaxf(y, 0.f);
          z[zn + (int64_t)m * HW + p] = y;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch(const float* x, const float* w, const float* const* bn, float eps, const float* res, int relu, float* z, int64_t N,
           int C, int K, int HW, hipStream_t st) {
  const dim3 grid((HW + BN - 1) / BN, (K + BM - 1) / BM, (unsigned)N);
#define CONV_CASE(R, S)                                                                                            \
  VITTA_LAUNCH((conv1x1_bn_act_kernel<BM, BN, WM, WN, R, S>), grid, dim3(CT), 0, st, x, w, bn[0], bn[1], bn[2], bn[3], eps, \
               res, z, C, K, HW)
  if (relu && res) CONV_CASE(true, true);
  else if (relu) CONV_CASE(true, false);
  else if (res) CONV_CASE(false, true);
  else CONV_CASE(false, false);
#undef CONV_CASE
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_conv1x1_bn_act_supported(int32_t C, int32_t K, int64_t HW) {
  return (C >= BK && C % BK == 0 && K >= 1 && HW >= 4 && HW % 4 == 0) ? 1 : 0;
}

int vitta_conv1x1_bn_act_fwd_f32(const float* d_x, const float* d_weight, const float* const* h_bn, float eps,
                                 const float* d_res, int32_t relu, float* d_z, int64_t N, int32_t C, int32_t K, int64_t HW,
                                 void* stream) {
  if (!d_x || !d_weight || !h_bn || !h_bn[0] || !h_bn[1] || !h_bn[2] || !h_bn[3] || !d_z || N <= 0 || N > 65535)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_conv1x1_bn_act_supported(C, K, HW)) return VITTA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_weight)) & 15u) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // few output channels: every workgroup takes all of them and a narrow pixel tile (more workgroups);
  // otherwise 128 x 128
  if (K <= 64) return launch<64, 128, 1, 4>(d_x, d_weight, h_bn, eps, d_res, relu, d_z, N, C, K, (int)HW, st);
  return launch<128, 128, 2, 2>(d_x, d_weight, h_bn, eps, d_res, relu, d_z, N, C, K, (int)HW, st);
}

}  // extern "C"
