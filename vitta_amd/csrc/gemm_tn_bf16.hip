// Dense WEIGHT gradients of the bf16 recipe (BASELINE config 5 under SGD over all parameters, the reference's default optimizer:
// corpus/basics.py:547-560; the products are autograd's d weight of swin_transformer.py:30-35, 144, 165, 304-311):
//   out[n][k] (+)= sum_m g[m][n] x[m][k],     g [M][N], x [M][K] bfloat16 in memory (the 2-byte activations / gradients of the data
//                                              flow, ops.bf16_flow), out [N][K] fp32 = nn.Linear's weight layout, fp32 accumulation.
// Until round 6 these ran on the exact-fp32 stream-K convolution kernel (ops._weight_grad_conv: tokens as the channel axis of a
// pointwise convolution) behind two .float() copies per product: 15 + 4.4 of config 5's 54 ms per video, at 81 % of the fp32 matrix
// peak -- only the operand width was left to gain.  Here both operands stay 2 bytes wide end to end.
//
// The contraction runs over TOKENS, the slow axis of both operands, so neither fragment of v_mfma_f32_16x16x16_bf16 (lane (i, g):
// four consecutive k of row / column i) is contiguous in memory.  The tiles sit in LDS as they sit in memory -- [32 tokens][128
// features], row pitch 144 -- and both fragments come out of them by the LDS TRANSPOSE read of gfx950 (ds_read_b64_tr_b16: lane k of a
// 16-lane group hands in the address of a quarter row of the [4 tokens][16 features] block and receives feature k's four tokens;
// semantics pinned by tools/ubench/tr16_probe.hip, used the same way by wmsa_bf16.hip::gather4).  Workgroup = 128 x 128 output tile,
// 2 x 2 waves of 64 x 64 (sixteen 16 x 16 accumulators, four A and four B fragments per 16 tokens), two LDS stages filled through
// registers (the padded pitch rules out one LDS-DMA instruction per four rows), one barrier per 32 tokens.  The token axis is cut into
// `splits` contiguous ranges so that tiles x splits ~ 3 workgroups per CU; a split writes its fp32 partial tile to the caller's
// workspace and ONE reduce launch adds the splits in split order into out (deterministic; splits = 1 writes out directly).
#include <cstdlib>

#include "common.h"

using namespace vitta;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BT = 128;        // output tile edge (features of g x features of x)
constexpr int MS = 32;         // tokens per stage
constexpr int P = 144;         // LDS row pitch in bf16 elements: 288 bytes, the four rows of a transpose read start 32 bytes apart in bank space
constexpr int TN_THREADS = 256;

struct TnArgs {
  const unsigned short* g;
  const unsigned short* x;
  float* out;
  float* part;   // [tiles][splits][BT * BT] when splits > 1
  float* dbias;  // [N] or null: += column sums of g (the bias gradient of the same nn.Linear), by the workgroups of the first k tile
  int64_t M;
  int N, K, tilesK, splits, accumulate;
};

__device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// feature f's four tokens 16 tb + 4 g .. + 3 of a row-major [tokens][features] tile: lane i of a 16-lane group asks for the quarter
// row (token 4 g + i / 4, features f0 + 4 (i % 4) ..) and receives feature f0 + i
__device__ __forceinline__ bf16x4 tr4(const unsigned short* tile, int tb, int g, int i, int f0) {
  const unsigned short* p = tile + (16 * tb + 4 * g + (i >> 2)) * P + f0 + 4 * (i & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

__global__ __launch_bounds__(TN_THREADS, 3) void gemm_tn_bf16_kernel(const TnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 2 * MS * P];  // [stage][g | x][MS][P]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int unit = blockIdx.x, tile = unit / a.splits, sp = unit - tile * a.splits;
  const int tn = tile / a.tilesK, tk = tile - tn * a.tilesK;
  const int n0 = tn * BT, k0 = tk * BT;
  // this split's token range, in stages of MS tokens
  const int64_t stages = a.M / MS;
  const int64_t s_lo = stages * sp / a.splits, s_hi = stages * (sp + 1) / a.splits;

  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a stage = 32 rows x 256 bytes per operand = 512 sixteen-byte pieces per operand: two of each per thread
  uint4 rg[2], rx[2];
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // column sums of g over this thread's rows (columns n0 + 8 (tid % 16) ..)
  const bool do_bias = a.dbias != nullptr && tk == 0;
  auto fetch = [&](int64_t s) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = tid + TN_THREADS * u, row = pc >> 4, c16 = pc & 15;
      const int64_t m = s * MS + row;
      rg[u] = *reinterpret_cast<const uint4*>(a.g + m * a.N + n0 + 8 * c16);
      rx[u] = *reinterpret_cast<const uint4*>(a.x + m * a.K + k0 + 8 * c16);
    }
  };
  auto commit = [&](int st) __attribute__((always_inline)) {
    unsigned short* gt = lds + st * 2 * MS * P;
    unsigned short* xt = gt + MS * P;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = tid + TN_THREADS * u, row = pc >> 4, c16 = pc & 15;
      *reinterpret_cast<uint4*>(gt + row * P + 8 * c16) = rg[u];
      *reinterpret_cast<uint4*>(xt + row * P + 8 * c16) = rx[u];
      if (do_bias) {
        const unsigned w[4] = {rg[u].x, rg[u].y, rg[u].z, rg[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bsum[2 * j] += __uint_as_float(w[j] << 16);
          bsum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    }
  };
  if (s_lo < s_hi) {
    fetch(s_lo);
    commit(0);
  }
  __syncthreads();
  int st = 0;
  for (int64_t s = s_lo; s < s_hi; ++s) {
    const bool more = s + 1 < s_hi;
    if (more) fetch(s + 1);  // in flight while this stage multiplies
    const unsigned short* gt = lds + st * 2 * MS * P;
    const unsigned short* xt = gt + MS * P;
#pragma unroll
    for (int tb = 0; tb < MS / 16; ++tb) {
      bf16x4 fa[4], fb[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        fa[p] = tr4(gt, tb, g, i, 64 * wn + 16 * p);
        fb[p] = tr4(xt, tb, g, i, 64 * wk + 16 * p);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = mfma(fa[p], fb[q], acc[p][q]);
    }
    if (more) commit(st ^ 1);
    __syncthreads();
    st ^= 1;
  }
  if (do_bias) {  // the 16 threads that share a column octet meet in LDS (the ring is free), one atomic per column
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(tid >> 4) * 128 + 8 * (tid & 15) + j] = bsum[j];
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
      atomicAdd(a.dbias + n0 + tid, t);
    }
  }
  // accumulator layout: lane (column j = i, row group g) holds rows 4 g + r of the 16 x 16 block (p, q)
  float* dst;
  int64_t ld;
  if (a.splits > 1) {
    dst = a.part + ((int64_t)tile * a.splits + sp) * (BT * BT);
    ld = BT;
  } else {
    dst = a.out + (int64_t)n0 * a.K + k0;
    ld = a.K;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* o = dst + (int64_t)(64 * wn + 16 * p + 4 * g + r) * ld + 64 * wk + 16 * q + i;
        if (a.splits == 1 && a.accumulate) *o += acc[p][q][r];
        else *o = acc[p][q][r];
      }
}

// out tile (+)= the splits' partial tiles, in split order; one thread per four consecutive floats of a tile row
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const TnArgs a) {
  const int tile = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;  // within the tile
  if (e >= BT * BT) return;
  const int row = e / BT, col = e - row * BT;
  const int tn = tile / a.tilesK, tk = tile - tn * a.tilesK;
  const float* p = a.part + (int64_t)tile * a.splits * (BT * BT) + e;
  float4 s = *reinterpret_cast<const float4*>(p);
  for (int z = 1; z < a.splits; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)z * (BT * BT));
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = a.out + (int64_t)(tn * BT + row) * a.K + tk * BT + col;
  if (a.accumulate) {
    const float4 c = *reinterpret_cast<const float4*>(o);
    s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
  }
  *reinterpret_cast<float4*>(o) = s;
}

inline int pick_splits(int64_t M, int tiles) {
  // ~3 workgroups per CU over the launch, at least four stages (128 tokens) per split, at most 512 splits
  const int64_t stages = M / MS;
  static const int target = [] {
    const char* e = std::getenv("VITTA_TN_WGRAD_WGS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 768;
  }();
  int64_t s = (target + tiles - 1) / tiles;
  if (s > stages / 4) s = stages / 4;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

}  // namespace

extern "C" {

int vitta_gemm_tn_bf16_supported(int64_t M, int32_t N, int32_t K) {
  return (M >= MS && M % MS == 0 && N >= BT && K >= BT && N % BT == 0 && K % BT == 0 && M * (int64_t)(N > K ? N : K) < (1ll << 40)) ? 1 : 0;
}

size_t vitta_gemm_tn_bf16_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (!vitta_gemm_tn_bf16_supported(M, N, K)) return 0;
  const int tiles = (N / BT) * (K / BT), splits = pick_splits(M, tiles);
  return splits > 1 ? (size_t)tiles * splits * BT * BT * sizeof(float) : 0;
}

int vitta_gemm_tn_bf16(const void* d_g, const void* d_x, float* d_out, int64_t M, int32_t N, int32_t K, int32_t accumulate,
                       float* d_bias_grad, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_g || !d_x || !d_out) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_tn_bf16_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(d_g) | reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_out)) & 15u) return VITTA_ERR_INVALID_ARG;
  const int tilesN = N / BT, tilesK = K / BT, tiles = tilesN * tilesK, splits = pick_splits(M, tiles);
  const size_t need = vitta_gemm_tn_bf16_workspace_bytes(M, N, K);
  if (splits > 1 && (!d_ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(d_ws) & 15u))) return VITTA_ERR_WORKSPACE;
  const TnArgs a{static_cast<const unsigned short*>(d_g), static_cast<const unsigned short*>(d_x), d_out, static_cast<float*>(d_ws), d_bias_grad,
                 M, N, K, tilesK, splits, accumulate ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  VITTA_LAUNCH(gemm_tn_bf16_kernel, dim3((unsigned)(tiles * splits)), dim3(TN_THREADS), 0, st, a);
  if (splits > 1) VITTA_LAUNCH(gemm_tn_reduce_kernel, dim3(BT * BT / 4 / 256, (unsigned)tiles), dim3(256), 0, st, a);
  return VITTA_OK;
}

}  // extern "C"
