// Dense product of Video Swin-B in exact fp32 (BASELINE config 3; SURVEY A10: swin_transformer.py:30-35, 144, 165, 304-311) on the
// LDS-DMA ring of gemm_bf16x.hip:   y[m][n] = epi(sum_k a[m][k] b[n][k]),   a [M][K], b [N][K] (nn.Linear's own layout), y fp32.
// gemm.hip stages both operands through registers (global load -> ds_write) into 64 x 64 tiles at four workgroups per CU and
// reaches 80-100 TF on config 3's shapes; here
//   * both operands go to LDS by LDS-DMA (no staging registers, no vector instruction): a k-step of 16 floats is 64 bytes per
//     row, byte for byte the stage geometry of gemm_bf16x.hip's 32-wide bf16 step -- [128 rows][4 chunks of 16 bytes, chunk ^
//     (row / 4) % 4], 8 KB per operand, THREE stages of both = 48 KB, three workgroups per CU, one barrier per step;
//   * 128 x 128 tile, four waves of 64 x 64 = 2 x 2 accumulators of v_mfma_f32_32x32x2_f32.  A lane's 16-byte LDS read holds
//     four consecutive k of its row; the two lane halves read DIFFERENT chunks (k = 8 j + 4 (lane / 32) + i), and MFMA i of
//     the group multiplies element i of both operands' vectors -- any common k order is a valid order (gemm.hip's trick): 8
//     operand reads feed 32 MFMAs (2048 cycles) per step and wave;
//   * the accumulator layout has a lane own one output column: every epilogue access (bias, pre-activation, gelu' operand, output)
//     is 128 contiguous bytes per row as it is -- no LDS turn-around.
// Epilogues as vitta_gemm_nt_f32: mode 0 bias, mode 1 bias + exact-erf GELU with the pre-activation kept, mode 2 times gelu'(aux).
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::xcd_remap;

namespace {

struct GemmF {
  const float* a;     // [M][K]
  const float* b;     // [N][K]
  const float* bias;  // [N] or null
  float* y;           // [M][N]
  const float* aux;   // mode 2: [M][N] pre-activation
  float* pre;         // mode 1: [M][N] pre-activation out (or null)
  int M, N, K;
  int nMt, nNt;
};

__device__ __forceinline__ float gelu_f(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float h) {
  return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void gemm_f32x_kernel(const GemmF g) {
  constexpr int BM = 128, BN = 128, BK = 16, NB = 3;
  constexpr int STAGE = BM * BK * 4;  // bytes per operand per stage (8 KB)
  constexpr int PER_STEP = 4;         // DMA instructions of a wave per step: two for a, two for b
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;               // [NB][128 rows][4 chunks of 16 bytes, chunk ^ (row / 4) % 4]
  unsigned char* const Bb = lds + NB * STAGE;  // the same for the 128 output columns

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
  const int M = g.M, N = g.N, K = g.K;
  const int S = K / BK;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)((int64_t)M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, (int)((int64_t)N * K * 4), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // requests: wave w loads rows 32 w .. 32 w + 31 of both operands, instruction u = rows 32 w + 16 u .. + 15;
  // lane -> row + lane / 4, LDS chunk lane % 4 = the row's global chunk (lane % 4) ^ (lane / 16)   [(row / 4) % 4 = lane / 16]
  int voff_a[2], voff_b[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rl = 32 * wave + 16 * u + (lane >> 2), ch = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    voff_a[u] = min(m0 + rl, M - 1) * K * 4 + ch;  // tail rows re-read the last valid one (never stored)
    voff_b[u] = (n0 + rl) * K * 4 + ch;
  }
  int q = 0;  // next step to request (clamped to the last one: the tail re-requests it into a stage nobody reads again)
  auto request = [&](int stage) __attribute__((always_inline)) {
    unsigned char* da = Ab + stage * STAGE + wave * 2048;
    unsigned char* db = Bb + stage * STAGE + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(da + u * 1024), 16, voff_a[u], q * (BK * 4), 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(db + u * 1024), 16, voff_b[u], q * (BK * 4), 0, 0);
    q += (q + 1 < S) ? 1 : 0;
  };

  // operand reads: row r of a stage, group j of eight k (chunks 2 j, 2 j + 1: the lane's half lk)
  int a_row[2], b_row[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    a_row[x] = 64 * wm + 32 * x + li;
    b_row[x] = 64 * wn + 32 * x + li;
  }
  auto frag = [&](const unsigned char* base, int row, int j) __attribute__((always_inline)) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(base + row * 64 + (((2 * j + lk) ^ ((row >> 2) & 3)) << 4));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;

  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // mode 2: the gelu' operand of this lane's outputs is NOT prefetched (64 registers): it is read in the epilogue, 128 contiguous
  // bytes per row and instruction
  request(0);
  request(1);
  int st = 0;
  for (int s = 0; s < S; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
    barrier();
    request(st + 2 >= NB ? st + 2 - NB : st + 2);
    const unsigned char* as_ = Ab + st * STAGE;
    const unsigned char* bs_ = Bb + st * STAGE;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        fa[x] = frag(as_, a_row[x], j);
        fb[x] = frag(bs_, b_row[x], j);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][i], fb[y][i], acc[x][y], 0, 0, 0);
    }
    st = st + 1 == NB ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's surplus requests must not land in the next workgroup's LDS

  // epilogue: accumulator v of block (x, y) = row 8 (v / 4) + 4 lk + v % 4, column li
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int n = n0 + 64 * wn + 32 * y + li;
    const float bv = (MODE != 2 && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + 64 * wm + 32 * x + 8 * (v >> 2) + 4 * lk + (v & 3);
        if (m < M) {
          const int64_t o = (int64_t)m * N + n;
          float h = acc[x][y][v] + bv;
          if constexpr (MODE == 1) {
            if (g.pre) g.pre[o] = h;
            h = gelu_f(h);
          }
          if constexpr (MODE == 2) h *= dgelu_f(g.aux[o]);
          g.y[o] = h;
        }
      }
    }
  }
}

template <int MODE>
int launch(const GemmF& g, hipStream_t st) {
  constexpr size_t lds = 2 * 3 * 128 * 16 * 4;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32x_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL((gemm_f32x_kernel<MODE>), dim3((unsigned)(g.nMt * g.nNt)), dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int vitta_gemm_f32x_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 16 == 0 && M * K * 4 < (1ll << 31) && N * K * 4 < (1ll << 31) && M * N < (1ll << 40);
}

int vitta_gemm_nt_f32x(const float* d_a, const float* d_b, const float* d_bias, const float* d_aux, float* d_y, float* d_pre, int64_t M,
                       int64_t N, int64_t K, int32_t mode, void* stream) {
  if (!d_a || !d_b || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_f32x_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmF g{d_a, d_b, d_bias, d_y, d_aux, d_pre, (int)M, (int)N, (int)K, (int)((M + 127) / 128), (int)(N / 128)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) return launch<0>(g, st);
  if (mode == 1) return launch<1>(g, st);
  return launch<2>(g, st);
}

}  // extern "C"
