// Dense product on bf16 OPERANDS IN MEMORY for the bf16 recipe of Video Swin-B (BASELINE config 5; SURVEY A10:
// swin_transformer.py:30-35, 144, 165, 304-311):   y[m][n] = sum_k a[m][k] b[n][k] (+ bias[n]),   a, b bfloat16, y fp32.
// gemm.hip's bf16 variant reads fp32 activations and rounds them while staging through registers (64 x 64 tiles: 21 flop per
// byte from L2, 215-260 TF); here both operands are 2-byte in HBM and go to LDS by LDS-DMA (no registers, no VALU), the tile is
// 128 x 128 (64 flop per byte):
//   * a stage = [128 rows][64 bytes] per operand (a 32-wide k-step), 8 KB each; THREE stages of both = 48 KB: three workgroups
//     per CU.  Step s: wait for its images (counted vmcnt), ONE barrier -- behind it every wave has also finished reading the
//     stage of step s - 1, which takes the request of step s + 2 --, then the step's 8 operand reads and 8 MFMAs per wave: two
//     steps of requests in flight at all times.  (Measured on the way: two stages of 64-wide steps, two barriers per step, two
//     workgroups per CU: 310 TF over the set; four stages of 32-wide steps with two barriers: 296; this form: 352.)
//   * one DMA instruction = 16 rows x 64 bytes; the four 16-byte chunks of a row are XOR-ed with (row / 4) % 4 on the SOURCE
//     side, so that the rows a ds_read_b128 serves per cycle land in different bank groups;
//   * four waves as 2 x 2, a wave = 64 x 64 outputs = 2 x 2 accumulators of v_mfma_f32_32x32x16_bf16.
// Measured (tools/debug/gemm_bf16x_probe.py, config 5's sixteen shapes): 352 TF over the set against 268 TF for gemm.hip's bf16
// kernel and 529 TF for hipBLASLt writing bf16; 500-610 TF on the K >= 1024 shapes, 165-183 TF on the K = 128 ones, whose fp32
// output (308 MB at 200 704 tokens x 384) is what the launch waits for: the next step is a bf16 output written in whole lines.
// Round-3 state: a stand-alone kernel with its test and bench; the model path still runs gemm.hip -- it has no bf16 activations
// to hand over yet.
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::xcd_remap;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmX {
  const void* a;      // [M][K] bf16
  const void* b;      // [N][K] bf16
  const float* bias;  // [N] or null
  float* y;           // [M][N]
  int M, N, K;
  int nMt, nNt;
};

__global__ __launch_bounds__(256, 3) void gemm_bf16x_kernel(const GemmX g) {
  constexpr int BM = 128, BN = 128, BK = 32, NB = 3;
  constexpr int STAGE = BM * BK * 2;  // bytes per operand per stage (8 KB): three stages of both = 48 KB, three workgroups per CU
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

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), 0, (int)((int64_t)M * K * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.b), 0, (int)((int64_t)N * K * 2), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // requests: wave w loads rows 32 w .. 32 w + 31 of both operands, instruction u = rows 32 w + 16 u .. + 15;
  // lane -> row + lane / 4, LDS chunk lane % 4 = the row's global chunk (lane % 4) ^ (lane / 16)   [(row / 4) % 4 = lane / 16]
  int voff_a[2], voff_b[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rl = 32 * wave + 16 * u + (lane >> 2), ch = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    voff_a[u] = min(m0 + rl, M - 1) * K * 2 + ch;  // tail rows re-read the last valid one (never stored)
    voff_b[u] = (n0 + rl) * K * 2 + ch;
  }
  int q = 0;  // next step to request (clamped to the last one: the tail re-requests it into a stage nobody reads again)
  auto request = [&](int stage) __attribute__((always_inline)) {
    unsigned char* da = Ab + stage * STAGE + wave * 2048;
    unsigned char* db = Bb + stage * STAGE + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(da + u * 1024), 16, voff_a[u], q * (BK * 2), 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(db + u * 1024), 16, voff_b[u], q * (BK * 2), 0, 0);
    q += (q + 1 < S) ? 1 : 0;
  };

  // operand reads: row r of a stage, 16-wide k-step kk (chunks 2 kk, 2 kk + 1: the lane's half lk)
  int a_row[2], b_row[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    a_row[x] = 64 * wm + 32 * x + li;
    b_row[x] = 64 * wn + 32 * x + li;
  }
  auto frag = [&](const unsigned char* base, int row, int kk) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(base + row * 64 + (((2 * kk + lk) ^ ((row >> 2) & 3)) << 4));
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

  request(0);
  request(1);
  int st = 0;
  for (int s = 0; s < S; ++s) {
    // step s has landed (step s + 1 may still be in flight); behind the barrier every wave has also finished reading the stage
    // of step s - 1, which takes the request of step s + 2: ONE barrier per step, two steps of requests in flight
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
    barrier();
    request(st + 2 >= NB ? st + 2 - NB : st + 2);
    const unsigned char* as_ = Ab + st * STAGE;
    const unsigned char* bs_ = Bb + st * STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        fa[x] = frag(as_, a_row[x], kk);
        fb[x] = frag(bs_, b_row[x], kk);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[x], fb[y], acc[x][y], 0, 0, 0);
    }
    st = st + 1 == NB ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's surplus requests must not land in the next workgroup's LDS

  // epilogue: accumulator v of block (x, y): row 8 (v / 4) + 4 lk + v % 4, column li
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int n = n0 + 64 * wn + 32 * y + li;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + 64 * wm + 32 * x + 8 * (v >> 2) + 4 * lk + (v & 3);
        if (m < M) g.y[(int64_t)m * N + n] = acc[x][y][v] + bv;
      }
    }
  }
}

}  // namespace

extern "C" {

int vitta_gemm_bf16x_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0 && M * K * 2 < (1ll << 31) && N * K * 2 < (1ll << 31) && M * N < (1ll << 40);
}

int vitta_gemm_nt_bf16x_f32(const void* d_a, const void* d_b, const float* d_bias, float* d_y, int64_t M, int64_t N, int64_t K,
                            void* stream) {
  if (!d_a || !d_b || !d_y) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_bf16x_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmX g{d_a, d_b, d_bias, d_y, (int)M, (int)N, (int)K, (int)((M + 127) / 128), (int)(N / 128)};
  constexpr size_t lds = 2 * 3 * 128 * 32 * 2;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(gemm_bf16x_kernel, dim3((unsigned)(g.nMt * g.nNt)), dim3(256), lds, static_cast<hipStream_t>(stream), g);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // extern "C"
