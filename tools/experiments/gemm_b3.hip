// Dense layers of Video Swin-B on the bf16 matrix pipe at fp32 accuracy (qkv / proj of WindowAttention3D,
// Mlp.fc1 + GELU + fc2, PatchMerging.reduction and their data gradients; swin_transformer.py:30-35, 144, 165, 304-311):
//   y[m][n] = epi(sum_k a[m][k] b[n][k])
// with every fp32 operand split into three bf16 terms and six v_mfma_f32_32x32x16_bf16 products per multiply-add, exactly
// as conv_b3.hip does for the TANet convolutions (same arithmetic, same error class; the weights arrive as the same
// pre-split image, vitta_gemm_pack_b3).  The activations are token-major here ([tokens][channels], the reduction axis
// contiguous), so the A image of a k-step is [128 rows][16 floats]:
//   * filled by LDS-DMA, 16 bytes per lane (one instruction = 16 rows x 64 bytes), with the four 16-byte units of a row
//     XOR-ed by (row / 4) % 4 on the SOURCE side so that the 16 lanes a ds_read_b128 serves per cycle (16 rows, same unit)
//     hit 16 different bank quads;
//   * an operand fragment (8 k of one row) = two ds_read_b128, split in registers (44 vector instructions) and used for
//     FOUR 32 x 32 column blocks: a wave = 32 token rows x 128 output columns, 24 MFMAs per k-step;
//   * step = one k-step of 16, ring of three stages (60 KB: two workgroups per CU), one barrier per step, counted vmcnt --
//     the pipeline of conv_b3.hip's 128 x 128 form.  No K split: these launches have hundreds to thousands of tiles.
// Epilogues as gemm.hip (lane = output column): + bias | + bias, GELU (pre-activation kept) | * gelu'(aux).
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::u32x4;
using vitta_conv::xcd_remap;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct GemmB3 {
  const float* a;     // [M][K]
  const void* b3;     // split image of b [N][K]: [K / 32][3 planes][4 octets][N][8] bf16
  const float* bias;  // [N] or null
  const float* aux;   // [M][N] (mode 2)
  float* y;           // [M][N]
  float* pre;         // [M][N] or null (mode 1)
  int M, N, K, mode;
  int nMt, nNt;
};

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

__device__ __forceinline__ float gelu_f(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float h) {
  return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__global__ __launch_bounds__(256, 2) void gemm_b3_kernel(const GemmB3 g) {
  constexpr int BM = 128, BN = 128, NY = 4, NB = 3;
  constexpr int A_STAGE = BM * 16 * 4, B_STAGE = 3 * 2 * BN * 16, PER_STEP = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;                  // [NB][128 rows][4 units of 16 bytes, unit ^ (row / 4) % 4]
  unsigned char* const Bb = lds + NB * A_STAGE;   // [NB][3 planes][2 octets][BN][8] bf16

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
  const int M = g.M, N = g.N, K = g.K;
  const int S = K / 16;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)((int64_t)M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.b3), 0, 0x7fffffff, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- requests --------------------------------------------------------------------------------------------------------
  // A: instruction j (two per wave: j = 2 w + u) = rows 16 j .. 16 j + 15; lane -> row 16 j + lane / 4, LDS unit lane % 4,
  // which holds the row's global unit (lane % 4) ^ (lane / 16) % 4  [(row / 4) % 4 = (lane / 16) % 4: 16 j is a multiple of 16]
  int voff_a[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = min(m0 + 16 * (2 * wave + u) + (lane >> 2), M - 1);  // tail rows re-read the last valid one (never stored)
    voff_a[u] = row * K * 4 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  }
  // B: 6 runs (plane, octet) of 128 x 16 bytes = 12 instructions, three per wave
  int b_soff[3], b_dst[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int i = wave * 3 + u, run = i >> 1, half = i & 1;  // run = plane * 2 + octet-of-the-step
    b_soff[u] = (((run >> 1) * 4 + (run & 1)) * N + n0 + 64 * half) * 16;
    b_dst[u] = run * 2048 + half * 1024;
  }
  int q = 0;  // next step to request (clamped to the last one: the tail re-requests it into a stage nobody reads again)
  auto request = [&](int stage) __attribute__((always_inline)) {
    const int base = (((q >> 1) * 12 + 2 * (q & 1)) * N) * 16;
    unsigned char* db = Bb + stage * B_STAGE;
#pragma unroll
    for (int u = 0; u < 3; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(db + b_dst[u]), 16, lane * 16, base + b_soff[u], 0, 0);
    unsigned char* da = Ab + stage * A_STAGE + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(da + u * 1024), 16, voff_a[u], q * 64, 0, 0);
    q += (q + 1 < S) ? 1 : 0;
  };

  // ---- operands ----------------------------------------------------------------------------------------------------------
  const int r = 32 * wave + li, sw = (r >> 2) & 3;
  const int a_lane0 = r * 64 + (((2 * lk) ^ sw) << 4), a_lane1 = r * 64 + (((2 * lk + 1) ^ sw) << 4);
  const int b_lane = (lk * BN + li) * 16;
  f32x16 acc[NY];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;

  auto read_a = [&](const unsigned char* as_, f32x4 (&raw)[2]) __attribute__((always_inline)) {
    raw[0] = *reinterpret_cast<const f32x4*>(as_ + a_lane0);
    raw[1] = *reinterpret_cast<const f32x4*>(as_ + a_lane1);
  };
  auto read_b = [&](const unsigned char* bs_, bf16x8 (&fb)[NY][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[y][p] = *reinterpret_cast<const bf16x8*>(bs_ + b_lane + (p * 2 * BN + 32 * y) * 16);
  };
  auto split = [&](const f32x4 (&raw)[2], bf16x8 (&fa)[3]) __attribute__((always_inline)) {
    u32x4 sp[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h_, m_, l_;
      split2(raw[j >> 1][2 * (j & 1)], raw[j >> 1][2 * (j & 1) + 1], h_, m_, l_);
      sp[0][j] = h_;
      sp[1][j] = m_;
      sp[2][j] = l_;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = __builtin_bit_cast(bf16x8, sp[p]);
  };
  // six products of one column block, small terms first
  auto mfma6 = [&](const bf16x8 (&fa)[3], const bf16x8 (&fby)[3], f32x16& c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fby[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fby[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fby[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fby[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fby[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fby[0], c, 0, 0, 0);
  };
  auto wait_ring = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER_STEP) : "memory");
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- pipeline: barrier X_s (step s + 1 landed) -> request step s + NB -> A reads of step s + 1 -> per column block: the six
  // MFMAs of step s, then that block's B operands of step s + 1 into the registers they leave; the split of step s + 1 rides
  // between the MFMAs (ONE set of B registers: two sets spill at 256 VGPRs) -----------------------------------------------
#pragma unroll
  for (int i = 0; i < NB; ++i) request(i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * PER_STEP) : "memory");
  barrier();
  f32x4 raw[2];
  bf16x8 fa0[3], fa1[3], fb[NY][3];
  read_a(Ab, raw);
  read_b(Bb, fb);
  split(raw, fa0);
  int st = 0;
  auto step = [&](bf16x8 (&fa_c)[3], bf16x8 (&fa_n)[3], bool last) __attribute__((always_inline)) {
    if (last) {
#pragma unroll
      for (int y = 0; y < NY; ++y) mfma6(fa_c, fb[y], acc[y]);
      return;
    }
    const int st1 = st + 1 == NB ? 0 : st + 1;
    wait_ring();
    barrier();
    request(st);
    read_a(Ab + st1 * A_STAGE, raw);
    split(raw, fa_n);
    const unsigned char* bn = Bb + st1 * B_STAGE + b_lane;
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      mfma6(fa_c, fb[y], acc[y]);
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[y][p] = *reinterpret_cast<const bf16x8*>(bn + (p * 2 * BN + 32 * y) * 16);
    }
    // schedule: the two A reads first; then per column block six MFMAs with the split's vector instructions between them
    // (44 + address arithmetic over 24 MFMAs) and the block's three B reads behind its last MFMA
    SGB(0x100, 2);
#pragma unroll
    for (int y = 0; y < NY; ++y) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        SGB(0x008, 1);
        SGB(0x002, 2);
      }
      SGB(0x100, 3);
    }
    st = st1;
  };
  for (int s = 0; s < S; s += 2) {
    step(fa0, fa1, s + 1 >= S);
    if (s + 1 < S) step(fa1, fa0, s + 2 >= S);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's surplus requests must not land in the next workgroup's LDS

  // ---- epilogue: register v of block y = row 32 w + 8 (v / 4) + 4 lk + v % 4, column 32 y + li ----------------------------------
  const int mode = g.mode;
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    const int col = n0 + 32 * y + li;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + 32 * wave + 8 * (v >> 2) + 4 * lk + (v & 3);
      if (row >= M) continue;
      const int64_t o = (int64_t)row * N + col;
      float h = acc[y][v] + bv;
      if (mode == 1) {
        if (g.pre) g.pre[o] = h;
        h = gelu_f(h);
      } else if (mode == 2) {
        h *= dgelu_f(g.aux[o]);
      }
      g.y[o] = h;
    }
  }
}
#undef SGB

// b [N][K] fp32 -> [K / 32][3 planes][4 octets][N][8] bf16: one lane per (slab, octet, n)
__global__ __launch_bounds__(256) void gemm_pack_b3_kernel(const float* __restrict__ b, u32x4* __restrict__ dst, int N, int K) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= (int64_t)(K / 8) * N) return;
  const int n = (int)(u % N);
  const int64_t r = u / N;  // slab * 4 + octet
  const int oct = (int)(r & 3);
  const int64_t cs = r >> 2;
  const float* s = b + (int64_t)n * K + cs * 32 + 8 * oct;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
  const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h_, m_, l_;
    split2(x[2 * j], x[2 * j + 1], h_, m_, l_);
    h[j] = h_;
    m[j] = m_;
    l[j] = l_;
  }
  u32x4* dp = dst + ((cs * 3) * 4 + oct) * N + n;
  const int64_t plane = 4 * (int64_t)N;
  dp[0] = h;
  dp[plane] = m;
  dp[2 * plane] = l;
}

bool supported(int64_t M, int N, int K) {
  return M >= 1 && N >= 128 && N % 128 == 0 && K >= 32 && K % 32 == 0 && M * K * 4 < (1ll << 31) && (int64_t)K * N * 6 < (1ll << 31);
}

}  // namespace

extern "C" {

int vitta_gemm_b3_supported(int64_t M, int32_t N, int32_t K) { return supported(M, N, K) ? 1 : 0; }

size_t vitta_gemm_pack_b3_bytes(int32_t N, int32_t K) { return (N > 0 && K > 0 && K % 32 == 0) ? (size_t)N * K * 6 : 0; }

int vitta_gemm_pack_b3(const float* d_b, void* d_dst, int32_t N, int32_t K, void* stream) {
  if (!d_b || !d_dst || N <= 0 || K <= 0) return VITTA_ERR_INVALID_ARG;
  if (K % 32 || (reinterpret_cast<uintptr_t>(d_b) & 15)) return VITTA_ERR_UNSUPPORTED;
  const int64_t units = (int64_t)(K / 8) * N;
  VITTA_LAUNCH(gemm_pack_b3_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_b,
               static_cast<u32x4*>(d_dst), N, K);
  return VITTA_OK;
}

int vitta_gemm_nt_b3_f32(const float* d_a, const void* d_b_b3, const float* d_bias, const float* d_aux, float* d_y, float* d_pre, int64_t M,
                         int32_t N, int32_t K, int32_t mode, void* stream) {
  if (!d_a || !d_b_b3 || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmB3 g{d_a, d_b_b3, d_bias, d_aux, d_y, d_pre, (int)M, N, K, mode, (int)((M + 127) / 128), N / 128};
  constexpr size_t lds = 3 * (128 * 16 * 4 + 3 * 2 * 128 * 16);
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_b3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  VITTA_LAUNCH(gemm_b3_kernel, dim3((unsigned)(g.nMt * g.nNt)), dim3(256), lds, static_cast<hipStream_t>(stream), g);
  return VITTA_OK;
}

}  // extern "C"
