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
// Round 4: the kernel of the bf16 recipe's dense layers (ops.py: --dense_bf16).  Epilogues as gemm.hip's -- mode 0 bias, mode 1
// bias + exact-erf GELU with the pre-activation kept, mode 2 times gelu'(aux) -- with the OUTPUT (and pre / aux) in bfloat16 where the
// consumer is another product or the residual update: fc1 -> fc2 passes 2-byte activations, the K = 128 products stop waiting for
// a 4-byte-per-element output.
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::xcd_remap;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemmX {
  const void* a;      // [M][K] bf16
  const void* b;      // [N][K] bf16
  const float* bias;  // [N] or null
  void* y;            // [M][N] fp32 or bf16 (OUT16)
  const unsigned short* aux;  // mode 2: [M][N] bf16 pre-activation
  unsigned short* pre;        // mode 1: [M][N] bf16 pre-activation out (or null)
  int M, N, K;
  int nMt, nNt;
};

// exact-erf GELU (nn.GELU, approximate = "none") to well below what a bfloat16 hand-over keeps: erf by Abramowitz-Stegun 7.1.26,
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0,   |error| <= 1.5e-7,
// with z = |h| / sqrt 2, so exp(-z^2) = exp(-h^2 / 2) is ALSO the density term of gelu' -- one v_exp, one v_rcp and ten multiply-adds
// per element where erff() is ~40 instructions (the fc1 epilogue at 200 704 tokens x 512 columns was 124 us of 232 with it).
__device__ __forceinline__ void erf_terms(float h, float& phi_cdf, float& e) {
  const float z = fabsf(h) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  e = __expf(-0.5f * h * h);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float half_erfc = 0.5f * poly * e;            // 0.5 (1 - erf(z))
  phi_cdf = h >= 0.f ? 1.f - half_erfc : half_erfc;   // Phi(h) = 0.5 (1 + erf(h / sqrt 2))
}
__device__ __forceinline__ float gelu_f(float h) {
  float cdf, e;
  erf_terms(h, cdf, e);
  return h * cdf;
}
__device__ __forceinline__ float dgelu_f(float h) {
  float cdf, e;
  erf_terms(h, cdf, e);
  return fmaf(h * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int MODE, bool OUT16>
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

  // epilogue.  Accumulator v of block (x, y) = row 8 (v / 4) + 4 lk + v % 4, column li: written from there a store is 2 or 4 bytes
  // per lane (and mode 2's gelu' operand a 2-byte load per element: fc2's data gradient at 200 704 tokens took 136 us against 46
  // for the plain product).  The ring is free now: every wave turns its 32 x 64 half tiles through its own 8.5 KB of LDS so that a
  // lane owns EIGHT consecutive columns of a row -- bias, pre-activation, gelu' operand and output move as 16-byte vectors, eight
  // lanes write one full 128-byte line of a bfloat16 row.
  __syncthreads();  // every wave has left the ring
  constexpr int TP = 68;  // floats per row of the turn-around tile (the two lane halves of a ds_write_b32 land 16 banks apart)
  float* const turn = reinterpret_cast<float*>(lds) + wave * (32 * TP);
  float* const yf = static_cast<float*>(g.y);
  unsigned short* const yh = static_cast<unsigned short*>(g.y);
  const int rl = lane >> 3, cg = lane & 7;  // reading side: row rl + 8 it of the half tile, columns 8 cg .. 8 cg + 7
  const int nb = n0 + 64 * wn + 8 * cg;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = (MODE != 2 && g.bias) ? g.bias[nb + j] : 0.f;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) turn[(8 * (v >> 2) + 4 * lk + (v & 3)) * TP + 32 * y + li] = acc[x][y][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = rl + 8 * it;
      const int m = m0 + 64 * wm + 32 * x + r;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(turn + r * TP + 8 * cg);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(turn + r * TP + 8 * cg + 4);
      if (m < M) {
        const int64_t o = (int64_t)m * N + nb;
        float h[8] = {lo.x + bv[0], lo.y + bv[1], lo.z + bv[2], lo.w + bv[3], hi.x + bv[4], hi.y + bv[5], hi.z + bv[6], hi.w + bv[7]};
        if constexpr (MODE == 1) {
          if (g.pre) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[j] = (unsigned)f2bf(h[2 * j]) | ((unsigned)f2bf(h[2 * j + 1]) << 16);
            *reinterpret_cast<u32x4*>(g.pre + o) = pk;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = gelu_f(h[j]);
        }
        if constexpr (MODE == 2) {
          const u32x4 ax = *reinterpret_cast<const u32x4*>(g.aux + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            h[2 * j] *= dgelu_f(__uint_as_float(ax[j] << 16));
            h[2 * j + 1] *= dgelu_f(__uint_as_float(ax[j] & 0xffff0000u));
          }
        }
        if constexpr (OUT16) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk[j] = (unsigned)f2bf(h[2 * j]) | ((unsigned)f2bf(h[2 * j + 1]) << 16);
          *reinterpret_cast<u32x4*>(yh + o) = pk;
        } else {
          *reinterpret_cast<f32x4*>(yf + o) = f32x4{h[0], h[1], h[2], h[3]};
          *reinterpret_cast<f32x4*>(yf + o + 4) = f32x4{h[4], h[5], h[6], h[7]};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this half tile are done before the next one overwrites it
  }
}

template <int MODE, bool OUT16>
int launch(const GemmX& g, hipStream_t st) {
  constexpr size_t lds = 2 * 3 * 128 * 32 * 2;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x_kernel<MODE, OUT16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL((gemm_bf16x_kernel<MODE, OUT16>), dim3((unsigned)(g.nMt * g.nNt)), dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int vitta_gemm_bf16x_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0 && M * K * 2 < (1ll << 31) && N * K * 2 < (1ll << 31) && M * N < (1ll << 40);
}

int vitta_gemm_nt_bf16x(const void* d_a, const void* d_b, const float* d_bias, const void* d_aux, void* d_y, void* d_pre, int64_t M,
                        int64_t N, int64_t K, int32_t mode, int32_t out_bf16, void* stream) {
  if (!d_a || !d_b || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_bf16x_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmX g{d_a, d_b, d_bias, d_y, static_cast<const unsigned short*>(d_aux), static_cast<unsigned short*>(d_pre), (int)M, (int)N, (int)K,
          (int)((M + 127) / 128), (int)(N / 128)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) return out_bf16 ? launch<0, true>(g, st) : launch<0, false>(g, st);
  if (mode == 1) return out_bf16 ? launch<1, true>(g, st) : launch<1, false>(g, st);
  return out_bf16 ? launch<2, true>(g, st) : launch<2, false>(g, st);
}

int vitta_gemm_nt_bf16x_f32(const void* d_a, const void* d_b, const float* d_bias, float* d_y, int64_t M, int64_t N, int64_t K,
                            void* stream) {
  return vitta_gemm_nt_bf16x(d_a, d_b, d_bias, nullptr, d_y, nullptr, M, N, K, 0, 0, stream);
}

}  // extern "C"
