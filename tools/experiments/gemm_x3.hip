// Dense product with fp32-GRADE arithmetic on the bf16 matrix pipe, both operands PRE-SPLIT in memory (Video Swin-B in fp32,
// BASELINE config 3; SURVEY A10: swin_transformer.py:30-35, 144, 165, 304-311):
//   y[m][n] = epi(sum_k a[m][k] b[n][k]),   a = a_hi + a_mid + a_lo,  b = b_hi + b_mid + b_lo  (three bfloat16 terms each, 24 mantissa bits),
// taken as the six partial products of weight >= 2^-16 (lo hi, hi lo, mid mid, mid hi, hi mid, hi hi; smallest first, fp32
// accumulation by v_mfma_f32_32x32x16_bf16) -- conv_b3.hip's arithmetic (<= 3e-7 of fp64 on the trunk's shapes).  gemm.hip's exact
// fp32 kernel stops at 80-100 TF on these shapes (v_mfma_f32_32x32x2_f32 at 64 x 64 tiles is LDS-bound); splitting A in registers
// (gemm_b3.hip) pays 44 vector instructions per fragment and measured no faster.  Here the PRODUCER of an activation writes the
// three terms once ("x3" layout below) and this kernel only moves bytes and issues MFMAs: six products per loaded fragment pair.
//
// x3 layout of a [rows][cols] fp32 tensor (cols % 16 == 0): [rows][cols / 16][3][16] bfloat16 -- the three terms of sixteen
// consecutive columns sit in 96 consecutive bytes (32 per term): a 16-wide k-step of a row is one contiguous 96-byte piece, a lane
// that owns eight columns writes three 16-byte vectors.  6 bytes per element.
//
// Tile 128 x 128, four waves of 64 x 64 (2 x 2 accumulators), k-steps of 16 through a THREE-stage LDS ring filled by LDS-DMA:
// a stage = [3 terms][128 rows][32 bytes] per operand = 12 KB, both operands 24 KB, the ring 72 KB -- TWO workgroups per CU (a
// 32-wide step is 144 KB: one workgroup per CU, one wave per SIMD, and everything around the MFMAs -- the LDS reads behind the
// barrier, the first stages, the epilogue -- ran exposed: 74 TF over config 3's shapes against 91 for gemm.hip).  A step of a wave
// = 12 operand reads and 24 MFMAs (768 cycles).
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::xcd_remap;

#ifndef X3_NPROD
#define X3_NPROD 6
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemmX3 {
  const void* a;      // x3 [M][K]
  const void* b;      // x3 [N][K]
  const float* bias;  // [N] or null
  void* y;            // [M][N] fp32, or x3 (OUT3)
  const float* aux;   // mode 2: [M][N] fp32 pre-activation
  float* pre;         // mode 1: [M][N] fp32 pre-activation out (or null)
  int M, N, K;
  int nMt, nNt;
};

__device__ __forceinline__ float gelu_f(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float h) {
  return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}

__device__ __forceinline__ unsigned pack2(float a, float b) {  // two round-to-nearest-even conversions, one v_cvt_pk_bf16_f32
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// the three terms of eight consecutive values (columns 8 j .. 8 j + 7 of a sixteen-column group) -> three 16-byte vectors, 32 bytes
// apart (dst = the hi term's half of the group)
__device__ __forceinline__ void store_x3(unsigned char* dst, const float (&h)[8]) {
  u32x4 t0, t1, t2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = h[2 * j], b = h[2 * j + 1];
    const unsigned p0 = pack2(a, b);
    const float ra = a - lo_f(p0), rb = b - hi_f(p0);
    const unsigned p1 = pack2(ra, rb);
    const unsigned p2 = pack2(ra - lo_f(p1), rb - hi_f(p1));
    t0[j] = p0;
    t1[j] = p1;
    t2[j] = p2;
  }
  *reinterpret_cast<u32x4*>(dst) = t0;
  *reinterpret_cast<u32x4*>(dst + 32) = t1;
  *reinterpret_cast<u32x4*>(dst + 64) = t2;
}

template <int MODE, bool OUT3>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmX3 g) {
  constexpr int BM = 128, BN = 128, BK = 16, NB = 3;
  constexpr int PLANE = BM * BK * 2;   // bytes of one term of one operand per stage (4 KB)
  constexpr int STAGE = 3 * PLANE;     // 12 KB per operand per stage
  constexpr int PER_STEP = 6;          // DMA instructions of a wave per step: (three terms) x (a, b), 32 rows x 32 bytes each
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;               // [NB][3 terms][128 rows][32 bytes]
  unsigned char* const Bb = lds + NB * STAGE;  // the same for the 128 output columns

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
  const int M = g.M, N = g.N, K = g.K;
  const int S = K / BK;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), 0, (int)((int64_t)M * K * 6), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.b), 0, (int)((int64_t)N * K * 6), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // requests: wave w loads rows 32 w .. 32 w + 31 of both operands, one instruction per term: lane -> row + lane / 2, the row's
  // 16-byte half lane % 2 (a lane pair fetches 32 contiguous bytes; the mapping lane -> row + lane % 32, half lane / 32, whose LDS
  // image is bank-conflict free for the fragment reads, measured 20 % SLOWER: the request side, not the LDS read, sets the pace);
  // global: row * 6 K + 32 p + 16 (lane % 2), the step = 96 bytes further
  const int rq = 32 * wave + (lane >> 1), ch = (lane & 1) * 16;
  const int voff_a = min(m0 + rq, M - 1) * K * 6 + ch;  // tail rows re-read the last valid one (never stored)
  const int voff_b = (n0 + rq) * K * 6 + ch;
  int q = 0;  // next step to request (clamped to the last one: the tail re-requests it into a stage nobody reads again)
  auto request = [&](int stage) __attribute__((always_inline)) {
    unsigned char* da = Ab + stage * STAGE + wave * 1024;
    unsigned char* db = Bb + stage * STAGE + wave * 1024;
#pragma unroll
    for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(da + p * PLANE), 16, voff_a + 32 * p, q * 96, 0, 0);
#pragma unroll
    for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(db + p * PLANE), 16, voff_b + 32 * p, q * 96, 0, 0);
    q += (q + 1 < S) ? 1 : 0;
  };

  int a_row[2], b_row[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    a_row[x] = 64 * wm + 32 * x + li;
    b_row[x] = 64 * wn + 32 * x + li;
  }
  auto frag = [&](const unsigned char* base, int row) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(base + row * 32 + lk * 16);
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
    // of step s - 1, which takes the request of step s + 2
    __builtin_amdgcn_sched_barrier(0);
    #ifndef X3_NODMA
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
#endif
    barrier();
#ifndef X3_NODMA
    request(st + 2 >= NB ? st + 2 - NB : st + 2);
#endif
    const unsigned char* as_ = Ab + st * STAGE;
    const unsigned char* bs_ = Bb + st * STAGE;
    {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          fa[x][p] = frag(as_ + p * PLANE, a_row[x]);
          fb[x][p] = frag(bs_ + p * PLANE, b_row[x]);
        }
      // (a term, b term): 0 = hi, 1 = mid, 2 = lo; smallest products first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int i = 6 - X3_NPROD; i < 6; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[x][PA[i]], fb[y][PB[i]], acc[x][y], 0, 0, 0);
    }
    st = st + 1 == NB ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's surplus requests must not land in the next workgroup's LDS

  // epilogue (as gemm_bf16x.hip): every wave turns its 32 x 64 half tiles through its own 8.5 KB of the free ring so that a lane owns
  // EIGHT consecutive columns of a row -- bias, pre-activation, gelu' operand and the output move as 16-byte vectors; an x3 output is
  // the lane's 48 contiguous bytes
  __syncthreads();  // every wave has left the ring
  constexpr int TP = 68;
  float* const turn = reinterpret_cast<float*>(lds) + wave * (32 * TP);
  float* const yf = static_cast<float*>(g.y);
  unsigned char* const y3 = static_cast<unsigned char*>(g.y);
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
            *reinterpret_cast<f32x4*>(g.pre + o) = f32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<f32x4*>(g.pre + o + 4) = f32x4{h[4], h[5], h[6], h[7]};
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = gelu_f(h[j]);
        }
        if constexpr (MODE == 2) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(g.aux + o), a1 = *reinterpret_cast<const f32x4*>(g.aux + o + 4);
          const float ax[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] *= dgelu_f(ax[j]);
        }
        if constexpr (OUT3) {
          store_x3(y3 + (o >> 4) * 96 + ((nb >> 3) & 1) * 16, h);  // group (m N + nb) / 16 (N % 16 == 0), half nb / 8 % 2
        } else {
          *reinterpret_cast<f32x4*>(yf + o) = f32x4{h[0], h[1], h[2], h[3]};
          *reinterpret_cast<f32x4*>(yf + o + 4) = f32x4{h[4], h[5], h[6], h[7]};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this half tile are done before the next one overwrites it
  }
}

template <int MODE, bool OUT3>
int launch(const GemmX3& g, hipStream_t st) {
  constexpr size_t lds = 2 * 3 * 3 * 128 * 16 * 2;  // 72 KB
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<MODE, OUT3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL((gemm_x3_kernel<MODE, OUT3>), dim3((unsigned)(g.nMt * g.nNt)), dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

// fp32 [rows][cols] -> x3: a thread owns eight consecutive columns
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + 8 * i));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + 8 * i + 4));
    const float h[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    store_x3(y + (i >> 1) * 96 + (i & 1) * 16, h);  // cols % 16 == 0: the pair (2 j, 2 j + 1) of eights is one group of a row
  }
}

}  // namespace

extern "C" {

int vitta_gemm_x3_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 16 == 0 && M * K * 6 < (1ll << 31) && N * K * 6 < (1ll << 31) && M * N < (1ll << 40);
}

int vitta_split3_f32(const float* d_x, void* d_y3, int64_t rows, int64_t cols, void* stream) {
  if (!d_x || !d_y3 || rows <= 0 || cols <= 0 || cols % 16) return VITTA_ERR_INVALID_ARG;
  const int64_t n8 = rows * cols / 8;
  const int64_t wg = (n8 + 255) / 256;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(wg < 16384 ? wg : 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), d_x,
                     static_cast<unsigned char*>(d_y3), n8);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

int vitta_gemm_nt_x3(const void* d_a3, const void* d_b3, const float* d_bias, const float* d_aux, void* d_y, float* d_pre, int64_t M,
                     int64_t N, int64_t K, int32_t mode, int32_t out_x3, void* stream) {
  if (!d_a3 || !d_b3 || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_x3_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmX3 g{d_a3, d_b3, d_bias, d_y, d_aux, d_pre, (int)M, (int)N, (int)K, (int)((M + 127) / 128), (int)(N / 128)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) return out_x3 ? launch<0, true>(g, st) : launch<0, false>(g, st);
  if (mode == 1) return out_x3 ? launch<1, true>(g, st) : launch<1, false>(g, st);
  return out_x3 ? launch<2, true>(g, st) : launch<2, false>(g, st);
}

}  // extern "C"
