// Dense layers of Video Swin-B: y[m][n] = epi(sum_k a[m][k] b[n][k]) -- the qkv / proj Linear of
// WindowAttention3D (models/videoswintransformer_models/swin_transformer.py:144, 165), Mlp.fc1 + GELU + fc2 (:30-35),
// PatchMerging.reduction (:304-311) and their data gradients (the same product against the transposed weight).
//
// Both operands are row-major with the REDUCTION axis contiguous (tokens x channels, nn.Linear's [out][in] weight), so
// a slab tile sits in LDS exactly as it sits in memory: [row][32 k + 4 pad].  A lane of the 32x32x2 fp32 MFMA needs
// (row = lane % 32, k = lane / 32) per k-step; since both operands may walk k in ANY common order, a lane takes the four
// k = 8 j + 4 (lane / 32) + {0..3} of its row with ONE 16-byte LDS read and feeds four k-steps from it.  Row stride 36
// floats: 9 (odd) sixteen-byte units, so the 16 lanes a ds_read_b128 services per cycle start in 16 different bank quads,
// and a global 16-byte load lands as one conflict-free ds_write_b128.
//
// Workgroup = 256 lanes = 2 x 2 waves over a BM x BN output tile: 64 x 64 (one 32 x 32 accumulator block per wave, 37 KB of
// LDS, four workgroups per CU), 64 x 128 or 128 x 128 (2 x 2 blocks per wave: 16 MFMAs per pair of 16-byte reads).  K is
// walked in slabs of 32 through two LDS stages; the global loads of slab s + 2 are in flight in registers while slab s is
// multiplied: one barrier per slab.  Tiles are numbered n-fastest and handed to XCDs in contiguous ranges (a token tile's
// rows are re-read by the n-tiles of one L2).  MEASURED on the 20 shapes of the Swin-B step (profiles/r2k_gemm_bench.json):
// the 64 x 64 tile wins everywhere -- at 1.6-6.6 GFLOP per launch four co-resident workgroups hiding each other's staging
// and epilogue latencies matter more than operand reuse, and ceil(tiles / 256 CUs) quantises the coarse tiles harder -- so
// choose_tile() returns it unless a cost model with the measured penalties says otherwise; the coarse tiles stay selectable.
//
// Epilogues (lane = output column, so every store instruction writes two full 128-byte lines):
//   mode 0  y = acc (+ bias[n])
//   mode 1  h = acc + bias[n];  (pre = h);  y = gelu(h)                 fc1 + nn.GELU (exact erf form)
//   mode 2  y = acc * gelu'(aux[m][n])                                   fc2's data gradient lands on fc1's output
#include <hip/hip_runtime.h>

#include "conv_common.h"

using vitta_conv::f32x16;
using vitta_conv::f32x4;
using vitta_conv::u32x4;
using vitta_conv::xcd_remap;

namespace {

constexpr int BK = 32, LS = 36, NTH = 256;

struct GemmArgs {
  const float* a;      // [M][K]
  const float* b;      // [N][K]
  const float* bias;   // [N] or null
  const float* aux;    // [M][N] (mode 2)
  float* y;            // [M][N]
  float* pre;          // [M][N] or null (mode 1)
  int M, N, K, mode;
  int nMt, nNt;
  // stream-K form (gemm_nt_sk_kernel): K slabs per workgroup, arrival counters [tiles] (zero at rest), partial tiles [2 * grid][64 * 64]
  int sk_per;
  unsigned* sk_cnt;
  float* sk_ws;
};

__device__ __forceinline__ float gelu_f(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float h) {
  return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}

// mode 2: the gelu' operand of this lane's outputs is requested BEFORE the K walk (its latency hides under the MFMAs; a
// load -> multiply -> store chain at the end of a short-K launch ran the epilogue at 1 TB/s)
template <int BM, int BN, bool AUX>
__device__ __forceinline__ void prefetch_aux(const GemmArgs& g, float (&auxv)[AUX ? BM / 64 : 1][AUX ? BN / 64 : 1][16], int m0, int n0,
                                             int wm, int wn, int li, int lk) {
  if constexpr (AUX) {
#pragma unroll
    for (int n = 0; n < BN / 64; ++n) {
      const int col = n0 + wn * (BN / 2) + n * 32 + li;
#pragma unroll
      for (int i = 0; i < BM / 64; ++i) {
        const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * lk;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = rbase + 8 * (v >> 2) + (v & 3);
          auxv[i][n][v] = (row < g.M && col < g.N) ? g.aux[(int64_t)row * g.N + col] : 0.f;
        }
      }
    }
  }
}

// register v of block (i, n): row 8 (v / 4) + 4 lk + v % 4, column li
template <int BM, int BN, bool AUX>
__device__ __forceinline__ void epilogue(const GemmArgs& g, const f32x16 (&acc)[BM / 64][BN / 64],
                                         const float (&auxv)[AUX ? BM / 64 : 1][AUX ? BN / 64 : 1][16], int m0, int n0, int wm, int wn,
                                         int li, int lk) {
  constexpr int MI = BM / 64, NI = BN / 64;
  const int mode = AUX ? 2 : g.mode;
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int col = n0 + wn * (BN / 2) + n * 32 + li;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * lk;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = rbase + 8 * (v >> 2) + (v & 3);
        if (row >= g.M) continue;
        const int64_t o = (int64_t)row * g.N + col;
        float h = acc[i][n][v] + bv;
        if (mode == 1) {
          if (g.pre) g.pre[o] = h;
          h = gelu_f(h);
        } else if constexpr (AUX) {
          h *= dgelu_f(auxv[i][n][v]);
        }
        g.y[o] = h;
      }
    }
  }
}

template <int BM, int BN, bool AUX>
__global__ __launch_bounds__(NTH) void gemm_nt_kernel(const GemmArgs g) {
  constexpr int MI = BM / 64, NI = BN / 64;                       // 32 x 32 blocks per wave (waves 2 x 2)
  constexpr int A4 = BM * BK / 4 / NTH, B4 = BN * BK / 4 / NTH;   // 16-byte staging loads per lane and slab
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const As = lds;                    // [2][BM][LS]
  float* const Bs = lds + 2 * BM * LS;      // [2][BN][LS]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
  const int K = g.K;

  // rows past M / N read zeros: the buffer's extent is the operand's true size
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)((int64_t)g.M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, (int)((int64_t)g.N * K * 4), 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  int voff_a[A4], voff_b[B4];
#pragma unroll
  for (int u = 0; u < A4; ++u) {
    const int r = m0 + (tid >> 3) + u * 32;
    voff_a[u] = r < g.M ? (r * K + (tid & 7) * 4) * 4 : OOB;
  }
#pragma unroll
  for (int u = 0; u < B4; ++u) {
    const int r = n0 + (tid >> 3) + u * 32;
    voff_b[u] = r < g.N ? (r * K + (tid & 7) * 4) * 4 : OOB;
  }
  f32x4 ra[A4], rb[B4];
  auto load_global = [&](int s) __attribute__((always_inline)) {
    const int so = s * BK * 4;
#pragma unroll
    for (int u = 0; u < A4; ++u) ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff_a[u], so, 0));
#pragma unroll
    for (int u = 0; u < B4; ++u) rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b[u], so, 0));
  };
  float* const st_a = As + (tid >> 3) * LS + (tid & 7) * 4;
  float* const st_b = Bs + (tid >> 3) * LS + (tid & 7) * 4;
  auto store_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A4; ++u) *reinterpret_cast<f32x4*>(st_a + buf * BM * LS + u * 32 * LS) = ra[u];
#pragma unroll
    for (int u = 0; u < B4; ++u) *reinterpret_cast<f32x4*>(st_b + buf * BN * LS + u * 32 * LS) = rb[u];
  };

  const float* const rd_a = As + (wm * (BM / 2) + li) * LS + 4 * lk;
  const float* const rd_b = Bs + (wn * (BN / 2) + li) * LS + 4 * lk;
  f32x4 fa[2][MI], fb[2][NI];
  auto read_frag = [&](int buf, int j, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(rd_a + buf * BM * LS + i * 32 * LS + 8 * j);
#pragma unroll
    for (int i = 0; i < NI; ++i) fb[set][i] = *reinterpret_cast<const f32x4*>(rd_b + buf * BN * LS + i * 32 * LS + 8 * j);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  float auxv[AUX ? MI : 1][AUX ? NI : 1][16];
  prefetch_aux<BM, BN, AUX>(g, auxv, m0, n0, wm, wn, li, lk);

  const int ns = K / BK;
  load_global(0);
  store_lds(0);
  if (ns > 1) load_global(1);
  __syncthreads();

  auto slab = [&](int s, int buf) __attribute__((always_inline)) {
    read_frag(buf, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) read_frag(buf, j + 1, (j + 1) & 1);
      if (j == 1 && s + 1 < ns) store_lds(buf ^ 1);   // slab s + 1: its stage was last read before the previous barrier
      if (j == 2 && s + 2 < ns) load_global(s + 2);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int n = 0; n < NI; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1][i][kk], fb[j & 1][n][kk], acc[i][n], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int s = 0; s < ns; s += 2) {
    slab(s, 0);
    if (s + 1 < ns) slab(s + 1, 1);
  }

  epilogue<BM, BN, AUX>(g, acc, auxv, m0, n0, wm, wn, li, lk);
}

// ---- stream-K form of the 64 x 64 kernel ------------------------------------------------------------------------------------
// The N = 512 products of stage 2 (proj, fc2, and the data gradients of qkv / proj / fc1: 18 of the 24 blocks) are 392 tiles: on
// 256 CUs 136 CUs hold two workgroups and 120 one, and the launch lasts two tile times for 1.53 tiles of work per CU.  Here the
// launch is a fixed number of workgroups (a multiple of the CU count) and the unit of work is a K slab of a tile: the tiles' slabs,
// tile after tile, are cut into equal contiguous ranges, so a workgroup walks the tail of one tile, whole tiles, the head of
// another.  A whole tile ends in the ordinary epilogue.  A partial tile is written through to the workspace (register order: every
// store instruction 4 KB contiguous), a ticket on the tile's counter follows, and the workgroup that arrives last adds the tile's
// partials IN RANGE ORDER (its own from registers: the sum does not depend on who was last), runs the epilogue and leaves the
// counter zero.  Nobody waits for anybody: no residency requirement.  Slot of workgroup c's partial: 2 c + (it starts at the
// tile's slab 0) -- a range has at most one partial segment that starts inside a tile (its first) and one that starts a tile (its last).
template <bool AUX>
__global__ __launch_bounds__(NTH) void gemm_nt_sk_kernel(const GemmArgs g) {
  constexpr int BM = 64, BN = 64, A4 = BM * BK / 4 / NTH, B4 = BN * BK / 4 / NTH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const As = lds;
  float* const Bs = lds + 2 * BM * LS;
  __shared__ int flag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int K = g.K, ns = K / BK;
  const int64_t total = (int64_t)g.nMt * g.nNt * ns;
  const int me = xcd_remap(blockIdx.x, gridDim.x);  // contiguous ranges per XCD: a token tile's rows stay in one L2
  int64_t it = (int64_t)me * g.sk_per;
  const int64_t end = it + g.sk_per < total ? it + g.sk_per : total;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)((int64_t)g.M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, (int)((int64_t)g.N * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(g.sk_ws, 0, (int)((int64_t)gridDim.x * 2 * BM * BN * 4), 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  float* const st_a = As + (tid >> 3) * LS + (tid & 7) * 4;
  float* const st_b = Bs + (tid >> 3) * LS + (tid & 7) * 4;
  const float* const rd_a = As + (wm * (BM / 2) + li) * LS + 4 * lk;
  const float* const rd_b = Bs + (wn * (BN / 2) + li) * LS + 4 * lk;

  while (it < end) {
    const int t = (int)(it / ns), s0 = (int)(it - (int64_t)t * ns);
    const int s1 = (int64_t)(ns - s0) < end - it ? ns : s0 + (int)(end - it);
    const int n = s1 - s0;
    it += n;
    const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
    int voff_a[A4], voff_b[B4];
#pragma unroll
    for (int u = 0; u < A4; ++u) {
      const int r = m0 + (tid >> 3) + u * 32;
      voff_a[u] = r < g.M ? (r * K + (tid & 7) * 4) * 4 : OOB;
    }
#pragma unroll
    for (int u = 0; u < B4; ++u) {
      const int r = n0 + (tid >> 3) + u * 32;
      voff_b[u] = r < g.N ? (r * K + (tid & 7) * 4) * 4 : OOB;
    }
    f32x4 ra[A4], rb[B4];
    auto load_global = [&](int s) __attribute__((always_inline)) {
      const int so = s * BK * 4;
#pragma unroll
      for (int u = 0; u < A4; ++u) ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff_a[u], so, 0));
#pragma unroll
      for (int u = 0; u < B4; ++u) rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b[u], so, 0));
    };
    auto store_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < A4; ++u) *reinterpret_cast<f32x4*>(st_a + buf * BM * LS + u * 32 * LS) = ra[u];
#pragma unroll
      for (int u = 0; u < B4; ++u) *reinterpret_cast<f32x4*>(st_b + buf * BN * LS + u * 32 * LS) = rb[u];
    };
    f32x4 fa[2], fb[2];
    auto read_frag = [&](int buf, int j, int set) __attribute__((always_inline)) {
      fa[set] = *reinterpret_cast<const f32x4*>(rd_a + buf * BM * LS + 8 * j);
      fb[set] = *reinterpret_cast<const f32x4*>(rd_b + buf * BN * LS + 8 * j);
    };
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;

    load_global(s0);
    store_lds(0);
    if (n > 1) load_global(s0 + 1);
    __syncthreads();
    auto slab = [&](int q, int buf) __attribute__((always_inline)) {
      read_frag(buf, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) read_frag(buf, j + 1, (j + 1) & 1);
        if (j == 1 && q + 1 < n) store_lds(buf ^ 1);
        if (j == 2 && q + 2 < n) load_global(s0 + q + 2);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1][kk], fb[j & 1][kk], acc, 0, 0, 0);
      }
      __syncthreads();
    };
    for (int q = 0; q < n; q += 2) {
      slab(q, 0);
      if (q + 1 < n) slab(q + 1, 1);
    }

    if (n < ns) {  // a partial tile: out through the workspace, ticket, the last arriver goes on
      const int first = (int)(((int64_t)t * ns) / g.sk_per), lastc = (int)(((int64_t)(t + 1) * ns - 1) / g.sk_per);
      const int slot = 2 * me + (s0 == 0 ? 1 : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_w, (q * NTH + tid) * 16, slot * (BM * BN * 4), 16);
      }
      // (the data registers of a 16-byte store are read a few cycles after it issues -- conv_b3.hip's split-K tail has the story:
      // the accumulator stays live across the stores and a few idle cycles)
      asm volatile("s_nop 7" ::: "memory");
      asm volatile("" ::"v"(acc));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(g.sk_cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned)(lastc - first);
        if (last) __hip_atomic_store(g.sk_cnt + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = last ? 1 : 0;
      }
      __syncthreads();
      const bool last = flag != 0;
      __syncthreads();  // (flag is rewritten by the next partial segment)
      if (!last) continue;
      f32x16 sum;
#pragma unroll
      for (int v = 0; v < 16; ++v) sum[v] = 0.f;
      for (int c = first; c <= lastc; ++c) {
        if (c == me) {
#pragma unroll
          for (int v = 0; v < 16; ++v) sum[v] += acc[v];
          continue;
        }
        const int so = (2 * c + (c == first ? 1 : 0)) * (BM * BN * 4);
        f32x4 pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (q * NTH + tid) * 16, so, 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sum[4 * q] += pv[q].x; sum[4 * q + 1] += pv[q].y; sum[4 * q + 2] += pv[q].z; sum[4 * q + 3] += pv[q].w;
        }
      }
      acc = sum;
    }
    float auxv[1][1][16];
    f32x16 accs[1][1] = {{acc}};
    prefetch_aux<BM, BN, AUX>(g, auxv, m0, n0, wm, wn, li, lk);
    epilogue<BM, BN, AUX>(g, accs, auxv, m0, n0, wm, wn, li, lk);
  }
}

// ---- bf16-operand variant (opt-in, BASELINE config 5's arithmetic: bf16 MFMA operands, fp32 accumulation) ----------------
// a stays fp32 in memory and is rounded to bf16 (nearest even) while it is staged; b is a bf16 copy of the weight the caller
// keeps (2 bytes / element: [N][K]).  v_mfma_f32_32x32x16_bf16: a lane supplies 8 consecutive k of its row = one 16-byte LDS
// read per operand block and k-step of 16.  Slabs of 64 k; rows of 64 bf16 + 8 pad = 144 bytes (the same odd multiple of 16
// bytes as the fp32 layout).  The epilogues are the fp32 kernel's.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKH = 64, LSH = 72;  // slab depth / row pitch in bf16 elements

template <int BM, int BN, bool AUX>
__global__ __launch_bounds__(NTH) void gemm_nt_bf16_kernel(const GemmArgs g) {
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int A4 = BM * BKH / 4 / NTH;   // fp32 16-byte loads per lane and slab (4 floats -> 4 bf16)
  constexpr int B8 = BN * BKH / 8 / NTH;   // bf16 16-byte loads per lane and slab
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __bf16* const As = reinterpret_cast<__bf16*>(lds);   // [2][BM][LSH]
  __bf16* const Bs = As + 2 * BM * LSH;                 // [2][BN][LSH]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / g.nNt) * BM, n0 = (t % g.nNt) * BN;
  const int K = g.K;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)((int64_t)g.M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, (int)((int64_t)g.N * K * 2), 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  int voff_a[A4], voff_b[B8];
#pragma unroll
  for (int u = 0; u < A4; ++u) {   // 16 lanes per row of 64 floats
    const int r = m0 + (tid >> 4) + u * 16;
    voff_a[u] = r < g.M ? (r * K + (tid & 15) * 4) * 4 : OOB;
  }
#pragma unroll
  for (int u = 0; u < B8; ++u) {   // 8 lanes per row of 64 bf16
    const int r = n0 + (tid >> 3) + u * 32;
    voff_b[u] = r < g.N ? (r * K + (tid & 7) * 8) * 2 : OOB;
  }
  f32x4 ra[A4];
  u32x4 rb[B8];
  auto load_global = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A4; ++u) ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff_a[u], s * BKH * 4, 0));
#pragma unroll
    for (int u = 0; u < B8; ++u) rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b[u], s * BKH * 2, 0);
  };
  __bf16* const st_a = As + (tid >> 4) * LSH + (tid & 15) * 4;
  __bf16* const st_b = Bs + (tid >> 3) * LSH + (tid & 7) * 8;
  auto store_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A4; ++u) *reinterpret_cast<bf16x4*>(st_a + buf * BM * LSH + u * 16 * LSH) = __builtin_convertvector(ra[u], bf16x4);
#pragma unroll
    for (int u = 0; u < B8; ++u) *reinterpret_cast<u32x4*>(st_b + buf * BN * LSH + u * 32 * LSH) = rb[u];
  };

  const __bf16* const rd_a = As + (wm * (BM / 2) + li) * LSH + 8 * lk;
  const __bf16* const rd_b = Bs + (wn * (BN / 2) + li) * LSH + 8 * lk;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  float auxv[AUX ? MI : 1][AUX ? NI : 1][16];
  prefetch_aux<BM, BN, AUX>(g, auxv, m0, n0, wm, wn, li, lk);

  const int ns = K / BKH;
  load_global(0);
  store_lds(0);
  if (ns > 1) load_global(1);
  __syncthreads();

  auto slab = [&](int s, int buf) __attribute__((always_inline)) {
    bf16x8 fa[4][MI], fb[4][NI];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[j][i] = *reinterpret_cast<const bf16x8*>(rd_a + buf * BM * LSH + i * 32 * LSH + 16 * j);
#pragma unroll
      for (int i = 0; i < NI; ++i) fb[j][i] = *reinterpret_cast<const bf16x8*>(rd_b + buf * BN * LSH + i * 32 * LSH + 16 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == 1 && s + 1 < ns) store_lds(buf ^ 1);
      if (j == 2 && s + 2 < ns) load_global(s + 2);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j][i], fb[j][n], acc[i][n], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int s = 0; s < ns; s += 2) {
    slab(s, 0);
    if (s + 1 < ns) slab(s + 1, 1);
  }
  epilogue<BM, BN, AUX>(g, acc, auxv, m0, n0, wm, wn, li, lk);
}

template <int BM, int BN, bool AUX>
int launch_one(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.nMt = (g.M + BM - 1) / BM;
  g.nNt = (g.N + BN - 1) / BN;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * LS;
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  VITTA_LAUNCH((gemm_nt_kernel<BM, BN, AUX>), dim3((unsigned)(g.nMt * g.nNt)), dim3(NTH), lds, st, g);
  return VITTA_OK;
}

template <int BM, int BN>
int launch(const GemmArgs& g, hipStream_t st) {
  return g.mode == 2 ? launch_one<BM, BN, true>(g, st) : launch_one<BM, BN, false>(g, st);
}

template <int BM, int BN, bool AUX>
int launch_one_bf16(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.nMt = (g.M + BM - 1) / BM;
  g.nNt = (g.N + BN - 1) / BN;
  const size_t lds = 2 * 2 * (BM + BN) * LSH;
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_kernel<BM, BN, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  VITTA_LAUNCH((gemm_nt_bf16_kernel<BM, BN, AUX>), dim3((unsigned)(g.nMt * g.nNt)), dim3(NTH), lds, st, g);
  return VITTA_OK;
}

template <int BM, int BN>
int launch_bf16(const GemmArgs& g, hipStream_t st) {
  return g.mode == 2 ? launch_one_bf16<BM, BN, true>(g, st) : launch_one_bf16<BM, BN, false>(g, st);
}

// Tile choice: the launch lasts as long as its busiest CU -- ceil(tiles / 256) tiles of MI x NI accumulator blocks each
// (workgroups sharing a CU share its matrix pipes) -- so the finest tile wins the quantisation and the coarsest the LDS /
// L2 traffic per flop; `pen` = measured relative cost of a block's work in each tile shape (profiles/r2h_gemm_bench.json: on
// every Swin-B shape the 64 x 64 tile -- four workgroups per CU hiding each other's staging -- is the fastest, 82-100 TF).
int choose_tile(int64_t M, int N, int mode) {
  const struct { int bm, bn; double pen; } cfg[3] = {{128, 128, 1.25}, {64, 128, 1.20}, {64, 64, 1.00}};
  int best = 1;
  double best_cost = 0;
  const int first = mode == 2 ? 1 : 0;  // the gelu' operand held in registers leaves the 128 x 128 tile one wave per SIMD
  for (int c = first; c < 3; ++c) {
    const int64_t tiles = ((M + cfg[c].bm - 1) / cfg[c].bm) * ((N + cfg[c].bn - 1) / cfg[c].bn);
    const double cost = (double)((tiles + 255) / 256) * (cfg[c].bm / 32) * (cfg[c].bn / 32) * cfg[c].pen;
    if (c == first || cost < best_cost) best = c + 1, best_cost = cost;
  }
  return best;
}

// bf16 operands: the matrix work of a tile is 1 / 16 of the fp32 kernel's and the launch is bound by what it moves and by
// how many workgroups hide each other's latencies: measured (profiles/r2i_gemm_bench.json) the 64 x 64 tile wins on 17 of
// the 20 Swin-B shapes, 64 x 128 on the short, wide ones (784 tokens x >= 2048 outputs) by 4-7 %
int choose_tile_bf16(int64_t M, int N, int mode) {
  (void)mode;
  return (M <= 1024 && N >= 2048) ? 2 : 3;
}

}  // namespace

extern "C" int vitta_gemm_nt_supported(int64_t M, int32_t N, int32_t K) {
  return M > 0 && N > 0 && K >= BK && K % BK == 0 && M * (int64_t)K * 4 < (int64_t)1 << 31 && (int64_t)N * K * 4 < (int64_t)1 << 31 &&
         M * (int64_t)N < (int64_t)1 << 40;
}

extern "C" int vitta_gemm_nt_f32(const float* d_a, const float* d_b, const float* d_bias, const float* d_aux, float* d_y,
                                 float* d_pre, int64_t M, int32_t N, int32_t K, int32_t mode, int32_t tile, void* stream) {
  if (!d_a || !d_b || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_nt_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  GemmArgs g{d_a, d_b, d_bias, d_aux, d_y, d_pre, (int)M, N, K, mode, 0, 0, 0, nullptr, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tile == 0) tile = choose_tile(M, N, mode);
  if (tile == 1) return launch<128, 128>(g, st);
  if (tile == 2) return launch<64, 128>(g, st);
  if (tile == 3) return launch<64, 64>(g, st);
  return VITTA_ERR_INVALID_ARG;
}

extern "C" int vitta_gemm_nt_bf16w_f32(const float* d_a, const uint16_t* d_b_bf16, const float* d_bias, const float* d_aux, float* d_y,
                                       float* d_pre, int64_t M, int32_t N, int32_t K, int32_t mode, int32_t tile, void* stream) {
  if (!d_a || !d_b_bf16 || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux)) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_nt_supported(M, N, K) || K % BKH) return VITTA_ERR_UNSUPPORTED;
  GemmArgs g{d_a, reinterpret_cast<const float*>(d_b_bf16), d_bias, d_aux, d_y, d_pre, (int)M, N, K, mode, 0, 0, 0, nullptr, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tile == 0) tile = choose_tile_bf16(M, N, mode);
  if (tile == 1) return launch_bf16<128, 128>(g, st);
  if (tile == 2) return launch_bf16<64, 128>(g, st);
  if (tile == 3) return launch_bf16<64, 64>(g, st);
  return VITTA_ERR_INVALID_ARG;
}

// stream-K form (64 x 64 tiles): workspace = [65536 arrival counters, ZERO when first used -- the kernel leaves them zero][2 * grid
// partial tiles of 16 KB], one per stream (launches that share it must not overlap).
constexpr int64_t SK_CNT_BYTES = 65536 * 4;
extern "C" int64_t vitta_gemm_nt_sk_workspace_bytes(int32_t grid) { return grid > 0 ? SK_CNT_BYTES + (int64_t)grid * 2 * 64 * 64 * 4 : 0; }

extern "C" int vitta_gemm_nt_sk_f32(const float* d_a, const float* d_b, const float* d_bias, const float* d_aux, float* d_y,
                                    float* d_pre, int64_t M, int32_t N, int32_t K, int32_t mode, int32_t grid, void* d_workspace,
                                    int64_t workspace_bytes, void* stream) {
  if (!d_a || !d_b || !d_y || mode < 0 || mode > 2 || (mode == 2 && !d_aux) || grid <= 0 || !d_workspace) return VITTA_ERR_INVALID_ARG;
  if (!vitta_gemm_nt_supported(M, N, K)) return VITTA_ERR_UNSUPPORTED;
  if (workspace_bytes < vitta_gemm_nt_sk_workspace_bytes(grid) || (reinterpret_cast<uintptr_t>(d_workspace) & 15u)) return VITTA_ERR_INVALID_ARG;
  GemmArgs g{d_a, d_b, d_bias, d_aux, d_y, d_pre, (int)M, N, K, mode, 0, 0, 0, nullptr, nullptr};
  g.nMt = (int)((M + 63) / 64);
  g.nNt = (N + 63) / 64;
  const int64_t tiles = (int64_t)g.nMt * g.nNt, total = tiles * (K / BK);
  if (tiles > 65536 || (int64_t)grid * 2 * 64 * 64 * 4 >= (1ll << 31)) return VITTA_ERR_UNSUPPORTED;
  g.sk_per = (int)((total + grid - 1) / grid);
  g.sk_cnt = static_cast<unsigned*>(d_workspace);
  g.sk_ws = reinterpret_cast<float*>(static_cast<char*>(d_workspace) + SK_CNT_BYTES);
  const size_t lds = sizeof(float) * 2 * (64 + 64) * LS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 2) VITTA_LAUNCH((gemm_nt_sk_kernel<true>), dim3((unsigned)grid), dim3(NTH), lds, st, g);
  else VITTA_LAUNCH((gemm_nt_sk_kernel<false>), dim3((unsigned)grid), dim3(NTH), lds, st, g);
  return VITTA_OK;
}
