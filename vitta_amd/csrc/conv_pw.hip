// Pointwise (1x1, stride 1) convolutions of the TANet trunk and their data gradients, one workgroup per 64 x 64 output tile
// (models/tanet_models/temporal_module.py:85-106 conv1 / conv3, tanet.py:125-150 downsample): the same implicit GEMM,
// LDS layouts, accumulator layout and epilogues (conv_epilogue.h) as the persistent stream-K kernel of conv_sk.hip, for the
// launches whose tiles all fit the chip at once.
//
// Why a second kernel: a pointwise launch of the trunk is 1.6-3.3 GFLOP = 10-20 us of matrix time, so what it costs is
// everything around the MFMAs.  The stream-K kernel pays per workgroup for two partial tiles (write-through, ticket,
// read-back) and holds 3 workgroups per CU (a three-slab LDS ring, 49 KB); with TWO LDS stages (32.5 KB) a CU holds four
// workgroups = 1024 slots, which is more than the 196-832 tiles of these launches: one round, no partial tiles, no
// workspace traffic, and four waves per SIMD hiding each other's staging and epilogue latencies.  Measured on the same
// GEMM shapes (tools/bench_gemm.py --tanet): 22-24 us against 27-30 us.  Launches with few tiles and long K (the
// 2048 -> 512 layer at 7 x 7: 104 tiles of 64 slabs) stay on the stream-K kernel.
#include <hip/hip_ext.h>

#include "conv_epilogue.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

// PRE: the launch's epilogue reads a residual / BatchNorm-backward input that can be requested before the K walk
template <bool PRE>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvK a) {
  constexpr int BM = 64, BN = 64, BK = 32, NTH = 256;
  constexpr int A4 = BK * BM / 4 / NTH, B4 = BK * BN / 4 / NTH;  // 16-byte staging loads per lane and slab (2 + 2)
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];    // [stage][channel of the slab][pixel]
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];    // [stage][channel of the slab][output channel]
  __shared__ float red[128];

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int C = d.C, K = d.K;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / a.nNt) * BM, k0 = (L % a.nNt) * BN;

  TileEpilogue epi(a, red, wm, wn, li, lk);

  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, 0x7fffffff, 0x00020000);
  const int row_bytes = (int)(a.xP * 4);
  // lane part of the addresses: slab row tid / 16 (+ 16 u), four consecutive pixels / output channels; the pixel tail of
  // the last tile re-reads valid pixels (its outputs are not stored)
  const int voff_a = (tid >> 4) * row_bytes + min(m0 + (tid & 15) * 4, a.Mtot - 4) * 4;
  const int voff_b = ((((a.tap[0] >> 16) * C + (tid >> 4)) * K) + k0 + (tid & 15) * 4) * 4;  // weight slot of the single tap
  f32x4 ra[A4], rb[B4];
  auto load_global = [&](int s) __attribute__((always_inline)) {
    const int c0 = s * BK;
#pragma unroll
    for (int u = 0; u < A4; ++u)
      ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_a, (c0 + u * 16) * row_bytes, 0));
#pragma unroll
    for (int u = 0; u < B4; ++u)
      rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff_b, (c0 + u * 16) * K * 4, 0));
  };
  auto store_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A4; ++u) *reinterpret_cast<f32x4*>(&As[buf][(tid >> 4) + u * 16][(tid & 15) * 4]) = ra[u];
#pragma unroll
    for (int u = 0; u < B4; ++u) *reinterpret_cast<f32x4*>(&Bs[buf][(tid >> 4) + u * 16][(tid & 15) * 4]) = rb[u];
  };

  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;

  const int ns = C / BK;
  load_global(0);
  epi.load_consts(L);
  float4 p0 = {}, p1 = {}, p2 = {}, p3 = {};
  if constexpr (PRE) tile_prefetch(a, L, wm, wn, li, lk, p0, p1, p2, p3);
  store_lds(0);
  if (ns > 1) load_global(1);
  __syncthreads();

  auto slab = [&](int s, int buf) __attribute__((always_inline)) {
    float af[16], bf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      af[ks] = As[buf][2 * ks + lk][wm * 32 + li];
      bf[ks] = Bs[buf][2 * ks + lk][wn * 32 + li];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks == 4 && s + 1 < ns) store_lds(buf ^ 1);  // slab s + 1: that stage was last read before the previous barrier
      if (ks == 8 && s + 2 < ns) load_global(s + 2);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks], bf[ks], acc, 0, 0, 0);
    }
    __syncthreads();
  };
  for (int s = 0; s < ns; s += 2) {
    slab(s, 0);
    if (s + 1 < ns) slab(s + 1, 1);
  }
  epi.template run<PRE>(L, acc, p0, p1, p2, p3);
}

}  // namespace

namespace vitta_conv {

int launch_pointwise(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid((unsigned)(a.nMt * a.nNt)), block(256);
  (void)hipGetLastError();
  if (a.pw_prefetch) {
    if (e0) hipExtLaunchKernelGGL(conv_pw_kernel<true>, grid, block, 0, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(conv_pw_kernel<true>, grid, block, 0, st, a);
  } else {
    if (e0) hipExtLaunchKernelGGL(conv_pw_kernel<false>, grid, block, 0, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(conv_pw_kernel<false>, grid, block, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace vitta_conv
