// Weight gradients of the trunk's convolutions (SGD over all parameters, the reference's default optimizer:
// corpus/basics.py:547-560; call sites as conv.hip) on v_mfma_f32_32x32x2_f32:
//
//   dW[k][c][tap] += sum_p  A[c][src(p, tap)] * dY[k][p]        p over the N * Hg * Wg output positions
//
// with A = x, or relu(bn(x)) applied on load (the forward convolution's BatchNorm + ReLU prologue).  Both operands are
// channel-major planes, i.e. the REDUCTION axis (pixels) is the contiguous one: a 64 (c) x 64 (k) output tile of one tap
// walks slabs of 32 pixels, both slab tiles sit in LDS as [row][33] (odd pitch: the MFMA operand read of 32 rows at one
// pixel is conflict free).  Work = (tile, pixel slab) units, cut into equal contiguous ranges over the resident workgroups
// (stream-K, as conv_sk.hip); a range's share of a tile is ADDED to dW with atomics, so there is no partial-tile protocol.
// The slab loop follows conv_sk.hip's rules: buffer loads whose per-slab part is scalar, compile-time LDS ring, no
// vector-ALU address arithmetic besides one lane-offset update per slab on the gathered (3x3 / strided) operand, whose
// source offsets and tap validity come from two per-pixel tables built once per geometry by the caller.
#include <hip/hip_ext.h>

#include "conv_common.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

template <int V>
struct IC {
  static constexpr int value = V;
};

struct WgradK {
  vitta_wgrad_desc d;
  int64_t xP;   // pixels per channel row of x
  int P;        // output positions = pixels per channel row of dy
  int nslab;    // ceil(P / 32)
  int nCt, nKt; // 64-row tiles of C and K
  int G;
  float* partials;  // [G][2][64 * 64] or NULL (atomics)
  int tapd[VITTA_CONV_MAX_TAPS];  // source offset of the tap relative to the table entry (pixels)
};

template <bool GATHER, bool PRO>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK a) {
  constexpr int BR = 64, BP = 32, LP = 33, NTH = 256;  // rows per tile, pixels per slab, LDS pitch
  constexpr int KS = BP / 2, PD = 3, KSB = 10;
  constexpr int NA = GATHER ? 8 : 2, NB = 2, NI = NA + NB;  // staging loads per thread and slab
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                 // [3][BR][LP]
  float* Bs = lds + 3 * BR * LP;   // [3][BR][LP]
  float* pro = Bs + 3 * BR * LP;   // [2][C]

  const vitta_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int C = d.C, K = d.K;
  const int G = (int)gridDim.x, g = xcd_remap(blockIdx.x, G);
  const int T = d.ntaps * a.nCt * a.nKt;
  const int64_t U = (int64_t)T * a.nslab;
  const int64_t u0 = (int64_t)g * U / G, u1 = (int64_t)(g + 1) * U / G;
  const int n_units = (int)(u1 - u0);
  if (n_units <= 0) return;

  if (PRO) {
    for (int c = tid; c < C; c += NTH) {
      const float s = d.pro_bn[0][c] * rsqrtf(d.pro_bn[3][c] + d.pro_eps);
      pro[c] = s;
      pro[C + c] = d.pro_bn[1][c] - d.pro_bn[2][c] * s;
    }
    __syncthreads();
  }

  // ---- load side -----------------------------------------------------------------------------------------------------
  // tile L = (tap * nCt + ct) * nKt + kt; unit = L * nslab + slab
  int ld_L = (int)(u0 / a.nslab), ld_s = (int)(u0 - (int64_t)ld_L * a.nslab);
  int ld_left = n_units - 1;
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy), 0, (int)((int64_t)K * a.P * 4), 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  const int xrow = (int)(a.xP * 4), yrow = a.P * 4;
  // lane parts: vector loads = (row tid / 8 (+ 32 per item), pixels 4 (tid % 8) ..); gathered loads = (row tid / 32 (+ 8 per
  // item), pixel tid % 32)
  const int vrow = tid >> 3, vpx = (tid & 7) * 4;
  const int grow = tid >> 5, gpx = tid & 31;
  int voff_a = 0, voff_b = 0;
  int soff_a = 0, soff_b = 0;
  int ld_c0 = 0;  // first channel of the A tile whose loads are in flight (PRO)
  int ld_kt = ld_L % a.nKt, ld_ct = (ld_L / a.nKt) % a.nCt, ld_tap = ld_L / (a.nKt * a.nCt);
  auto load_begin = [&]() __attribute__((always_inline)) {
    const int kt = ld_kt, ct = ld_ct, tap = ld_tap;
    const int p0 = ld_s * BP;
    ld_c0 = ct * BR;
    soff_a = ct * BR * xrow;
    soff_b = kt * BR * yrow + p0 * 4;
    voff_b = (p0 + vpx < a.P) ? (vrow * yrow + vpx * 4) : OOB;
    if (GATHER) {
      const int p = p0 + gpx;
      const bool in = p < a.P;
      const int pp = in ? p : 0;
      const int off = d.src_off[pp], msk = d.src_mask[pp];
      voff_a = (in && ((msk >> tap) & 1)) ? (grow * xrow + (off + a.tapd[tap]) * 4) : OOB;
    } else {
      soff_a += p0 * 4;
      voff_a = (p0 + vpx < a.P) ? (vrow * xrow + vpx * 4) : OOB;
    }
    if (ld_left > 0) {
      --ld_left;
      if (++ld_s == a.nslab) {
        ld_s = 0;
        ++ld_L;
        if (++ld_kt == a.nKt) {
          ld_kt = 0;
          if (++ld_ct == a.nCt) {
            ld_ct = 0;
            ++ld_tap;
          }
        }
      }
    }
  };
  f32x4 ra4[GATHER ? 1 : NA];
  float ra1[GATHER ? NA : 1];
  f32x4 rb[NB];
  float ps[PRO ? NA : 1], pt[PRO ? NA : 1];
  bool a_ok = true;
  auto load_item = [&](auto u_) __attribute__((always_inline)) {
    constexpr int u = decltype(u_)::value;
    if constexpr (u < NA) {
      if constexpr (GATHER) ra1[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_a, soff_a + u * 8 * xrow, 0));
      else ra4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_a, soff_a + u * 32 * xrow, 0));
      if constexpr (PRO) {
        const int row = GATHER ? (grow + u * 8) : (vrow + u * 32);
        ps[u] = pro[ld_c0 + row];
        pt[u] = pro[C + ld_c0 + row];
        if (u == 0) a_ok = voff_a != OOB;
      }
    } else {
      rb[u - NA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, voff_b, soff_b + (u - NA) * 32 * yrow, 0));
    }
  };
  float* const st_av = As + vrow * LP + vpx;
  float* const st_ag = As + grow * LP + gpx;
  float* const st_b = Bs + vrow * LP + vpx;
  auto store_item = [&](auto buf_, auto u_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value, u = decltype(u_)::value;
    if constexpr (u < NA) {
      if constexpr (GATHER) {
        float v = ra1[u];
        if constexpr (PRO) v = a_ok ? fmaxf(fmaf(v, ps[u], pt[u]), 0.f) : 0.f;
        st_ag[buf * BR * LP + u * 8 * LP] = v;
      } else {
        f32x4 v = ra4[u];
        if constexpr (PRO) {
          const float s_ = a_ok ? ps[u] : 0.f, t_ = a_ok ? pt[u] : 0.f;
          v.x = fmaxf(fmaf(v.x, s_, t_), 0.f);
          v.y = fmaxf(fmaf(v.y, s_, t_), 0.f);
          v.z = fmaxf(fmaf(v.z, s_, t_), 0.f);
          v.w = fmaxf(fmaf(v.w, s_, t_), 0.f);
        }
        float* p = st_av + buf * BR * LP + u * 32 * LP;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      }
    } else {
      const f32x4 v = rb[u - NA];
      float* p = st_b + buf * BR * LP + (u - NA) * 32 * LP;
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
  };
  auto for_items = [&](auto lo_, auto hi_, auto&& fn) __attribute__((always_inline)) {
    constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
    if constexpr (lo + 0 < hi) fn(IC<lo + 0>{});
    if constexpr (lo + 1 < hi) fn(IC<lo + 1>{});
    if constexpr (lo + 2 < hi) fn(IC<lo + 2>{});
    if constexpr (lo + 3 < hi) fn(IC<lo + 3>{});
    if constexpr (lo + 4 < hi) fn(IC<lo + 4>{});
    if constexpr (lo + 5 < hi) fn(IC<lo + 5>{});
    if constexpr (lo + 6 < hi) fn(IC<lo + 6>{});
    if constexpr (lo + 7 < hi) fn(IC<lo + 7>{});
    if constexpr (lo + 8 < hi) fn(IC<lo + 8>{});
    if constexpr (lo + 9 < hi) fn(IC<lo + 9>{});
    static_assert(hi - lo <= 10, "items per group");
  };

  // ---- multiplying side ------------------------------------------------------------------------------------------
  float af[PD + 1], bf[PD + 1];
  const float* const rd_a = As + (wm * 32 + li) * LP + lk;
  const float* const rd_b = Bs + (wn * 32 + li) * LP + lk;
  auto read_ops = [&](auto buf_, auto ks_, auto slot_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value, ks = decltype(ks_)::value, slot = decltype(slot_)::value;
    af[slot] = rd_a[buf * BR * LP + 2 * ks];
    bf[slot] = rd_b[buf * BR * LP + 2 * ks];
  };
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  int cp_L = (int)(u0 / a.nslab), cp_s = (int)(u0 - (int64_t)cp_L * a.nslab);

  // this range's share of tile L: with a workspace, stored as a partial tile in register order (slot = segment number of
  // this workgroup; wgrad_reduce_kernel adds the partials of a tile into dW), else added to dW with atomics
  int seg = 0;
  auto flush = [&](int L) __attribute__((always_inline)) {
    if (a.partials) {
      f32x4* dst = reinterpret_cast<f32x4*>(a.partials + ((int64_t)g * 2 + seg) * (BR * BR));
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        dst[qd * NTH + tid] = v;
      }
      ++seg;
      return;
    }
    const int kt = L % a.nKt, r = L / a.nKt, ct = r % a.nCt, tap = r / a.nCt;
    const int k = kt * BR + wn * 32 + li;
    if (k >= K) return;
    float* row = d.grad_w + ((int64_t)k * C) * d.wtaps + d.wt[tap];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int c = ct * BR + wm * 32 + 8 * (v >> 2) + 4 * lk + (v & 3);
      if (c < C) atomicAdd(row + (int64_t)c * d.wtaps, acc[v]);
    }
  };

  // ---- pipeline (as conv_sk.hip) -----------------------------------------------------------------------------------------
  load_begin();
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { load_item(u); });
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { store_item(IC<0>{}, u); });
  load_begin();
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { load_item(u); });
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { store_item(IC<1>{}, u); });
  __syncthreads();
  read_ops(IC<0>{}, IC<0>{}, IC<0>{});
  read_ops(IC<0>{}, IC<1>{}, IC<1>{});
  read_ops(IC<0>{}, IC<2>{}, IC<2>{});

  auto kstep = [&](auto R0_, auto R1_, auto R2_, auto ks_) __attribute__((always_inline)) {
    constexpr int R0 = decltype(R0_)::value, R1 = decltype(R1_)::value, R2 = decltype(R2_)::value, ks = decltype(ks_)::value;
    if constexpr (ks + PD < KS) read_ops(IC<R0>{}, IC<ks + PD>{}, IC<(ks + PD) % (PD + 1)>{});
    else read_ops(IC<R1>{}, IC<ks + PD - KS>{}, IC<(ks + PD) % (PD + 1)>{});
    if constexpr (ks == KSB) __syncthreads();
    if constexpr (ks < KSB) {
      constexpr int lo = (ks * NI) / KSB, hi = ((ks + 1) * NI) / KSB;
      for_items(IC<lo>{}, IC<hi>{}, [&](auto u) { load_item(u); });
    } else {
      constexpr int lo = ((ks - KSB) * NI) / (KS - KSB), hi = ((ks - KSB + 1) * NI) / (KS - KSB);
      for_items(IC<lo>{}, IC<hi>{}, [&](auto u) { store_item(IC<R2>{}, u); });
    }
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks % (PD + 1)], bf[ks % (PD + 1)], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto slab = [&](auto R0, auto R1, auto R2) __attribute__((always_inline)) {
    load_begin();
    __builtin_amdgcn_sched_barrier(0);
    kstep(R0, R1, R2, IC<0>{}); kstep(R0, R1, R2, IC<1>{}); kstep(R0, R1, R2, IC<2>{}); kstep(R0, R1, R2, IC<3>{});
    kstep(R0, R1, R2, IC<4>{}); kstep(R0, R1, R2, IC<5>{}); kstep(R0, R1, R2, IC<6>{}); kstep(R0, R1, R2, IC<7>{});
    kstep(R0, R1, R2, IC<8>{}); kstep(R0, R1, R2, IC<9>{}); kstep(R0, R1, R2, IC<10>{}); kstep(R0, R1, R2, IC<11>{});
    kstep(R0, R1, R2, IC<12>{}); kstep(R0, R1, R2, IC<13>{}); kstep(R0, R1, R2, IC<14>{}); kstep(R0, R1, R2, IC<15>{});
  };
  int j = 0;
  auto after_slab = [&]() __attribute__((always_inline)) -> bool {
    ++cp_s;
    ++j;
    const bool tile_end = cp_s == a.nslab, range_end = j == n_units;
    if (__builtin_expect(tile_end || range_end, 0)) {
      flush(cp_L);
      if (!range_end) {
        ++cp_L;
        cp_s = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      }
    }
    return range_end;
  };
  for (;;) {
    slab(IC<0>{}, IC<1>{}, IC<2>{});
    if (after_slab()) break;
    slab(IC<1>{}, IC<2>{}, IC<0>{});
    if (after_slab()) break;
    slab(IC<2>{}, IC<0>{}, IC<1>{});
    if (after_slab()) break;
  }
}

// grid (tiles, 16): thread = one element of a 64 x 64 tile of one tap; sums the partial tiles of the workgroups whose
// ranges cover the tile and adds the result to dW (one writer per element, or gridDim.z atomic adds where a tile has
// hundreds of partials)
__device__ __forceinline__ void wgrad_reduce_tile(const WgradK& a, int L, int Z, int bz) {
  const vitta_wgrad_desc& d = a.d;
  const int G = a.G;
  const int T = d.ntaps * a.nCt * a.nKt;
  const int64_t U = (int64_t)T * a.nslab;
  const int e = blockIdx.y * 256 + threadIdx.x;  // element index in register order: ((qd * 256 + tid) * 4 + j)
  const int j = e & 3, t = (e >> 2) & 255, qd = e >> 10;
  const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int kt = L % a.nKt, r = L / a.nKt, ct = r % a.nCt, tap = r / a.nCt;
  const int k = kt * 64 + wn * 32 + li, c = ct * 64 + wm * 32 + 8 * qd + 4 * lk + j;
  const int g_first = (int)((((int64_t)L * a.nslab + 1) * G + U - 1) / U) - 1;
  const int g_last = (int)(((((int64_t)L + 1) * a.nslab) * G + U - 1) / U) - 1;
  float s = 0.f;
  // Z: tiles covered by many ranges (few tiles, long pixel walks) split their partial list Z ways
  for (int gg = g_first + bz; gg <= g_last; gg += Z) {
    const int first_tile = (int)(((int64_t)gg * U / G) / a.nslab);
    s += a.partials[((int64_t)gg * 2 + (L - first_tile)) * 4096 + e];
  }
  if (k < d.K && c < d.C) {
    float* dst = d.grad_w + ((int64_t)k * d.C + c) * d.wtaps + d.wt[tap];
    if (Z > 1) atomicAdd(dst, s);
    else *dst += s;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradK a) { wgrad_reduce_tile(a, blockIdx.x, gridDim.z, blockIdx.z); }

// the reductions of up to four weight gradients (the convolutions of one bottleneck) as ONE launch
constexpr int REDUCE_MAX = 4;
struct WgradReduceMulti {
  WgradK a[REDUCE_MAX];
  int n;
  int tile_end[REDUCE_MAX];  // running tile counts
  int z[REDUCE_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradReduceMulti m) {
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.tile_end[i]) ++i;
  if ((int)blockIdx.z >= m.z[i]) return;
  wgrad_reduce_tile(m.a[i], (int)blockIdx.x - (i ? m.tile_end[i - 1] : 0), m.z[i], blockIdx.z);
}

int fill(const vitta_wgrad_desc* h, WgradK& a) {
  if (!h || !h->x || !h->dy || !h->grad_w) return VITTA_ERR_INVALID_ARG;
  a.d = *h;
  const vitta_wgrad_desc& d = a.d;
  if (d.C <= 0 || d.K <= 0 || d.N <= 0 || d.ntaps < 1 || d.ntaps > VITTA_CONV_MAX_TAPS || d.wtaps < d.ntaps || d.sstride < 1)
    return VITTA_ERR_INVALID_ARG;
  a.xP = (int64_t)d.N * d.Hs * d.Ws;
  const int64_t P = (int64_t)d.N * d.Hg * d.Wg;
  if (P % 4 || a.xP % 4 || (int64_t)d.C * a.xP * 4 >= (1ll << 31) || (int64_t)d.K * P * 4 >= (1ll << 31)) return VITTA_ERR_UNSUPPORTED;
  if (d.C % 64 || d.K % 64) return VITTA_ERR_UNSUPPORTED;  // row tails would need the scalar offset range-checked
  a.P = (int)P;
  a.nslab = (int)((P + 31) / 32);
  a.nCt = (d.C + 63) / 64;
  a.nKt = (d.K + 63) / 64;
  const bool pointwise = d.ntaps == 1 && d.sstride == 1 && d.dh[0] == 0 && d.dw[0] == 0 && d.Hg == d.Hs && d.Wg == d.Ws;
  if (!pointwise && (!d.src_off || !d.src_mask)) return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_PRO_BN_RELU) && (d.C > PRO_MAX || !d.pro_bn[0] || !d.pro_bn[1] || !d.pro_bn[2] || !d.pro_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t) a.tapd[t] = t < d.ntaps ? d.dh[t] * d.Ws + d.dw[t] : 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  (void)hipGetLastError();
  const int64_t U = (int64_t)d.ntaps * a.nCt * a.nKt * a.nslab;
  a.G = (int)(U < 3 * cus ? U : 3 * cus);
  // partial tiles through the workspace when every workgroup's range touches at most two tiles (tiles <= workgroups)
  const int64_t T = (int64_t)d.ntaps * a.nCt * a.nKt;
  const size_t need = (size_t)a.G * 2 * 4096 * sizeof(float);
  a.partials = (d.workspace && (size_t)d.workspace_bytes >= need && T <= a.G) ? static_cast<float*>(d.workspace) : nullptr;
  return VITTA_OK;
}

int reduce_z(const WgradK& a) {
  const int T = a.d.ntaps * a.nCt * a.nKt, per_tile = (a.G + T - 1) / T;
  const int z = per_tile / 8;
  return z < 1 ? 1 : (z > 32 ? 32 : z);
}

template <bool GATHER, bool PRO>
int launch_one(const WgradK& a, hipStream_t st) {
  const size_t lds = sizeof(float) * (6 * 64 * 33 + (PRO ? 2 * a.d.C : 0));
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<GATHER, PRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  VITTA_LAUNCH((conv_wgrad_kernel<GATHER, PRO>), dim3((unsigned)a.G), dim3(256), lds, st, a);
  if (a.partials && !(a.d.flags & VITTA_WGRAD_DEFER_REDUCE)) {
    const int T = a.d.ntaps * a.nCt * a.nKt, z = reduce_z(a);
    VITTA_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)T, 16, (unsigned)z), dim3(256), 0, st, a);
  }
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_conv_wgrad_reduce_f32(const vitta_wgrad_desc* const* h_descs, int32_t n, void* stream) {
  if (!h_descs || n < 1 || n > REDUCE_MAX) return VITTA_ERR_INVALID_ARG;
  WgradReduceMulti m;
  m.n = 0;
  int tiles = 0, zmax = 1;
  for (int i = 0; i < n; ++i) {
    WgradK a;
    const int rc = fill(h_descs[i], a);
    if (rc != VITTA_OK) return rc;
    if (!a.partials) continue;  // that launch added its partial tiles with atomics: nothing left to do
    for (int j = 0; j < m.n; ++j)
      if (m.a[j].partials == a.partials) return VITTA_ERR_INVALID_ARG;  // deferred launches need a workspace each
    m.a[m.n] = a;
    tiles += a.d.ntaps * a.nCt * a.nKt;
    m.tile_end[m.n] = tiles;
    m.z[m.n] = reduce_z(a);
    zmax = m.z[m.n] > zmax ? m.z[m.n] : zmax;
    ++m.n;
  }
  if (!m.n) return VITTA_OK;
  for (int i = m.n; i < REDUCE_MAX; ++i) {
    m.tile_end[i] = tiles;
    m.z[i] = 0;
  }
  VITTA_LAUNCH(wgrad_reduce_multi_kernel, dim3((unsigned)tiles, 16, (unsigned)zmax), dim3(256), 0, static_cast<hipStream_t>(stream), m);
  return VITTA_OK;
}

int vitta_conv_wgrad_f32(const vitta_wgrad_desc* h_desc, void* stream) {
  WgradK a;
  const int rc = fill(h_desc, a);
  if (rc != VITTA_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const vitta_wgrad_desc& d = a.d;
  const bool pointwise = d.ntaps == 1 && d.sstride == 1 && d.dh[0] == 0 && d.dw[0] == 0 && d.Hg == d.Hs && d.Wg == d.Ws;
  const bool pro = d.flags & VITTA_CONV_PRO_BN_RELU;
  if (pointwise) return pro ? launch_one<false, true>(a, st) : launch_one<false, false>(a, st);
  return pro ? launch_one<true, true>(a, st) : launch_one<true, false>(a, st);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------
// Trainable weights change every step: their packed copies ([tap][C][K] forward, [tap][K][C] data gradient) are rebuilt by
// ONE launch over all convolutions (a table of {source, destinations, K, C, taps, first element}) instead of two small
// permute launches per convolution (159 launches, ~0.7 ms per SGD-all step).
// ------------------------------------------------------------------------------------------------------------------------
namespace {

// One workgroup = a 32 (output channels) x 32 (input channel, tap) tile of one convolution's [K][C * taps] parameter, moved
// through LDS so that the read (along c, tap) and both writes (forward pack along k; data-gradient pack along c within a
// tap plane) are row-wise: the element-per-lane form read the parameter with a stride of C * taps floats between lanes and
// ran at 1 TB/s (282 us for the 23.5 M weights of the trunk, twice per step).
__global__ __launch_bounds__(256) void conv_repack_kernel(const vitta_repack_entry* __restrict__ tab, int n) {
  __shared__ float tile[32][33];
  __shared__ int where[2];
  if (threadIdx.x < 64) {
    // which entry owns tile blockIdx.x: the first wave scans the entries' tile counts 64 at a time
    const int lane = threadIdx.x;
    int64_t base = 0;
    int found = n, local = 0;
    for (int e0 = 0; e0 < n && found == n; e0 += 64) {
      const int e = e0 + lane;
      int64_t cnt = e < n ? (int64_t)(tab[e].K / 32) * ((tab[e].C * tab[e].taps + 31) / 32) : 0;
      int64_t incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      const bool mine = e < n && (int64_t)blockIdx.x >= base + incl - cnt && (int64_t)blockIdx.x < base + incl;
      const unsigned long long hit = __ballot(mine);
      if (hit) {
        const int src = __ffsll((long long)hit) - 1;
        found = e0 + src;
        local = (int)((int64_t)blockIdx.x - (base + __shfl(incl - cnt, src, 64)));
      }
      base += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      where[0] = found;
      where[1] = local;
    }
  }
  __syncthreads();
  if (where[0] >= n) return;
  const vitta_repack_entry e = tab[where[0]];
  const int CT = e.C * e.taps, nct = (CT + 31) / 32;
  const int k0 = (where[1] / nct) * 32, ct0 = (where[1] % nct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kl = ty + 8 * r;
    tile[kl][tx] = ct0 + tx < CT ? e.src[(int64_t)(k0 + kl) * CT + ct0 + tx] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ctl = ty + 8 * r, ct = ct0 + ctl;
    if (ct < CT) {
      const int c = ct / e.taps, t = ct - c * e.taps;
      e.dst_fwd[((int64_t)t * e.C + c) * e.K + k0 + tx] = tile[tx][ctl];  // 32 consecutive k
    }
  }
  if (e.dst_bwd) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kl = ty + 8 * r, ct = ct0 + tx;
      if (ct < CT) {
        const int c = ct / e.taps, t = ct - c * e.taps;
        e.dst_bwd[((int64_t)t * e.K + k0 + kl) * e.C + c] = tile[kl][tx];  // lanes of one tap: consecutive c
      }
    }
  }
}

}  // namespace

extern "C" int vitta_conv_repack_f32(const vitta_repack_entry* d_table, int32_t n_entries, int64_t total_elements, void* stream) {
  if (!d_table || n_entries <= 0 || total_elements <= 0) return VITTA_ERR_INVALID_ARG;
  // tiles of 32 x 32 elements; an entry whose C * taps is not a multiple of 32 has one partial tile per 32 output channels
  // (K <= 2048: at most 64 of them), hence the slack -- surplus workgroups exit
  const int64_t tiles = (total_elements + 1023) / 1024 + (int64_t)n_entries * 64;
  VITTA_LAUNCH(conv_repack_kernel, dim3((unsigned)tiles), dim3(256), 0, static_cast<hipStream_t>(stream), d_table, n_entries);
  return VITTA_OK;
}
