// Persistent stream-K form of the implicit-GEMM convolution (same arithmetic, tiles, LDS layouts and epilogues as the
// tile-per-workgroup kernel of conv.hip; reference call sites: models/tanet_models/temporal_module.py:85-106,
// tanet.py:125-150).
//
// Why: with one workgroup per 64 x 64 output tile the trunk's launches have 784 / 832 / 1568 / 3136 workgroups for
// 768 resident slots (3 per CU x 256 CUs): a second, nearly empty round doubles the launch (61 TF at 784 tiles against
// 77 TF at 768, tools/debug/conv_quant_probe.py), and every tile pays its own prologue / epilogue latency (fitted from the
// asymptotic rates of 8- and 32-slab tiles: ~6 slabs' worth of idle matrix pipe per tile).  Here the work is the stream
// of K-SLABS (unit = one 32-channel x 1-tap slab of one tile), cut into as many equal contiguous ranges as there are
// resident slots:
//   * every workgroup multiplies ~U / G slabs, whatever the tile count: no tail round;
//   * a workgroup walks its range as ONE software pipeline -- the loads of slab j + 2 are in flight under the MFMAs of
//     slab j across tile boundaries, so a new tile's first slabs are already in LDS when the previous tile's epilogue ends;
//   * a range that starts / ends inside a tile leaves a partial accumulator tile in the workspace (at most two per
//     workgroup: its first and its last segment); the last workgroup to arrive at a tile (ticket counter) sums the partials
//     in workgroup order (deterministic) and runs the epilogue.  Ranges of launches with short K (< 4 slabs per tile) are
//     aligned to tile boundaries instead (no partials).
//
// The slab loop carries NO vector-ALU instruction besides the MFMAs (tools/ubench/mfma_loop_probe.hip: five integer VALU
// instructions per MFMA take the matrix pipe from 152 to 119 TF even with three waves per SIMD; the same loop with its
// addresses in immediates / scalar registers runs at 152 TF with operand reads, staging loads, LDS stores and the barrier):
//   * the LDS ring slots are compile-time (the loop body exists once per ring rotation): every ds_read / ds_write is a
//     lane-constant base register + immediate offset;
//   * global -> register staging uses BUFFER loads: the lane part of the address (pixel, row-in-slab; padding taps and tile
//     tails as an out-of-range offset that reads 0) is a register computed once per (tile, tap), everything that changes per
//     slab (channel slab, tap's weight block, output-channel tile) is the instruction's SCALAR offset;
//   * K is walked tap OUTER, channel slab INNER, so the lane register changes only every C / 32 slabs.
#include <hip/hip_ext.h>

#include "conv_epilogue.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

template <int V>
struct IC {
  static constexpr int value = V;
};

template <bool GATHER, bool PRO>
__global__ __launch_bounds__(256) void conv_sk_kernel(const ConvK a) {
  constexpr int BM = 64, BN = 64, BK = 32, NTH = 256;
  constexpr int A4 = BK * BM / 4 / NTH, A1 = BK * BM / NTH, B4 = BK * BN / 4 / NTH, RSTEP = NTH / BM;
  constexpr int NA = GATHER ? A1 : A4, NI = NA + B4;
  constexpr int KS = BK / 2, PD = 3, KSB = 10;
  static_assert(KS % (PD + 1) == 0 && KSB + PD < KS, "operand ring");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                 // [3][BK][BM]
  float* Bs = lds + 3 * BK * BM;   // [3][BK][BN]
  int* flag = reinterpret_cast<int*>(Bs + 3 * BK * BN);  // [4] ticket result
  float* red = Bs + 3 * BK * BN + 4;                     // [2][32][2] per-channel sums of the upper pixel half of a tile
  float* pro = red + 128;                                // [2][C] prologue BN scale / shift

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int flags = d.flags, C = d.C, K = d.K;
  const int nslab = (C / BK) * d.ntaps, T = a.nMt * a.nNt, G = (int)gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  const int64_t U = (int64_t)T * nslab;
  int64_t u0, u1;
  if (a.sk_aligned) {
    u0 = ((int64_t)g * T / G) * nslab;
    u1 = ((int64_t)(g + 1) * T / G) * nslab;
  } else {
    u0 = (int64_t)g * U / G;
    u1 = (int64_t)(g + 1) * U / G;
  }
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

  // ---- load side: runs two slabs ahead of the multiplying side, across tile boundaries --------------------------------
  // slab q of a tile = (tap q / ncs, channel slab q % ncs)
  const int HWs = d.Hs * d.Ws, ncs = C / BK;
  int ld_L = (int)(u0 / nslab), ld_q = (int)(u0 - (int64_t)ld_L * nslab);
  int ld_t = ld_q / ncs, ld_cs = ld_q - ld_t * ncs;
  int ld_left = n_units - 1;  // advances left: past the range's last slab the loads repeat it (never read)
  int ld_k0 = (ld_L % a.nNt) * BN;
  bool ld_dirty = true;       // the lane offset of the A loads must be recomputed (new tile or new tap)
  // buffer resources: x with its true size (an offset >= size reads 0: padding taps), weights unbounded
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, 0x7fffffff, 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  const int row_bytes = (int)(a.xP * 4);
  int voff_a = 0;  // lane part of the A address (bytes)
  int voff_b[B4];
#pragma unroll
  for (int u = 0; u < B4; ++u) {
    const int idx = tid + u * NTH;
    voff_b[u] = ((idx / (BN / 4)) * K + (idx % (BN / 4)) * 4) * 4;
  }
  auto lane_offset = [&]() __attribute__((always_inline)) {
    const int m0 = (ld_L / a.nNt) * BM;
    if (GATHER) {
      const int m = m0 + (tid % BM);
      const int hw = d.Hg * d.Wg, mm = m < a.Mtot ? m : 0;
      const int n = mm / hw, r = mm - n * hw, gi = r / d.Wg, gj = r - gi * d.Wg;
      const int tp = a.tap[ld_t];
      const int sh = gi * d.sstride + (int)(int8_t)(tp & 0xff), sw = gj * d.sstride + (int)(int8_t)((tp >> 8) & 0xff);
      const bool ok = m < a.Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws;
      voff_a = ok ? ((tid / BM) * row_bytes + (n * HWs + sh * d.Ws + sw) * 4) : OOB;
    } else {
      voff_a = (tid / (BM / 4)) * row_bytes + min(m0 + (tid % (BM / 4)) * 4, a.Mtot - 4) * 4;
    }
  };

  f32x4 ra4[2][GATHER ? 1 : A4];
  float ra1[2][GATHER ? A1 : 1];
  f32x4 rb[2][B4];
  float ps[2][PRO ? NA : 1], pt[2][PRO ? NA : 1];
  bool ra_ok[2] = {true, true};  // PRO + GATHER only: relu(bn(0)) of a padding tap is not 0
  int st_c0[2] = {0, 0};
  int soff_a = 0, soff_b = 0;  // scalar parts of the addresses of the loads in flight (bytes)

  // scalar (and, on a tile / tap change, lane) addresses of the next slab; then the load-side position advances
  auto load_begin = [&](auto S_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    if (ld_dirty) {
      lane_offset();
      ld_dirty = false;
    }
    const int c0 = ld_cs * BK;
    soff_a = c0 * row_bytes;
    soff_b = (((a.tap[ld_t] >> 16) * C + c0) * K + ld_k0) * 4;
    st_c0[S] = c0;
    if (PRO && GATHER) ra_ok[S] = voff_a != OOB;
    if (ld_left > 0) {
      --ld_left;
      if (++ld_cs == ncs) {
        ld_cs = 0;
        ld_dirty = GATHER;
        if (++ld_t == d.ntaps) {
          ld_t = 0;
          ++ld_L;
          ld_k0 = (ld_L % a.nNt) * BN;
          ld_dirty = true;
        }
      }
    }
  };
  auto load_item = [&](auto S_, auto u_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value, u = decltype(u_)::value;
    if constexpr (u < NA) {
      if constexpr (GATHER)
        ra1[S][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_a, soff_a + u * RSTEP * row_bytes, 0));
      else
        ra4[S][u] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_a, soff_a + u * (NTH / (BM / 4)) * row_bytes, 0));
      if constexpr (PRO) {
        const int kk = GATHER ? (tid / BM + u * RSTEP) : ((tid + u * NTH) / (BM / 4));
        ps[S][u] = pro[st_c0[S] + kk];
        pt[S][u] = pro[C + st_c0[S] + kk];
      }
    } else {
      rb[S][u - NA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff_b[u - NA], soff_b, 0));
    }
  };
  // LDS store bases of this thread (floats); ring slot and item offsets are immediates
  float* const st_a = As + (GATHER ? (tid / BM) * BM + (tid % BM) : (tid / (BM / 4)) * BM + (tid % (BM / 4)) * 4);
  float* const st_b = Bs + (tid / (BN / 4)) * BN + (tid % (BN / 4)) * 4;
  auto store_item = [&](auto S_, auto buf_, auto u_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value, buf = decltype(buf_)::value, u = decltype(u_)::value;
    if constexpr (u < NA) {
      if constexpr (GATHER) {
        float v = ra1[S][u];
        if constexpr (PRO) v = ra_ok[S] ? fmaxf(fmaf(v, ps[S][u], pt[S][u]), 0.f) : 0.f;
        st_a[buf * BK * BM + u * RSTEP * BM] = v;
      } else {
        f32x4 v = ra4[S][u];
        if constexpr (PRO) {
          const float s_ = ps[S][u], t_ = pt[S][u];
          v.x = fmaxf(fmaf(v.x, s_, t_), 0.f);
          v.y = fmaxf(fmaf(v.y, s_, t_), 0.f);
          v.z = fmaxf(fmaf(v.z, s_, t_), 0.f);
          v.w = fmaxf(fmaf(v.w, s_, t_), 0.f);
        }
        *reinterpret_cast<f32x4*>(st_a + buf * BK * BM + u * (NTH / (BM / 4)) * BM) = v;
      }
    } else {
      *reinterpret_cast<f32x4*>(st_b + buf * BK * BN + (u - NA) * (NTH / (BN / 4)) * BN) = rb[S][u - NA];
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
  const float* const rd_a = As + lk * BM + wm * 32 + li;
  const float* const rd_b = Bs + lk * BN + wn * 32 + li;
  auto read_ops = [&](auto buf_, auto ks_, auto slot_) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_)::value, ks = decltype(ks_)::value, slot = decltype(slot_)::value;
    af[slot] = rd_a[buf * BK * BM + 2 * ks * BM];
    bf[slot] = rd_b[buf * BK * BN + 2 * ks * BN];
  };
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;

  int cp_L = (int)(u0 / nslab), cp_q = (int)(u0 - (int64_t)(u0 / nslab) * nslab);  // tile / slab-in-tile being multiplied
  int seg_q0 = cp_q;             // where this workgroup's segment of the current tile started
  bool first_seg = true;

  // epilogue (conv_epilogue.h): constants of a tile's channel are loaded at the tile's start, used at its end
  TileEpilogue epi(a, red, wm, wn, li, lk);
  auto load_consts = [&](int L) __attribute__((always_inline)) { epi.load_consts(L); };
  auto epilogue = [&](int L) __attribute__((always_inline)) { epi.run(L, acc); };

  // ---- end of this workgroup's segment of tile L: whole tile -> epilogue; else partial + ticket (+ reduction) ----------
  constexpr int tile_bytes = BM * BN * 4;
  auto owner = [&](int64_t u) __attribute__((always_inline)) { return (int)(((u + 1) * G + U - 1) / U) - 1; };  // workgroup whose range holds unit u
  auto flush = [&](int L, bool whole) __attribute__((always_inline)) {
    if (!whole) {
      // Partials travel with WRITE-THROUGH (sc1) stores and are read back with sc1 loads; order: stores -> vmcnt(0) in
      // every wave -> barrier -> ticket (relaxed, agent scope), as in conv.hip's split-K.
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, G * 2 * tile_bytes, 0x00020000);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (qd * NTH + tid) * 16,
                                               (g * 2 + (first_seg ? 0 : 1)) * tile_bytes, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int g_first = owner((int64_t)L * nslab), g_last = owner((int64_t)(L + 1) * nslab - 1);
      if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(a.cnt + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned)(g_last - g_first);
        if (last) __hip_atomic_store(a.cnt + L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = last ? 1 : 0;
      }
      __syncthreads();
      const bool last = flag[0] != 0;
      if (!last) return;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      for (int gg = g_first; gg <= g_last; ++gg) {
        const int slot = ((int)(((int64_t)gg * U / G) / nslab) == L) ? 0 : 1;  // its first segment lies in this tile
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 pv = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (qd * NTH + tid) * 16, (gg * 2 + slot) * tile_bytes, 16));
          acc[4 * qd] += pv.x;
          acc[4 * qd + 1] += pv.y;
          acc[4 * qd + 2] += pv.z;
          acc[4 * qd + 3] += pv.w;
        }
      }
    }
    epilogue(L);
  };

  // ---- pipeline ------------------------------------------------------------------------------------------------------
  load_begin(IC<0>{});
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { load_item(IC<0>{}, u); });
  load_begin(IC<1>{});
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { load_item(IC<1>{}, u); });
  load_consts(cp_L);
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { store_item(IC<0>{}, IC<0>{}, u); });
  for_items(IC<0>{}, IC<NI>{}, [&](auto u) { store_item(IC<1>{}, IC<1>{}, u); });
  __syncthreads();
  read_ops(IC<0>{}, IC<0>{}, IC<0>{});
  read_ops(IC<0>{}, IC<1>{}, IC<1>{});
  read_ops(IC<0>{}, IC<2>{}, IC<2>{});

  // k-step ks of the slab in ring slot R0: [operand reads of k-step ks + PD] [barrier at KSB] [one or two staging items:
  // global loads of slab j + 2 before the barrier, its LDS stores into ring slot R2 after it] [MFMA]
  auto kstep = [&](auto R0_, auto R1_, auto R2_, auto ks_) __attribute__((always_inline)) {
    constexpr int R0 = decltype(R0_)::value, R1 = decltype(R1_)::value, R2 = decltype(R2_)::value, ks = decltype(ks_)::value;
    if constexpr (ks + PD < KS) read_ops(IC<R0>{}, IC<ks + PD>{}, IC<(ks + PD) % (PD + 1)>{});
    else read_ops(IC<R1>{}, IC<ks + PD - KS>{}, IC<(ks + PD) % (PD + 1)>{});
    if constexpr (ks == KSB) __syncthreads();
    if constexpr (ks < KSB) {
      constexpr int lo = (ks * NI) / KSB, hi = ((ks + 1) * NI) / KSB;
      for_items(IC<lo>{}, IC<hi>{}, [&](auto u) { load_item(IC<0>{}, u); });
    } else {
      constexpr int lo = ((ks - KSB) * NI) / (KS - KSB), hi = ((ks - KSB + 1) * NI) / (KS - KSB);
      for_items(IC<lo>{}, IC<hi>{}, [&](auto u) { store_item(IC<0>{}, IC<R2>{}, u); });
    }
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks % (PD + 1)], bf[ks % (PD + 1)], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto slab = [&](auto R0, auto R1, auto R2) __attribute__((always_inline)) {
    load_begin(IC<0>{});
    __builtin_amdgcn_sched_barrier(0);
    kstep(R0, R1, R2, IC<0>{}); kstep(R0, R1, R2, IC<1>{}); kstep(R0, R1, R2, IC<2>{}); kstep(R0, R1, R2, IC<3>{});
    kstep(R0, R1, R2, IC<4>{}); kstep(R0, R1, R2, IC<5>{}); kstep(R0, R1, R2, IC<6>{}); kstep(R0, R1, R2, IC<7>{});
    kstep(R0, R1, R2, IC<8>{}); kstep(R0, R1, R2, IC<9>{}); kstep(R0, R1, R2, IC<10>{}); kstep(R0, R1, R2, IC<11>{});
    kstep(R0, R1, R2, IC<12>{}); kstep(R0, R1, R2, IC<13>{}); kstep(R0, R1, R2, IC<14>{}); kstep(R0, R1, R2, IC<15>{});
  };

  // after every slab: end of tile / end of range bookkeeping; true when the range is done
  int j = 0;
  auto after_slab = [&]() __attribute__((always_inline)) -> bool {
    ++cp_q;
    ++j;
    const bool tile_end = cp_q == nslab, range_end = j == n_units;
    if (__builtin_expect(tile_end || range_end, 0)) {
      flush(cp_L, seg_q0 == 0 && tile_end);
      first_seg = false;
      if (!range_end) {
        ++cp_L;
        cp_q = 0;
        seg_q0 = 0;
        load_consts(cp_L);
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      }
    }
    return range_end;
  };
  // one ring rotation per trip, straight-line (the accumulator stays in its registers across the three slab bodies)
  for (;;) {
    slab(IC<0>{}, IC<1>{}, IC<2>{});
    if (after_slab()) break;
    slab(IC<1>{}, IC<2>{}, IC<0>{});
    if (after_slab()) break;
    slab(IC<2>{}, IC<0>{}, IC<1>{});
    if (after_slab()) break;
  }
}

template <bool GATHER, bool PRO>
int launch_one(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = sizeof(float) * (3 * 32 * 64 * 2 + 4 + 128 + (PRO ? 2 * a.d.C : 0));
  if (lds > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_sk_kernel<GATHER, PRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  const dim3 grid((unsigned)a.sk_G), block(256);
  (void)hipGetLastError();
  if (e0) hipExtLaunchKernelGGL((conv_sk_kernel<GATHER, PRO>), grid, block, lds, st, e0, e1, 0, a);
  else hipLaunchKernelGGL((conv_sk_kernel<GATHER, PRO>), grid, block, lds, st, a);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace

namespace vitta_conv {

int launch_stream_k(const ConvK& a, bool gather, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const bool pro = a.d.flags & VITTA_CONV_PRO_BN_RELU;
  if (gather) return pro ? launch_one<true, true>(a, st, e0, e1) : launch_one<true, false>(a, st, e0, e1);
  return pro ? launch_one<false, true>(a, st, e0, e1) : launch_one<false, false>(a, st, e0, e1);
}

}  // namespace vitta_conv
