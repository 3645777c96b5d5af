// The 2D convolutions of the TANet trunk as one implicit-GEMM kernel family on v_mfma_f32_32x32x2_f32 (exact fp32).
// Reference call sites: models/tanet_models/temporal_module.py:85-106 (TemporalBottleneck: conv1x1 -> BN -> ReLU -> TAM ->
// conv3x3 -> BN -> ReLU -> conv1x1 -> BN -> +identity -> ReLU) over torchvision's ResNet-50 (tanet.py:125-150).
//
// Data layout: channel-major planes, tensor[c][p] with p = frame * H*W + h * W + w over ALL frames of the clip.
//   D[p][k] = sum_{tap, c} X[c][src(p, tap)] * Wp[tap][c][k]
// Pixels sit on the MFMA ROW axis (A operand), output channels on the COLUMN axis (B operand):
//   * both LDS tiles are [k-slab][pixels | channels] with the fast axis contiguous: an operand read is one ds_read_b32 of 32
//     consecutive floats per half-wave (conflict free), and fp32 MFMA (64 cycles per 32x32x2) leaves the LDS idle anyway;
//   * the 32x32 accumulator layout gives every lane ONE output channel (lane & 31) and 4 CONSECUTIVE pixels per register
//     quad: the epilogue reads / writes 16 bytes per lane, per-channel constants (BatchNorm scale / shift, statistics
//     shift, injection coefficients) are per-lane scalars, and per-channel reductions (hooked-layer moments, d gamma /
//     d beta) are in-register sums + one cross-half shuffle + one atomic per (wave, channel).
// K is walked input-channel slab OUTER, tap INNER: the nine shifted re-reads of a 3x3 convolution's input rows follow
// each other and hit the L1 / L2; each weight slab is read once.  Global -> LDS goes through registers (double-buffered
// LDS, the next slab's loads are in flight under the current slab's MFMAs, one barrier per slab).
// Workgroups are numbered so that the N-tiles of one M-tile run on the same XCD (shared A rows in one L2).
#include <hip/hip_ext.h>

#include <cstdlib>

#include "conv_common.h"
#include "conv_epilogue.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

template <int BM, int BN, int BK, int WM, int WN, bool GATHER, bool PRO>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvK a) {
  constexpr int NTH = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
  constexpr int A4 = BK * BM / 4 / NTH;  // float4 per thread (vector path)
  constexpr int A1 = BK * BM / NTH;      // floats per thread (gather path)
  constexpr int B4 = BK * BN / 4 / NTH;
  constexpr int RSTEP = NTH / BM;        // gather path: slab rows covered per pass
  static_assert(TM % 32 == 0 && TN % 32 == 0 && A4 >= 1 && B4 >= 1 && NTH % BM == 0 && BK % 2 == 0, "tile configuration");
  static_assert((BK * BM / 4) % NTH == 0 && (BK * BN / 4) % NTH == 0, "slab must split evenly over the threads");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                   // [3][BK][BM]  ring of slabs
  float* Bs = lds + 3 * BK * BM;     // [3][BK][BN]
  float* cst = Bs + 3 * BK * BN;     // per-output-channel epilogue constants [9][BN]
  float* pro = cst + 9 * BN;         // prologue BN scale / shift [2][C] (only with PRO_BN_RELU)

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  // logical id -> (tile, K slice): the slices of a tile and the N-tiles of an M-tile are neighbours, i.e. on one XCD
  const int Lz = xcd_remap(blockIdx.x, a.nMt * a.nNt * a.ksplit);
  const int L = Lz / a.ksplit, kz = Lz - L * a.ksplit;
  const int m0 = (L / a.nNt) * BM, k0 = (L % a.nNt) * BN;
  const int flags = d.flags;
  const int C = d.C, K = d.K;

  // ---- per-channel constants ------------------------------------------------------------------------------
  if (tid < BN) {
    const int k = k0 + tid;
    float es = 1.f, et = 0.f, sh = 0.f;
    if (d.epi_bn[0]) {
      es = d.epi_bn[0][k] * rsqrtf(d.epi_bn[3][k] + d.epi_eps);
      et = d.epi_bn[1][k] - d.epi_bn[2][k] * es;
    }
    if (d.st_shift) sh = d.st_shift[k];
    cst[0 * BN + tid] = es;
    cst[1 * BN + tid] = et;
    cst[2 * BN + tid] = sh;
    if (flags & VITTA_CONV_BWD_BN) {
      const float rstd = rsqrtf(d.bwd_bn[3][k] + d.bwd_eps);
      const float bs = d.bwd_bn[0][k] * rstd;
      cst[3 * BN + tid] = bs;
      cst[4 * BN + tid] = d.bwd_bn[1][k] - d.bwd_bn[2][k] * bs;
      cst[5 * BN + tid] = d.bwd_bn[2][k];
      cst[6 * BN + tid] = rstd;
      float gs = 0.f, ia = 0.f, ib = 0.f, mu = 0.f;
      if (d.inj_mu) {
        gs = d.inj_gscale ? d.inj_gscale[0] : 1.f;
        ia = gs * d.inj_a[k];
        ib = gs * d.inj_b[k];
        mu = d.inj_mu[k];
      }
      cst[7 * BN + tid] = ia;
      cst[8 * BN + tid] = ib;
      // mu rides in slot 2 (the forward's statistics shift is unused in backward launches)
      cst[2 * BN + tid] = mu;
    }
  }
  if (PRO) {
    for (int c = tid; c < C; c += NTH) {
      const float s = d.pro_bn[0][c] * rsqrtf(d.pro_bn[3][c] + d.pro_eps);
      pro[c] = s;
      pro[C + c] = d.pro_bn[1][c] - d.pro_bn[2][c] * s;
    }
  }

  // ---- staging geometry -----------------------------------------------------------------------------------
  // Every thread keeps the same pixel(s) and the same weight columns for the whole K walk; what changes per slab is the
  // channel row base (c0) and the tap.  Invalid elements (tile tail, padding taps) are loaded from a clamped address and
  // zeroed by a select: no branch around a load, all loads of a slab are in flight together.
  const int HWs = d.Hs * d.Ws;
  int g_n = 0, g_i = 0, g_j = 0;
  bool g_valid = false;
  if (GATHER) {
    const int m = m0 + (tid % BM);
    g_valid = m < a.Mtot;
    const int hw = d.Hg * d.Wg;
    const int mm = g_valid ? m : 0;
    g_n = mm / hw;
    const int r = mm - g_n * hw;
    g_i = r / d.Wg;
    g_j = r - g_i * d.Wg;
  }
  // per-thread element offsets (32-bit, in floats) from the uniform slab base: constant for the whole K walk
  unsigned a_off[GATHER ? 1 : A4];
  if (!GATHER) {
#pragma unroll
    for (int u = 0; u < A4; ++u) {
      const int idx = tid + u * NTH;
      const int kk = idx / (BM / 4), i4 = (idx % (BM / 4)) * 4;
      const int m = min(m0 + i4, a.Mtot - 4);
      a_off[u] = (unsigned)(kk * a.xP + m);
    }
  }
  unsigned b_off[B4];
#pragma unroll
  for (int u = 0; u < B4; ++u) {
    const int idx = tid + u * NTH;
    b_off[u] = (unsigned)((idx / (BN / 4)) * K + (idx % (BN / 4)) * 4);
  }

  // two staging register sets: both are in flight in the prologue (slabs 0 and 1), the steady state uses set 0
  constexpr int NA = GATHER ? A1 : A4;  // A-side staging items (one load / one LDS store each)
  constexpr int NI = NA + B4;           // staging items per slab
  f32x4 ra4[2][GATHER ? 1 : A4];
  float ra1[2][GATHER ? A1 : 1];
  f32x4 rb[2][B4];
  bool ra_valid[2] = {false, false};
  float ps[2][PRO ? NA : 1], pt[2][PRO ? NA : 1];  // prologue BN scale / shift of the rows held in each staging set
  int st_c0[2] = {0, 0};    // first channel of the slab held in each staging set
  const int nslab_all = (C / BK) * d.ntaps;
  const int q_first = (int)(((int64_t)nslab_all * kz) / a.ksplit), q_end = (int)(((int64_t)nslab_all * (kz + 1)) / a.ksplit);
  int ld_t = q_first % d.ntaps, ld_c0 = (q_first / d.ntaps) * BK;  // tap / first channel of the next slab to load (uniform)
  const float* ld_x = d.x;  // slab bases of the loads in flight
  const float* ld_w = d.w;

  // slab setup (uniform address arithmetic + the tap's validity), then one staging item per call
#define LOAD_BEGIN(S_)                                                                                           \
  do {                                                                                                           \
    const int tp = a.tap[ld_t];                                                                                  \
    if (GATHER) {                                                                                                \
      const int sh = g_i * d.sstride + (int)(int8_t)(tp & 0xff), sw = g_j * d.sstride + (int)(int8_t)((tp >> 8) & 0xff); \
      ra_valid[S_] = g_valid && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws;                   \
      const int64_t off = ra_valid[S_] ? ((int64_t)g_n * HWs + sh * d.Ws + sw) : 0;                               \
      ld_x = d.x + (int64_t)(ld_c0 + tid / BM) * a.xP + off;                                                      \
    } else {                                                                                                     \
      ld_x = d.x + (int64_t)ld_c0 * a.xP;                                                                        \
    }                                                                                                            \
    ld_w = d.w + ((int64_t)(tp >> 16) * C + ld_c0) * K + k0;                                                     \
    st_c0[S_] = ld_c0;                                                                                           \
    if (++ld_t == d.ntaps) {                                                                                     \
      ld_t = 0;                                                                                                  \
      ld_c0 += BK;                                                                                               \
    }                                                                                                            \
  } while (0)

#define LOAD_ITEM(S_, u)                                                                                         \
  do {                                                                                                           \
    if ((u) < NA) {                                                                                              \
      const int ua = (u) < NA ? (u) : 0;                                                                         \
      if (GATHER) ra1[S_][ua] = ld_x[(int64_t)ua * RSTEP * a.xP];                                                \
      else ra4[S_][ua] = *reinterpret_cast<const f32x4*>(ld_x + a_off[ua]);                                      \
      if (PRO) {                                                                                                 \
        const int kk = GATHER ? (tid / BM + ua * RSTEP) : ((tid + ua * NTH) / (BM / 4));                         \
        ps[S_][PRO ? ua : 0] = pro[st_c0[S_] + kk];                                                              \
        pt[S_][PRO ? ua : 0] = pro[C + st_c0[S_] + kk];                                                          \
      }                                                                                                          \
    } else {                                                                                                     \
      rb[S_][(u) >= NA ? (u) - NA : 0] = *reinterpret_cast<const f32x4*>(ld_w + b_off[(u) >= NA ? (u) - NA : 0]); \
    }                                                                                                            \
  } while (0)

#define STORE_ITEM(S_, buf, u)                                                                                   \
  do {                                                                                                           \
    float* as_ = As + (buf) * BK * BM;                                                                           \
    float* bs_ = Bs + (buf) * BK * BN;                                                                           \
    if ((u) < NA) {                                                                                              \
      constexpr int ua = (u) < NA ? (u) : 0;                                                                     \
      if (GATHER) {                                                                                              \
        const int kk = tid / BM + ua * RSTEP;                                                                    \
        float v = ra1[S_][ua];                                                                                   \
        if (PRO) v = fmaxf(fmaf(v, ps[S_][PRO ? ua : 0], pt[S_][PRO ? ua : 0]), 0.f);                             \
        as_[kk * BM + (tid % BM)] = ra_valid[S_] ? v : 0.f;                                                      \
      } else {                                                                                                   \
        const int idx = tid + ua * NTH;                                                                          \
        const int kk = idx / (BM / 4), i4 = (idx % (BM / 4)) * 4;                                                \
        f32x4 v = ra4[S_][ua];                                                                                   \
        if (PRO) {                                                                                               \
          const float s_ = ps[S_][PRO ? ua : 0], t_ = pt[S_][PRO ? ua : 0];                                      \
          v.x = fmaxf(fmaf(v.x, s_, t_), 0.f);                                                                   \
          v.y = fmaxf(fmaf(v.y, s_, t_), 0.f);                                                                   \
          v.z = fmaxf(fmaf(v.z, s_, t_), 0.f);                                                                   \
          v.w = fmaxf(fmaf(v.w, s_, t_), 0.f);                                                                   \
        }                                                                                                        \
        *reinterpret_cast<f32x4*>(as_ + kk * BM + i4) = v;                                                       \
      }                                                                                                          \
    } else {                                                                                                     \
      constexpr int ub = (u) >= NA ? (u) - NA : 0;                                                               \
      const int idx = tid + ub * NTH;                                                                            \
      *reinterpret_cast<f32x4*>(bs_ + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = rb[S_][ub];                 \
    }                                                                                                            \
  } while (0)

  // MFMA operand registers: a ring of PD + 1 k-steps (k-step s + PD is being read from LDS while k-step s multiplies)
  constexpr int KS = BK / 2;
  constexpr int PD = 3;
  constexpr int KSB = ((2 * KS) / 3 < KS - PD - 1) ? (2 * KS) / 3 : KS - PD - 1;  // k-step of the slab barrier
  static_assert(KS % (PD + 1) == 0 && KSB + PD < KS && KSB >= 1, "operand ring");
  float af[PD + 1][MT], bf[PD + 1][NT];

#define READ_OPS(buf, ks, slot)                                                                                  \
  do {                                                                                                           \
    const float* as_ = As + (buf) * BK * BM + (2 * (ks) + lk) * BM + wm * TM + li;                               \
    const float* bs_ = Bs + (buf) * BK * BN + (2 * (ks) + lk) * BN + wn * TN + li;                               \
    _Pragma("unroll") for (int x = 0; x < MT; ++x) af[slot][x] = as_[32 * x];                                     \
    _Pragma("unroll") for (int y = 0; y < NT; ++y) bf[slot][y] = bs_[32 * y];                                     \
  } while (0)

  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;

  // Pipeline (3 LDS slabs, ONE barrier per slab, placed INSIDE the MFMA stream):
  //   iteration q:  k-steps [0, KSB): MFMAs on slab q, the global loads of slab q+2 issued ONE PER K-STEP between them
  //                 barrier
  //                 k-steps [KSB, KS): MFMAs on slab q, the LDS stores of slab q+2 spread over them the same way
  // * the barrier of iteration q orders (WAR) every wave's reads of slab q-1 before the stores into its ring slot and
  //   (RAW) the stores of slab q+1 (made in iteration q-1) before any read of it: the last k-steps of iteration q already
  //   prefetch the first operands of slab q+1, so the MFMA stream runs across the slab boundary without a bubble;
  // * a wave reaches the barrier with operands for the next k-steps in registers and MFMAs queued: the skew between the
  //   four waves is covered;
  // * a vector-memory or LDS-store instruction holds the wave's issue for tens of cycles (1 KB per wave-instruction):
  //   bunched at the top of the slab they starve the matrix pipe (measured: 144 TF without them, 107 TF with), one
  //   per 64-cycle MFMA they disappear.
  const int nslab = q_end - q_first;
  if (PRO) __syncthreads();  // prologue constants visible before the first loads pick them up
  LOAD_BEGIN(0);
#pragma unroll
  for (int u = 0; u < NI; ++u) LOAD_ITEM(0, u);
  if (nslab > 1) {
    LOAD_BEGIN(1);
#pragma unroll
    for (int u = 0; u < NI; ++u) LOAD_ITEM(1, u);
  }
  __syncthreads();  // prologue constants visible before the first stores read them
#define STORE_ALL(S_, buf, u) STORE_ITEM(S_, buf, u)
  {
    // (constant item indices: the macro needs them at compile time)
#define ST8(S_, B_, o)                                                                     \
  do {                                                                                     \
    if ((o) + 0 < NI) STORE_ITEM(S_, B_, ((o) + 0 < NI ? (o) + 0 : 0));                     \
    if ((o) + 1 < NI) STORE_ITEM(S_, B_, ((o) + 1 < NI ? (o) + 1 : 0));                     \
    if ((o) + 2 < NI) STORE_ITEM(S_, B_, ((o) + 2 < NI ? (o) + 2 : 0));                     \
    if ((o) + 3 < NI) STORE_ITEM(S_, B_, ((o) + 3 < NI ? (o) + 3 : 0));                     \
    if ((o) + 4 < NI) STORE_ITEM(S_, B_, ((o) + 4 < NI ? (o) + 4 : 0));                     \
    if ((o) + 5 < NI) STORE_ITEM(S_, B_, ((o) + 5 < NI ? (o) + 5 : 0));                     \
    if ((o) + 6 < NI) STORE_ITEM(S_, B_, ((o) + 6 < NI ? (o) + 6 : 0));                     \
    if ((o) + 7 < NI) STORE_ITEM(S_, B_, ((o) + 7 < NI ? (o) + 7 : 0));                     \
  } while (0)
    static_assert(NI <= 24, "staging items");
    ST8(0, 0, 0);
    ST8(0, 0, 8);
    ST8(0, 0, 16);
    if (nslab > 1) {
      ST8(1, 1, 0);
      ST8(1, 1, 8);
      ST8(1, 1, 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < PD; ++s) READ_OPS(0, s, s);
  int r0 = 0, r1 = 1, r2 = 2;  // ring slots of slabs q, q + 1, q + 2
  int q = 0;
    // k-step ks: [operand reads of k-step ks + PD] [barrier at KSB] [staging items of this k-step] [MFMAs]
#define KSTEP(more, ks)                                                                                                \
  do {                                                                                                           \
    if ((ks) + PD < KS) READ_OPS(r0, ((ks) + PD < KS ? (ks) + PD : 0), ((ks) + PD) % (PD + 1));                   \
    else READ_OPS(r1, ((ks) + PD >= KS ? (ks) + PD - KS : 0), ((ks) + PD) % (PD + 1));                            \
    if ((ks) == KSB) __syncthreads();                                                                            \
    if (more) {                                                                                                  \
      if ((ks) < KSB) {                                                                                          \
        constexpr int lo = ((ks) * NI) / KSB, hi = (((ks) + 1) * NI) / KSB;                                      \
        if (lo + 0 < hi) LOAD_ITEM(0, (lo + 0 < NI ? lo + 0 : 0));                                               \
        if (lo + 1 < hi) LOAD_ITEM(0, (lo + 1 < NI ? lo + 1 : 0));                                               \
        if (lo + 2 < hi) LOAD_ITEM(0, (lo + 2 < NI ? lo + 2 : 0));                                               \
        if (lo + 3 < hi) LOAD_ITEM(0, (lo + 3 < NI ? lo + 3 : 0));                                               \
        if (lo + 4 < hi) LOAD_ITEM(0, (lo + 4 < NI ? lo + 4 : 0));                                               \
        if (lo + 5 < hi) LOAD_ITEM(0, (lo + 5 < NI ? lo + 5 : 0));                                               \
        static_assert(hi - lo <= 6, "loads per k-step");                                                         \
      } else {                                                                                                   \
        constexpr int lo = (((ks) - KSB) * NI) / (KS - KSB), hi = (((ks) - KSB + 1) * NI) / (KS - KSB);          \
        if (lo + 0 < hi) STORE_ITEM(0, r2, (lo + 0 < NI ? lo + 0 : 0));                                          \
        if (lo + 1 < hi) STORE_ITEM(0, r2, (lo + 1 < NI ? lo + 1 : 0));                                          \
        if (lo + 2 < hi) STORE_ITEM(0, r2, (lo + 2 < NI ? lo + 2 : 0));                                          \
        if (lo + 3 < hi) STORE_ITEM(0, r2, (lo + 3 < NI ? lo + 3 : 0));                                          \
        if (lo + 4 < hi) STORE_ITEM(0, r2, (lo + 4 < NI ? lo + 4 : 0));                                          \
        if (lo + 5 < hi) STORE_ITEM(0, r2, (lo + 5 < NI ? lo + 5 : 0));                                          \
        static_assert(hi - lo <= 6, "stores per k-step");                                                        \
      }                                                                                                          \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    _Pragma("unroll") for (int x = 0; x < MT; ++x)                                                                \
      _Pragma("unroll") for (int y = 0; y < NT; ++y)                                                              \
        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(ks) % (PD + 1)][x], bf[(ks) % (PD + 1)][y], acc[x][y], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)
#define SLAB(more)                                                                                      \
  do {                                                                                                  \
    if (more) LOAD_BEGIN(0);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    KSTEP(more, 0); KSTEP(more, 1); KSTEP(more, 2); KSTEP(more, 3);                                     \
    KSTEP(more, 4); KSTEP(more, 5); KSTEP(more, 6); KSTEP(more, 7);                                     \
    if constexpr (KS == 16) {                                                                           \
      KSTEP(more, 8); KSTEP(more, 9); KSTEP(more, 10); KSTEP(more, 11);                                 \
      KSTEP(more, 12); KSTEP(more, 13); KSTEP(more, 14); KSTEP(more, 15);                               \
    }                                                                                                   \
    const int t_ = r0;                                                                                  \
    r0 = r1;                                                                                            \
    r1 = r2;                                                                                            \
    r2 = t_;                                                                                            \
  } while (0)
  for (; q + 2 < nslab; ++q) SLAB(true);   // steady state: slab q + 2 is staged while slab q multiplies
  for (; q < nslab; ++q) SLAB(false);      // the last two slabs
#undef SLAB
#undef KSTEP
#undef LOAD_BEGIN
#undef LOAD_ITEM
#undef STORE_ITEM
#undef STORE_ALL
#undef ST8
#undef READ_OPS

  // ---- split-K: partial tiles meet in the last-arriving workgroup ---------------------------------------------------
  // Each slice stores its accumulators as a slab (16 bytes per lane, the register order IS the slab order, so the reducer
  // reads exactly its own registers' values), then draws a ticket; the workgroup that draws the last one sums the slabs in
  // slice order (deterministic) and runs the epilogue.  The counter is back at zero when the launch ends.
  if (a.ksplit > 1) {
    // Slabs travel with WRITE-THROUGH (sc1) stores and are read back with sc1 loads: no agent-scope release / acquire
    // (a release writes back every dirty line of the XCD's L2 -- with dozens of workgroups finishing per XCD each of
    // them paid for everybody's fresh slabs: 20 us per launch).  Order: slab stores -> vmcnt(0) in every wave -> barrier ->
    // ticket (relaxed, agent scope).  The ticket cannot overtake the slab: both leave through the same write-through path
    // after the wait.
    // one buffer descriptor over this tile's slabs (buffer instructions carry the sc1 bit as `aux`; the compiler counts
    // them like any other load / store)
    const int tile_bytes = BM * BN * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (int64_t)L * a.ksplit * (BM * BN), 0,
                                                                  a.ksplit * tile_bytes, 0x00020000);
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 v = {acc[x][y][4 * qd], acc[x][y][4 * qd + 1], acc[x][y][4 * qd + 2], acc[x][y][4 * qd + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                 (((x * NT + y) * 4 + qd) * NTH + tid) * 16, kz * tile_bytes, 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(a.cnt + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)(a.ksplit - 1);
      if (last) __hip_atomic_store(a.cnt + L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      As[0] = last ? 1.f : 0.f;  // the slab ring is idle now
    }
    __syncthreads();
    if (As[0] == 0.f) return;
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;
    for (int z = 0; z < a.ksplit; ++z) {
#pragma unroll
      for (int i = 0; i < MT * NT * 4; ++i) {
        const f32x4 pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * NTH + tid) * 16, z * tile_bytes, 16));
        const int xy = i / 4, qd = i % 4;
        acc[xy / NT][xy % NT][4 * qd] += pv.x;
        acc[xy / NT][xy % NT][4 * qd + 1] += pv.y;
        acc[xy / NT][xy % NT][4 * qd + 2] += pv.z;
        acc[xy / NT][xy % NT][4 * qd + 3] += pv.w;
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // register v of tile (x, y): pixel row = 32x + 8 (v / 4) + 4 lk + (v % 4), channel column = 32y + li
  const bool BWD = flags & VITTA_CONV_BWD_BN;
  const bool STATS = (flags & VITTA_CONV_STATS) && d.st_s1;
  const bool RAWST = flags & VITTA_CONV_STATS_RAW;
  const bool APPLY = flags & VITTA_CONV_EPI_APPLY;
  const bool RELU = flags & VITTA_CONV_EPI_RELU;
  const bool RES = (flags & VITTA_CONV_RES) && d.res;
  const bool RESH = (flags & VITTA_CONV_RES_HALF) && d.res;
  const bool BRELU = flags & VITTA_CONV_BWD_RELU;
  const bool IRAW = (flags & VITTA_CONV_INJ_RAW) && d.inj_mu;
  const bool POOL = (flags & VITTA_CONV_POOL) && d.pool;
  const int HWy = d.Hy * d.Wy;
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int jl = wn * TN + 32 * y + li;
    const int k = k0 + jl;
    const float es = cst[jl], et = cst[BN + jl], sh = cst[2 * BN + jl];
    float bsc = 0.f, bt = 0.f, brm = 0.f, brs = 0.f, ia = 0.f, ib = 0.f;
    if (BWD) {
      bsc = cst[3 * BN + jl];
      bt = cst[4 * BN + jl];
      brm = cst[5 * BN + jl];
      brs = cst[6 * BN + jl];
      ia = cst[7 * BN + jl];
      ib = cst[8 * BN + jl];
    }
    float r1 = 0.f, r2 = 0.f;  // statistics (forward) or d gamma / d beta (backward) partial sums of this lane
    const int64_t yrow = (int64_t)k * a.yP;
#pragma unroll
    for (int x = 0; x < MT; ++x) {
      PoolSums pool(a, m0 + wm * TM + 32 * x, POOL);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int m = m0 + wm * TM + 32 * x + 8 * qd + 4 * lk;
        if (m >= a.Mtot) continue;
        float v[4] = {acc[x][y][4 * qd], acc[x][y][4 * qd + 1], acc[x][y][4 * qd + 2], acc[x][y][4 * qd + 3]};
        if (a.contig) {
          float* yp = d.y + yrow + m;
          if (RES && BWD) {
            const float4 r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
            if (BWD) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
          }
          if (RESH) {
            const int Hh = (d.Hy + 1) >> 1, Wh = (d.Wy + 1) >> 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int p = m + e;
              const int n = p / HWy, r = p - n * HWy, h = r / d.Wy, w = r - h * d.Wy;
              if (!((h | w) & 1)) v[e] += d.res[(int64_t)k * a.rP + (int64_t)n * Hh * Wh + (h >> 1) * Wh + (w >> 1)];
            }
          }
          if (BWD) {
            const float4 xr = *reinterpret_cast<const float4*>(d.bwd_x + (int64_t)k * a.yP + m);
            const float xv[4] = {xr.x, xr.y, xr.z, xr.w};
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (BRELU && d.bwd_mask) {
              const float4 mr = *reinterpret_cast<const float4*>(d.bwd_mask + (int64_t)k * a.yP + m);
              mk[0] = mr.x > 0.f; mk[1] = mr.y > 0.f; mk[2] = mr.z > 0.f; mk[3] = mr.w > 0.f;
            }
            float o[4], gm[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaf(xv[e], bsc, bt);
              const float mm = (BRELU && !d.bwd_mask) ? (z > 0.f ? 1.f : 0.f) : mk[e];
              gm[e] = v[e] * mm;
              const float dz = IRAW ? gm[e] : gm[e] + fmaf(ib, z - sh, ia);
              r1 += dz * (xv[e] - brm) * brs;
              r2 += dz;
              o[e] = IRAW ? fmaf(dz, bsc, fmaf(ib, xv[e] - sh, ia)) : dz * bsc;
            }
            *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
            if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(gm[0], gm[1], gm[2], gm[3]);
          } else {
            if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(v[0], v[1], v[2], v[3]);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaf(v[e], es, et);
              if (STATS) {
                const float dd = (RAWST ? v[e] : z) - sh;
                r1 += dd;
                r2 = fmaf(dd, dd, r2);
              }
              if (POOL) pool.add(m + e, z);
              o[e] = APPLY ? z : v[e];
            }
            if (RES) {
              const float4 r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
              o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
            }
            if (RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? 0.f : o[e];  // (keeps NaN, as torch's relu)
            }
            *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
          }
        } else {
          // scattered destination (data gradient of a stride-2 convolution, one parity class per launch): plain values
          const int hwg = d.Hg * d.Wg;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = m + e;
            const int n = p / hwg, r = p - n * hwg, gi = r / d.Wg, gj = r - gi * d.Wg;
            const int h = gi * d.ostride + d.oa, w = gj * d.ostride + d.ob;
            if (h < d.Hy && w < d.Wy) d.y[yrow + (int64_t)n * HWy + h * d.Wy + w] = v[e];
          }
        }
      }
      if (POOL) pool.flush(a, k, lk);
    }
    if (STATS || BWD) {
      r1 += __shfl_xor(r1, 32, 64);
      r2 += __shfl_xor(r2, 32, 64);
      if (lk == 0) {
        if (BWD) {
          if (d.dgamma) atomicAdd(d.dgamma + k, r1);
          if (d.dbeta) atomicAdd(d.dbeta + k, r2);
        } else {
          atomicAdd(d.st_s1 + k, r1);
          atomicAdd(d.st_s2 + k, r2);
        }
      }
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const ConvK& a, bool gather, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = sizeof(float) * (3 * BK * BM + 3 * BK * BN + 9 * BN +
                                      ((a.d.flags & VITTA_CONV_PRO_BN_RELU) ? 2 * a.d.C : 0));
  const dim3 grid((unsigned)(a.nMt * a.nNt * a.ksplit)), block(WM * WN * 64);
  if (lds > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  const bool pro = a.d.flags & VITTA_CONV_PRO_BN_RELU;
  const int v = (gather ? 2 : 0) | (pro ? 1 : 0);
  const void* fns[4] = {reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, false, false>),
                        reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, false, true>),
                        reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, true, false>),
                        reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, true, true>)};
  if (lds > 48 * 1024) {  // raise the dynamic-LDS ceiling of this instantiation once (not a stream operation)
    static bool raised[4] = {false, false, false, false};
    if (!raised[v]) {
      if (hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return VITTA_ERR_LAUNCH;
      raised[v] = true;
    }
  }
  if (e0) {  // events attached to this dispatch (bench.py's live per-kernel timing)
    (void)hipGetLastError();
    switch (v) {
      case 0: hipExtLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WM, WN, false, false>), grid, block, lds, st, e0, e1, 0, a); break;
      case 1: hipExtLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WM, WN, false, true>), grid, block, lds, st, e0, e1, 0, a); break;
      case 2: hipExtLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WM, WN, true, false>), grid, block, lds, st, e0, e1, 0, a); break;
      default: hipExtLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WM, WN, true, true>), grid, block, lds, st, e0, e1, 0, a); break;
    }
    return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
  }
  switch (v) {
    case 0: VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, false, false>), grid, block, lds, st, a); break;
    case 1: VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, false, true>), grid, block, lds, st, a); break;
    case 2: VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, true, false>), grid, block, lds, st, a); break;
    default: VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, true, true>), grid, block, lds, st, a); break;
  }
  return VITTA_OK;
}

bool is_vector_geometry(const vitta_conv_desc& d) {
  return d.ntaps == 1 && d.sstride == 1 && d.ostride == 1 && d.dh[0] == 0 && d.dw[0] == 0 && d.Hg == d.Hs && d.Wg == d.Ws &&
         d.Hy == d.Hg && d.Wy == d.Wg;
}

// conv_b3.hip's patch form: taps are shifts inside one flat pixel range of the source planes -- the source grid is the
// output grid (stride 1) and every shift dh * Ws + dw stays inside the halo (63 pixels: 128 x 64 tiles, 32: 128 x 128)
bool b3_patch_geometry(const vitta_conv_desc& d, int halo) {
  if (d.sstride != 1 || d.Hg != d.Hs || d.Wg != d.Ws) return false;
  for (int t = 0; t < d.ntaps; ++t) {
    const int sh = d.dh[t] * d.Ws + d.dw[t];
    if (sh < -halo || sh > halo) return false;
  }
  return true;
}

// Tile choice: the largest tile that still gives the launch about two workgroups per CU; small problems fall through to
// 64 x 64 / 64 x 32 (more, shorter workgroups).  BN must divide K.
// The counters sit in a FIXED prefix of the workspace: launches with different tile counts share one workspace, and a
// slab of one must never land on a counter of another (counters are zero at rest, slabs are not).
constexpr int MAX_SPLIT_TILES = 16384;
size_t counter_bytes(int) { return (size_t)MAX_SPLIT_TILES * sizeof(unsigned); }

// workgroups of the 64 x 64 x 32 configuration the chip holds at once: 3 per CU (LDS: 3 x 49 KB)
int resident_slots() {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 256;
    (void)hipGetLastError();
    slots = 3 * cus;
  }
  return slots;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

// VITTA_CONV_B3=0: exact-fp32 MFMA kernels even where a split-bf16 weight image is supplied (A/B measurements)
bool b3_enabled() {
  static const int on = env_int("VITTA_CONV_B3", 1);
  return on != 0;
}

// VITTA_CONV_STREAM_K=0 keeps the tile-per-workgroup kernels (A/B measurements)
bool sk_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VITTA_CONV_STREAM_K");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

// VITTA_CONV_PW=0 keeps the pointwise launches on the stream-K kernel (A/B measurements)
bool pw_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VITTA_CONV_PW");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

// smallest tile count / longest K walk (slabs per tile) conv_pw.hip takes.  Measured with bench.py: 96-192 tiles and 32-64
// slabs are within noise of each other (148.9-150.3 videos/s); a gathered (3x3) variant of the kernel was measured and
// dropped (64-channel layer 46.8 vs 47.6 us, 128-channel layer 49.2 vs 47.4 us: the stream-K kernel keeps those), and so
// was a shifted-load form of it for the stride-1 3x3 layers (one 16-byte load + four v_cndmask per four pixels instead of four
// gathered loads: 49.7 vs 47.3 us) -- with 784 / 392 / 196 tiles on 256 CUs a tile-per-workgroup launch is bound by
// ceil(tiles / CUs) (77 % at best), which is exactly what the stream-K form removes at the price of its partial tiles.
int pw_min_tiles() {
  static int v = env_int("VITTA_CONV_PW_MIN_TILES", 192);
  return v;
}
int pw_max_slabs() {
  static int v = env_int("VITTA_CONV_PW_MAX_SLABS", 32);
  return v;
}

void choose_tile(const vitta_conv_desc& d, int64_t M, int& bm, int& bn) {
  if (d.tile) {
    bm = d.tile >> 16;
    bn = d.tile & 0xffff;
    return;
  }
  // measured on the TANet shapes (tools/bench_conv.py --tiles): 64 x 64 with 3-4 workgroups per CU beats the larger
  // tiles everywhere below ~3000 workgroups; K % 64 != 0 falls to 64 x 32
  (void)M;
  bm = 64;
  bn = (d.K % 64 == 0) ? 64 : 32;
}

int fill(const vitta_conv_desc* h, ConvK& a) {
  if (!h || !h->x || (!h->w && !h->w_b3) || !h->y) return VITTA_ERR_INVALID_ARG;
  a.d = *h;
  a.nfast = 0;
  a.hot = B3Hot{};
  const vitta_conv_desc& d = a.d;
  if (d.C <= 0 || d.K <= 0 || d.N <= 0 || d.ntaps < 1 || d.ntaps > VITTA_CONV_MAX_TAPS || d.sstride < 1 || d.ostride < 1 ||
      d.ostride > 2)
    return VITTA_ERR_INVALID_ARG;
  if (d.C % 16 || d.K % 32) return VITTA_ERR_UNSUPPORTED;
  a.xP = (int64_t)d.N * d.Hs * d.Ws;
  a.yP = (int64_t)d.N * d.Hy * d.Wy;
  const int64_t M = (int64_t)d.N * d.Hg * d.Wg;
  if (M >= (1ll << 31) || a.xP % 4 || a.yP % 4 || M % 4) return VITTA_ERR_UNSUPPORTED;
  a.Mtot = (int)M;
  a.contig = (d.ostride == 1 && d.oa == 0 && d.ob == 0 && d.Hg == d.Hy && d.Wg == d.Wy) ? 1 : 0;
  a.rP = (d.flags & VITTA_CONV_RES_HALF) ? (int64_t)d.N * ((d.Hy + 1) / 2) * ((d.Wy + 1) / 2) : a.yP;
  if (!a.contig && (d.flags & (VITTA_CONV_BWD_BN | VITTA_CONV_STATS | VITTA_CONV_RES | VITTA_CONV_RES_HALF | VITTA_CONV_EPI_APPLY |
                               VITTA_CONV_EPI_RELU)))
    return VITTA_ERR_UNSUPPORTED;
  if ((d.flags & VITTA_CONV_PRO_BN_RELU) && (d.C > PRO_MAX || !d.pro_bn[0] || !d.pro_bn[1] || !d.pro_bn[2] || !d.pro_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & (VITTA_CONV_EPI_APPLY | VITTA_CONV_STATS)) && (!d.epi_bn[0] || !d.epi_bn[1] || !d.epi_bn[2] || !d.epi_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_STATS) && (!d.st_shift || !d.st_s1 || !d.st_s2)) return VITTA_ERR_INVALID_ARG;
  if (d.flags & VITTA_CONV_POOL) {
    if (!d.pool || !d.epi_bn[0] || !d.epi_bn[1] || !d.epi_bn[2] || !d.epi_bn[3]) return VITTA_ERR_INVALID_ARG;
    if (!a.contig || (d.flags & VITTA_CONV_BWD_BN) || d.Hy * d.Wy < 32) return VITTA_ERR_UNSUPPORTED;
  }
  if ((d.flags & VITTA_CONV_BWD_BN) && (!d.bwd_x || !d.bwd_bn[0] || !d.bwd_bn[1] || !d.bwd_bn[2] || !d.bwd_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_BWD_BN) && (d.flags & (VITTA_CONV_STATS | VITTA_CONV_EPI_APPLY | VITTA_CONV_EPI_RELU)))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & (VITTA_CONV_RES | VITTA_CONV_RES_HALF)) && !d.res) return VITTA_ERR_INVALID_ARG;
  a.sk_G = a.sk_aligned = a.pw = a.pw_prefetch = a.b3 = 0;
  // Split-bf16 operands on the bf16 matrix pipe (conv_b3.hip) whenever the caller supplied the split weight image and the
  // shape fits its one configuration: 128 x 64 tiles, 32-channel slabs (VITTA_CONV_B3=0 keeps the exact-fp32 kernels)
  if (d.w_b3 && b3_enabled() && d.C % 32 == 0 && d.K % 64 == 0 && !(d.flags & VITTA_CONV_PRO_BN_RELU) &&
      (h->tile == 0 || h->tile == ((128 << 16) | 64)) && (int64_t)d.C * a.xP * 4 < (1ll << 31)) {
    // form: pointwise rows, one patch per channel slab (stride-1 taps inside the halo), or gathered (anything else)
    const int form = is_vector_geometry(d) ? 1 : b3_patch_geometry(d, 63) ? 2 : 3;
    static const int gather_on = env_int("VITTA_CONV_B3_GATHER", 1);  // 0: gathered geometries stay on the exact-fp32 kernels (A/B)
    if (form == 3 && !gather_on) goto exact_fp32;
    // the pooled-means epilogue exists in the plain pointwise instantiation (and in the tile kernel below)
    if ((d.flags & VITTA_CONV_POOL) && (form != 1 || (d.flags & (VITTA_CONV_RES | VITTA_CONV_RES_HALF)))) goto exact_fp32;
    const bool parity4 = d.flags & VITTA_CONV_PARITY4;
    if (parity4) {
      int sum = 0;
      for (int c = 0; c < 4; ++c) {
        if (d.cls_ntaps[c] < 1) return VITTA_ERR_INVALID_ARG;
        sum += d.cls_ntaps[c];
      }
      if (sum != d.ntaps || d.ostride != 2 || form != 2) return VITTA_ERR_UNSUPPORTED;
    }
    const int bm = 128, bn = 64;  // the one tile shape of conv_b3.hip (round 3 measured 128 x 128 and 64 x 64 tiles: slower)
    a.nMt = (int)((M + bm - 1) / bm);
    a.nNt = d.K / bn;
    a.d.tile = (bm << 16) | bn;
    // K is split over CHANNEL slabs (a slice walks all taps of its slabs): aim at >= ~1.5 workgroups per CU
    const int ncs = d.C / 32, tiles = a.nMt * a.nNt * (parity4 ? 4 : 1);
    a.cls_tiles = parity4 ? a.nMt * a.nNt : 0;
    a.cls_tap0[0] = 0;
    for (int c = 0; c < 4; ++c) a.cls_tap0[c + 1] = a.cls_tap0[c] + (parity4 ? d.cls_ntaps[c] : 0);
    int ks = 1;
    if (tiles > MAX_SPLIT_TILES) ks = 1;
    else if (d.ksplit != 0) ks = d.ksplit > 0 ? d.ksplit : 1;
    else {
      static const int min_wgs = env_int("VITTA_CONV_B3_MIN_WGS", 384), min_steps = env_int("VITTA_CONV_B3_MIN_STEPS", 4);
      // tools/debug/b3_ks_sweep.sh forces a factor for a per-layer sweep.  Round 3: the best factor per layer beats this
      // heuristic by 1.2 % (16 frames) / 6 % (8 frames) stand-alone and by nothing measurable in the step (5.83 vs 5.84 ms
      // with the twelve differing entries as a table) -- no table kept
      static const int force_ks = env_int("VITTA_CONV_B3_FORCE_KS", 0);
      const int taps_min = parity4 ? 1 : d.ntaps;  // (the lightest class of a parity-merged launch has one tap)
      while (tiles * ks < min_wgs && (ncs / (ks * 2)) * taps_min >= min_steps && ncs % (ks * 2) == 0 && ks < 16) ks *= 2;
      if (force_ks > 0) ks = force_ks;
    }
    if (ks > ncs) ks = ncs;
    const size_t need = ks > 1 ? counter_bytes(tiles) + (size_t)tiles * ks * bm * bn * sizeof(float) : 0;
    if (ks > 1 && (!d.workspace || (size_t)d.workspace_bytes < need)) {
      if (d.ksplit > 0) return VITTA_ERR_WORKSPACE;
      ks = 1;
    }
    a.ksplit = ks;
    a.ws_need = need;
    a.cnt = ks > 1 ? static_cast<unsigned*>(d.workspace) : nullptr;
    a.slabs = ks > 1 ? reinterpret_cast<float*>(static_cast<char*>(d.workspace) + counter_bytes(tiles)) : nullptr;
    a.b3 = form;
    {
      // operand bytes one XCD pulls through its L2 under either tile order: its share of one operand, all of the other
      static const int nfast_mode = env_int("VITTA_CONV_B3_NFAST", -1);  // -1: by operand size, 0 / 1: forced
      const double w_bytes = 6.0 * d.C * d.K * d.ntaps, a_bytes = 4.0 * d.C * (double)a.xP;
      a.nfast = (nfast_mode < 0 ? (w_bytes > 1.5 * a_bytes && a.nNt >= 8) : nfast_mode) ? 1 : 0;
    }
    static const int pf = env_int("VITTA_CONV_PW_PREFETCH", 1);
    a.pw_prefetch = (pf && a.contig &&
                     ((d.flags & VITTA_CONV_BWD_BN) || ((d.flags & VITTA_CONV_RES) && d.res && !(d.flags & VITTA_CONV_RES_HALF)))) ? 1 : 0;
    for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t)
      a.tap[t] = t < d.ntaps ? ((d.dh[t] & 0xff) | ((d.dw[t] & 0xff) << 8) | ((int)d.wt[t] << 16)) : 0;
    // the workgroup prologue's launch constants, divisions as reciprocal multiplications (conv_common.h)
    a.hot.d_ks = make_fastdiv(ks);
    a.hot.d_nNt = make_fastdiv(a.nNt);
    a.hot.d_nMt = make_fastdiv(a.nMt);
    a.hot.d_hw = make_fastdiv((int64_t)d.Hg * d.Wg);
    a.hot.d_w = make_fastdiv(d.Wg);
    a.hot.nwg = tiles * ks;
    a.hot.ksplit = ks;
    a.hot.nNt = a.nNt;
    a.hot.nMt = a.nMt;
    a.hot.ncs = ncs;
    a.hot.flags = (a.nfast ? 1 : 0) | (parity4 ? 2 : 0);
    return VITTA_OK;
  }
exact_fp32:
  if (d.flags & VITTA_CONV_PARITY4) return VITTA_ERR_UNSUPPORTED;  // one launch per parity class on the exact-fp32 kernels
  a.cls_tiles = 0;
  int bm = 64, bn = 32;
  choose_tile(d, M, bm, bn);
  if (d.K % bn) return VITTA_ERR_UNSUPPORTED;
  a.nMt = (int)((M + bm - 1) / bm);
  a.nNt = d.K / bn;
  a.d.tile = (bm << 16) | bn;
  // split K over workgroups when the tiles alone leave CUs idle: aim at >= ~3 workgroups per CU, >= 4 slabs per slice
  const int bk_ = (d.C % 32 == 0) ? 32 : 16;
  const int nslab = (d.C / bk_) * d.ntaps, tiles = a.nMt * a.nNt;
  int ks = 1;
  if (tiles > MAX_SPLIT_TILES) ks = 1;
  else if (d.ksplit > 0) ks = d.ksplit;
  else if (d.ksplit == 0) {
    while (tiles * ks < 640 && nslab / (ks * 2) >= 4 && ks < 16) ks *= 2;
  }
  if (ks > nslab) ks = nslab;
  const size_t need = ks > 1 ? counter_bytes(tiles) + (size_t)tiles * ks * bm * bn * sizeof(float) : 0;
  if (ks > 1 && (!d.workspace || (size_t)d.workspace_bytes < need)) {
    if (d.ksplit > 0) return VITTA_ERR_WORKSPACE;
    ks = 1;  // no (or too small a) workspace: one workgroup per tile
  }
  a.ksplit = ks;
  a.ws_need = need;
  a.cnt = ks > 1 ? static_cast<unsigned*>(d.workspace) : nullptr;
  a.slabs = ks > 1 ? reinterpret_cast<float*>(static_cast<char*>(d.workspace) + counter_bytes(tiles)) : nullptr;
  // Persistent stream-K form (conv_sk.hip) for the 64 x 64 x 32 configuration whenever the caller left tile and split to
  // the library and lent a workspace: as many workgroups as the chip holds at once (3 per CU), each walking an equal
  // share of the launch's K-slabs.  Short-K launches (< 4 slabs per tile) and launches with >= 8 tiles per workgroup
  // keep their ranges on tile boundaries (no partial tiles).
  // Pointwise launches whose tiles fit the chip in one round of four workgroups per CU, with K short enough that a tile
  // is not the whole launch's critical path: conv_pw.hip (VITTA_CONV_PW=0 keeps them on the stream-K kernel)
  // (VITTA_CONV_POOL outside conv_b3.hip: the tile kernel, whose epilogue has it)
  if (d.flags & VITTA_CONV_POOL) {
  } else
  if (bm == 64 && bn == 64 && bk_ == 32 && h->tile == 0 && h->ksplit == 0 && pw_enabled() && is_vector_geometry(d) && a.contig &&
      !(d.flags & VITTA_CONV_PRO_BN_RELU) && tiles >= pw_min_tiles() && tiles <= MAX_SPLIT_TILES && nslab <= pw_max_slabs() &&
      (int64_t)d.C * a.xP * 4 < (1ll << 31)) {
    a.pw = 1;
    // the epilogue's residual / BatchNorm-backward input can be requested at the tile's start (VITTA_CONV_PW_PREFETCH=0: A/B)
    static const int pf = env_int("VITTA_CONV_PW_PREFETCH", 1);
    a.pw_prefetch = (pf && ((d.flags & VITTA_CONV_BWD_BN) ||
                            ((d.flags & VITTA_CONV_RES) && d.res && !(d.flags & VITTA_CONV_RES_HALF)))) ? 1 : 0;
    a.ksplit = 1;
    a.ws_need = 0;
    a.cnt = nullptr;
    a.slabs = nullptr;
  } else
  if (bm == 64 && bn == 64 && bk_ == 32 && h->tile == 0 && h->ksplit == 0 && d.workspace && tiles <= MAX_SPLIT_TILES && sk_enabled() &&
      (int64_t)d.C * a.xP * 4 < (1ll << 31)) {  // buffer addressing: 31-bit byte offsets into x
    const int64_t units = (int64_t)tiles * nslab;
    int G = (int)(units < resident_slots() ? units : resident_slots());
    const int aligned = (nslab < 4 || tiles >= 8 * G) ? 1 : 0;
    if (aligned && G > tiles) G = tiles;
    const size_t sk_need = counter_bytes(tiles) + (size_t)G * 2 * 64 * 64 * sizeof(float);
    if ((size_t)d.workspace_bytes >= sk_need) {
      a.sk_G = G;
      a.sk_aligned = aligned;
      a.ksplit = 1;
      a.ws_need = sk_need;
      a.cnt = static_cast<unsigned*>(d.workspace);
      a.slabs = reinterpret_cast<float*>(static_cast<char*>(d.workspace) + counter_bytes(tiles));
    }
  }
  for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t)
    a.tap[t] = t < d.ntaps ? ((d.dh[t] & 0xff) | ((d.dw[t] & 0xff) << 8) | ((int)d.wt[t] << 16)) : 0;
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_conv_supported(const vitta_conv_desc* h_desc) {
  ConvK a;
  return fill(h_desc, a) == VITTA_OK ? 1 : 0;
}

int64_t vitta_conv_num_blocks(const vitta_conv_desc* h_desc) {
  ConvK a;
  if (fill(h_desc, a) != VITTA_OK) return -1;
  return (int64_t)a.nMt * a.nNt * (a.cls_tiles ? 4 : 1);
}

size_t vitta_conv_workspace_bytes(const vitta_conv_desc* h_desc) {
  if (!h_desc) return 0;
  vitta_conv_desc d = *h_desc;
  static char probe;  // any non-null pointer: only sizes matter here
  d.workspace = &probe;
  d.workspace_bytes = INT64_MAX;
  ConvK a;
  if (fill(&d, a) != VITTA_OK) return 0;
  return a.ws_need;
}

int vitta_conv_kernel(const vitta_conv_desc* h_desc) {
  ConvK a;
  if (fill(h_desc, a) != VITTA_OK) return -1;
  return a.b3 ? VITTA_CONV_KERNEL_B3 : a.pw ? VITTA_CONV_KERNEL_PW : a.sk_G ? VITTA_CONV_KERNEL_SK : VITTA_CONV_KERNEL_TILE;
}

int64_t vitta_conv_fastdiv_host(int64_t n, int64_t d) {
  if (n < 0 || n >= (1ll << 31) || d < 1 || d >= (1ll << 31)) return -1;
  const FastDiv f = make_fastdiv(d);  // the device's fdiv(): __umulhi(n, mul) >> sh
  return f.sh < 0 ? n : (int64_t)((((uint64_t)(unsigned)n * f.mul) >> 32) >> f.sh);
}

int64_t vitta_conv_flops(const vitta_conv_desc* h_desc) {
  ConvK a;
  if (fill(h_desc, a) != VITTA_OK) return -1;
  return 2ll * a.Mtot * a.d.K * a.d.C * a.d.ntaps;  // (parity-merged: Mtot positions per class x the classes' taps = the same sum)
}

int vitta_conv_f32(const vitta_conv_desc* h_desc, void* stream) { return vitta_conv_timed_f32(h_desc, stream, nullptr, nullptr); }

int vitta_conv_timed_f32(const vitta_conv_desc* h_desc, void* stream, void* ev_start, void* ev_stop) {
  ConvK a;
  const int rc = fill(h_desc, a);
  if (rc != VITTA_OK) return rc;
  if ((ev_start == nullptr) != (ev_stop == nullptr)) return VITTA_ERR_INVALID_ARG;
  hipEvent_t e0 = static_cast<hipEvent_t>(ev_start), e1 = static_cast<hipEvent_t>(ev_stop);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool gather = !is_vector_geometry(a.d);
  if (a.b3) return launch_b3(a, st, e0, e1);
  if (!a.d.w) return VITTA_ERR_UNSUPPORTED;  // only the split image was given and the shape does not qualify for it
  if (a.pw) return launch_pointwise(a, st, e0, e1);
  if (a.sk_G) return launch_stream_k(a, gather, st, e0, e1);
  const int bm = a.d.tile >> 16, bn = a.d.tile & 0xffff;
  const int bk = (a.d.C % 32 == 0) ? 32 : 16;
#define CFG(M_, N_, K_, WM_, WN_) \
  if (bm == M_ && bn == N_ && bk == K_) return launch_cfg<M_, N_, K_, WM_, WN_>(a, gather, st, e0, e1)
  CFG(128, 128, 32, 2, 2);
  CFG(128, 64, 32, 2, 2);
  CFG(64, 64, 32, 2, 2);
  CFG(64, 32, 32, 2, 1);
  CFG(128, 128, 16, 2, 2);
  CFG(128, 64, 16, 2, 2);
  CFG(64, 64, 16, 2, 2);
  CFG(64, 32, 16, 2, 1);
#undef CFG
  return VITTA_ERR_UNSUPPORTED;
}

}  // extern "C"
