// The trunk's convolutions on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split into three bf16 terms
//   x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (round to nearest even: |lo| <= 2^-16 |x|)
// and a product is accumulated as the six terms of order <= 2,
//   a b ~= a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi      (dropped: <= 2^-23 |a b|),
// each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Six bf16 MFMAs do 16 k of a 32 x 32 block in 6 x 32 cycles where
// the exact-fp32 instruction (v_mfma_f32_32x32x2_f32, conv_sk.hip / conv_pw.hip) needs 8 x 64: 2.67x the matrix rate
// (2.5 PF / 6 = 417 TF of fp32-grade multiply-adds), error of the fp32-roundoff class (tests/test_gpu_conv.py holds both
// arithmetic forms to the same bounds against fp64: 2e-5 of the tensor's maximum, and element-wise 4 x 2^-23 x sqrt(R) x sum|a b|
// under 16 decades of per-channel dynamic range; an Inf / NaN operand makes exactly the elements non-finite that fp32
// arithmetic makes non-finite, as NaN -- x - bf16(x) is NaN for x = Inf).
// Reference call sites: models/tanet_models/temporal_module.py:85-106, tanet.py:125-150 (as conv.hip).
//
// What ships (round 4; the round-3 variants that measured equal or slower -- 128 x 128 tiles, a persistent pointwise form, 64 x 64
// two-wave tiles, three 16-channel stages -- are gone, their measurements are in DESIGN.md section 4a'):
//   * same implicit GEMM as conv.hip -- D[p][k] = sum_{tap, c} X[c][src(p, tap)] W[tap][c][k], pixels of all frames on the MFMA row
//     axis, output channels on the column axis, the epilogues of conv_epilogue.h -- with a 128 x 64 output tile per workgroup: four
//     waves stacked along the pixels (32 rows x 64 columns each: one A fragment feeds twelve MFMAs), K walked tap OUTER in steps of
//     32 channels x 1 tap;
//   * the activations stay fp32 in memory AND in LDS and are split in registers per MFMA fragment (44 vector instructions per
//     32 rows x 16 channels); the weights are split ONCE per weight version into [tap][C / 32][3 planes][4 channel octets][K][8] bf16
//     (vitta_conv_pack_b3): a B fragment is one 16-byte LDS read;
//   * both operands reach LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane: no registers, no vector ALU) into a ring of
//     stages; a step = counted s_waitcnt vmcnt for its stage, ONE s_barrier in its middle, the requests of the step after next,
//     then reads -> split -> MFMA software-pipelined across the two 16-channel halves;
//   * MODE 0 pointwise: the A stage is the tile's 128 pixels x 32 channels, two stages.  MODE 1 "patch" (3x3, stride 1): the A
//     image of a channel slab is the flat pixel range [m0 - 64, m0 + 192) of each channel row, loaded ONCE per slab; a tap is an
//     address shift of the operand reads (a lane whose tap falls outside the plane reads a position the DMA keeps at zero).
//     MODE 2 gathered (stride 2, planes wider than the halo): four bytes per lane, out-of-plane taps requested out of range;
//   * launches with few tiles split K over channel slabs (partial tiles through the write-through workspace, the last arriver of a
//     tile sums them in slice order and runs the epilogue); the stride-2 data gradient's four parity classes are ONE launch;
//   * XCD-aware tile order (an XCD owns a contiguous range of logical workgroup ids; pixel tiles fastest where the weight image is
//     the larger operand).
// Round 4, the workgroup prologue: a workgroup used to need 2.3 (pointwise) to 3.8 us (3x3) from its first instruction to its
// first stage in LDS (tools/debug/b3_trace.py) -- ~500 instructions with 23 serialised scalar-memory waits and nine integer
// divisions by launch constants.  The launch constants now arrive as one 64-byte block (ConvK::hot), every division is a multiply
// by a host-made reciprocal (FastDiv), the tap table is read by independent loads, the tap word of a request is fetched one step
// ahead of its use, and the epilogue gets the tile origin handed over instead of dividing again.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "conv_epilogue.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> packed bf16 pairs of the three terms
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#ifdef B3_TRACE
// tools/debug/b3_trace.py: shader-clock stamps of wave 0 at the phase boundaries of a workgroup, 16 words per workgroup
__device__ unsigned long long b3_trace_buf[16 * 8192];
#define B3_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 8192) b3_trace_buf[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#define B3_CLK() __builtin_readcyclecounter()
#else
#define B3_STAMP(i) \
  do {              \
  } while (0)
#endif

// ---- round 5: the epilogue of a CONTIGUOUS output through an LDS turn-around -------------------------------------------------------
// In the accumulator layout a lane owns ONE output channel and four consecutive pixels per register quad, so every epilogue access of a
// wave -- output stores, residual / BatchNorm-backward input / mask loads -- is 64 pieces of 16 bytes in 64 different channel rows:
// 64 cache lines per instruction for 1 KB of data.  A 128 x 64 tile with one input stream and one or two outputs is ~6 000 such line
// accesses per workgroup through the CU's one L1 port (~2.7 us: "epilogue until its stores have landed", profiles/r4_conv_b3_phase_timeline.txt),
// and the launches whose workgroups are short -- 64 -> 256 at 56 x 56: 6 workgroups per CU of 1.8 us of K walk each -- are bound by it.
// Here every wave writes its two 32 x 32 accumulator blocks to a private 8 KB LDS area as [channel][32 pixels] (16-byte pieces XOR-
// swizzled by channel pair: writes and reads are both bank-conflict free) and reads them back with lane = (channel row lane / 8,
// piece lane % 8): an access instruction then covers 8 channel rows x 128 contiguous bytes = 8 lines, the same arithmetic as
// TileEpilogue::body runs on (channel, four pixels) items, the per-channel sums meet over the 8 lanes of a row by shuffles.
struct EpiConst {  // per-channel constants of the tile, derived once: forward es, et, shift; backward s, t, mean, rstd, ia, ib, mu
  float c0, c1, c2, c3, c4, c5, c6;
};

template <bool PRE, bool POOLT>
__device__ __forceinline__ void epilogue_t(const ConvK& a, int m0, int k0, int wave, int lane, const f32x16 (&acc)[2], const TileEpilogue& e0,
                                           const TileEpilogue& e1, const float4 (&pre)[8], unsigned char* lds, float* red) {
  const vitta_conv_desc& d = a.d;
  const int li = lane & 31, lk = lane >> 5;
  const int flags = d.flags;
  const bool BWD = flags & VITTA_CONV_BWD_BN;
  const bool STATS = (flags & VITTA_CONV_STATS) && d.st_s1;
  const bool RAWST = flags & VITTA_CONV_STATS_RAW;
  const bool APPLY = flags & VITTA_CONV_EPI_APPLY;
  const bool RELU = flags & VITTA_CONV_EPI_RELU;
  const bool RES = (flags & VITTA_CONV_RES) && d.res;
  const bool RESH = (flags & VITTA_CONV_RES_HALF) && d.res;
  const bool BRELU = flags & VITTA_CONV_BWD_RELU;
  const bool IRAW = (flags & VITTA_CONV_INJ_RAW) && d.inj_mu;
  const bool POOL = POOLT && (flags & VITTA_CONV_POOL) && d.pool;
  float* const W = reinterpret_cast<float*>(lds + wave * 8192);             // [64 channels][32 pixels], pieces swizzled
  float* const cst = reinterpret_cast<float*>(lds + 32768 + wave * 2048);   // [7][64]
  // ---- accumulators -> LDS (lane = channel li of block y, pixel quad 8 qd + 4 lk) ----
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int ch = 32 * y + li, f = (ch >> 1) & 3;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int q = 2 * qd + lk;
      const f32x4 v = {acc[y][4 * qd], acc[y][4 * qd + 1], acc[y][4 * qd + 2], acc[y][4 * qd + 3]};
      *reinterpret_cast<f32x4*>(W + ch * 32 + ((q ^ f) << 2)) = v;
    }
  }
  // ---- derived per-channel constants (the lanes of the lower half hold channel li of either block) ----
  if (lk == 0) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const TileEpilogue& e = y ? e1 : e0;
      const int ch = 32 * y + li;
      float c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f, c6 = e.c_sh;
      if (BWD) {
        c3 = rsqrtf(e.c_var + d.bwd_eps);   // rstd
        c0 = e.c_gam * c3;                  // s
        c1 = e.c_bet - e.c_mean * c0;       // t
        c2 = e.c_mean;
        c4 = e.c_gs * e.c_a;
        c5 = e.c_gs * e.c_b;
        c6 = e.c_mu;
      } else if (d.epi_bn[0]) {
        c0 = e.c_gam * rsqrtf(e.c_var + d.epi_eps);
        c1 = e.c_bet - e.c_mean * c0;
      }
      cst[ch] = c0; cst[64 + ch] = c1; cst[128 + ch] = c2; cst[192 + ch] = c3; cst[256 + ch] = c4; cst[320 + ch] = c5; cst[384 + ch] = c6;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private areas: no barrier)
  // ---- items: (channel row 8 i + lane / 8, 16-byte piece lane % 8) ----
  const int r = lane >> 3, q = lane & 7;
  const int m = m0 + 32 * wave + 4 * q;
  const bool live = m < a.Mtot;
  const bool counted = STATS && live;
  const int HWy = d.Hy * d.Wy;
  PoolSums pool(a, m0 + 32 * wave, POOL);
  float s1v[8], s2v[8];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 v[4], x2[4], x3[4];
    // the tile values and every remaining input piece of these four items first ...
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = 8 * (4 * half + j) + r;
      v[j] = *reinterpret_cast<const f32x4*>(W + ch * 32 + ((q ^ ((ch >> 1) & 3)) << 2));
      const int64_t k = k0 + ch;
      const int mm = live ? m : a.Mtot - 4;
      x2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      x3[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (BWD) {
        if (!PRE) x2[j] = *reinterpret_cast<const f32x4*>(d.bwd_x + k * a.yP + mm);
        if (RES) x3[j] = *reinterpret_cast<const f32x4*>(d.res + k * a.rP + mm);
      } else if (RES && !PRE) {
        x2[j] = *reinterpret_cast<const f32x4*>(d.res + k * a.rP + mm);
      }
    }
    f32x4 mk[4];
    if (BWD && BRELU && d.bwd_mask) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[j] = *reinterpret_cast<const f32x4*>(d.bwd_mask + (int64_t)(k0 + 8 * (4 * half + j) + r) * a.yP + (live ? m : a.Mtot - 4));
    }
    // ... then the arithmetic of TileEpilogue::body per item
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * half + j, ch = 8 * i + r;
      const int64_t k = k0 + ch;
      const float c0 = cst[ch], c1 = cst[64 + ch], c6 = cst[384 + ch];
      float vv[4] = {v[j][0], v[j][1], v[j][2], v[j][3]};
      float r1 = 0.f, r2 = 0.f;
      float* yp = d.y + k * a.yP + m;
      if (RESH) {
        const int Hh = (d.Hy + 1) >> 1, Wh = (d.Wy + 1) >> 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = (live ? m : 0) + e;
          const int n = p / HWy, rr = p - n * HWy, h = rr / d.Wy, w = rr - h * d.Wy;
          if (!((h | w) & 1)) vv[e] += d.res[k * a.rP + (int64_t)n * Hh * Wh + (h >> 1) * Wh + (w >> 1)];
        }
      }
      if (BWD) {
        const float c2 = cst[128 + ch], c3 = cst[192 + ch], c4 = cst[256 + ch], c5 = cst[320 + ch];
        f32x4 xr = x2[j];
        if constexpr (PRE) xr = f32x4{pre[i].x, pre[i].y, pre[i].z, pre[i].w};
        if (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] += x3[j][e];
        }
        float o[4], gm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fmaf(xr[e], c0, c1);
          const float mmk = BRELU ? (d.bwd_mask ? (mk[j][e] > 0.f ? 1.f : 0.f) : (z > 0.f ? 1.f : 0.f)) : 1.f;
          gm[e] = vv[e] * mmk;
          const float dz = IRAW ? gm[e] : gm[e] + fmaf(c5, z - c6, c4);
          r1 += dz * (xr[e] - c2) * c3;
          r2 += dz;
          o[e] = IRAW ? fmaf(dz, c0, fmaf(c5, xr[e] - c6, c4)) : dz * c0;
        }
        if (live) {
          *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
          if (d.y_raw) *reinterpret_cast<f32x4*>(d.y_raw + k * a.yP + m) = f32x4{gm[0], gm[1], gm[2], gm[3]};
        }
      } else {
        if (live && d.y_raw) *reinterpret_cast<f32x4*>(d.y_raw + k * a.yP + m) = f32x4{vv[0], vv[1], vv[2], vv[3]};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fmaf(vv[e], c0, c1);
          if (counted) {
            const float dd = (RAWST ? vv[e] : z) - c6;
            r1 += dd;
            r2 = fmaf(dd, dd, r2);
          }
          if (POOL && live) pool.add(m + e, z);
          o[e] = APPLY ? z : vv[e];
        }
        if (RES) {
          f32x4 rr = x2[j];
          if constexpr (PRE) rr = f32x4{pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rr[e];
        }
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.f ? 0.f : o[e];  // (keeps NaN, as torch's relu)
        }
        if (live) *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
      }
      if (!live) r1 = r2 = 0.f;
      if (POOL) {  // the item's two frame sums over the row's eight lanes -> one atomic per (block, channel, frame)
        pool.s0 += __shfl_xor(pool.s0, 1, 64); pool.s1 += __shfl_xor(pool.s1, 1, 64);
        pool.s0 += __shfl_xor(pool.s0, 2, 64); pool.s1 += __shfl_xor(pool.s1, 2, 64);
        pool.s0 += __shfl_xor(pool.s0, 4, 64); pool.s1 += __shfl_xor(pool.s1, 4, 64);
        if (q == 0 && pool.mblk < a.Mtot) {
          unsigned long long* pl = reinterpret_cast<unsigned long long*>(d.pool);
          pool_add(pl + (int64_t)pool.fA * d.K + k, pool.s0 * d.pool_scale);
          if (pool.mB < pool.mblk + 32 && pool.mB < a.Mtot) pool_add(pl + (int64_t)(pool.fA + 1) * d.K + k, pool.s1 * d.pool_scale);
        }
        pool.s0 = pool.s1 = 0.f;
      }
      s1v[i] = r1;
      s2v[i] = r2;
    }
  }
  // ---- per-channel sums: the eight lanes of a row, then the four waves (pixel quarters of the tile), one atomic per (tile, channel) ----
  if (STATS || BWD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s1v[i] += __shfl_xor(s1v[i], 1, 64); s2v[i] += __shfl_xor(s2v[i], 1, 64);
      s1v[i] += __shfl_xor(s1v[i], 2, 64); s2v[i] += __shfl_xor(s2v[i], 2, 64);
      s1v[i] += __shfl_xor(s1v[i], 4, 64); s2v[i] += __shfl_xor(s2v[i], 4, 64);
    }
    if (wave > 0 && q == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[((wave - 1) * 64 + 8 * i + r) * 2] = s1v[i];
        red[((wave - 1) * 64 + 8 * i + r) * 2 + 1] = s2v[i];
      }
    }
    __syncthreads();
    if (wave == 0 && q == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ch = 8 * i + r;
        float t1 = s1v[i], t2 = s2v[i];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          t1 += red[(w * 64 + ch) * 2];
          t2 += red[(w * 64 + ch) * 2 + 1];
        }
        if (BWD) {
          if (d.dgamma) atomicAdd(d.dgamma + k0 + ch, t1);
          if (d.dbeta) atomicAdd(d.dbeta + k0 + ch, t2);
        } else {
          atomicAdd(d.st_s1 + k0 + ch, t1);
          atomicAdd(d.st_s2 + k0 + ch, t2);
        }
      }
    }
  }
}

template <int MODE, bool PRE, bool POOLT>
__global__ __launch_bounds__(256, 2) void conv_b3_kernel(const ConvK a) {
  constexpr bool PATCH = MODE == 1, GATHER = MODE == 2;
  constexpr int NB = PATCH ? 3 : 2;       // B stages (A: NA)
  constexpr int NW = 4, BM = 128, BN = 64, BK = 32, HALO = 64, NTHR = 256;
  constexpr int PL = PATCH ? 256 : BM;    // pixels per channel row of the A image
  constexpr int NA = PATCH ? 1 : NB;
  constexpr int PER_STEP = PATCH ? 3 : GATHER ? 19 : 7;  // LDS-DMA instructions of a wave per step
  constexpr int RPW = BK / NW;            // channel rows of an A stage a wave requests
  constexpr int A_BYTES = BK * PL * 4, B_BYTES = 12 * BN * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;                        // [NA][BK][PL] fp32
  unsigned char* const Bb = lds + NA * A_BYTES;         // [NB][3 planes][4 channel octets][BN][8] bf16
  float* const red = reinterpret_cast<float*>(Bb + NB * B_BYTES);  // [NW - 1][2][32][2]
  int* const flag = reinterpret_cast<int*>(red + 384);

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  B3_STAMP(0);
  // ---- which tile, which slice of K: launch constants in one block, divisions as multiplications -------------------------------
  const B3Hot h = a.hot;
  const int C = d.C, K = d.K, Mtot = a.Mtot;
  const int Lz = xcd_remap(blockIdx.x, h.nwg);
  const int Lg = fdiv(Lz, h.d_ks), kz = Lz - Lg * h.ksplit;  // tile of the launch (arrival counter, partial tiles), slice
  // parity-merged data gradient: the launch holds four classes of tiles, each with its own run of the tap table (class fastest:
  // the classes have 1, 2, 2 and 4 taps -- as four contiguous blocks of ids the XCDs, which own contiguous id ranges, would get
  // one class each: a 4x imbalance; interleaved, every XCD holds all four classes of its tiles, which read the same input pixels)
  const bool par = h.flags & 2;
  const int cls = par ? (Lg & 3) : 0, Lc = par ? (Lg >> 2) : Lg;
  // an XCD owns a contiguous range of logical ids.  With column tiles fastest (default) that is a few pixel tiles x ALL output
  // channels -- every XCD pulls the whole weight image through its own L2; where the image is the larger operand (layer 4: 14 MB
  // against 1.6 MB of activations) the host asks for pixel tiles fastest instead: an XCD then covers all pixels of a few column
  // tiles and reads only their share of the weights
  int L = Lc;
  if (h.flags & 1) {
    const int qn = fdiv(Lc, h.d_nMt);
    L = (Lc - qn * h.nMt) * h.nNt + qn;
  }
  const int mt = fdiv(L, h.d_nNt), m0 = mt * BM, k0 = (L - mt * h.nNt) * BN;
  const int tap0 = par ? a.cls_tap0[cls] : 0, ntaps = par ? a.cls_tap0[cls + 1] - tap0 : d.ntaps;
  const int ncs = h.ncs;
  const int cs0 = fdiv(ncs * kz, h.d_ks), cs1 = fdiv(ncs * (kz + 1), h.d_ks);
  const int S = (cs1 - cs0) * ntaps;  // steps = (channel slab, tap) pairs
  B3_STAMP(9);

  // wave w owns pixel rows 32 w .. 32 w + 31 and all 64 output channels (two 32 x 32 accumulators)
  TileEpilogue epi0(a, red, wave >> 1, 0, li, lk, BM), epi1(a, red, wave >> 1, 1, li, lk, BM);
  epi0.set_tile(m0, k0);
  epi1.set_tile(m0, k0);
  const int xb = wave & 1;
  if (par) {
    epi0.oa = epi1.oa = cls >> 1;
    epi0.ob = epi1.ob = cls & 1;
  }

  const int row_bytes = (int)(a.xP * 4);
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, C * row_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_b3), 0, 0x7fffffff, 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  B3_STAMP(10);

  // ---- LDS-DMA side ----------------------------------------------------------------------------------------------------
  // A, patch: one instruction = one channel row (256 pixels; lane 0 out of range: positions 0..3 stay zero); wave w loads
  // rows 8 w .. 8 w + 7.  A, pointwise: one instruction = two channel rows (2 x 128 pixels), four per wave.
  const int voff_a = PATCH ? (lane == 0 ? OOB : (m0 - HALO + 4 * lane) * 4)
                           : (lane / (PL / 4)) * row_bytes + min(m0 + 4 * (lane % (PL / 4)), Mtot - 4) * 4;
  // the tap table (nine words at most): independent scalar loads, one wait
  int tapw[VITTA_CONV_MAX_TAPS];
#pragma unroll
  for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t) tapw[t] = (PATCH || GATHER) ? a.tap[min(tap0 + t, VITTA_CONV_MAX_TAPS - 1)] : 0;
  // gathered: the lane's two pixels (lane, lane + 64 of the tile): source offset of tap (0, 0), validity bit per tap
  int g_base[2] = {0, 0};
  unsigned g_valid[2] = {0, 0};
  if constexpr (GATHER) {
    const int hwg = d.Hg * d.Wg;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int m = m0 + lane + 64 * hh;
      const int mm = m < Mtot ? m : 0;
      const int n = fdiv(mm, h.d_hw), r = mm - n * hwg, gi = fdiv(r, h.d_w), gj = r - gi * d.Wg;
      g_base[hh] = (n * d.Hs * d.Ws + gi * d.sstride * d.Ws + gj * d.sstride) * 4;
#pragma unroll
      for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t) {
        const int tp = tapw[t];
        const int sh = gi * d.sstride + (int)(int8_t)(tp & 0xff), sw = gj * d.sstride + (int)(int8_t)((tp >> 8) & 0xff);
        if (t < ntaps && m < Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws) g_valid[hh] |= 1u << t;
      }
    }
  }
  auto dma_a = [&](int cs, int stage, int tp = 0, int t = 0) __attribute__((always_inline)) {
    unsigned char* dst = Ab + stage * A_BYTES + wave * RPW * PL * 4;
    const int c0 = cs * BK + wave * RPW;
    if constexpr (GATHER) {
      const int sh = ((int)(int8_t)(tp & 0xff) * d.Ws + (int)(int8_t)((tp >> 8) & 0xff)) * 4;
      const int v0 = ((g_valid[0] >> t) & 1) ? g_base[0] + sh : OOB, v1 = ((g_valid[1] >> t) & 1) ? g_base[1] + sh : OOB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 512), 4, v0, (c0 + i) * row_bytes, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 512 + 256), 4, v1, (c0 + i) * row_bytes, 0, 0);
      }
    } else if constexpr (PATCH) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 1024), 16, voff_a, (c0 + i) * row_bytes, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 1024), 16, voff_a, (c0 + (256 / PL) * i) * row_bytes, 0, 0);
    }
  };
  // B: the slab image of this tile's 64 output channels = 12 runs (plane, channel octet) of 64 x 16 bytes, three per wave
  auto dma_b = [&](int cs, int tp, int stage) __attribute__((always_inline)) {
    unsigned char* dst = Bb + stage * B_BYTES + wave * 3 * 1024;
    const int run0 = (((tp >> 16) * ncs + cs) * 12 + wave * 3);
#pragma unroll
    for (int u = 0; u < 3; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + u * 1024), 16, lane * 16, ((run0 + u) * K + k0) * 16, 0, 0);
  };

  // ---- multiplying side ------------------------------------------------------------------------------------------------
  // the lane's row: position in the A image, tap validity bits
  const int pos = (PATCH ? HALO : 0) + 32 * wave + li;
  unsigned valid = 0x1ff;
  if constexpr (PATCH) {
    const int m = m0 + 32 * wave + li;
    const int hw = d.Hs * d.Ws, mm = m < Mtot ? m : 0;
    const int n = fdiv(mm, h.d_hw), r = mm - n * hw, hh = fdiv(r, h.d_w), ww = r - hh * d.Ws;
    valid = 0;
#pragma unroll
    for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t) {
      const int tp = tapw[t];
      const int sh = hh + (int)(int8_t)(tp & 0xff), sw = ww + (int)(int8_t)((tp >> 8) & 0xff);
      if (t < ntaps && m < Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws) valid |= 1u << t;
    }
  }
  B3_STAMP(11);
  const int a_lane = (8 * lk * PL + pos) * 4;   // byte offset of the lane's first operand word in an A stage
  const int a_zero = 8 * lk * PL * 4;           // ... of the always-zero position of the same rows
  const int b_lane = (lk * BN + li) * 16;       // ... of its first B operand in a B stage
  // operand address of the lane in stage `stage` for the tap with table word tp (validity bit t)
  auto a_addr = [&](int tp, int t, int stage) __attribute__((always_inline)) -> const float* {
    int off = a_lane;
    if constexpr (PATCH) {
      const int sh = (int)(int8_t)(tp & 0xff) * d.Ws + (int)(int8_t)((tp >> 8) & 0xff);
      off = ((valid >> t) & 1) ? a_lane + sh * 4 : a_zero;
    }
    return reinterpret_cast<const float*>(Ab + stage * A_BYTES + off);
  };

  f32x16 acc[2];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;

  // operand reads of k-step ks: 8 words of A (channels 16 ks + 8 lk + j), 6 x 16 bytes of B (plane p, column block y)
  auto read_ops = [&](const float* ap, const unsigned char* bs_, int ks, float (&raw)[8], bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = ap[(16 * ks + j) * PL];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int y = 0; y < 2; ++y)
        fb[y][p] = *reinterpret_cast<const bf16x8*>(bs_ + b_lane + ((p * 4 + 2 * ks) * BN + 32 * y) * 16);
  };
  auto split = [&](const float (&raw)[8], bf16x8 (&fa)[3]) __attribute__((always_inline)) {
#ifdef B3_ABL_NOSPLIT
    {
      const u32x4 v0 = {__builtin_bit_cast(unsigned, raw[0]), __builtin_bit_cast(unsigned, raw[1]), __builtin_bit_cast(unsigned, raw[2]),
                        __builtin_bit_cast(unsigned, raw[3])};
      const u32x4 v1 = {__builtin_bit_cast(unsigned, raw[4]), __builtin_bit_cast(unsigned, raw[5]), __builtin_bit_cast(unsigned, raw[6]),
                        __builtin_bit_cast(unsigned, raw[7])};
      fa[0] = __builtin_bit_cast(bf16x8, v0);
      fa[1] = __builtin_bit_cast(bf16x8, v1);
      fa[2] = __builtin_bit_cast(bf16x8, v0);
      return;
    }
#endif
    u32x4 sp[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h_, m_, l_;
      split2(raw[2 * j], raw[2 * j + 1], h_, m_, l_);
      sp[0][j] = h_;
      sp[1][j] = m_;
      sp[2][j] = l_;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = __builtin_bit_cast(bf16x8, sp[p]);
  };
  // six products per accumulator, small terms first; the two accumulators alternate
  auto mfma12 = [&](const bf16x8 (&fa)[3], const bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
#ifdef B3_ABL_NOMFMA
    asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]));
    return;
#endif
#define B3_MFMA(PA, PB)                                                                                  \
  _Pragma("unroll") for (int y = 0; y < 2; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA], fb[y][PB], acc[y], 0, 0, 0)
    B3_MFMA(2, 0);
    B3_MFMA(0, 2);
    B3_MFMA(1, 1);
    B3_MFMA(1, 0);
    B3_MFMA(0, 1);
    B3_MFMA(0, 0);
#undef B3_MFMA
  };
  // the MFMAs of one k-step with the operand reads (issued first) and the split of the NEXT k-step between them
  auto interleave = [&]() __attribute__((always_inline)) {
    SGB(0x100, 10);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      SGB(0x008, 1);
      SGB(0x002, 4);
    }
  };

  // ---- pipeline ----------------------------------------------------------------------------------------------------------
  // Step s = (channel slab, tap); its images sit in stage s % NB.  One barrier per step, in its MIDDLE: the first k-step's
  // MFMAs cover the second k-step's operand reads and split; at the barrier every wave's DMA of step s + 1 has landed
  // (counted vmcnt: the steps behind it stay in flight), step s + NB is requested into the stage step s has left, and the
  // second k-step's MFMAs cover the reads and split of step s + 1's first k-step.  (A patch has one stage: at a channel-slab
  // change its DMA is requested after the barrier and the next step starts cold.)
  // The compiler does not see the DMA -> ds_read dependence (one LDS array, no alias information) and is not asked to:
  // every wait is explicit, the barriers are bare s_barrier.
  auto wait_ring = [&]() __attribute__((always_inline)) {  // every DMA of this wave except those of the last NB - 2 steps has landed
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER_STEP) : "memory");
  };
  auto wait_all = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);  // (the interleave groups of the neighbouring regions must not pull reads across)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  float4 pre[8] = {};  // the epilogue's first input stream in the mapping of epilogue_t: item i = channel row 8 i + lane / 8, piece lane % 8
  // requests run NB steps ahead: (cs_q, t_q) = the step to request next, clamped to the slice's last step (the tail
  // re-requests it into a stage nobody reads again: the instruction count per step stays fixed for the counted waits).
  // tq_w = the table word of that step's tap, fetched when the step BEFORE it was requested (pointwise: one tap, a constant):
  // its scalar load has a whole step to return instead of sitting between the barrier and the step's first request
  int cs_q = cs0, t_q = 0, q = 0;
  int tq_w = a.tap[tap0];
  auto request = [&](int stage, bool with_a) __attribute__((always_inline)) {
#ifdef B3_ABL_NODMA
    if (q >= NB) return;
#endif
    dma_b(cs_q, tq_w, stage);
    if constexpr (!PATCH) {
      if (with_a) dma_a(cs_q, stage, tq_w, t_q);
    }
    const bool adv = q + 1 < S;
    q += adv ? 1 : 0;
    const bool wrap = adv && t_q + 1 == ntaps;
    t_q = adv ? (wrap ? 0 : t_q + 1) : t_q;
    cs_q += wrap ? 1 : 0;
    if constexpr (MODE != 0) tq_w = a.tap[tap0 + t_q];
  };
  dma_a(cs0, 0, tq_w, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) request(i, i > 0);
  B3_STAMP(13);
  epi0.load_consts(L);
  epi1.load_consts(L);
  if constexpr (PRE) {  // (the host sets pw_prefetch for contiguous outputs only: BWD_BN -> its input, else the residual)
    const bool bwd = d.flags & VITTA_CONV_BWD_BN;
    const float* row = (bwd ? d.bwd_x : d.res) + (int64_t)(k0 + (lane >> 3)) * (bwd ? a.yP : a.rP) + min(m0 + 32 * wave + 4 * (lane & 7), Mtot - 4);
    const int64_t step = 8 * (bwd ? a.yP : a.rP);
#pragma unroll
    for (int i = 0; i < 8; ++i) pre[i] = *reinterpret_cast<const float4*>(row + i * step);
  }
  // step 0 (requested first) has landed; loads the compiler placed behind the requests only make this wait longer
  B3_STAMP(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * PER_STEP) : "memory");
  barrier();
  B3_STAMP(2);
  float raw[8];
  bf16x8 fa0[3], fa1[3], fb0[2][3], fb1[2][3];
  read_ops(a_addr(tapw[0], 0, 0), Bb, 0, raw, fb0);
  split(raw, fa0);
  int st = 0;  // stage of the current step
#ifdef B3_TRACE
  unsigned long long tr_wait = 0, tr_req = 0;  // shader-clock ticks of wave 0 in: wait for the ring + barrier; issuing the requests
#endif
  // one step whose successor's images are (or will be, after the barrier) in the ring: both k-steps covered
  auto full_step = [&](int tp, int t, int tp1, int t1) __attribute__((always_inline)) {
    const int st1 = st + 1 == NB ? 0 : st + 1;
    read_ops(a_addr(tp, t, PATCH ? 0 : st), Bb + st * B_BYTES, 1, raw, fb1);
    split(raw, fa1);
    mfma12(fa0, fb0);
    interleave();
#ifdef B3_TRACE
    const unsigned long long c0 = B3_CLK();
#endif
    wait_ring();
    barrier();
#ifdef B3_TRACE
    const unsigned long long c1 = B3_CLK();
#endif
    request(st, true);
#ifdef B3_TRACE
    const unsigned long long c2 = B3_CLK();
    tr_wait += c1 - c0;
    tr_req += c2 - c1;
#endif
    read_ops(a_addr(tp1, t1, PATCH ? 0 : st1), Bb + st1 * B_BYTES, 0, raw, fb0);
    split(raw, fa0);
    mfma12(fa1, fb1);
    interleave();
    st = st1;
  };
  if constexpr (!PATCH) {
    for (int s = 0; s + 1 < S; ++s) full_step(0, 0, 0, 0);
    read_ops(a_addr(0, 0, st), Bb + st * B_BYTES, 1, raw, fb1);
    split(raw, fa1);
    mfma12(fa0, fb0);
    interleave();
    mfma12(fa1, fb1);
  } else {
    for (int cs = cs0; cs < cs1; ++cs) {
      int tp = a.tap[tap0];
      for (int t = 0; t + 1 < ntaps; ++t) {
        const int tp1 = a.tap[tap0 + t + 1];
        full_step(tp, t, tp1, t + 1);
        tp = tp1;
      }
      // last tap of the channel slab: behind its barrier the patch is free
      const int st1 = st + 1 == NB ? 0 : st + 1;
      read_ops(a_addr(tp, ntaps - 1, 0), Bb + st * B_BYTES, 1, raw, fb1);
      split(raw, fa1);
      mfma12(fa0, fb0);
      interleave();
      if (cs + 1 < cs1) {
        wait_ring();
        barrier();
        dma_a(cs + 1, 0);
        request(st, false);
        mfma12(fa1, fb1);
        wait_all();
        barrier();
        read_ops(a_addr(a.tap[tap0], 0, 0), Bb + st1 * B_BYTES, 0, raw, fb0);
        split(raw, fa0);
        st = st1;
      } else {
        mfma12(fa1, fb1);
      }
    }
  }
#ifdef B3_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    b3_trace_buf[blockIdx.x * 16 + 14] = tr_wait;
    b3_trace_buf[blockIdx.x * 16 + 15] = tr_req;
  }
#endif
  B3_STAMP(3);
  wait_all();       // the tail's surplus requests: nothing may land in LDS that the next workgroup of this CU owns
  __syncthreads();
  B3_STAMP(4);
#ifdef B3_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    b3_trace_buf[blockIdx.x * 16 + 12] = ((unsigned long long)xcc << 32) | (unsigned)(kz | (h.ksplit << 8) | (S << 16));
  }
#endif

  // ---- split K: partial tiles meet in the last-arriving workgroup (write-through slabs, ticket; as conv.hip) --------
  if (h.ksplit > 1) {
    constexpr int tile_bytes = BM * BN * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (int64_t)Lg * h.ksplit * (BM * BN), 0, h.ksplit * tile_bytes,
                                                                  0x00020000);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[y][4 * qd], acc[y][4 * qd + 1], acc[y][4 * qd + 2], acc[y][4 * qd + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((y * 4 + qd) * NTHR + tid) * 16, kz * tile_bytes, 16);
      }
    // (A 16-byte store reads its data registers a few cycles AFTER it issues.  The accumulators are dead behind this loop, and in
    // one build of this file -- round 5, a different tail behind these stores -- the register allocator handed the first data
    // register of each store to the NEXT store's address: `buffer_store_dwordx4 v[18:21], ...` directly followed by `v_or_b32 v18,
    // 0x1000, v34`, with no wait state from this compiler for gfx950.  The first float of a piece then went out corrupted now and
    // then: O(1) errors in a few elements of a split tile (tools/debug/dist_probe.py found them, tools/isa_store_hazard.py finds the
    // pattern in the assembly).  The accumulators therefore stay live across the stores and a few idle cycles: the address
    // temporaries get other registers.)
    asm volatile("s_nop 7" ::: "memory");
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    B3_STAMP(5);
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(a.cnt + Lg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)(h.ksplit - 1);
      if (last) __hip_atomic_store(a.cnt + Lg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = last ? 1 : 0;
    }
    __syncthreads();
    B3_STAMP(6);
    if (flag[0] == 0) return;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;
    // the slices' tiles are added in slice order (the result does not depend on who arrived last).  This re-read takes 2.0 /
    // 3.9 / 7.6 us for 2 / 4 / 8 slices (tools/debug/b3_trace.py) -- ~1 us per 32 KB slice; several slices in flight together
    // change nothing (measured in round 3): the consumer CU's memory queue, not the round-trip latency, sets the pace
    // (MI355X_MICROARCH.md "handoff-payload": 47-75 GB/s per block at these sizes)
    for (int z = 0; z < h.ksplit; ++z) {
      f32x4 pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * NTHR + tid) * 16, z * tile_bytes, 16));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i / 4][4 * (i % 4)] += pv[i].x;
        acc[i / 4][4 * (i % 4) + 1] += pv[i].y;
        acc[i / 4][4 * (i % 4) + 2] += pv[i].z;
        acc[i / 4][4 * (i % 4) + 3] += pv[i].w;
      }
    }
  }

  // ---- epilogue: the wave's two column blocks; per-channel sums of the four waves meet in LDS ----------------------------
  B3_STAMP(7);
  if (a.contig) {  // through the LDS turn-around (every access a whole 128-byte line per channel row)
    epilogue_t<PRE, POOLT>(a, m0, k0, wave, lane, acc, epi0, epi1, pre, lds, red);
#ifdef B3_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B3_STAMP(8);
#endif
    return;
  }
  float r1[2] = {0.f, 0.f}, r2[2] = {0.f, 0.f};
  epi0.template body<false, false>(L, xb, acc[0], r1[0], r2[0]);
  epi1.template body<false, false>(L, xb, acc[1], r1[1], r2[1]);
#ifdef B3_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  B3_STAMP(8);
#endif
  const bool BWD = d.flags & VITTA_CONV_BWD_BN;
  if (((d.flags & VITTA_CONV_STATS) && d.st_s1) || BWD) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      r1[y] += __shfl_xor(r1[y], 32, 64);
      r2[y] += __shfl_xor(r2[y], 32, 64);
    }
    if (wave > 0 && lk == 0) {
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        red[(((wave - 1) * 2 + y) * 32 + li) * 2] = r1[y];
        red[(((wave - 1) * 2 + y) * 32 + li) * 2 + 1] = r2[y];
      }
    }
    __syncthreads();
    if (wave == 0 && lk == 0) {
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        float s1 = r1[y], s2 = r2[y];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) {
          s1 += red[((w * 2 + y) * 32 + li) * 2];
          s2 += red[((w * 2 + y) * 32 + li) * 2 + 1];
        }
        const int k = k0 + 32 * y + li;
        if (BWD) {
          if (d.dgamma) atomicAdd(d.dgamma + k, s1);
          if (d.dbeta) atomicAdd(d.dbeta + k, s2);
        } else {
          atomicAdd(d.st_s1 + k, s1);
          atomicAdd(d.st_s2 + k, s2);
        }
      }
    }
  }
}
#undef SGB

template <int MODE, bool PRE, bool POOLT = false>
int launch_one(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr int NB = MODE == 1 ? 3 : 2;
  constexpr size_t lds = (size_t)(MODE == 1 ? 1 : NB) * 32 * (MODE == 1 ? 256 : 128) * 4 + NB * 12 * 64 * 16 + 384 * 4 + 16;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b3_kernel<MODE, PRE, POOLT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  const dim3 grid((unsigned)a.hot.nwg), block(256);
  (void)hipGetLastError();
  if (e0) hipExtLaunchKernelGGL((conv_b3_kernel<MODE, PRE, POOLT>), grid, block, lds, st, e0, e1, 0, a);
  else hipLaunchKernelGGL((conv_b3_kernel<MODE, PRE, POOLT>), grid, block, lds, st, a);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

// fp32 [taps][R][O] pack -> [taps][R / 32][3 planes][4 channel octets][O][8] bf16 (16-byte units: one lane per
// (tap, slab, octet, o) writes its three planes)
struct PackB3 {
  const float* src;
  u32x4* dst;
  int64_t first;  // units before this entry (table form)
  int taps, R, O, pad;
};

__device__ __forceinline__ void pack_unit(const PackB3& e, int64_t u) {
  const int o = (int)(u % e.O);
  int64_t r = u / e.O;
  const int g = (int)(r & 3);
  r >>= 2;  // tap * (R / 32) + slab
  const int ncs = e.R / 32, cs = (int)(r % ncs), tap = (int)(r / ncs);
  const float* s = e.src + ((int64_t)tap * e.R + cs * 32 + 8 * g) * e.O + o;
  u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h_, m_, l_;
    split2(s[(2 * j) * (int64_t)e.O], s[(2 * j + 1) * (int64_t)e.O], h_, m_, l_);
    h[j] = h_;
    m[j] = m_;
    l[j] = l_;
  }
  u32x4* dp = e.dst + ((r * 3) * 4 + g) * e.O + o;
  const int64_t plane = 4 * (int64_t)e.O;
  dp[0] = h;
  dp[plane] = m;
  dp[2 * plane] = l;
}

__global__ __launch_bounds__(256) void pack_b3_kernel(const PackB3 e, int64_t units) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u < units) pack_unit(e, u);
}

__global__ __launch_bounds__(256) void pack_b3_table_kernel(const PackB3* __restrict__ tab, int n, int64_t units) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  int lo = 0, hi = n - 1;  // last entry with first <= u
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first <= u) lo = mid;
    else hi = mid - 1;
  }
  const PackB3 e = tab[lo];
  pack_unit(e, u - e.first);
}

}  // namespace

namespace vitta_conv {

int launch_b3(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  if (a.b3 == 2) return a.pw_prefetch ? launch_one<1, true>(a, st, e0, e1) : launch_one<1, false>(a, st, e0, e1);
  if (a.b3 == 3) return a.pw_prefetch ? launch_one<2, true>(a, st, e0, e1) : launch_one<2, false>(a, st, e0, e1);
  // (the pooled-means epilogue is its own instantiation of the pointwise form without an epilogue input stream: conv1 of a bottleneck)
  if ((a.d.flags & VITTA_CONV_POOL) && a.d.pool) return launch_one<0, false, true>(a, st, e0, e1);
  return a.pw_prefetch ? launch_one<0, true>(a, st, e0, e1) : launch_one<0, false>(a, st, e0, e1);
}

}  // namespace vitta_conv

extern "C" {

size_t vitta_conv_pack_b3_bytes(int32_t taps, int32_t R, int32_t O) {
  if (taps <= 0 || R <= 0 || O <= 0 || R % 32) return 0;
  return (size_t)taps * R * O * 6;
}

int vitta_conv_pack_b3(const float* d_src, void* d_dst, int32_t taps, int32_t R, int32_t O, void* stream) {
  if (!d_src || !d_dst || taps <= 0 || R <= 0 || O <= 0) return VITTA_ERR_INVALID_ARG;
  if (R % 32) return VITTA_ERR_UNSUPPORTED;
  const PackB3 e{d_src, static_cast<u32x4*>(d_dst), 0, taps, R, O, 0};
  const int64_t units = (int64_t)taps * (R / 8) * O;
  VITTA_LAUNCH(pack_b3_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), e, units);
  return VITTA_OK;
}

int vitta_conv_pack_b3_table(const vitta_pack_b3_entry* d_table, int32_t n_entries, int64_t total_units, void* stream) {
  static_assert(sizeof(vitta_pack_b3_entry) == sizeof(PackB3), "table entry layout");
  if (!d_table || n_entries <= 0 || total_units <= 0) return VITTA_ERR_INVALID_ARG;
  VITTA_LAUNCH(pack_b3_table_kernel, dim3((unsigned)((total_units + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
               reinterpret_cast<const PackB3*>(d_table), n_entries, total_units);
  return VITTA_OK;
}

}  // extern "C"

#ifdef B3_TRACE
extern "C" int vitta_conv_b3_trace_read(void* h_dst, int64_t bytes, int32_t clear) {
  if (hipDeviceSynchronize() != hipSuccess) return VITTA_ERR_LAUNCH;
  if (h_dst && hipMemcpyFromSymbol(h_dst, HIP_SYMBOL(b3_trace_buf), (size_t)bytes) != hipSuccess) return VITTA_ERR_LAUNCH;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(b3_trace_buf)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 16 * 8192) != hipSuccess)
      return VITTA_ERR_LAUNCH;
  }
  return VITTA_OK;
}
#endif
