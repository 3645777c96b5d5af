// The trunk's convolutions on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split into three bf16 terms
//   x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (round to nearest even: |lo| <= 2^-16 |x|)
// and a product is accumulated as the six terms of order <= 2,
//   a b ~= a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi      (dropped: <= 2^-23 |a b|),
// each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Six bf16 MFMAs do 16 k of a 32 x 32 block in 6 x 32 cycles where
// the exact-fp32 instruction (v_mfma_f32_32x32x2_f32, conv_sk.hip / conv_pw.hip) needs 8 x 64: 2.67x the matrix rate
// (2.5 PF / 6 = 417 TF of fp32-grade multiply-adds), error of the fp32-roundoff class (tests/test_gpu_conv.py holds both
// arithmetic forms to the same 2e-5-of-max bound against fp64).
// Reference call sites: models/tanet_models/temporal_module.py:85-106, tanet.py:125-150 (as conv.hip).
//
// Same implicit GEMM as conv.hip -- D[p][k] = sum_{tap, c} X[c][src(p, tap)] W[tap][c][k], pixels of all frames on the MFMA row
// axis, output channels on the column axis, the epilogues of conv_epilogue.h -- with a 128 x 64 output tile per workgroup
// (2 x 2 waves of 64 x 32: two 32 x 32 accumulators per wave), K walked tap OUTER in slabs of 32 channels:
//   * the activations are split WHILE THEY ARE STAGED (global -> registers -> three bf16 planes in LDS): a lane owns four
//     consecutive pixels x four channels (pointwise: four 16-byte loads) or one pixel x eight channels (gathered taps), so a
//     pixel's channel group is 8 bytes of a plane and the LDS image is [plane][channel group of 4][pixel row][4 bf16]: the
//     stores are 16 / 8 contiguous bytes per lane, an MFMA operand (8 k of one row) is two conflict-free ds_read_b64;
//   * the weights are split ONCE per weight version into exactly that image ([tap][slab][plane][group][K][4] bf16,
//     vitta_conv_pack_b3): a tile's slab piece is 24 runs of 512 bytes, copied 16 bytes per lane;
//   * two LDS stages (72 KB: two workgroups per CU), the next slab's split + stores and the loads of the slab after next
//     ride between the MFMAs of the current one; one barrier per slab;
//   * launches with few tiles split K over workgroups (partial tiles through the write-through workspace, last arriver
//     reduces, as conv.hip).
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

// PATCH: 3x3 (any stride-1 tap set on a source grid equal to the output grid): the A image of a channel slab is the flat
// pixel range [m0 - 64, m0 + 192) of each channel row (the tile's 128 pixels + a halo that covers every tap shift
// dh * W + dw for W <= 62), loaded ONCE per channel slab; a tap is an address shift of the operand reads, and a lane whose
// tap falls outside the plane reads position 0 of the row instead, which the DMA keeps at zero.
// MODE 0: pointwise, the A image is the tile's 128 pixels, two stages.  MODE 2: any other tap set (strided 3x3, strided
// pointwise, planes wider than the halo): the A image of a (channel slab, tap) step is GATHERED, four bytes per lane
// (one instruction = 64 pixels of one channel row; a pixel whose tap falls outside the plane requests out of range and
// lands as zero), two stages like the pointwise form.
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// NW = waves per workgroup = 32-row blocks of the tile: 4 (128 x 64 tiles, two workgroups per CU) or, pointwise only, 2
// (64 x 64 tiles, 40 KB of LDS: FOUR workgroups per CU -- twice the independent request chains for the same wave count)
#ifdef B3_TRACE
// tools/debug/b3_trace.py: shader-clock stamps of wave 0 at the phase boundaries of a workgroup, 16 words per workgroup
__device__ unsigned long long b3_trace_buf[16 * 8192];
#define B3_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 8192) b3_trace_buf[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define B3_STAMP(i) \
  do {              \
  } while (0)
#endif

// Q (pointwise, four waves): THREE stages of SIXTEEN channels (14 KB each, 44 KB with the epilogue scratch: three workgroups per
// CU) and one barrier per 16-channel step -- behind it every wave has the step's images and has finished reading the stage of
// the step before, which takes the request of the step after next: two steps of requests in flight.  The pipeline of
// gemm_bf16x.hip, where it was worth 14 % against two 64-wide stages at two workgroups per CU.
template <int MODE, bool PRE, int NB, int NW = 4, bool Q = false>
__global__ __launch_bounds__(64 * NW, Q ? 3 : 2) void conv_b3_kernel(const ConvK a) {
  constexpr bool PATCH = MODE == 1, GATHER = MODE == 2;
  static_assert(NW == 4 || (NW == 2 && MODE == 0), "two-wave workgroups: pointwise form only");
  static_assert(!Q || (MODE == 0 && NW == 4 && NB == 3), "Q: the pointwise form with three stages");
  constexpr int BM = 32 * NW, BN = 64, BK = 32, HALO = 64, NTHR = 64 * NW;
  constexpr int PL = PATCH ? 256 : BM;    // pixels per channel row of the A image
  constexpr int NA = PATCH ? 1 : NB;      // A stages (B: NB)
  constexpr int PER_STEP = Q ? 4 : PATCH ? 12 / NW : GATHER ? 16 + 12 / NW : 4 + 12 / NW;  // LDS-DMA instructions of a wave per step
  constexpr int RPW = (Q ? 16 : BK) / NW;  // channel rows of an A stage a wave requests
  constexpr int A_BYTES = (Q ? 16 : BK) * PL * 4, B_BYTES = (Q ? 6 : 12) * BN * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;                        // [NA][BK][PL] fp32
  unsigned char* const Bb = lds + NA * A_BYTES;         // [NB][3 planes][4 channel octets][BN][8] bf16
  float* const red = reinterpret_cast<float*>(Bb + NB * B_BYTES);  // [NW - 1][2][32][2]
  int* const flag = reinterpret_cast<int*>(red + 384);

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int C = d.C, K = d.K;
  B3_STAMP(0);
  const int Lz = xcd_remap(blockIdx.x, gridDim.x);
  const int Lg = Lz / a.ksplit, kz = Lz - Lg * a.ksplit;  // tile of the launch (arrival counter, partial tiles)
  // parity-merged data gradient: the launch holds four classes of tiles, each with its own run of the tap table
  // (class fastest: the classes have 1, 2, 2 and 4 taps -- as four contiguous blocks of ids the XCDs, which own contiguous id
  // ranges, would get one class each: a 4x imbalance between them; interleaved, every XCD holds all four classes of its
  // tiles, which also read the same input pixels)
  const int cls = a.cls_tiles ? (Lg & 3) : 0, Lc = a.cls_tiles ? (Lg >> 2) : Lg;
  // which tile: an XCD owns a contiguous range of logical ids (xcd_remap).  With column tiles fastest (default) that range is a
  // few pixel tiles x ALL output channels -- every XCD pulls the whole weight image through its own L2; where the image is the
  // larger operand (layer 4: 14 MB against 1.6 MB of activations) the host asks for pixel tiles fastest instead (a.nfast):
  // an XCD then covers all pixels of a few column tiles and reads only their share of the weights
  const int L = a.nfast ? (Lc % a.nMt) * a.nNt + Lc / a.nMt : Lc;
  const int tap0 = a.cls_tiles ? a.cls_tap0[cls] : 0, ntaps = a.cls_tiles ? a.cls_tap0[cls + 1] - tap0 : d.ntaps;
  const int m0 = (L / a.nNt) * BM, k0 = (L % a.nNt) * BN;
  const int ncs = C / BK;
  const int cs0 = (int)(((int64_t)ncs * kz) / a.ksplit), cs1 = (int)(((int64_t)ncs * (kz + 1)) / a.ksplit);
  const int S = (cs1 - cs0) * ntaps * (Q ? 2 : 1);  // steps = (channel slab, tap) pairs (Q: 16-channel halves of a slab)
  B3_STAMP(9);

  // wave w owns pixel rows 32 w .. 32 w + 31 and all 64 output channels (two 32 x 32 accumulators)
  TileEpilogue epi0(a, red, NW == 4 ? wave >> 1 : wave, 0, li, lk, BM), epi1(a, red, NW == 4 ? wave >> 1 : wave, 1, li, lk, BM);
  const int xb = NW == 4 ? wave & 1 : 0;
  if (a.cls_tiles) {
    epi0.oa = epi1.oa = cls >> 1;
    epi0.ob = epi1.ob = cls & 1;
  }

  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_b3), 0, 0x7fffffff, 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  const int row_bytes = (int)(a.xP * 4);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  B3_STAMP(10);

  // ---- LDS-DMA side ----------------------------------------------------------------------------------------------------
  // A, patch: one instruction = one channel row (256 pixels; lane 0 out of range: positions 0..3 stay zero); wave w loads
  // rows 8 w .. 8 w + 7.  A, pointwise: one instruction = two channel rows (2 x 128 pixels), four per wave.
  // (pointwise: one instruction = 256 / PL channel rows of PL pixels)
  const int voff_a = PATCH ? (lane == 0 ? OOB : (m0 - HALO + 4 * lane) * 4)
                           : (lane / (PL / 4)) * row_bytes + min(m0 + 4 * (lane % (PL / 4)), a.Mtot - 4) * 4;
  // gathered: the lane's two pixels (lane, lane + 64 of the tile): source offset of tap (0, 0), validity bit per tap
  int g_base[2] = {0, 0};
  unsigned g_valid[2] = {0, 0};
  if constexpr (GATHER) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int m = m0 + lane + 64 * hh;
      const int hw = d.Hg * d.Wg, mm = m < a.Mtot ? m : 0;
      const int n = mm / hw, r = mm - n * hw, gi = r / d.Wg, gj = r - gi * d.Wg;
      g_base[hh] = (n * d.Hs * d.Ws + gi * d.sstride * d.Ws + gj * d.sstride) * 4;
      for (int t = 0; t < ntaps; ++t) {
        const int tp = a.tap[tap0 + t];
        const int sh = gi * d.sstride + (int)(int8_t)(tp & 0xff), sw = gj * d.sstride + (int)(int8_t)((tp >> 8) & 0xff);
        if (m < a.Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws) g_valid[hh] |= 1u << t;
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
    } else if constexpr (Q) {  // cs = 16-channel step: the wave's four channel rows, two per instruction
      const int c16 = cs * 16 + wave * RPW;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 1024), 16, voff_a, (c16 + 2 * i) * row_bytes, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + i * 1024), 16, voff_a, (c0 + (256 / PL) * i) * row_bytes, 0, 0);
    }
  };
  // B: the slab image of this tile's 64 output channels = 12 runs (plane, channel octet) of 64 x 16 bytes, three per wave
  auto dma_b = [&](int cs, int tp, int stage) __attribute__((always_inline)) {
    if constexpr (Q) {
      // cs = 16-channel step: the two octets 2 h, 2 h + 1 of each plane of slab cs / 2 = six runs, LDS order (plane, octet).
      // Waves 0, 1 bring two runs each, waves 2, 3 one each -- issued TWICE, so that every wave has the same number of
      // instructions in flight for the counted waits (same bytes to the same place)
      const int h = cs & 1, r0 = wave < 2 ? 2 * wave : 2 + wave, r1 = wave < 2 ? r0 + 1 : r0;
      const int base = ((tp >> 16) * ncs + (cs >> 1)) * 12 + 2 * h;
      unsigned char* dst = Bb + stage * B_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + r0 * 1024), 16, lane * 16, ((base + (r0 >> 1) * 4 + (r0 & 1)) * K + k0) * 16, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + r1 * 1024), 16, lane * 16, ((base + (r1 >> 1) * 4 + (r1 & 1)) * K + k0) * 16, 0, 0);
      return;
    }
    unsigned char* dst = Bb + stage * B_BYTES + wave * (12 / NW) * 1024;
    const int run0 = (((tp >> 16) * ncs + cs) * 12 + wave * (12 / NW));
#pragma unroll
    for (int u = 0; u < 12 / NW; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + u * 1024), 16, lane * 16, ((run0 + u) * K + k0) * 16, 0, 0);
  };

  // ---- multiplying side ------------------------------------------------------------------------------------------------
  // the lane's row: position in the A image, tap validity bits
  const int pos = (PATCH ? HALO : 0) + 32 * wave + li;
  unsigned valid = 0x1ff;
  if constexpr (PATCH) {
    const int m = m0 + 32 * wave + li;
    const int hw = d.Hs * d.Ws, mm = m < a.Mtot ? m : 0;
    const int r = mm % hw, h = r / d.Ws, w = r - h * d.Ws;
    valid = 0;
    for (int t = 0; t < ntaps; ++t) {
      const int tp = a.tap[tap0 + t];
      const int sh = h + (int)(int8_t)(tp & 0xff), sw = w + (int)(int8_t)((tp >> 8) & 0xff);
      if (m < a.Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws) valid |= 1u << t;
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
  float4 pre[2][4] = {};
  // requests run NB steps ahead: (cs_q, t_q) = the step to request next, clamped to the slice's last step (the tail
  // re-requests it into a stage nobody reads again: the instruction count per step stays fixed for the counted waits)
  int cs_q = Q ? 2 * cs0 : cs0, t_q = 0, q = 0;
  auto request = [&](int stage, bool with_a) __attribute__((always_inline)) {
#ifdef B3_ABL_NODMA
    if (q >= NB) return;
#endif
    dma_b(cs_q, a.tap[tap0 + t_q], stage);
    if constexpr (!PATCH) {
      if (with_a) dma_a(cs_q, stage, a.tap[tap0 + t_q], t_q);
    }
    const bool adv = q + 1 < S;
    q += adv ? 1 : 0;
    const bool wrap = adv && t_q + 1 == ntaps;
    t_q = adv ? (wrap ? 0 : t_q + 1) : t_q;
    cs_q += wrap ? 1 : 0;
  };
  if constexpr (Q) {
    request(0, true);
    request(1, true);
  } else {
    dma_a(cs0, 0, a.tap[tap0], 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) request(i, i > 0);
  }
  B3_STAMP(13);
  epi0.load_consts(L);
  epi1.load_consts(L);
  if constexpr (PRE) {
#pragma unroll
    for (int y = 0; y < 2; ++y) tile_prefetch(a, L, NW == 4 ? wave >> 1 : wave, y, li, lk, pre[y][0], pre[y][1], pre[y][2], pre[y][3], BM, xb);
  }
  if constexpr (Q) {
    float raw[8];
    bf16x8 fa[3], fb[2][3];
    int st = 0;
    B3_STAMP(1);
    for (int s = 0; s < S; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");  // step s has landed (step s + 1 may be in flight)
      barrier();
      if (s == 0) B3_STAMP(2);
      request(st >= 1 ? st - 1 : 2, true);  // step s + 2 into the stage of step s - 1
      const float* ap = reinterpret_cast<const float*>(Ab + st * A_BYTES + a_lane);
      const unsigned char* bs_ = Bb + st * B_BYTES;
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = ap[j * PL];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int y = 0; y < 2; ++y) fb[y][p] = *reinterpret_cast<const bf16x8*>(bs_ + b_lane + (p * 2 * BN + 32 * y) * 16);
      split(raw, fa);
      mfma12(fa, fb);
      st = st == 2 ? 0 : st + 1;
    }
  } else {
  // step 0 (requested first) has landed; loads the compiler placed behind the requests only make this wait longer
  B3_STAMP(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * PER_STEP) : "memory");
  barrier();
  B3_STAMP(2);
  float raw[8];
  bf16x8 fa0[3], fa1[3], fb0[2][3], fb1[2][3];
  read_ops(a_addr(a.tap[tap0], 0, 0), Bb, 0, raw, fb0);
  split(raw, fa0);
  int st = 0;  // stage of the current step
  // one step whose successor's images are (or will be, after the barrier) in the ring: both k-steps covered
  auto full_step = [&](int tp, int t, int tp1, int t1) __attribute__((always_inline)) {
    const int st1 = st + 1 == NB ? 0 : st + 1;
    read_ops(a_addr(tp, t, PATCH ? 0 : st), Bb + st * B_BYTES, 1, raw, fb1);
    split(raw, fa1);
    mfma12(fa0, fb0);
    interleave();
    wait_ring();
    barrier();
    request(st, true);
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
  }
  B3_STAMP(3);
  wait_all();       // the tail's surplus requests: nothing may land in LDS that the next workgroup of this CU owns
  __syncthreads();
  B3_STAMP(4);
#ifdef B3_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    b3_trace_buf[blockIdx.x * 16 + 12] = ((unsigned long long)xcc << 32) | (unsigned)(kz | (a.ksplit << 8) | (S << 16));
  }
#endif

  // ---- split K: partial tiles meet in the last-arriving workgroup (write-through slabs, ticket; as conv.hip) --------
  if (a.ksplit > 1) {
    constexpr int tile_bytes = BM * BN * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (int64_t)Lg * a.ksplit * (BM * BN), 0, a.ksplit * tile_bytes,
                                                                  0x00020000);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[y][4 * qd], acc[y][4 * qd + 1], acc[y][4 * qd + 2], acc[y][4 * qd + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((y * 4 + qd) * NTHR + tid) * 16, kz * tile_bytes, 16);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    B3_STAMP(5);
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(a.cnt + Lg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)(a.ksplit - 1);
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
    // 3.9 / 7.6 us for 2 / 4 / 8 slices (tools/debug/b3_trace.py) -- ~1 us per 32 KB slice; B3_RU slices in flight together
    // change nothing (launch totals 1.427 / 1.421 / 1.448 ms for 1 / 2 / 4): the consumer CU's memory queue, not the
    // round-trip latency, sets the pace (MI355X_MICROARCH.md "handoff-payload": 47-75 GB/s per block at these sizes)
#ifndef B3_RU
#define B3_RU 1
#endif
    for (int z0 = 0; z0 < a.ksplit; z0 += B3_RU) {
      f32x4 pv[B3_RU][8];
#pragma unroll
      for (int u = 0; u < B3_RU; ++u) {
        const int z = min(z0 + u, a.ksplit - 1);  // (past the end: a valid address, the value is not added)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pv[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * NTHR + tid) * 16, z * tile_bytes, 16));
      }
#pragma unroll
      for (int u = 0; u < B3_RU; ++u) {
        if (z0 + u < a.ksplit) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[i / 4][4 * (i % 4)] += pv[u][i].x;
            acc[i / 4][4 * (i % 4) + 1] += pv[u][i].y;
            acc[i / 4][4 * (i % 4) + 2] += pv[u][i].z;
            acc[i / 4][4 * (i % 4) + 3] += pv[u][i].w;
          }
        }
      }
    }
  }

  // ---- epilogue: the wave's two column blocks; per-channel sums of the four waves meet in LDS ----------------------------
  B3_STAMP(7);
  float r1[2] = {0.f, 0.f}, r2[2] = {0.f, 0.f};
  epi0.template body<PRE>(L, xb, acc[0], r1[0], r2[0], pre[0][0], pre[0][1], pre[0][2], pre[0][3]);
  epi1.template body<PRE>(L, xb, acc[1], r1[1], r2[1], pre[1][0], pre[1][1], pre[1][2], pre[1][3]);
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
// ---- pointwise launches with more tiles than resident workgroups: persistent form ----------------------------------------
// A pointwise tile of a short-K layer is two to eight steps: with one tile per workgroup the launch is a chain of
// request latency -> a few steps -> epilogue per workgroup, three rounds of it on the 64 -> 256 layer at 56 x 56 (1568 tiles on
// 512 resident workgroups, 23 us with the output stores removed).  Here a workgroup owns a contiguous range of tiles and the
// request ring simply runs on across tile boundaries: the next tile's first images are in flight under the current tile's
// last MFMAs and land during its epilogue; the operand reads never depend on the tile (a wave's rows are rows 32 w .. of
// whatever the stage holds), only the request addresses and the epilogue do.
template <bool PRE>
__global__ __launch_bounds__(256, 2) void conv_b3p_kernel(const ConvK a) {
  constexpr int BM = 128, BN = 64, BK = 32, PL = 128, NB = 2;
  constexpr int A_BYTES = BK * PL * 4, B_BYTES = 12 * BN * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;
  unsigned char* const Bb = lds + NB * A_BYTES;
  float* const red = reinterpret_cast<float*>(Bb + NB * B_BYTES);

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int C = d.C, K = d.K, ncs = C / BK;
  const int tiles = a.nMt * a.nNt, G = (int)gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  const int T0 = (int)((int64_t)g * tiles / G), T1 = (int)((int64_t)(g + 1) * tiles / G);
  if (T1 <= T0) return;
  const int S = (T1 - T0) * ncs;  // steps of this workgroup = (tile, channel slab)
  const int wslot = a.tap[0] >> 16;

  TileEpilogue epi0(a, red, wave >> 1, 0, li, lk, BM), epi1(a, red, wave >> 1, 1, li, lk, BM);
  const int xb = wave & 1;
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_b3), 0, 0x7fffffff, 0x00020000);
  const int row_bytes = (int)(a.xP * 4);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- requests: (tile Lq, channel slab csq), clamped to the range's last step --------------------------------------------
  int q = 0, Lq = T0, csq = 0;
  auto tile_voff = [&](int Lx) __attribute__((always_inline)) { return lk * row_bytes + min((Lx / a.nNt) * BM + 4 * li, a.Mtot - 4) * 4; };
  int voff_q = tile_voff(Lq);
  auto request = [&](int stage) __attribute__((always_inline)) {
    const int k0q = (Lq % a.nNt) * BN;
    unsigned char* db = Bb + stage * B_BYTES + wave * 3 * 1024;
    const int run0 = (wslot * ncs + csq) * 12 + wave * 3;
#pragma unroll
    for (int u = 0; u < 3; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(db + u * 1024), 16, lane * 16, ((run0 + u) * K + k0q) * 16, 0, 0);
    unsigned char* da = Ab + stage * A_BYTES + wave * 8 * PL * 4;
    const int c0 = csq * BK + wave * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(da + i * 1024), 16, voff_q, (c0 + 2 * i) * row_bytes, 0, 0);
    if (q + 1 < S) {
      ++q;
      if (++csq == ncs) {
        csq = 0;
        ++Lq;
        voff_q = tile_voff(Lq);
      }
    }
  };

  // ---- operands (tile independent) ---------------------------------------------------------------------------------------------
  const int a_lane = (8 * lk * PL + 32 * wave + li) * 4, b_lane = (lk * BN + li) * 16;
  f32x16 acc[2];
  auto read_ops = [&](int stage, int ks, float (&raw)[8], bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
    const float* ap = reinterpret_cast<const float*>(Ab + stage * A_BYTES + a_lane);
    const unsigned char* bs_ = Bb + stage * B_BYTES;
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = ap[(16 * ks + j) * PL];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int y = 0; y < 2; ++y) fb[y][p] = *reinterpret_cast<const bf16x8*>(bs_ + b_lane + ((p * 4 + 2 * ks) * BN + 32 * y) * 16);
  };
  auto split = [&](const float (&raw)[8], bf16x8 (&fa)[3]) __attribute__((always_inline)) {
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
  auto mfma12 = [&](const bf16x8 (&fa)[3], const bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
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
  auto interleave = [&]() __attribute__((always_inline)) {
    SGB(0x100, 10);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      SGB(0x008, 1);
      SGB(0x002, 4);
    }
  };
  auto barrier_all_landed = [&]() __attribute__((always_inline)) {  // NB = 2: every request of this wave has landed
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  request(0);
  request(1);
  barrier_all_landed();
  float raw[8];
  bf16x8 fa0[3], fa1[3], fb0[2][3], fb1[2][3];
  read_ops(0, 0, raw, fb0);
  split(raw, fa0);
  int st = 0, s = 0;
  for (int L = T0; L < T1; ++L) {
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;
    epi0.load_consts(L);
    epi1.load_consts(L);
    float4 pre[2][4] = {};
    if constexpr (PRE) {
#pragma unroll
      for (int y = 0; y < 2; ++y) tile_prefetch(a, L, wave >> 1, y, li, lk, pre[y][0], pre[y][1], pre[y][2], pre[y][3], BM, xb);
    }
    for (int cs = 0; cs < ncs; ++cs, ++s) {
      read_ops(st, 1, raw, fb1);
      split(raw, fa1);
      mfma12(fa0, fb0);
      interleave();
      if (s + 1 < S) {  // the next step (of this tile or the next) is in the other stage once the barrier is passed
        barrier_all_landed();
        request(st);
        read_ops(st ^ 1, 0, raw, fb0);
        split(raw, fa0);
        mfma12(fa1, fb1);
        interleave();
        st ^= 1;
      } else {
        mfma12(fa1, fb1);
      }
    }
    float r1[2] = {0.f, 0.f}, r2[2] = {0.f, 0.f};
    epi0.template body<PRE>(L, xb, acc[0], r1[0], r2[0], pre[0][0], pre[0][1], pre[0][2], pre[0][3]);
    epi1.template body<PRE>(L, xb, acc[1], r1[1], r2[1], pre[1][0], pre[1][1], pre[1][2], pre[1][3]);
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
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wave == 0 && lk == 0) {
        const int k0 = (L % a.nNt) * BN;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          float s1 = r1[y], s2 = r2[y];
#pragma unroll
          for (int w = 0; w < 3; ++w) {
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
      // (`red` is written again only after the next tile's barriers)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool PRE>
int launch_persistent(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr size_t lds = (size_t)2 * 32 * 128 * 4 + 2 * 12 * 64 * 16 + 384 * 4 + 16;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b3p_kernel<PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  const dim3 grid((unsigned)a.sk_G), block(256);
  (void)hipGetLastError();
  if (e0) hipExtLaunchKernelGGL((conv_b3p_kernel<PRE>), grid, block, lds, st, e0, e1, 0, a);
  else hipLaunchKernelGGL((conv_b3p_kernel<PRE>), grid, block, lds, st, a);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

// ---- 128 x 128 tiles ("wide"): a wave = 32 pixel rows x 128 output channels, four accumulators --------------------------
// The split of the activations (44 vector instructions per 32 rows x 16 channels) is the same whatever the tile's width:
// with four column blocks it feeds 24 MFMAs (768 matrix-pipe cycles) instead of 12, and the step overheads (requests,
// address arithmetic, barrier) halve per MFMA as well.  Step = ONE k-step of 16 channels (x one tap), one barrier per step:
//   barrier X_s (step s + 1's images have landed, counted vmcnt) -> request step s + NB into the stage step s has left
//   (its operands sit in registers) -> operand reads of step s + 1 -> the 24 MFMAs of step s with the split of step s + 1
//   between them.
// Images: A = the 16 channel rows of the step (pointwise, ring of NB stages) or the 32-row patch of a channel slab
// (pixels [m0 - 32, m0 + 160): every tap shift |dh W + dw| <= 32, i.e. W <= 31; one stage, rows out of the plane masked
// per lane); B = [3 planes][2 channel octets][128 output channels][8] bf16 = 12 KB per step, ring of NB stages.
// The nine per-channel epilogue constants wait in LDS, not in registers.
template <int PATCH_I>
__global__ __launch_bounds__(256, 2) void conv_b3w_kernel(const ConvK a) {
  constexpr bool PATCH = PATCH_I != 0;
  constexpr int BM = 128, BN = 128, NY = 4, HALO = 32;
  constexpr int PL = PATCH ? 192 : 128;
  constexpr int NB = PATCH ? 4 : 3;
  constexpr int A_STAGE = PATCH ? 32 * PL * 4 : 16 * PL * 4, NA = PATCH ? 1 : NB;
  constexpr int B_STAGE = 3 * 2 * BN * 16;
  constexpr int PER_STEP = PATCH ? 3 : 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ab = lds;
  unsigned char* const Bb = lds + NA * A_STAGE;
  float* const cst = reinterpret_cast<float*>(Bb + NB * B_STAGE);  // [9][BN]
  float* const red = cst + 9 * BN;                                  // [3][NY][32][2]
  int* const flag = reinterpret_cast<int*>(red + 3 * NY * 64);

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int C = d.C, K = d.K;
  const int Lz = xcd_remap(blockIdx.x, gridDim.x);
  const int L = Lz / a.ksplit, kz = Lz - L * a.ksplit;
  const int m0 = (L / a.nNt) * BM, k0 = (L % a.nNt) * BN;
  const int ncs = C / 32, ntaps = d.ntaps;
  const int cs0 = (int)(((int64_t)ncs * kz) / a.ksplit), cs1 = (int)(((int64_t)ncs * (kz + 1)) / a.ksplit);
  const int S = (cs1 - cs0) * ntaps * 2;  // steps = (channel slab, tap, k-step)

  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)((int64_t)C * a.xP * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_b3), 0, 0x7fffffff, 0x00020000);
  const int row_bytes = (int)(a.xP * 4);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- requests (LDS-DMA) ------------------------------------------------------------------------------------------------
  // patch: 32 rows x 48 sixteen-byte units = 24 instructions, six per wave; unit u -> row u / 48, pixels 4 (u % 48) ..
  int voff_p[6] = {};  // (fixed bounds: a bound that depends on the template argument, captured by the lambdas below, makes
                       // hipcc drop the host stub of the instantiation silently)
  int row0_p[6] = {};  // first row an instruction touches (wave-uniform): the scalar part of its address
  if constexpr (PATCH) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int u0 = (wave * 6 + i) * 64, u = u0 + lane;
      row0_p[i] = u0 / 48;
      voff_p[i] = (u / 48 - row0_p[i]) * row_bytes + (m0 - HALO + 4 * (u % 48)) * 4;
    }
  }
  auto dma_patch = [&](int cs) __attribute__((always_inline)) {
    if constexpr (PATCH) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(Ab + (wave * 6 + i) * 1024), 16, voff_p[i], (cs * 32 + row0_p[i]) * row_bytes,
                                                 0, 0);
    }
  };
  // pointwise A: 16 rows x 512 bytes = 8 instructions (two rows each), two per wave
  const int voff_a = lk * row_bytes + min(m0 + 4 * li, a.Mtot - 4) * 4;
  // B: 6 runs (plane, octet) of 128 x 16 bytes = 12 instructions, three per wave: instruction 3 w + u -> run, half
  int b_soff[3], b_dst[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int i = wave * 3 + u, run = i >> 1, half = i & 1;  // run = plane * 2 + j
    b_soff[u] = (((run >> 1) * 4 + (run & 1)) * K + k0 + 64 * half) * 16;
    b_dst[u] = run * 2048 + half * 1024;
  }
  // step q -> (channel slab, tap, k-step); the request index runs NB ahead and stops at the slice's last step (re-requested)
  int q = 0, q_cs = cs0, q_t = 0, q_ks = 0;
  auto request = [&](int stage) __attribute__((always_inline)) {
    const int base = ((((a.tap[q_t] >> 16) * ncs + q_cs) * 12 + 2 * q_ks) * K) * 16;
    unsigned char* dst = Bb + stage * B_STAGE;
#pragma unroll
    for (int u = 0; u < 3; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + b_dst[u]), 16, lane * 16, base + b_soff[u], 0, 0);
    if constexpr (!PATCH) {
      unsigned char* da = Ab + stage * A_STAGE + wave * 4 * PL * 4;
      const int c0 = q_cs * 32 + q_ks * 16 + wave * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(da + u * 1024), 16, voff_a, (c0 + 2 * u) * row_bytes, 0, 0);
    }
    const bool adv = q + 1 < S;
    q += adv ? 1 : 0;
    const bool w1 = adv && q_ks == 1;
    q_ks = adv ? (q_ks ^ 1) : q_ks;
    const bool w2 = w1 && q_t + 1 == ntaps;
    q_t = w1 ? (w2 ? 0 : q_t + 1) : q_t;
    q_cs += w2 ? 1 : 0;
  };

  // ---- operands ----------------------------------------------------------------------------------------------------------
  const int pos = (PATCH ? HALO : 0) + 32 * wave + li;
  unsigned valid = 0x1ff;
  if constexpr (PATCH) {
    const int m = m0 + 32 * wave + li;
    const int hw = d.Hs * d.Ws, mm = m < a.Mtot ? m : 0;
    const int r = mm % hw, h = r / d.Ws, w = r - h * d.Ws;
    valid = 0;
    for (int t = 0; t < ntaps; ++t) {
      const int tp = a.tap[t];
      const int sh = h + (int)(int8_t)(tp & 0xff), sw = w + (int)(int8_t)((tp >> 8) & 0xff);
      if (m < a.Mtot && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws) valid |= 1u << t;
    }
  }
  const int a_lane = (8 * lk * PL + pos) * 4;
  const int b_lane = (lk * BN + li) * 16;

  f32x16 acc[NY];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;

  // operand reads of the step whose A rows start at `ap` (lane address incl. tap shift) and whose B image is `bs_`
  auto read_a = [&](const float* ap, bool ok, float (&raw)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = ap[j * PL];
    if constexpr (PATCH) {
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = ok ? raw[j] : 0.f;
    }
  };
  auto read_b = [&](const unsigned char* bs_, bf16x8 (&fb)[NY][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[y][p] = *reinterpret_cast<const bf16x8*>(bs_ + b_lane + (p * 2 * BN + 32 * y) * 16);
  };
  auto split = [&](const float (&raw)[8], bf16x8 (&fa)[3]) __attribute__((always_inline)) {
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
  auto mfma24 = [&](const bf16x8 (&fa)[3], const bf16x8 (&fb)[NY][3]) __attribute__((always_inline)) {
#define B3_MFMA(PA, PB)                                                                                  \
  _Pragma("unroll") for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA], fb[y][PB], acc[y], 0, 0, 0)
    B3_MFMA(2, 0);
    B3_MFMA(0, 2);
    B3_MFMA(1, 1);
    B3_MFMA(1, 0);
    B3_MFMA(0, 1);
    B3_MFMA(0, 0);
#undef B3_MFMA
  };
  // A address of the lane for (tap word, k-step) in `stage`
  auto a_ptr = [&](int tp, int ks, int stage) __attribute__((always_inline)) -> const float* {
    if constexpr (PATCH) {
      const int sh = (int)(int8_t)(tp & 0xff) * d.Ws + (int)(int8_t)((tp >> 8) & 0xff);
      return reinterpret_cast<const float*>(Ab + a_lane + (sh + 16 * ks * PL) * 4);
    } else {
      return reinterpret_cast<const float*>(Ab + stage * A_STAGE + a_lane);
    }
  };
  auto wait_ring = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PER_STEP) : "memory");
  };
  auto wait_all = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // 24 MFMAs with the next step's reads in front and its split between them
  auto interleave = [&]() __attribute__((always_inline)) {
    SGB(0x100, 16);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      SGB(0x008, 1);
      SGB(0x002, 3);
    }
  };

  // ---- pipeline ----------------------------------------------------------------------------------------------------------
  dma_patch(cs0);
#pragma unroll
  for (int i = 0; i < NB; ++i) request(i);
  if (tid < BN) TileEpilogue::stage_consts(a, L, BN, cst, tid);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * PER_STEP) : "memory");  // step 0 has landed (later loads only lengthen the wait)
  barrier();
  float raw[8];
  bf16x8 fa0[3], fa1[3], fb0[NY][3], fb1[NY][3];
  int tp = a.tap[0];
  read_a(a_ptr(tp, 0, 0), valid & 1, raw);
  read_b(Bb, fb0);
  split(raw, fa0);
  int st = 0, t = 0, ks = 0, cs = cs0;
  // step s in (fa0, fb0); two steps per trip so that the register sets alternate by name
  auto step = [&](bf16x8 (&fa_c)[3], bf16x8 (&fb_c)[NY][3], bf16x8 (&fa_n)[3], bf16x8 (&fb_n)[NY][3], bool last) __attribute__((always_inline)) {
    if (last) {
      mfma24(fa_c, fb_c);
      return;
    }
    // successor of step (cs, t, ks)
    int t1 = t, ks1 = ks ^ 1, cs_n = cs;
    if (ks == 1) {
      t1 = t + 1;
      if (t1 == ntaps) {
        t1 = 0;
        ++cs_n;
      }
    }
    const int st1 = st + 1 == NB ? 0 : st + 1;
    const int tp1 = a.tap[t1];
    const bool new_patch = PATCH && cs_n != cs;
    wait_ring();
    barrier();
    request(st);
    if (!new_patch) {
      read_a(a_ptr(tp1, ks1, st1), (valid >> t1) & 1, raw);
      read_b(Bb + st1 * B_STAGE, fb_n);
      split(raw, fa_n);
      mfma24(fa_c, fb_c);
      interleave();
    } else {
      dma_patch(cs_n);
      mfma24(fa_c, fb_c);
      wait_all();
      barrier();
      read_a(a_ptr(tp1, ks1, st1), (valid >> t1) & 1, raw);
      read_b(Bb + st1 * B_STAGE, fb_n);
      split(raw, fa_n);
    }
    t = t1;
    ks = ks1;
    cs = cs_n;
    st = st1;
  };
  for (int s = 0; s < S; s += 2) {
    step(fa0, fb0, fa1, fb1, s + 1 >= S);
    if (s + 1 < S) step(fa1, fb1, fa0, fb0, s + 2 >= S);
  }
  wait_all();
  __syncthreads();

  // ---- split K (as conv.hip: write-through partial tiles, ticket, last arriver reduces) ----------------------------------
  if (a.ksplit > 1) {
    constexpr int tile_bytes = BM * BN * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (int64_t)L * a.ksplit * (BM * BN), 0, a.ksplit * tile_bytes,
                                                                  0x00020000);
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[y][4 * qd], acc[y][4 * qd + 1], acc[y][4 * qd + 2], acc[y][4 * qd + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((y * 4 + qd) * 256 + tid) * 16, kz * tile_bytes, 16);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(a.cnt + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool lastw = ticket == (unsigned)(a.ksplit - 1);
      if (lastw) __hip_atomic_store(a.cnt + L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = lastw ? 1 : 0;
    }
    __syncthreads();
    if (flag[0] == 0) return;
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[y][v] = 0.f;
    for (int z = 0; z < a.ksplit; ++z) {
#pragma unroll
      for (int i = 0; i < NY * 4; ++i) {
        const f32x4 pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 256 + tid) * 16, z * tile_bytes, 16));
        acc[i / 4][4 * (i % 4)] += pv.x;
        acc[i / 4][4 * (i % 4) + 1] += pv.y;
        acc[i / 4][4 * (i % 4) + 2] += pv.z;
        acc[i / 4][4 * (i % 4) + 3] += pv.w;
      }
    }
  }

  // ---- epilogue: four column blocks, constants from LDS; per-channel sums of the four waves meet in LDS ---------------------
  float r1[NY], r2[NY];
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    r1[y] = r2[y] = 0.f;
    TileEpilogue epi(a, red, wave >> 1, y, li, lk, BM, BN);
    epi.consts_from_lds(cst);
    epi.template body<false>(L, wave & 1, acc[y], r1[y], r2[y]);
  }
  const bool BWD = d.flags & VITTA_CONV_BWD_BN;
  if (((d.flags & VITTA_CONV_STATS) && d.st_s1) || BWD) {
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      r1[y] += __shfl_xor(r1[y], 32, 64);
      r2[y] += __shfl_xor(r2[y], 32, 64);
    }
    if (wave > 0 && lk == 0) {
#pragma unroll
      for (int y = 0; y < NY; ++y) {
        red[(((wave - 1) * NY + y) * 32 + li) * 2] = r1[y];
        red[(((wave - 1) * NY + y) * 32 + li) * 2 + 1] = r2[y];
      }
    }
    __syncthreads();
    if (wave == 0 && lk == 0) {
#pragma unroll
      for (int y = 0; y < NY; ++y) {
        float s1 = r1[y], s2 = r2[y];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          s1 += red[((w * NY + y) * 32 + li) * 2];
          s2 += red[((w * NY + y) * 32 + li) * 2 + 1];
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

int launch_wide(const ConvK& a, bool patch, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = (size_t)(patch ? 32 * 192 * 4 : 3 * 16 * 128 * 4) + (patch ? 4 : 3) * 3 * 2 * 128 * 16 + 9 * 128 * 4 + 3 * 4 * 64 * 4 + 16;
  void (*const kp)(ConvK) = conv_b3w_kernel<1>;
  void (*const kf)(ConvK) = conv_b3w_kernel<0>;
  static bool raised[2] = {false, false};
  if (!raised[patch]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(patch ? kp : kf), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised[patch] = true;
  }
  const dim3 grid((unsigned)(a.nMt * a.nNt * a.ksplit)), block(256);
  (void)hipGetLastError();
  if (patch) {
    if (e0) hipExtLaunchKernelGGL(conv_b3w_kernel<1>, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(conv_b3w_kernel<1>, grid, block, lds, st, a);
  } else {
    if (e0) hipExtLaunchKernelGGL(conv_b3w_kernel<0>, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(conv_b3w_kernel<0>, grid, block, lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

#undef SGB

template <int MODE, bool PRE, int NW = 4, bool Q = false>
int launch_one(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr int NB = Q ? 3 : MODE == 1 ? 3 : 2;
  constexpr size_t lds = Q ? (size_t)3 * (16 * 128 * 4 + 6 * 64 * 16) + 384 * 4 + 16
                           : (size_t)(MODE == 1 ? 1 : NB) * 32 * (MODE == 1 ? 256 : 32 * NW) * 4 + NB * 12 * 64 * 16 + 384 * 4 + 16;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b3_kernel<MODE, PRE, NB, NW, Q>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  const dim3 grid((unsigned)(a.nMt * a.nNt * a.ksplit * (a.cls_tiles ? 4 : 1))), block(64 * NW);
  (void)hipGetLastError();
  if (e0) hipExtLaunchKernelGGL((conv_b3_kernel<MODE, PRE, NB, NW, Q>), grid, block, lds, st, e0, e1, 0, a);
  else hipLaunchKernelGGL((conv_b3_kernel<MODE, PRE, NB, NW, Q>), grid, block, lds, st, a);
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
  if (a.sk_G > 0) return a.pw_prefetch ? launch_persistent<true>(a, st, e0, e1) : launch_persistent<false>(a, st, e0, e1);
  if ((a.d.tile & 0xffff) == 128) return launch_wide(a, a.b3 == 2, st, e0, e1);
  if (a.b3 == 2) return a.pw_prefetch ? launch_one<1, true>(a, st, e0, e1) : launch_one<1, false>(a, st, e0, e1);
  if (a.b3 == 3) return a.pw_prefetch ? launch_one<2, true>(a, st, e0, e1) : launch_one<2, false>(a, st, e0, e1);
  if ((a.d.tile >> 16) == 64) return a.pw_prefetch ? launch_one<0, true, 2>(a, st, e0, e1) : launch_one<0, false, 2>(a, st, e0, e1);
  if (a.q) return a.pw_prefetch ? launch_one<0, true, 4, true>(a, st, e0, e1) : launch_one<0, false, 4, true>(a, st, e0, e1);
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
