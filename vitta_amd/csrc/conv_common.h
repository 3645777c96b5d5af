// Shared between conv.hip (tile-per-workgroup kernels) and conv_sk.hip (persistent stream-K kernel).
#pragma once
#include "common.h"

namespace vitta_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int PRO_MAX = 2048;  // input channels whose prologue BatchNorm constants are held in LDS

// n / d for 0 <= n < 2^31 and a launch constant d >= 1, without a division on the device (an integer division is ~30
// instructions there, a 64-bit one ~150 -- the start of a convolution workgroup had nine of them on its path to the first
// memory request): mul = ceil(2^(31 + l) / d), l = ceil(log2 d); n / d == (n * mul) >> (31 + l), exact for every n < 2^31
// (Granlund-Montgomery with a 31-bit dividend).  sh = l - 1 applies to the HIGH word of the product; sh < 0 marks d == 1.
struct FastDiv {
  unsigned mul;
  int sh;
};
inline FastDiv make_fastdiv(int64_t d) {
  if (d <= 1) return FastDiv{0u, -1};
  int l = 0;
  while ((1ll << l) < d) ++l;
  const uint64_t num = 1ull << (31 + l);  // d < 2^31: l <= 31
  return FastDiv{(unsigned)((num + (uint64_t)d - 1) / (uint64_t)d), l - 1};
}
__device__ __forceinline__ int fdiv(int n, const FastDiv f) { return f.sh < 0 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.sh); }

// conv_b3.hip: what a workgroup needs before its first memory request, as ONE 64-byte block (one scalar load)
struct B3Hot {
  FastDiv d_ks, d_nNt, d_nMt;  // divisions by ksplit, nNt, nMt
  FastDiv d_hw, d_w;           // ... by the pixels of a plane of the M grid (Hg * Wg) and its width (patch / gathered forms, scattered output)
  int nwg;               // workgroups of the launch (the grid)
  int ksplit, nNt, nMt, ncs;
  int flags;             // 1: pixel tiles fastest (nfast), 2: four parity classes of tiles (cls_tiles)
};

struct ConvK {
  vitta_conv_desc d;
  int64_t xP, yP, rP;  // pixels per channel row of x, y, res
  int Mtot;            // N * Hg * Wg
  int nMt, nNt;        // tiles
  int contig;          // 1: output pixel index == M index (float4 epilogue)
  int tap[VITTA_CONV_MAX_TAPS];  // (dh & 0xff) | (dw & 0xff) << 8 | weight slot << 16
  int ksplit;          // workgroups sharing one output tile (each walks 1 / ksplit of the K slabs)
  unsigned* cnt;       // [nMt * nNt] arrival counters (zero at rest)
  float* slabs;        // [nMt * nNt][ksplit][BM * BN] partial accumulators (stream-K: [workgroups][2][BM * BN])
  size_t ws_need;      // host only
  int sk_G;            // stream-K: persistent workgroups (0: the tile-per-workgroup kernels)
  int sk_aligned;      // stream-K: unit ranges end on tile boundaries (no partial tiles)
  int pw_prefetch;     // conv_pw.hip: request the epilogue's residual / BatchNorm-backward input at the tile's start
  int pw;              // 1: conv_pw.hip (pointwise, one workgroup per tile, four workgroups per CU)
  int cls_tiles;       // VITTA_CONV_PARITY4: tiles (nMt * nNt) per parity class; 0: one class
  int cls_tap0[5];     // ... first tap of class c (and the end of the last)
  int nfast;           // conv_b3.hip: logical ids walk pixel tiles fastest (an XCD = all pixels of a few column tiles)
  int b3;              // conv_b3.hip (split-bf16 operands on the bf16 matrix pipe): 1 pointwise, 2 patch (3x3), 3 gathered; 0: not
  B3Hot hot;           // conv_b3.hip: launch constants of the workgroup prologue (hot.nwg == 0: not filled)
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // dispatcher places workgroup b on XCD b % 8: give every XCD a contiguous range of logical ids (bijective for any nwg)
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// conv_pw.hip
int launch_pointwise(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1);

// conv_b3.hip
int launch_b3(const ConvK& a, hipStream_t st, hipEvent_t e0, hipEvent_t e1);

// conv_sk.hip
int launch_stream_k(const ConvK& a, bool gather, hipStream_t st, hipEvent_t e0, hipEvent_t e1);

}  // namespace vitta_conv
