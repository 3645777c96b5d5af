// Shared device/host helpers for libvitta_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitta_hip.h"

#define VITTA_WAVE 64
#define VITTA_BLOCK 256                 // 4 waves: one per SIMD of a CU
#define VITTA_CHUNK (VITTA_BLOCK * 4)   // floats of one plane covered by a workgroup (one float4 per lane)

namespace vitta {

// (count, mean, M2) triple of a set of samples; merge = Chan et al. pairwise update.
struct Moments {
  float n, mean, m2;
};

__device__ __forceinline__ Moments merge(Moments a, Moments b) {
  const float n = a.n + b.n;
  if (n == 0.f) return Moments{0.f, 0.f, 0.f};
  const float d = b.mean - a.mean;
  const float w = b.n / n;
  Moments r;
  r.n = n;
  r.mean = a.mean + d * w;
  r.m2 = a.m2 + b.m2 + d * d * a.n * w;
  return r;
}

__device__ __forceinline__ Moments wave_merge(Moments v) {
#pragma unroll
  for (int m = VITTA_WAVE / 2; m >= 1; m >>= 1) {
    Moments o;
    o.n = __shfl_xor(v.n, m, VITTA_WAVE);
    o.mean = __shfl_xor(v.mean, m, VITTA_WAVE);
    o.m2 = __shfl_xor(v.m2, m, VITTA_WAVE);
    v = merge(v, o);
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = VITTA_WAVE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, VITTA_WAVE);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = VITTA_WAVE / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, VITTA_WAVE));
  return v;
}

// fp64 merge for the tiny cross-block combine.
struct MomentsD {
  double n, mean, m2;
};
__device__ __forceinline__ MomentsD merge(MomentsD a, MomentsD b) {
  const double n = a.n + b.n;
  if (n == 0.0) return MomentsD{0.0, 0.0, 0.0};
  const double d = b.mean - a.mean;
  const double w = b.n / n;
  MomentsD r;
  r.n = n;
  r.mean = a.mean + d * w;
  r.m2 = a.m2 + b.m2 + d * d * a.n * w;
  return r;
}

// Static per-layer record living in device memory (built once by vitta_plan_create).
struct LayerInfo {
  int64_t outer;    // NCHW: frames ; NHWC: rows
  int64_t inner;    // NCHW: H*W    ; NHWC: 1
  int64_t plane;    // NCHW: C*H*W  ; NHWC: C
  int64_t ws_off;   // first partial triple of this layer in the workspace
  int32_t C;
  int32_t layout;
  int32_t chan_off; // offset into the packed per-channel arrays
  int32_t nchunks;  // NCHW: ceil(plane/CHUNK) ; NHWC: channel tiles
  int32_t nsplit;   // splits of the outer dimension
  int32_t slots;    // NCHW: max channels touched by one chunk ; NHWC: C
  int32_t vec;      // 4: float4 path legal (plane % 4 == 0) ; 1: scalar path
  int32_t tx;       // NHWC: lanes across the channel dimension of one workgroup
};

struct BlockEnt {
  int32_t layer, chunk, split, pad;
};

struct PtrPack {
  const void* x[VITTA_MAX_LAYERS];  // fp32 or bf16 features, as the launch's element type says
};

}  // namespace vitta

struct vitta_plan {
  int n_layers = 0;
  int64_t total_channels = 0;
  int64_t ws_triples = 0;
  int n_blocks_nchw = 0, n_blocks_nhwc = 0;
  bool nt_loads = false;
  vitta::LayerInfo h_info[VITTA_MAX_LAYERS];
  // host image of the device tables: [LayerInfo x L | BlockEnt nchw | BlockEnt nhwc | chan2layer | ticket]
  void* h_tables = nullptr;
  size_t table_bytes = 0, off_nchw = 0, off_nhwc = 0, off_c2l = 0, off_ticket = 0;
  // device tables: views into the CALLER-OWNED buffer handed to vitta_plan_upload (never freed here)
  vitta::LayerInfo* d_info = nullptr;
  vitta::BlockEnt* d_tab_nchw = nullptr;
  vitta::BlockEnt* d_tab_nhwc = nullptr;
  int32_t* d_chan2layer = nullptr;
  unsigned* d_ticket = nullptr;  // arrival counter of the alignment launch (uploaded zero, zero at rest)
};

// hipGetLastError() is sticky per host thread: an unrelated earlier runtime call (e.g. a probing call
// of the framework that owns the context) may have left an error behind.  VITTA_LAUNCH clears it,
// launches, and returns VITTA_ERR_LAUNCH from the enclosing ABI function if THIS launch failed.
#define VITTA_LAUNCH(...)                                          \
  do {                                                             \
    (void)hipGetLastError();                                       \
    hipLaunchKernelGGL(__VA_ARGS__);                               \
    if (hipGetLastError() != hipSuccess) return VITTA_ERR_LAUNCH;  \
  } while (0)

