// The ResNet stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels; torchvision ResNet.conv1 under
// models/tanet_models/tanet.py:125-150) as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
//
//   D[k][q] = sum_t Wp[t][k] * X[c(t)][2 oh(q) + kh(t) - 3][2 ow(q) + kw(t) - 3],   t = (c, kh, kw), 147 taps (+1 zero row)
//
// Output CHANNELS sit on the MFMA row axis (A operand = weights), output PIXELS on the column axis (B operand = the input
// patch): a lane of the 32x32 accumulator owns ONE pixel column and 16 channel rows, so every store instruction writes 32
// consecutive pixels of a channel plane = one 128-byte line per half-wave, and the result lands in the reference's NCHW
// layout (what stem.hip's BN + ReLU + max-pool pass reads).
// A workgroup = 256 consecutive output pixels of one frame (linear index q = oh * OW + ow, so any OW works: 112 -> 49
// tiles per frame) x all 64 channels; 4 waves = 2 channel halves x 2 pixel halves, 4 accumulator tiles each.
// LDS: the packed weights [148][64] and the input patch [3][PR][PW] of the rows the tile touches (zero padded, patch
// column 0 = input column -4 so that the 16-byte staging loads are aligned).  The B operand of pixel q, tap t is
// patch[off(t) + lane_off(q)]: the lane part is fixed for the whole K walk, the tap part is wave-uniform; the stride-2
// pixel walk maps 32 lanes onto 32 distinct even banks (the other half-wave reads tap t + 1: odd banks).
#include "common.h"

using namespace vitta;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 148;    // 3 * 7 * 7 taps + one zero row (the MFMA consumes taps in pairs)
constexpr int NPIX = 256;  // output pixels per workgroup

struct StemConvArgs {
  const float* x;   // [N][3][H][W]
  const float* wp;  // [KT][64]
  float* y;         // [N][64][OH][OW]
  int H, W, OH, OW;
  int PR, PW;       // patch rows, row pitch in floats (PW % 4 == 0)
  int tiles;        // tiles per frame
};

__device__ __forceinline__ int tap_off(int t, int prpw, int pw) {
  t = t < 147 ? t : 146;  // the zero row multiplies a finite in-patch value
  const int c = t / 49, r = t - c * 49, kh = r / 7, kw = r - kh * 7;
  return c * prpw + kh * pw + kw;
}

__global__ __launch_bounds__(256) void stem_conv7_kernel(const StemConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;               // [KT][64]
  float* patch = lds + KT * 64;  // [3][PR][PW]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, kk = lane >> 5;
  const int wm = wave & 1, wg = wave >> 1;
  const int n = blockIdx.x / a.tiles, tile = blockIdx.x - n * a.tiles;
  const int q0 = tile * NPIX, npix = a.OH * a.OW;
  const int r0 = q0 / a.OW, ih0 = 2 * r0 - 3;

  for (int i = tid; i < KT * 16; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(a.wp)[i];
  const int pw4 = a.PW >> 2, rows4 = a.PR * pw4, total4 = 3 * rows4;
  const float* xn = a.x + (int64_t)n * 3 * a.H * a.W;
  for (int i = tid; i < total4; i += 256) {
    const int c = i / rows4, rem = i - c * rows4, pr = rem / pw4, p4 = rem - pr * pw4;
    const int ih = ih0 + pr, iw = p4 * 4 - 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
      v = *reinterpret_cast<const f32x4*>(xn + ((int64_t)c * a.H + ih) * a.W + iw);
    reinterpret_cast<f32x4*>(patch)[i] = v;
  }

  // lane part of the patch offset of this lane's pixel in each of its four 32-pixel groups
  int loff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    int q = q0 + 32 * (wg * 4 + g) + li;
    q = q < npix ? q : npix - 1;
    const int r = q / a.OW, col = q - r * a.OW;
    loff[g] = (r - r0) * 2 * a.PW + 2 * col + 1;  // + 1: patch column 0 is input column -4, tap kw = 0 reads 2 col - 3
  }
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[g][v] = 0.f;

  const int prpw = a.PR * a.PW;
  const float* wlane = wl + kk * 64 + wm * 32 + li;
  float af[2], bf[2][4];
#define READ_STEP(s, slot)                                                          \
  do {                                                                              \
    const int o0 = tap_off(2 * (s), prpw, a.PW), o1 = tap_off(2 * (s) + 1, prpw, a.PW); \
    const float* pk = patch + (kk ? o1 : o0);                                       \
    af[slot] = wlane[(s) * 128];                                                    \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) bf[slot][g] = pk[loff[g]];        \
  } while (0)
  READ_STEP(0, 0);
#pragma unroll
  for (int s = 0; s < KT / 2; ++s) {
    if (s + 1 < KT / 2) READ_STEP(s + 1, (s + 1) & 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1], bf[s & 1][g], acc[g], 0, 0, 0);
  }
#undef READ_STEP

  // register v of group g: channel 32 wm + 8 (v / 4) + 4 kk + (v % 4), pixel q0 + 32 (4 wg + g) + li
  float* yn = a.y + (int64_t)n * 64 * npix;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int q = q0 + 32 * (wg * 4 + g) + li;
    if (q >= npix) continue;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int ch = 32 * wm + 8 * (v >> 2) + 4 * kk + (v & 3);
      yn[(int64_t)ch * npix + q] = acc[g][v];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient (SGD over all parameters): dW[k][t] = sum over frames and output pixels q of dY[k][q] * patch[t][q],
// t = (c, kh, kw).  Taps on the MFMA row axis (5 row tiles of 32 = 160 >= 147: one wave each), output channels on the
// column axis (2 tiles per wave), the reduction runs over chunks of 128 consecutive output pixels of one frame: the input
// patch of the chunk and the dY tile [64][129] sit in LDS.  Persistent workgroups (2 per CU) accumulate over their chunks
// in registers and leave ONE partial [160][64] each; a second launch adds the partials to dW.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int WQ = 128;      // output pixels per chunk
constexpr int WTH = 320;     // 5 waves
constexpr int WDP = 129;     // LDS pitch of the dY tile
constexpr int WG_MAX = 512;  // persistent workgroups

struct StemWgradArgs {
  const float* x;   // [N][3][H][W]
  const float* dy;  // [N][64][OH][OW]
  float* partial;   // [G][160][64]
  float* dw;        // [64][147]
  int H, W, OH, OW, PR, PW, tiles, chunks, G;
};

__global__ __launch_bounds__(WTH) void stem_wgrad_kernel(const StemWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;                                  // [3][PR][PW]
  float* dyl = lds + 3 * a.PR * a.PW;                  // [64][WDP]
  int* pixoff = reinterpret_cast<int*>(dyl + 64 * WDP);  // [WQ]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
  const int npix = a.OH * a.OW, prpw = a.PR * a.PW;
  const int toff = tap_off(32 * wave + li, prpw, a.PW);  // rows >= 147 repeat tap 146 and are never written out
  f32x16 acc0, acc1;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc0[v] = acc1[v] = 0.f;
  const int pw4 = a.PW >> 2, rows4 = a.PR * pw4, total4 = 3 * rows4;
  for (int chunk = blockIdx.x; chunk < a.chunks; chunk += gridDim.x) {
    const int n = chunk / a.tiles, tile = chunk - n * a.tiles;
    const int q0 = tile * WQ, r0 = q0 / a.OW, ih0 = 2 * r0 - 3;
    __syncthreads();  // the previous chunk's operand reads are done
    const float* xn = a.x + (int64_t)n * 3 * a.H * a.W;
    for (int i = tid; i < total4; i += WTH) {
      const int c = i / rows4, rem = i - c * rows4, pr = rem / pw4, p4 = rem - pr * pw4;
      const int ih = ih0 + pr, iw = p4 * 4 - 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
        v = *reinterpret_cast<const f32x4*>(xn + ((int64_t)c * a.H + ih) * a.W + iw);
      reinterpret_cast<f32x4*>(patch)[i] = v;
    }
    const float* dyn = a.dy + (int64_t)n * 64 * npix;
    for (int i = tid; i < 64 * WQ; i += WTH) {
      const int k = i / WQ, j = i - k * WQ, q = q0 + j;
      dyl[k * WDP + j] = q < npix ? dyn[(int64_t)k * npix + q] : 0.f;
    }
    for (int j = tid; j < WQ; j += WTH) {
      int q = q0 + j;
      q = q < npix ? q : npix - 1;
      const int r = q / a.OW, col = q - r * a.OW;
      pixoff[j] = (r - r0) * 2 * a.PW + 2 * col + 1;
    }
    __syncthreads();
    const float* pa = patch + toff;
    const float* pb0 = dyl + li * WDP + lk;
    const float* pb1 = dyl + (32 + li) * WDP + lk;
#pragma unroll 4
    for (int s = 0; s < WQ / 2; ++s) {
      const float av = pa[pixoff[2 * s + lk]];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb0[2 * s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb1[2 * s], acc1, 0, 0, 0);
    }
  }
  // register v: tap row 32 wave + 8 (v / 4) + 4 lk + (v % 4), channel column li (acc0) / 32 + li (acc1)
  float* pp = a.partial + (int64_t)blockIdx.x * 160 * 64;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int row = 32 * wave + 8 * (v >> 2) + 4 * lk + (v & 3);
    pp[row * 64 + li] = acc0[v];
    pp[row * 64 + 32 + li] = acc1[v];
  }
}

// grid (37, Z): the partial list of an element is cut Z ways (one thread walking all G partials was a chain of G dependent
// loads on 37 workgroups: 119 us per step); the Z sums meet in dW by atomics
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const StemWgradArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // (t, k)
  if (i >= 147 * 64) return;
  const int t = i >> 6, k = i & 63;
  float s0 = 0.f, s1 = 0.f;
  const int Z = (int)gridDim.y;
  int g = (int)blockIdx.y;
  for (; g + Z < a.G; g += 2 * Z) {
    s0 += a.partial[((int64_t)g * 160 + t) * 64 + k];
    s1 += a.partial[((int64_t)(g + Z) * 160 + t) * 64 + k];
  }
  if (g < a.G) s0 += a.partial[((int64_t)g * 160 + t) * 64 + k];
  atomicAdd(a.dw + k * 147 + t, s0 + s1);
}

}  // namespace

extern "C" {

size_t vitta_stem_conv7_wgrad_workspace_bytes(void) { return (size_t)WG_MAX * 160 * 64 * sizeof(float); }

int vitta_stem_conv7_wgrad_f32(const float* d_x, const float* d_dy, int64_t N, int32_t H, int32_t W, float* d_dw, void* d_ws,
                               size_t ws_bytes, void* stream) {
  if (!d_x || !d_dy || !d_dw || !d_ws || N <= 0 || H < 7 || W < 7) return VITTA_ERR_INVALID_ARG;
  if (W % 4) return VITTA_ERR_UNSUPPORTED;
  if (ws_bytes < vitta_stem_conv7_wgrad_workspace_bytes()) return VITTA_ERR_WORKSPACE;
  StemWgradArgs a;
  a.x = d_x;
  a.dy = d_dy;
  a.partial = static_cast<float*>(d_ws);
  a.dw = d_dw;
  a.H = H;
  a.W = W;
  a.OH = (H - 1) / 2 + 1;
  a.OW = (W - 1) / 2 + 1;
  const int npix = a.OH * a.OW;
  a.tiles = (npix + WQ - 1) / WQ;
  if (N * a.tiles > 0x7fffffffll) return VITTA_ERR_UNSUPPORTED;
  a.chunks = (int)(N * a.tiles);
  int rows = (WQ - 1) / a.OW + 2;
  if (rows > a.OH) rows = a.OH;
  a.PR = 2 * (rows - 1) + 7;
  a.PW = ((2 * a.OW + 5 + 4 + 3) / 4) * 4;
  a.G = a.chunks < WG_MAX ? a.chunks : WG_MAX;
  const size_t lds = sizeof(float) * ((size_t)3 * a.PR * a.PW + 64 * WDP + WQ);
  if (lds > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  VITTA_LAUNCH(stem_wgrad_kernel, dim3((unsigned)a.G), dim3(WTH), lds, st, a);
  VITTA_LAUNCH(stem_wgrad_reduce_kernel, dim3((147 * 64 + 255) / 256, a.G >= 64 ? 16 : 1), dim3(256), 0, st, a);
  return VITTA_OK;
}

int vitta_stem_conv7_f32(const float* d_x, const float* d_wp, int64_t N, int32_t H, int32_t W, float* d_y, void* stream) {
  if (!d_x || !d_wp || !d_y || N <= 0 || H < 7 || W < 7) return VITTA_ERR_INVALID_ARG;
  if (W % 4) return VITTA_ERR_UNSUPPORTED;
  StemConvArgs a;
  a.x = d_x;
  a.wp = d_wp;
  a.y = d_y;
  a.H = H;
  a.W = W;
  a.OH = (H - 1) / 2 + 1;
  a.OW = (W - 1) / 2 + 1;
  const int npix = a.OH * a.OW;
  a.tiles = (npix + NPIX - 1) / NPIX;
  int rows = (NPIX - 1) / a.OW + 2;  // output rows a tile of NPIX consecutive pixels can touch
  if (rows > a.OH) rows = a.OH;
  a.PR = 2 * (rows - 1) + 7;
  a.PW = ((2 * a.OW + 5 + 4 + 3) / 4) * 4;  // taps reach patch column 2 (OW - 1) + 1 + 6, + the zero row's neighbour
  const size_t lds = sizeof(float) * ((size_t)KT * 64 + (size_t)3 * a.PR * a.PW);
  if (lds > 160 * 1024 || N * a.tiles > 0x7fffffffll) return VITTA_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return VITTA_ERR_LAUNCH;
      raised = true;
    }
  }
  VITTA_LAUNCH(stem_conv7_kernel, dim3((unsigned)(N * a.tiles)), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return VITTA_OK;
}

}  // extern "C"
