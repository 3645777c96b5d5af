// N1 -- decoded RGB frames -> network input on the device, bit-identical to the reference's PIL pipeline:
//   crop (per view)  ->  PIL Image.resize(BILINEAR)  ->  stack on the channel axis  ->  /255  ->  (x - mean_c) / std_c
// (models/tanet_models/transforms.py:277-384 multi-scale crop + resize, :46-54 / :170-184 scale + centre crop, :637-678
// stack / to-tensor, :140-152 normalise).  The arithmetic that defines "identical" is Pillow's 8-bit resampler
// (third-party, Pillow's src/libImaging/Resample.c): per output column a window of `count` taps starting at `first`,
// taps as 22-bit fixed point, accumulate in int32 from 1 << 21, shift, clip to a byte; the horizontal pass produces a
// BYTE image, the vertical pass resamples that.  Tap tables depend only on (crop size, output size) and are built on
// the host in double precision exactly as Pillow does (vitta_amd/frames.py); the two integer passes, the byte
// rounding between them and the normalisation (a 3 x 256 table built with the reference's own float ops) run here.
// An output window inside a larger resize (eval: short edge -> 256, centre 224) is just a table slice.
//
// One workgroup = one frame x `tile_rows` output rows, one lane per output column: the horizontal pass of the input rows
// that tile needs stays in LDS (one packed R|G|B dword per (row, column) in the unrolled path, planar bytes in the general
// one; a lane's taps stay in registers over the rows), the vertical pass reads LDS and writes coalesced fp32 rows through
// the LDS copy of the normalisation table.  HBM traffic = the crop's bytes once + the fp32 output once
// (profiles/r1p_frames_pmc.json: writes 1.000 x, total <= 1.08 x algorithmic).
#include <algorithm>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow: 8 bits of pixel, 2 bits of headroom for the accumulation

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct FramesArgs {
  const uint8_t* frames;  // [F][in_h][in_w][3]
  const int32_t* origin;  // [V][2] crop origin (x0, y0)
  const int32_t* xb;      // [V][out_w][2] first tap (crop-relative), tap count
  const int32_t* xc;      // [V][out_w][kx]
  const int32_t* yb;      // [V][out_h][2]
  const int32_t* yc;      // [V][out_h][ky]
  const float* lut;       // [3][256]
  float* out;             // [F*3][out_h][out_w]
  int in_h, in_w, fpv, kx, ky, out_h, out_w, tile_rows, lds_rows;
};

__device__ __forceinline__ int byte_of(uint32_t w, int i) { return (int)((w >> (8 * i)) & 0xffu); }
// pixel (<= 255) x 22-bit weight: the 24-bit multiply-add is full rate (a 32-bit integer multiply is quarter rate and
// was what bound this kernel)
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t w;
  __builtin_memcpy(&w, p, 4);  // one dword load at any byte address
  return w;
}

// Unrolled path: every window has at most four taps and the weight tables have row stride four, zero padded.  A lane
// (= output column) reads its window as 12 contiguous bytes (4 pixels x RGB) per input row with three dword loads; near
// the right edge of the FRAME the 4-pixel read is shifted left and the weights with it (reads stay inside the row).
// The horizontal pass is kept in LDS as one dword per (row, column): R | G << 8 | B << 16.
__global__ __launch_bounds__(VITTA_BLOCK) void frames_resample4_kernel(FramesArgs a) {
  extern __shared__ uint8_t smem[];
  float* lut = reinterpret_cast<float*>(smem);                            // [3][256]
  uint32_t* mid = reinterpret_cast<uint32_t*>(smem + 768 * sizeof(float));  // [lds_rows][out_w]
  const int f = blockIdx.y, v = f / a.fpv;
  const int oy0 = blockIdx.x * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.out_h);
  const int32_t* yb = a.yb + (int64_t)v * a.out_h * 2;
  for (int i = threadIdx.x; i < 768; i += VITTA_BLOCK) lut[i] = a.lut[i];
  const int r0 = yb[oy0 * 2];
  int r1 = r0;
  for (int y = oy0; y < oy1; ++y) r1 = max(r1, yb[y * 2] + yb[y * 2 + 1]);
  const int rows = min(r1 - r0, a.lds_rows);
  const int x0 = a.origin[v * 2], y0 = a.origin[v * 2 + 1];
  const int32_t* xb = a.xb + (int64_t)v * a.out_w * 2;
  const int32_t* xc = a.xc + (int64_t)v * a.out_w * 4;
  const uint8_t* src = a.frames + ((int64_t)f * a.in_h + y0 + r0) * a.in_w * 3;
  const int W = a.out_w;
  const int64_t row_bytes = (int64_t)a.in_w * 3;
  constexpr int HALF = 1 << (PRECISION_BITS - 1);

  for (int x = threadIdx.x; x < W; x += VITTA_BLOCK) {
    const int first = x0 + xb[x * 2];                  // frame column of the first tap
    const int shift = max(first + 4 - a.in_w, 0);      // 0 except in the last three columns of the frame
    const int4 k4 = *reinterpret_cast<const int4*>(xc + (int64_t)x * 4);
    const int kin[4] = {k4.x, k4.y, k4.z, k4.w};
    int kk[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // kk[t] = kin[t - shift] (zero below the window), as selects: no indexed registers
      int w = kin[t];
      if (t >= 1) w = shift == 1 ? kin[t - 1] : w;
      if (t >= 2) w = shift == 2 ? kin[t - 2] : w;
      if (t >= 3) w = shift == 3 ? kin[t - 3] : w;
      kk[t] = t < shift ? 0 : w;
    }
    const uint8_t* p = src + (int64_t)(first - shift) * 3;
#pragma unroll 3
    for (int r = 0; r < rows; ++r) {
      const uint8_t* q = p + r * row_bytes;
      const uint32_t w0 = load_u32_unaligned(q), w1 = load_u32_unaligned(q + 4), w2 = load_u32_unaligned(q + 8);
      // bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
      int s0 = HALF, s1 = HALF, s2 = HALF;
      s0 = mad24(byte_of(w0, 0), kk[0], s0);
      s1 = mad24(byte_of(w0, 1), kk[0], s1);
      s2 = mad24(byte_of(w0, 2), kk[0], s2);
      s0 = mad24(byte_of(w0, 3), kk[1], s0);
      s1 = mad24(byte_of(w1, 0), kk[1], s1);
      s2 = mad24(byte_of(w1, 1), kk[1], s2);
      s0 = mad24(byte_of(w1, 2), kk[2], s0);
      s1 = mad24(byte_of(w1, 3), kk[2], s1);
      s2 = mad24(byte_of(w2, 0), kk[2], s2);
      s0 = mad24(byte_of(w2, 1), kk[3], s0);
      s1 = mad24(byte_of(w2, 2), kk[3], s1);
      s2 = mad24(byte_of(w2, 3), kk[3], s2);
      mid[r * W + x] = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16);
    }
  }
  __syncthreads();

  const int32_t* yc = a.yc + (int64_t)v * a.out_h * 4;
  float* out = a.out + (int64_t)f * 3 * a.out_h * W;
  const int64_t plane = (int64_t)a.out_h * W;
  for (int x = threadIdx.x; x < W; x += VITTA_BLOCK) {
    for (int y = oy0; y < oy1; ++y) {
      const int first = yb[y * 2] - r0, cnt = yb[y * 2 + 1];  // uniform over the workgroup
      const int32_t* k = yc + (int64_t)y * 4;
      int s0 = HALF, s1 = HALF, s2 = HALF;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t w = mid[min(first + min(t, cnt - 1), rows - 1) * W + x];  // padded taps: zero weight
        const int kt = k[t];
        s0 = mad24(byte_of(w, 0), kt, s0);
        s1 = mad24(byte_of(w, 1), kt, s1);
        s2 = mad24(byte_of(w, 2), kt, s2);
      }
      float* o = out + (int64_t)y * W + x;
      o[0] = lut[clip8(s0)];
      o[plane] = lut[256 + clip8(s1)];
      o[2 * plane] = lut[512 + clip8(s2)];
    }
  }
}

// Windows of any length (strong down-scaling): plain loops over the taps, bytes in LDS (planar [row][channel][x]).
__global__ __launch_bounds__(VITTA_BLOCK) void frames_resample_kernel(FramesArgs a) {
  extern __shared__ uint8_t smem[];
  float* lut = reinterpret_cast<float*>(smem);  // [3][256]
  uint8_t* mid = smem + 768 * sizeof(float);    // [lds_rows][3][out_w]
  const int f = blockIdx.y, v = f / a.fpv;
  const int oy0 = blockIdx.x * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.out_h);
  const int32_t* yb = a.yb + (int64_t)v * a.out_h * 2;
  for (int i = threadIdx.x; i < 768; i += VITTA_BLOCK) lut[i] = a.lut[i];
  const int r0 = yb[oy0 * 2];
  int r1 = r0;
  for (int y = oy0; y < oy1; ++y) r1 = max(r1, yb[y * 2] + yb[y * 2 + 1]);
  const int rows = min(r1 - r0, a.lds_rows);
  const int x0 = a.origin[v * 2], y0 = a.origin[v * 2 + 1];
  const int32_t* xb = a.xb + (int64_t)v * a.out_w * 2;
  const int32_t* xc = a.xc + (int64_t)v * a.out_w * a.kx;
  const uint8_t* src = a.frames + ((int64_t)f * a.in_h + y0 + r0) * a.in_w * 3;
  const int W = a.out_w, plane = 3 * a.out_w;
  const int64_t row_bytes = (int64_t)a.in_w * 3;
  constexpr int HALF = 1 << (PRECISION_BITS - 1);

  for (int x = threadIdx.x; x < W; x += VITTA_BLOCK) {
    const int first = xb[x * 2], cnt = xb[x * 2 + 1];
    const uint8_t* p = src + (int64_t)(x0 + first) * 3;
    const int32_t* k = xc + (int64_t)x * a.kx;
    for (int r = 0; r < rows; ++r) {
      const uint8_t* q = p + r * row_bytes;
      int s0 = HALF, s1 = HALF, s2 = HALF;
      for (int t = 0; t < cnt; ++t) {
        const int w = k[t];
        s0 = mad24(q[t * 3], w, s0);
        s1 = mad24(q[t * 3 + 1], w, s1);
        s2 = mad24(q[t * 3 + 2], w, s2);
      }
      uint8_t* m = mid + r * plane + x;
      m[0] = (uint8_t)clip8(s0);
      m[W] = (uint8_t)clip8(s1);
      m[2 * W] = (uint8_t)clip8(s2);
    }
  }
  __syncthreads();

  const int32_t* yc = a.yc + (int64_t)v * a.out_h * a.ky;
  float* out = a.out + (int64_t)f * 3 * a.out_h * W;
  for (int x = threadIdx.x; x < W; x += VITTA_BLOCK) {
    for (int y = oy0; y < oy1; ++y) {
      const int first = yb[y * 2] - r0, cnt = yb[y * 2 + 1];
      const int32_t* k = yc + (int64_t)y * a.ky;
      int s0 = HALF, s1 = HALF, s2 = HALF;
      for (int t = 0; t < cnt && first + t < rows; ++t) {
        const uint8_t* m = mid + (first + t) * plane + x;
        const int w = k[t];
        s0 = mad24(m[0], w, s0);
        s1 = mad24(m[W], w, s1);
        s2 = mad24(m[2 * W], w, s2);
      }
      float* o = out + (int64_t)y * W + x;
      o[0] = lut[clip8(s0)];
      o[(int64_t)a.out_h * W] = lut[256 + clip8(s1)];
      o[(int64_t)2 * a.out_h * W] = lut[512 + clip8(s2)];
    }
  }
}

}  // namespace

extern "C" {

int vitta_frames_resample_norm_f32(const uint8_t* d_frames, int32_t n_frames, int32_t in_h, int32_t in_w,
                                   int32_t frames_per_view, const int32_t* d_origin, const int32_t* d_xbounds,
                                   const int32_t* d_xcoef, int32_t kx, const int32_t* d_ybounds, const int32_t* d_ycoef,
                                   int32_t ky, const float* d_lut, float* d_out, int32_t out_h, int32_t out_w,
                                   int32_t tile_rows, int32_t lds_rows, void* stream) {
  if (!d_frames || !d_origin || !d_xbounds || !d_xcoef || !d_ybounds || !d_ycoef || !d_lut || !d_out)
    return VITTA_ERR_INVALID_ARG;
  if (n_frames <= 0 || n_frames > 65535 || in_h <= 0 || in_w <= 0 || frames_per_view <= 0 || kx <= 0 || ky <= 0 ||
      out_h <= 0 || out_w <= 0 || tile_rows <= 0 || lds_rows <= 0)
    return VITTA_ERR_INVALID_ARG;
  const unsigned tiles = (unsigned)((out_h + tile_rows - 1) / tile_rows);
  hipStream_t st = static_cast<hipStream_t>(stream);
  FramesArgs a{d_frames, d_origin, d_xbounds, d_xcoef, d_ybounds, d_ycoef, d_lut, d_out, in_h, in_w, frames_per_view,
               kx,       ky,       out_h,     out_w,   tile_rows, lds_rows};
  // row strides bound every window's length: 4 / 4 (zero padded) selects the unrolled path
  if (kx == 4 && ky == 4 && in_w >= 4 && (reinterpret_cast<uintptr_t>(d_xcoef) & 15u) == 0) {
    const int64_t lds = (int64_t)lds_rows * 4 * out_w + 768 * sizeof(float);
    if (lds > 64 * 1024) return VITTA_ERR_UNSUPPORTED;
    VITTA_LAUNCH(frames_resample4_kernel, dim3(tiles, (unsigned)n_frames), dim3(VITTA_BLOCK), (size_t)lds, st, a);
  } else {
    const int64_t lds = (int64_t)lds_rows * 3 * out_w + 768 * sizeof(float);
    if (lds > 64 * 1024) return VITTA_ERR_UNSUPPORTED;
    VITTA_LAUNCH(frames_resample_kernel, dim3(tiles, (unsigned)n_frames), dim3(VITTA_BLOCK), (size_t)lds, st, a);
  }
  return VITTA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------
// Video Swin pipeline (N1, second half): cv2.resize(INTER_LINEAR) of uint8 frames as mmcv.imresize runs it
// (models/videoswintransformer_models/transforms_backup.py:193-349 Resize, RandomResizedCrop + Resize, CenterCrop), restated
// from OpenCV's resize.cpp (cv2 is not in this image: unpinned, see vitta_amd/frames.py).  One thread per destination pixel;
// mode 0 = copy of the crop, 1 = the exact-2x area shortcut, 2 = the fixed-point bilinear with host-built tables
// [x0 | x1 | a0 | a1 | y0 | y1 | b0 | b1] (absolute source indices, 11-bit weights).  Output: uint8 frames (the intermediate
// of the two-resize TTA chain) or the normalised fp32 clip in NCTHW (mmcv.imnormalize + FormatShape).
// ------------------------------------------------------------------------------------------------------------------------
namespace {

struct Cv2Args {
  const uint8_t* src; int F, sh, sw, x0, y0, mode; const int* tab; int dh, dw;
  uint8_t* out_u8; float* out_f32; int clip_len; const float* mean; const float* stdinv;
};

__global__ __launch_bounds__(256) void frames_cv2_resize_kernel(const Cv2Args a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)a.dh * a.dw;
  if (i >= per * a.F) return;
  const int f = (int)(i / per), r = (int)(i - f * per), y = r / a.dw, x = r - y * a.dw;
  const uint8_t* s = a.src + (int64_t)f * a.sh * a.sw * 3;
  int v[3];
  if (a.mode == 0) {
    const uint8_t* p = s + ((int64_t)(a.y0 + y) * a.sw + a.x0 + x) * 3;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  } else if (a.mode == 1) {
    const uint8_t* p = s + ((int64_t)(a.y0 + 2 * y) * a.sw + a.x0 + 2 * x) * 3;
    const uint8_t* q = p + (int64_t)a.sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2;
  } else {
    const int* t = a.tab;
    const int xa = t[x], xb = t[a.dw + x], wa = t[2 * a.dw + x], wb = t[3 * a.dw + x];
    const int* ty = t + 4 * a.dw;
    const int ya = ty[y], yb = ty[a.dh + y], ba = ty[2 * a.dh + y], bb = ty[3 * a.dh + y];
    const uint8_t* r0 = s + (int64_t)ya * a.sw * 3;
    const uint8_t* r1 = s + (int64_t)yb * a.sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[xa * 3 + c] * wa + r0[xb * 3 + c] * wb;
      const int h1 = r1[xa * 3 + c] * wa + r1[xb * 3 + c] * wb;
      int o = (((ba * (h0 >> 4)) >> 16) + ((bb * (h1 >> 4)) >> 16) + 2) >> 2;
      v[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
    }
  }
  if (a.out_u8) {
    uint8_t* o = a.out_u8 + i * 3;
    o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
  } else {
    const int view = f / a.clip_len, tt = f - view * a.clip_len;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      a.out_f32[(((int64_t)view * 3 + c) * a.clip_len + tt) * per + r] = ((float)v[c] - a.mean[c]) * a.stdinv[c];
  }
}

}  // namespace

extern "C" int vitta_frames_cv2_resize(const uint8_t* d_frames, int32_t n_frames, int32_t in_h, int32_t in_w, int32_t x0, int32_t y0,
                                       int32_t mode, const int32_t* d_tables, int32_t out_h, int32_t out_w, uint8_t* d_out_u8,
                                       float* d_out_f32, int32_t clip_len, const float* d_mean, const float* d_stdinv, void* stream) {
  if (!d_frames || n_frames <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || mode < 0 || mode > 2 || x0 < 0 || y0 < 0)
    return VITTA_ERR_INVALID_ARG;
  if ((d_out_u8 == nullptr) == (d_out_f32 == nullptr)) return VITTA_ERR_INVALID_ARG;
  if (mode == 2 && !d_tables) return VITTA_ERR_INVALID_ARG;
  if (d_out_f32 && (clip_len <= 0 || n_frames % clip_len || !d_mean || !d_stdinv)) return VITTA_ERR_INVALID_ARG;
  const int sx = mode == 1 ? 2 : 1;
  if (mode != 2 && (x0 + sx * out_w > in_w || y0 + sx * out_h > in_h)) return VITTA_ERR_INVALID_ARG;
  const Cv2Args a{d_frames, n_frames, in_h, in_w, x0, y0, mode, d_tables, out_h, out_w, d_out_u8, d_out_f32, clip_len, d_mean, d_stdinv};
  const int64_t n = (int64_t)n_frames * out_h * out_w;
  VITTA_LAUNCH(frames_cv2_resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return VITTA_OK;
}
