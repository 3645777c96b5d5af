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
// One workgroup = one frame x `tile_rows` output rows: the horizontal pass of the input rows that tile needs goes to LDS
// as bytes (planar [row][channel][x]), the vertical pass reads LDS and writes coalesced fp32 rows.  HBM traffic = the
// crop's bytes once + the fp32 output once.
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

__global__ __launch_bounds__(VITTA_BLOCK) void frames_resample_kernel(FramesArgs a) {
  extern __shared__ uint8_t mid[];  // [lds_rows][3][out_w]
  const int f = blockIdx.y, v = f / a.fpv;
  const int oy0 = blockIdx.x * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.out_h);
  const int32_t* yb = a.yb + (int64_t)v * a.out_h * 2;
  // input rows (crop-relative) this tile's vertical taps touch: windows start monotonically, so first row of the first
  // output row .. last row of the last one
  const int r0 = yb[oy0 * 2];
  int r1 = r0;
  for (int y = oy0; y < oy1; ++y) r1 = max(r1, yb[y * 2] + yb[y * 2 + 1]);
  const int rows = min(r1 - r0, a.lds_rows);
  const int x0 = a.origin[v * 2], y0 = a.origin[v * 2 + 1];
  const int32_t* xb = a.xb + (int64_t)v * a.out_w * 2;
  const int32_t* xc = a.xc + (int64_t)v * a.out_w * a.kx;
  const uint8_t* src = a.frames + (int64_t)f * a.in_h * a.in_w * 3;
  const int per_row = 3 * a.out_w;

  for (int i = threadIdx.x; i < rows * per_row; i += VITTA_BLOCK) {
    const int r = i / per_row, j = i - r * per_row, c = j / a.out_w, x = j - c * a.out_w;
    const int first = xb[x * 2], cnt = xb[x * 2 + 1];
    const uint8_t* p = src + ((int64_t)(y0 + r0 + r) * a.in_w + x0 + first) * 3 + c;
    const int32_t* k = xc + x * a.kx;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t) ss += (int)p[t * 3] * k[t];
    mid[i] = (uint8_t)clip8(ss);
  }
  __syncthreads();

  const int32_t* yc = a.yc + (int64_t)v * a.out_h * a.ky;
  const int n_out = (oy1 - oy0) * per_row;
  for (int i = threadIdx.x; i < n_out; i += VITTA_BLOCK) {
    const int yy = i / per_row, j = i - yy * per_row, c = j / a.out_w, x = j - c * a.out_w;
    const int y = oy0 + yy;
    const int first = yb[y * 2] - r0, cnt = yb[y * 2 + 1];
    const int32_t* k = yc + y * a.ky;
    const uint8_t* p = mid + first * per_row + j;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t)
      if (first + t < rows) ss += (int)p[t * per_row] * k[t];
    a.out[((int64_t)(f * 3 + c) * a.out_h + y) * a.out_w + x] = a.lut[c * 256 + clip8(ss)];
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
  const int64_t lds = (int64_t)lds_rows * 3 * out_w;
  if (lds > 64 * 1024) return VITTA_ERR_UNSUPPORTED;
  FramesArgs a{d_frames, d_origin, d_xbounds, d_xcoef, d_ybounds, d_ycoef, d_lut, d_out, in_h, in_w, frames_per_view,
               kx,       ky,       out_h,     out_w,   tile_rows, lds_rows};
  const unsigned tiles = (unsigned)((out_h + tile_rows - 1) / tile_rows);
  VITTA_LAUNCH(frames_resample_kernel, dim3(tiles, (unsigned)n_frames), dim3(VITTA_BLOCK), (size_t)lds,
               static_cast<hipStream_t>(stream), a);
  return VITTA_OK;
}

}  // extern "C"
