// The classification head of the TANet path: new_fc = Linear(2048 -> num_class) on the [frames, 2048] pooled features
// (models/tanet_models/tanet.py:105-123, 243-251).  A [16 x 2048] x [2048 x 101] product is microseconds of HBM / L2
// traffic and far too small for a matrix-core tile walk; what it costs is latency, so it is a wave-per-output-element dot product
// (forward), a thread-per-element walk over the classes (data gradient) and over the frames (weight gradient).
//   y[m][n] = b[n] + sum_k x[m][k] w[n][k]
#include "common.h"

using namespace vitta;

namespace {

// one wave per output element (m, n): M * N waves of K / 256 float4 loads per lane (x and w rows stay in L2)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int M, int N, int K) {
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= (int64_t)M * N) return;
  const int m = (int)(o / N), n = (int)(o - (int64_t)m * N);
  const float4* wr = reinterpret_cast<const float4*>(w + (int64_t)n * K);
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)m * K);
  const int k4n = K >> 2;
  float s0 = 0.f, s1 = 0.f;
  int k4 = lane;
  for (; k4 + 64 < k4n; k4 += 128) {
    const float4 w0 = wr[k4], x0 = xr[k4], w1 = wr[k4 + 64], x1 = xr[k4 + 64];
    s0 = fmaf(x0.x, w0.x, fmaf(x0.y, w0.y, fmaf(x0.z, w0.z, fmaf(x0.w, w0.w, s0))));
    s1 = fmaf(x1.x, w1.x, fmaf(x1.y, w1.y, fmaf(x1.z, w1.z, fmaf(x1.w, w1.w, s1))));
  }
  for (; k4 < k4n; k4 += 64) {
    const float4 w0 = wr[k4], x0 = xr[k4];
    s0 = fmaf(x0.x, w0.x, fmaf(x0.y, w0.y, fmaf(x0.z, w0.z, fmaf(x0.w, w0.w, s0))));
  }
  const float t = wave_sum(s0 + s1);
  if (lane == 0) y[o] = t + (b ? b[n] : 0.f);
}

// dx[m][k] = sum_n dy[m][n] w[n][k]
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, int M, int N, int K) {
  const int k4n = K >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * k4n) return;
  const int m = (int)(i / k4n), k4 = (int)(i - (int64_t)m * k4n);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < N; ++n) {
    const float g = dy[(int64_t)m * N + n];
    const float4 wv = reinterpret_cast<const float4*>(w + (int64_t)n * K)[k4];
    s.x = fmaf(g, wv.x, s.x);
    s.y = fmaf(g, wv.y, s.y);
    s.z = fmaf(g, wv.z, s.z);
    s.w = fmaf(g, wv.w, s.w);
  }
  reinterpret_cast<float4*>(dx)[i] = s;
}

// dw[n][k] += sum_m dy[m][n] x[m][k] ; db[n] += sum_m dy[m][n]  (single writer per element: plain read-modify-write)
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int M, int N, int K) {
  const int k4n = K >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * k4n) return;
  const int n = (int)(i / k4n), k4 = (int)(i - (int64_t)n * k4n);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int m = 0; m < M; ++m) {
    const float g = dy[(int64_t)m * N + n];
    const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)m * K)[k4];
    s.x = fmaf(g, xv.x, s.x);
    s.y = fmaf(g, xv.y, s.y);
    s.z = fmaf(g, xv.z, s.z);
    s.w = fmaf(g, xv.w, s.w);
    sb += g;
  }
  if (dw) {
    float4* p = reinterpret_cast<float4*>(dw) + i;
    float4 o = *p;
    o.x += s.x;
    o.y += s.y;
    o.z += s.z;
    o.w += s.w;
    *p = o;
  }
  if (db && k4 == 0) db[n] += sb;
}

}  // namespace

extern "C" {

int vitta_linear_fwd_f32(const float* d_x, const float* d_w, const float* d_b, int64_t M, int32_t N, int32_t K, float* d_y,
                         void* stream) {
  if (!d_x || !d_w || !d_y || M <= 0 || N <= 0 || K <= 0) return VITTA_ERR_INVALID_ARG;
  if (K % 4 || M > (1 << 20)) return VITTA_ERR_UNSUPPORTED;
  VITTA_LAUNCH(linear_fwd_kernel, dim3((unsigned)((M * N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), d_x, d_w, d_b,
               d_y, (int)M, N, K);
  return VITTA_OK;
}

int vitta_linear_bwd_f32(const float* d_dy, const float* d_x, const float* d_w, int64_t M, int32_t N, int32_t K, float* d_dx,
                         float* d_dw, float* d_db, void* stream) {
  if (!d_dy || !d_w || M <= 0 || N <= 0 || K <= 0 || ((d_dw || d_db) && !d_x)) return VITTA_ERR_INVALID_ARG;
  if (K % 4 || M > (1 << 20)) return VITTA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d_dx) {
    const int64_t n = M * (K / 4);
    VITTA_LAUNCH(linear_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_dy, d_w, d_dx, (int)M, N, K);
  }
  if (d_dw || d_db) {
    const int64_t n = (int64_t)N * (K / 4);
    VITTA_LAUNCH(linear_bwd_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_dy, d_x, d_dw, d_db, (int)M, N, K);
  }
  return VITTA_OK;
}

}  // extern "C"
