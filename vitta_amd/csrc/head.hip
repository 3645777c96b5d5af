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

// ====================================================================================================================
// Round 5: the TANet head of the ADAPTATION pass as two launches (tanet.py:243-251, corpus/basics.py:640-668,
// utils/pred_consistency_utils.py:15-31).  What followed the trunk was dropout -> Linear over the frames -> consensus mean over the
// T segments -> view logits -> compute_pred_consis -> mean over the views: seven launches forward and seven backward of 4-13 us each,
// every one a dependent hop of the adaptation chain (profiles/r5a_timeline.csv).  The consensus is linear, so the segment mean is
// taken BEFORE the product (mean_t (y_t W^T + b) = (mean_t y_t) W^T + b: an eighth of the multiply-adds); the workgroup that
// arrives last at a ticket computes the consistency loss, its gradient and the video logits from the finished view logits.
//   y      [B V T][D]  features after dropout (row (b V + v) T + t)
//   forward : ybar[b v] = mean_t y;  lv[b v][k] = ybar . w[k] + bias[k];  loss = sum_b consis(lv[b]);  gradc = d loss / d lv;
//             out[b][k] = mean_v lv[b v][k]
//   backward: dl[b v][k] = g_loss * gradc + g_out[b][k] / V;  dybar = dl W;  dfeat[row][d] = dybar[b v][d] * mask[row][d] * scale
// ====================================================================================================================
constexpr int HK = 4;  // classes per workgroup of the forward launch (one wave each)

__device__ __forceinline__ float sgnf(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__global__ __launch_bounds__(256) void tanet_head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int V, int T, int K, int D,
                                                             float* __restrict__ ybar_out, float* lv, unsigned* ticket,
                                                             float* __restrict__ out, float* __restrict__ loss, float* __restrict__ gradc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // ybar [B V][D]; the last workgroup: p [V K] | gp [V K] | red [8]
  __shared__ int last_flag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int BV = B * V, D4 = D >> 2;
  const int k = blockIdx.x * HK + wave;
  const float4* y4 = reinterpret_cast<const float4*>(y);
  // this wave's weight row: requested first, used after the staging of ybar
  float4 wr[8];
  const float4* w4 = reinterpret_cast<const float4*>(w + (int64_t)min(k, K - 1) * D);
#pragma unroll
  for (int j = 0; j < 8; ++j) wr[j] = w4[min(lane + 64 * j, D4 - 1)];
  const float bk = bias ? bias[min(k, K - 1)] : 0.f;
  // ybar into LDS: item = (view, column quad); T independent 16-byte loads per item
  const float invT = 1.f / (float)T;
  for (int it = tid; it < BV * D4; it += 256) {
    const int bv = it / D4, d4 = it - bv * D4;
    const float4* src = y4 + (int64_t)bv * T * D4 + d4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int t = 0; t < T; ++t) {
      const float4 v = src[(int64_t)t * D4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= invT; s.y *= invT; s.z *= invT; s.w *= invT;
    reinterpret_cast<float4*>(sm)[it] = s;
    if (blockIdx.x == 0 && ybar_out) reinterpret_cast<float4*>(ybar_out)[it] = s;
  }
  __syncthreads();
  // the wave's class against every view
  for (int bv = 0; bv < BV; ++bv) {
    const float4* yb = reinterpret_cast<const float4*>(sm) + (int64_t)bv * D4;
    float acc = 0.f;
    if (D4 <= 512) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < D4) {
          const float4 a = yb[i];
          acc = fmaf(a.x, wr[j].x, fmaf(a.y, wr[j].y, fmaf(a.z, wr[j].z, fmaf(a.w, wr[j].w, acc))));
        }
      }
    } else {
      for (int i = lane; i < D4; i += 64) {
        const float4 a = yb[i], b4 = w4[i];
        acc = fmaf(a.x, b4.x, fmaf(a.y, b4.y, fmaf(a.z, b4.z, fmaf(a.w, b4.w, acc))));
      }
    }
    acc = wave_sum(acc);
    if (lane == 0 && k < K) {  // write-through: the last workgroup reads it with coherent loads
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(lv, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc + bk), rs, (bv * K + k) * 4, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == gridDim.x - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = last ? 1 : 0;
  }
  __syncthreads();
  if (!last_flag) return;
  // ---- the last workgroup: consistency loss + gradient per video (pred_consis_kernel's arithmetic), video logits ----
  float* p = sm;
  float* gp = sm + (size_t)V * K;
  float* red = gp + (size_t)V * K;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(lv, 0, BV * K * 4, 0x00020000);
  float total = 0.f;
  for (int b = 0; b < B; ++b) {
    __syncthreads();
    for (int i = tid; i < V * K; i += 256)
      gp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (b * V * K + i) * 4, 0, 16));  // z, for now
    __syncthreads();
    for (int kk = tid; kk < K; kk += 256) {
      float m = 0.f;
      for (int v = 0; v < V; ++v) m += gp[v * K + kk];
      out[(int64_t)b * K + kk] = m / (float)V;
    }
    for (int v = wave; v < V; v += 4) {
      float mx = -INFINITY;
      for (int kk = lane; kk < K; kk += 64) mx = fmaxf(mx, gp[v * K + kk]);
      mx = wave_max(mx);
      float s = 0.f;
      for (int kk = lane; kk < K; kk += 64) {
        const float e = expf(gp[v * K + kk] - mx);
        p[v * K + kk] = e;
        s += e;
      }
      s = wave_sum(s);
      for (int kk = lane; kk < K; kk += 64) p[v * K + kk] = p[v * K + kk] / s;
    }
    __syncthreads();
    float acc = 0.f;
    const float invV = 1.f / (float)V;
    for (int kk = tid; kk < K; kk += 256) {
      float pb = 0.f;
      for (int v = 0; v < V; ++v) pb += p[v * K + kk];
      pb *= invV;
      float ssum = 0.f;
      for (int v = 0; v < V; ++v) {
        const float d = p[v * K + kk] - pb;
        acc += fabsf(d);
        ssum += sgnf(d);
      }
      for (int v = 0; v < V; ++v) gp[v * K + kk] = invV * (sgnf(p[v * K + kk] - pb) - invV * ssum);
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) total += (red[0] + red[1] + red[2] + red[3]) * invV;
    for (int v = wave; v < V; v += 4) {
      float dot = 0.f;
      for (int kk = lane; kk < K; kk += 64) dot = fmaf(gp[v * K + kk], p[v * K + kk], dot);
      dot = wave_sum(dot);
      for (int kk = lane; kk < K; kk += 64) gradc[((int64_t)b * V + v) * K + kk] = p[v * K + kk] * (gp[v * K + kk] - dot);
    }
  }
  if (tid == 0) *loss = total;
}

// dfeat: grid = D / 32 workgroups; a thread = (column quad of the tile, slice of the classes); the slices meet in LDS
__global__ __launch_bounds__(256) void tanet_head_bwd_kernel(const float* __restrict__ gradc, const float* __restrict__ g_loss,
                                                             const float* __restrict__ g_out, const float* __restrict__ w,
                                                             const unsigned char* __restrict__ mask, float scale, int B, int V, int T,
                                                             int K, int D, float* __restrict__ dfeat, float* __restrict__ dl_out) {
  constexpr int MAXBV = 8;
  __shared__ float4 part[32][8][MAXBV + 1];
  const int tid = threadIdx.x, q = tid & 7, sl = tid >> 3;  // 8 column quads x 32 class slices
  const int BV = B * V, D4 = D >> 2, d4 = blockIdx.x * 8 + q;
  const float gl = g_loss ? g_loss[0] : 0.f, invV = 1.f / (float)V;
  float4 acc[MAXBV];
#pragma unroll
  for (int i = 0; i < MAXBV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int k = sl; k < K; k += 32) {
    const float4 wv = w4[(int64_t)k * D4 + min(d4, D4 - 1)];
#pragma unroll
    for (int bv = 0; bv < MAXBV; ++bv) {
      if (bv < BV) {
        float g = gl * gradc[bv * K + k];
        if (g_out) g += g_out[(bv / V) * K + k] * invV;
        if (dl_out && blockIdx.x == 0 && q == 0) dl_out[bv * K + k] = g;
        acc[bv].x = fmaf(g, wv.x, acc[bv].x);
        acc[bv].y = fmaf(g, wv.y, acc[bv].y);
        acc[bv].z = fmaf(g, wv.z, acc[bv].z);
        acc[bv].w = fmaf(g, wv.w, acc[bv].w);
      }
    }
  }
#pragma unroll
  for (int bv = 0; bv < MAXBV; ++bv)
    if (bv < BV) part[sl][q][bv] = acc[bv];
  __syncthreads();
  // thread = (column quad, view) sums the 32 slices and writes the T frames of its view
  for (int it = tid; it < 8 * BV; it += 256) {
    const int qq = it & 7, bv = it >> 3, dd4 = blockIdx.x * 8 + qq;
    if (dd4 >= D4) continue;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const float4 v = part[i][qq][bv];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)bv * T + t;
      float4 o = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
      if (mask) {
        const uchar4 m = reinterpret_cast<const uchar4*>(mask)[row * D4 + dd4];
        o.x = m.x ? o.x : 0.f; o.y = m.y ? o.y : 0.f; o.z = m.z ? o.z : 0.f; o.w = m.w ? o.w : 0.f;
      }
      reinterpret_cast<float4*>(dfeat)[row * D4 + dd4] = o;
    }
  }
}

// trainable head (SGD over all parameters): dw[k][d] += sum_bv dl[bv][k] ybar[bv][d];  db[k] += sum_bv dl[bv][k]
__global__ __launch_bounds__(256) void tanet_head_bwd_w_kernel(const float* __restrict__ dl, const float* __restrict__ ybar, int BV, int K, int D,
                                                               float* __restrict__ dw, float* __restrict__ db) {
  const int D4 = D >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * D4) return;
  const int k = (int)(i / D4), d4 = (int)(i - (int64_t)k * D4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int bv = 0; bv < BV; ++bv) {
    const float g = dl[bv * K + k];
    const float4 yv = reinterpret_cast<const float4*>(ybar)[(int64_t)bv * D4 + d4];
    s.x = fmaf(g, yv.x, s.x); s.y = fmaf(g, yv.y, s.y); s.z = fmaf(g, yv.z, s.z); s.w = fmaf(g, yv.w, s.w);
    sb += g;
  }
  if (dw) {
    float4* p = reinterpret_cast<float4*>(dw) + i;
    float4 o = *p;
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    *p = o;
  }
  if (db && d4 == 0) db[k] += sb;
}

// out = la * a + lb * b (the step's total loss, corpus/basics.py:668);  backward: ga = la * g, gb = lb * g in one launch
__global__ void loss_axpby_kernel(const float* a, const float* b, float la, float lb, float* out, float* ga, float* gb) {
  if (threadIdx.x == 0) {
    out[0] = la * a[0] + lb * (b ? b[0] : 0.f);
    if (ga) ga[0] = la;  // the two upstream gradients for d out = 1 (what `loss.backward()` starts from): no backward launch then
    if (gb) gb[0] = lb;
  }
}
__global__ void loss_axpby_bwd_kernel(const float* g, float la, float lb, float* ga, float* gb) {
  if (threadIdx.x == 0) {
    const float v = g ? g[0] : 1.f;
    ga[0] = la * v;
    if (gb) gb[0] = lb * v;
  }
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

size_t vitta_tanet_head_lds_bytes(int32_t B, int32_t V, int32_t K, int32_t D) {
  const size_t a = (size_t)B * V * D * 4, b = ((size_t)2 * V * K + 8) * 4;
  return a > b ? a : b;
}

int vitta_tanet_head_fwd_f32(const float* d_y, const float* d_w, const float* d_b, int32_t B, int32_t V, int32_t T, int32_t K, int32_t D,
                             float* d_ybar, float* d_view_logits, void* d_ticket, float* d_out, float* d_loss, float* d_gradc,
                             void* stream) {
  if (!d_y || !d_w || !d_view_logits || !d_ticket || !d_out || !d_loss || !d_gradc || B <= 0 || V <= 0 || T <= 0 || K <= 0 || D <= 0)
    return VITTA_ERR_INVALID_ARG;
  const size_t lds = vitta_tanet_head_lds_bytes(B, V, K, D);
  if (D % 4 || B * V > 8 || lds > 128 * 1024) return VITTA_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&tanet_head_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(tanet_head_fwd_kernel, dim3((unsigned)((K + HK - 1) / HK)), dim3(256), lds, static_cast<hipStream_t>(stream), d_y, d_w, d_b, B, V,
               T, K, D, d_ybar, d_view_logits, static_cast<unsigned*>(d_ticket), d_out, d_loss, d_gradc);
  return VITTA_OK;
}

int vitta_tanet_head_bwd_f32(const float* d_gradc, const float* d_g_loss, const float* d_g_out, const float* d_w, const void* d_mask,
                             float scale, int32_t B, int32_t V, int32_t T, int32_t K, int32_t D, float* d_dfeat, const float* d_ybar,
                             float* d_dl, float* d_dw, float* d_db, void* stream) {
  if (!d_gradc || !d_w || !d_dfeat || B <= 0 || V <= 0 || T <= 0 || K <= 0 || D <= 0) return VITTA_ERR_INVALID_ARG;
  if (D % 4 || B * V > 8) return VITTA_ERR_UNSUPPORTED;
  if ((d_dw || d_db) && (!d_ybar || !d_dl)) return VITTA_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  VITTA_LAUNCH(tanet_head_bwd_kernel, dim3((unsigned)((D / 4 + 7) / 8)), dim3(256), 0, st, d_gradc, d_g_loss, d_g_out, d_w,
               static_cast<const unsigned char*>(d_mask), scale, B, V, T, K, D, d_dfeat, d_dl);
  if (d_dw || d_db) {
    const int64_t n = (int64_t)K * (D / 4);
    VITTA_LAUNCH(tanet_head_bwd_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_dl, d_ybar, B * V, K, D, d_dw, d_db);
  }
  return VITTA_OK;
}

int vitta_loss_axpby_f32(const float* d_a, const float* d_b, float la, float lb, float* d_out, float* d_ga1, float* d_gb1, void* stream) {
  if (!d_a || !d_out) return VITTA_ERR_INVALID_ARG;
  VITTA_LAUNCH(loss_axpby_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_a, d_b, la, lb, d_out, d_ga1, d_gb1);
  return VITTA_OK;
}

int vitta_loss_axpby_bwd_f32(const float* d_g, float la, float lb, float* d_ga, float* d_gb, void* stream) {
  if (!d_ga) return VITTA_ERR_INVALID_ARG;
  VITTA_LAUNCH(loss_axpby_bwd_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_g, la, lb, d_ga, d_gb);
  return VITTA_OK;
}

}  // extern "C"
