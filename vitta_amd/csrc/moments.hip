// Per-channel spatio-temporal moments of hooked norm-layer outputs.
//
// Reference behaviour being replaced (utils/norm_stats_utils.py:238-243):
//   feature.view(bz*m,t,c,h,w).permute(0,2,1,3,4).contiguous()     -> full copy
//   output.mean((0,2,3,4))                                          -> read
//   output.permute(1,0,2,3,4).contiguous().view([c,-1]).var(1,unbiased=False) -> copy + read
// i.e. ~6x the algorithmic traffic in 6 launches per layer.  Here: ONE pass over x
// (4 B/element), all hooked layers in one launch, no copies.
//
// NCHW kernel ("flat chunk"): a frame is a plane of C*HW contiguous floats.  A
// workgroup owns CHUNK=1024 consecutive floats of the plane and walks over the
// frames of its split: lane l always touches plane offsets 4l..4l+3, so each of
// its four accumulator slots belongs to ONE channel for the whole walk, whatever
// HW is (49, 196, 784 ... none of them needs to divide anything).  Loads are 16 B
// per lane, perfectly coalesced, and independent across frames (deep MLP).  The
// per-slot sums are shifted by the slot's first sample, converted to (n, mean, M2)
// and merged per channel (Chan) through LDS once per workgroup.
//
// NHWC kernel (LayerNorm outputs, rows x C): lane owns 4 consecutive channels,
// walks over rows; ty-rows of the workgroup are merged through LDS.
//
// A tiny finalize kernel merges the per-workgroup triples in fp64.
#include <hip/hip_ext.h>

#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace vitta;

namespace {

constexpr int kUnroll = 16;

// ----------------------------------------------------------------------------
// NCHW partial kernel
// ----------------------------------------------------------------------------
// streaming load: NT = non-temporal hint (the features are read exactly once by this kernel).  T = element type as
// stored: float, or uint16_t holding bfloat16 (widened to fp32 in registers: accumulation is always fp32).
template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<uint16_t> { using type = uint2; };  // four bf16 = 8 bytes

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <bool NT>
__device__ __forceinline__ float4 ld4(const float4* p) {
  if constexpr (NT) {
    float4 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z);
    v.w = __builtin_nontemporal_load(&p->w);
    return v;
  } else {
    return *p;
  }
}

template <bool NT>
__device__ __forceinline__ float4 ld4(const uint2* p) {
  uint2 w;
  if constexpr (NT) {
    w.x = __builtin_nontemporal_load(&p->x);
    w.y = __builtin_nontemporal_load(&p->y);
  } else {
    w = *p;
  }
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }

template <bool NT, typename T = float>
__global__ __launch_bounds__(VITTA_BLOCK) void moments_nchw_partial_kernel(
    const LayerInfo* __restrict__ linfo, const BlockEnt* __restrict__ tab, PtrPack ptrs,
    float* __restrict__ ws) {
  __shared__ float lds_mean[VITTA_CHUNK];
  __shared__ float lds_m2[VITTA_CHUNK];

  const BlockEnt e = tab[blockIdx.x];
  const LayerInfo L = linfo[e.layer];
  const T* __restrict__ x = static_cast<const T*>(ptrs.x[e.layer]);
  using V4 = typename Vec4<T>::type;
  const int tid = threadIdx.x;
  const int64_t plane = L.plane;
  const int64_t base = (int64_t)e.chunk * VITTA_CHUNK;
  const int64_t per = (L.outer + L.nsplit - 1) / L.nsplit;
  const int64_t n0 = (int64_t)e.split * per;
  const int64_t n1 = n0 + per < L.outer ? n0 + per : L.outer;
  const float cnt = n1 > n0 ? (float)(n1 - n0) : 0.f;

  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, x0[4] = {0.f, 0.f, 0.f, 0.f};

  if (L.vec == 4) {
    const int64_t j = base + 4 * (int64_t)tid;
    if (j < plane && n1 > n0) {
      const V4* p = reinterpret_cast<const V4*>(x + n0 * plane + j);
      const int64_t stride4 = plane >> 2;
      const float4 f = ld4<NT>(p);
      x0[0] = f.x; x0[1] = f.y; x0[2] = f.z; x0[3] = f.w;
      const int64_t nn = n1 - n0;
      int64_t n = 0;
      for (; n + kUnroll <= nn; n += kUnroll) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = ld4<NT>(p + (n + u) * stride4);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const float d0 = v[u].x - x0[0], d1 = v[u].y - x0[1], d2 = v[u].z - x0[2], d3 = v[u].w - x0[3];
          s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
          q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]);
          q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
        }
      }
      for (; n < nn; ++n) {
        const float4 v = ld4<NT>(p + n * stride4);
        const float d0 = v.x - x0[0], d1 = v.y - x0[1], d2 = v.z - x0[2], d3 = v.w - x0[3];
        s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
        q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]);
        q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
      lds_mean[4 * tid + k] = x0[k] + s[k] * inv;
      lds_m2[4 * tid + k] = fmaxf(q[k] - s[k] * s[k] * inv, 0.f);
    }
  } else {
    // scalar path (plane not a multiple of 4): slot k of lane l is plane offset base + l + 256*k
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = base + tid + (int64_t)VITTA_BLOCK * k;
      float sk = 0.f, qk = 0.f, xk = 0.f;
      if (j < plane && n1 > n0) {
        const T* p = x + n0 * plane + j;
        xk = ld1(p);
        for (int64_t n = 0; n < n1 - n0; ++n) {
          const float d = ld1(p + n * plane) - xk;
          sk += d;
          qk = fmaf(d, d, qk);
        }
      }
      const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
      lds_mean[tid + VITTA_BLOCK * k] = xk + sk * inv;
      lds_m2[tid + VITTA_BLOCK * k] = fmaxf(qk - sk * sk * inv, 0.f);
    }
  }
  __syncthreads();

  // segmented merge: one wave per channel touched by this chunk
  const int64_t HW = L.inner;
  const int64_t end = base + VITTA_CHUNK < plane ? base + VITTA_CHUNK : plane;
  const int64_t c_lo = base / HW;
  const int64_t c_hi = (end - 1) / HW;
  const int wave = tid >> 6, lane = tid & 63;
  float* out = ws + 3 * (L.ws_off + ((int64_t)e.split * L.nchunks + e.chunk) * L.slots);
  for (int64_t c = c_lo + wave; c <= c_hi; c += VITTA_BLOCK / VITTA_WAVE) {
    const int64_t lo = (c * HW > base ? c * HW : base) - base;
    const int64_t hi = ((c + 1) * HW < end ? (c + 1) * HW : end) - base;
    Moments acc{0.f, 0.f, 0.f};
    for (int64_t o = lo + lane; o < hi; o += VITTA_WAVE) {
      acc = merge(acc, Moments{cnt, lds_mean[o], lds_m2[o]});
    }
    acc = wave_merge(acc);
    if (lane == 0) {
      float* t = out + 3 * (c - c_lo);
      t[0] = acc.n; t[1] = acc.mean; t[2] = acc.m2;
    }
  }
}

// ----------------------------------------------------------------------------
// NHWC partial kernel
// ----------------------------------------------------------------------------
template <bool NT, typename T = float>
__global__ __launch_bounds__(VITTA_BLOCK) void moments_nhwc_partial_kernel(
    const LayerInfo* __restrict__ linfo, const BlockEnt* __restrict__ tab, PtrPack ptrs,
    float* __restrict__ ws) {
  __shared__ float lds_n[VITTA_CHUNK];
  __shared__ float lds_mean[VITTA_CHUNK];
  __shared__ float lds_m2[VITTA_CHUNK];

  const BlockEnt e = tab[blockIdx.x];
  const LayerInfo L = linfo[e.layer];
  const T* __restrict__ x = static_cast<const T*>(ptrs.x[e.layer]);
  using V4 = typename Vec4<T>::type;
  const int tid = threadIdx.x;
  const int C = L.C;
  const int TX = L.tx;
  const int TY = VITTA_BLOCK / TX;
  const int tx = tid % TX, ty = tid / TX;
  const int64_t per = (L.outer + L.nsplit - 1) / L.nsplit;
  const int64_t r0 = (int64_t)e.split * per;
  const int64_t r1 = r0 + per < L.outer ? r0 + per : L.outer;
  const int W = L.vec;  // channels per lane (4 or 1)

  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, x0[4] = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;
  const int64_t c0 = ((int64_t)e.chunk * TX + tx) * W;
  const bool active = ty < TY && c0 < C;
  if (active && r0 + ty < r1) {
    if (W == 4) {
      const V4* p = reinterpret_cast<const V4*>(x + (r0 + ty) * C + c0);
      const int64_t stride4 = ((int64_t)TY * C) >> 2;
      const int64_t nn = (r1 - r0 - ty + TY - 1) / TY;
      const float4 f = ld4<NT>(p);
      x0[0] = f.x; x0[1] = f.y; x0[2] = f.z; x0[3] = f.w;
      cnt = (float)nn;
      int64_t n = 0;
      for (; n + kUnroll <= nn; n += kUnroll) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = ld4<NT>(p + (n + u) * stride4);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const float d0 = v[u].x - x0[0], d1 = v[u].y - x0[1], d2 = v[u].z - x0[2], d3 = v[u].w - x0[3];
          s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
          q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]);
          q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
        }
      }
      for (; n < nn; ++n) {
        const float4 v = ld4<NT>(p + n * stride4);
        const float d0 = v.x - x0[0], d1 = v.y - x0[1], d2 = v.z - x0[2], d3 = v.w - x0[3];
        s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
        q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]);
        q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
      }
    } else {
      const T* p = x + (r0 + ty) * C + c0;
      const int64_t stride = (int64_t)TY * C;
      const int64_t nn = (r1 - r0 - ty + TY - 1) / TY;
      x0[0] = ld1(p);
      cnt = (float)nn;
      for (int64_t n = 0; n < nn; ++n) {
        const float d = ld1(p + n * stride) - x0[0];
        s[0] += d;
        q[0] = fmaf(d, d, q[0]);
      }
    }
  }
  // stage per-lane triples: column = tx*W + k, row = ty
  const int cols = TX * W;
  if (ty < TY) {
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    for (int k = 0; k < W; ++k) {
      const int o = ty * cols + tx * W + k;
      lds_n[o] = cnt;
      lds_mean[o] = x0[k] + s[k] * inv;
      lds_m2[o] = fmaxf(q[k] - s[k] * s[k] * inv, 0.f);
    }
  }
  __syncthreads();
  float* out = ws + 3 * (L.ws_off + (int64_t)e.split * C);
  for (int col = tid; col < cols; col += VITTA_BLOCK) {
    const int64_t c = (int64_t)e.chunk * cols + col;
    if (c >= C) continue;
    Moments acc{0.f, 0.f, 0.f};
    for (int r = 0; r < TY; ++r) {
      const int o = r * cols + col;
      acc = merge(acc, Moments{lds_n[o], lds_mean[o], lds_m2[o]});
    }
    float* t = out + 3 * c;
    t[0] = acc.n; t[1] = acc.mean; t[2] = acc.m2;
  }
}

// ----------------------------------------------------------------------------
// finalize: merge the per-workgroup triples of one channel (fp64), emit either
// additive shifted sums (mode 0) or mean / biased variance (mode 1).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(VITTA_BLOCK) void moments_finalize_kernel(
    const LayerInfo* __restrict__ linfo, const int32_t* __restrict__ chan2layer, int64_t total_c,
    const float* __restrict__ ws, const float* __restrict__ shift, int mode, float* __restrict__ out_cnt,
    float* __restrict__ out_a, float* __restrict__ out_b) {
  const int64_t g = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x;
  if (g >= total_c) return;
  const int l = chan2layer[g];
  const LayerInfo L = linfo[l];
  const int64_t c = g - L.chan_off;
  MomentsD acc{0.0, 0.0, 0.0};
  if (L.layout == VITTA_LAYOUT_NCHW) {
    const int64_t HW = L.inner;
    const int64_t k0 = (c * HW) / VITTA_CHUNK;
    const int64_t k1 = ((c + 1) * HW - 1) / VITTA_CHUNK;
    for (int sp = 0; sp < L.nsplit; ++sp) {
      for (int64_t k = k0; k <= k1; ++k) {
        const int64_t slot = c - (k * VITTA_CHUNK) / HW;
        const float* t = ws + 3 * (L.ws_off + ((int64_t)sp * L.nchunks + k) * L.slots + slot);
        acc = merge(acc, MomentsD{(double)t[0], (double)t[1], (double)t[2]});
      }
    }
  } else {
    for (int sp = 0; sp < L.nsplit; ++sp) {
      const float* t = ws + 3 * (L.ws_off + (int64_t)sp * L.C + c);
      acc = merge(acc, MomentsD{(double)t[0], (double)t[1], (double)t[2]});
    }
  }
  if (mode == 0) {
    const double k = shift ? (double)shift[g] : 0.0;
    const double d = acc.mean - k;
    out_a[g] = (float)(acc.n * d);
    out_b[g] = (float)(acc.m2 + acc.n * d * d);
    if (c == 0 && out_cnt) out_cnt[l] = (float)acc.n;
  } else {
    out_a[g] = (float)acc.mean;
    out_b[g] = (float)(acc.n > 0.0 ? acc.m2 / acc.n : 0.0);
  }
}

__global__ __launch_bounds__(VITTA_BLOCK) void moments_to_meanvar_kernel(
    const LayerInfo* __restrict__ linfo, const int32_t* __restrict__ chan2layer, int64_t total_c,
    const float* __restrict__ shift, const float* __restrict__ cnt, const float* __restrict__ s1,
    const float* __restrict__ s2, float* __restrict__ mean, float* __restrict__ var) {
  const int64_t g = (int64_t)blockIdx.x * VITTA_BLOCK + threadIdx.x;
  if (g >= total_c) return;
  const int l = chan2layer[g];
  const double n = (double)cnt[l];
  const double k = shift ? (double)shift[g] : 0.0;
  const double m1 = n > 0.0 ? (double)s1[g] / n : 0.0;
  const double m2 = n > 0.0 ? (double)s2[g] / n : 0.0;
  mean[g] = (float)(k + m1);
  const double v = m2 - m1 * m1;
  var[g] = (float)(v > 0.0 ? v : 0.0);
}

int fill_layer(const vitta_layer_shape& s, LayerInfo* L) {
  if (s.outer <= 0 || s.C <= 0 || s.inner <= 0) return VITTA_ERR_INVALID_ARG;
  L->outer = s.outer;
  L->C = s.C;
  L->layout = s.layout;
  if (s.layout == VITTA_LAYOUT_NCHW) {
    L->inner = s.inner;
    L->plane = (int64_t)s.C * s.inner;
    L->vec = (L->plane % 4 == 0) ? 4 : 1;
    L->nchunks = (int32_t)((L->plane + VITTA_CHUNK - 1) / VITTA_CHUNK);
    L->slots = (int32_t)((VITTA_CHUNK + s.inner - 2) / s.inner + 1);
    L->tx = 0;
  } else if (s.layout == VITTA_LAYOUT_NHWC) {
    if (s.inner != 1) return VITTA_ERR_INVALID_ARG;
    L->inner = 1;
    L->plane = s.C;
    L->vec = (s.C % 4 == 0) ? 4 : 1;
    const int lanes = (s.C + L->vec - 1) / L->vec;
    L->tx = lanes < VITTA_BLOCK ? lanes : VITTA_BLOCK;
    L->nchunks = (lanes + L->tx - 1) / L->tx;
    L->slots = s.C;
  } else {
    return VITTA_ERR_INVALID_ARG;
  }
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_abi_version(void) { return VITTA_ABI_VERSION; }

const char* vitta_status_string(int status) {
  switch (status) {
    case VITTA_OK: return "ok";
    case VITTA_ERR_INVALID_ARG: return "invalid argument";
    case VITTA_ERR_LAUNCH: return "kernel launch failed";
    case VITTA_ERR_ALLOC: return "device allocation failed";
    case VITTA_ERR_UNSUPPORTED: return "unsupported configuration";
    case VITTA_ERR_WORKSPACE: return "workspace too small or missing";
    default: return "unknown status";
  }
}

int vitta_plan_create(const vitta_layer_shape* h_shapes, int n_layers, int target_blocks,
                      vitta_plan** out_plan) {
  return vitta_plan_create_split(h_shapes, n_layers, target_blocks, nullptr, out_plan);
}

int vitta_plan_create_split(const vitta_layer_shape* h_shapes, int n_layers, int target_blocks,
                            const int32_t* h_nsplit, vitta_plan** out_plan) {
  if (!h_shapes || !out_plan || n_layers <= 0 || n_layers > VITTA_MAX_LAYERS) return VITTA_ERR_INVALID_ARG;
  vitta_plan* p = new (std::nothrow) vitta_plan();
  if (!p) return VITTA_ERR_ALLOC;
  p->n_layers = n_layers;
  // measured on MI355X (tools/bench_moments.py): 2-3k workgroups with the whole frame walk per lane beat
  // 5-20k shorter ones by 3-30 % (less merge/epilogue work per byte)
  if (target_blocks <= 0) target_blocks = 2048;

  double work_nchw = 0.0, work_nhwc = 0.0;
  int64_t coff = 0;
  for (int l = 0; l < n_layers; ++l) {
    LayerInfo* L = &p->h_info[l];
    const int st = fill_layer(h_shapes[l], L);
    if (st != VITTA_OK) { delete p; return st; }
    L->chan_off = (int32_t)coff;
    coff += L->C;
    if (L->layout == VITTA_LAYOUT_NCHW) work_nchw += (double)L->nchunks * (double)L->outer;
    else work_nhwc += (double)L->nchunks * (double)L->outer / (double)(VITTA_BLOCK / L->tx);
  }
  p->total_channels = coff;
  // every feature is read exactly once by this launch: non-temporal loads by default (VITTA_OPT_NT_LOADS switches them
  // off).  Measured: +8 % on operands beyond the 256 MiB Infinity Cache (6.58 vs 6.07 TB/s at 2.85 GB), no difference
  // on a fully cache-resident operand set, and -7 % time inside the TTA step (35.0 vs 37.7 us: the misses on the older
  // layers no longer evict the still-resident newer ones)
  p->nt_loads = true;
  const double goal_nchw = std::max(4.0, work_nchw / target_blocks);
  const double goal_nhwc = std::max(8.0, work_nhwc / target_blocks);

  std::vector<BlockEnt> tab_nchw, tab_nhwc;
  int64_t ws = 0;
  for (int l = 0; l < n_layers; ++l) {
    LayerInfo* L = &p->h_info[l];
    L->ws_off = ws;
    if (L->layout == VITTA_LAYOUT_NCHW) {
      int64_t ns = (int64_t)((double)L->outer / goal_nchw + 0.999);
      if (h_nsplit && h_nsplit[l] > 0) ns = h_nsplit[l];  // the caller's frame split (fused BN passes write the triples)
      ns = std::max<int64_t>(1, std::min<int64_t>(ns, L->outer));
      L->nsplit = (int32_t)ns;
      ws += ns * L->nchunks * L->slots;
      for (int sp = 0; sp < L->nsplit; ++sp)
        for (int k = 0; k < L->nchunks; ++k) tab_nchw.push_back(BlockEnt{l, k, sp, 0});
    } else {
      const int TY = VITTA_BLOCK / L->tx;
      const double per_thread = (double)L->outer / TY;
      int64_t ns = (int64_t)(per_thread / goal_nhwc + 0.999);
      ns = std::max<int64_t>(1, std::min<int64_t>(ns, std::max<int64_t>(1, L->outer / TY)));
      L->nsplit = (int32_t)ns;
      ws += ns * L->C;
      for (int sp = 0; sp < L->nsplit; ++sp)
        for (int k = 0; k < L->nchunks; ++k) tab_nhwc.push_back(BlockEnt{l, k, sp, 0});
    }
  }
  // workgroups are dispatched in table order: walk the layers LAST FIRST.  In a training step the hooked features were
  // written in layer order, so the last layers are the ones still in the Infinity Cache; reading the oldest (evicted)
  // ones first would push them out before they are reached.
  // (measured in the TANet step: 35.5-38 us against 41 us in layer order, r1p)
  std::stable_sort(tab_nchw.begin(), tab_nchw.end(), [](const BlockEnt& a, const BlockEnt& b) { return a.layer > b.layer; });
  std::stable_sort(tab_nhwc.begin(), tab_nhwc.end(), [](const BlockEnt& a, const BlockEnt& b) { return a.layer > b.layer; });
  p->ws_triples = ws;
  p->n_blocks_nchw = (int)tab_nchw.size();
  p->n_blocks_nhwc = (int)tab_nhwc.size();

  std::vector<int32_t> c2l((size_t)coff);
  for (int l = 0; l < n_layers; ++l)
    for (int c = 0; c < p->h_info[l].C; ++c) c2l[(size_t)p->h_info[l].chan_off + c] = l;

  auto pad = [](size_t v) { return (v + 255) / 256 * 256; };
  p->off_nchw = pad(sizeof(LayerInfo) * n_layers);
  p->off_nhwc = p->off_nchw + pad(sizeof(BlockEnt) * tab_nchw.size());
  p->off_c2l = p->off_nhwc + pad(sizeof(BlockEnt) * tab_nhwc.size());
  p->off_ticket = p->off_c2l + pad(sizeof(int32_t) * (size_t)coff);
  p->table_bytes = p->off_ticket + 256;
  p->h_tables = calloc(1, p->table_bytes);
  if (!p->h_tables) { delete p; return VITTA_ERR_ALLOC; }
  char* h = static_cast<char*>(p->h_tables);
  memcpy(h, p->h_info, sizeof(LayerInfo) * n_layers);
  if (!tab_nchw.empty()) memcpy(h + p->off_nchw, tab_nchw.data(), sizeof(BlockEnt) * tab_nchw.size());
  if (!tab_nhwc.empty()) memcpy(h + p->off_nhwc, tab_nhwc.data(), sizeof(BlockEnt) * tab_nhwc.size());
  memcpy(h + p->off_c2l, c2l.data(), sizeof(int32_t) * (size_t)coff);
  *out_plan = p;
  return VITTA_OK;
}

int vitta_plan_set_option(vitta_plan* p, int option, int value) {
  if (!p) return VITTA_ERR_INVALID_ARG;
  if (option == VITTA_OPT_NT_LOADS) { p->nt_loads = value != 0; return VITTA_OK; }
  return VITTA_ERR_UNSUPPORTED;
}

size_t vitta_plan_table_bytes(const vitta_plan* p) { return p ? p->table_bytes : 0; }

int vitta_plan_upload(vitta_plan* p, void* d_tables, size_t bytes, void* stream) {
  if (!p || !d_tables || bytes < p->table_bytes) return VITTA_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(d_tables) & 255u) return VITTA_ERR_INVALID_ARG;
  if (hipMemcpyAsync(d_tables, p->h_tables, p->table_bytes, hipMemcpyHostToDevice,
                     static_cast<hipStream_t>(stream)) != hipSuccess)
    return VITTA_ERR_LAUNCH;
  char* d = static_cast<char*>(d_tables);
  p->d_info = reinterpret_cast<LayerInfo*>(d);
  p->d_tab_nchw = reinterpret_cast<BlockEnt*>(d + p->off_nchw);
  p->d_tab_nhwc = reinterpret_cast<BlockEnt*>(d + p->off_nhwc);
  p->d_chan2layer = reinterpret_cast<int32_t*>(d + p->off_c2l);
  p->d_ticket = reinterpret_cast<unsigned*>(d + p->off_ticket);
  return VITTA_OK;
}

void vitta_plan_destroy(vitta_plan* p) {
  if (!p) return;
  free(p->h_tables);  // device tables belong to the caller
  delete p;
}

int64_t vitta_plan_total_channels(const vitta_plan* p) { return p ? p->total_channels : -1; }
int64_t vitta_plan_channel_offset(const vitta_plan* p, int layer) {
  if (!p || layer < 0 || layer >= p->n_layers) return -1;
  return p->h_info[layer].chan_off;
}
size_t vitta_plan_workspace_bytes(const vitta_plan* p) {
  // partial triples + per-channel loss terms used by the align launch
  return p ? sizeof(float) * (3 * (size_t)p->ws_triples + (size_t)p->total_channels) : 0;
}
int64_t vitta_plan_num_blocks(const vitta_plan* p) { return p ? p->n_blocks_nchw + p->n_blocks_nhwc : -1; }

// ev_start / ev_stop (optional): hipEvents attached to the DISPATCH of the streaming kernel (hipExtLaunchKernelGGL):
// they take the kernel's own begin / end timestamps, like a profiler, instead of bracketing it with barrier packets
// (an event pair around a lone launch measured 50 us for a kernel rocprofv3 times at 40 us, r1j).
extern "C++" {
template <typename T>
static int launch_partials(const vitta_plan* p, const void* const* h_x, float* ws, hipStream_t st,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  if (!p->d_info) return VITTA_ERR_INVALID_ARG;  // vitta_plan_upload not called
  PtrPack pack;
  for (int l = 0; l < VITTA_MAX_LAYERS; ++l) pack.x[l] = nullptr;
  for (int l = 0; l < p->n_layers; ++l) {
    const void* x = h_x[l];
    if (!x) return VITTA_ERR_INVALID_ARG;
    if (p->h_info[l].vec == 4 && (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(T) - 1))) return VITTA_ERR_INVALID_ARG;
    pack.x[l] = x;
  }
  if (p->n_blocks_nchw && ev_start) {
    (void)hipGetLastError();
    if (p->nt_loads)
      hipExtLaunchKernelGGL((moments_nchw_partial_kernel<true, T>), dim3(p->n_blocks_nchw), dim3(VITTA_BLOCK), 0, st, ev_start,
                            ev_stop, 0, p->d_info, p->d_tab_nchw, pack, ws);
    else
      hipExtLaunchKernelGGL((moments_nchw_partial_kernel<false, T>), dim3(p->n_blocks_nchw), dim3(VITTA_BLOCK), 0, st, ev_start,
                            ev_stop, 0, p->d_info, p->d_tab_nchw, pack, ws);
    ev_start = ev_stop = nullptr;
  } else if (p->n_blocks_nchw) {
    if (p->nt_loads)
      VITTA_LAUNCH((moments_nchw_partial_kernel<true, T>), dim3(p->n_blocks_nchw), dim3(VITTA_BLOCK), 0, st,
                   p->d_info, p->d_tab_nchw, pack, ws);
    else
      VITTA_LAUNCH((moments_nchw_partial_kernel<false, T>), dim3(p->n_blocks_nchw), dim3(VITTA_BLOCK), 0, st,
                   p->d_info, p->d_tab_nchw, pack, ws);
  }
  if (p->n_blocks_nhwc && ev_start) {  // channels-last plan (Swin): the events go to its kernel
    (void)hipGetLastError();
    if (p->nt_loads)
      hipExtLaunchKernelGGL((moments_nhwc_partial_kernel<true, T>), dim3(p->n_blocks_nhwc), dim3(VITTA_BLOCK), 0, st, ev_start,
                            ev_stop, 0, p->d_info, p->d_tab_nhwc, pack, ws);
    else
      hipExtLaunchKernelGGL((moments_nhwc_partial_kernel<false, T>), dim3(p->n_blocks_nhwc), dim3(VITTA_BLOCK), 0, st, ev_start,
                            ev_stop, 0, p->d_info, p->d_tab_nhwc, pack, ws);
  } else if (p->n_blocks_nhwc) {
    if (p->nt_loads)
      VITTA_LAUNCH((moments_nhwc_partial_kernel<true, T>), dim3(p->n_blocks_nhwc), dim3(VITTA_BLOCK), 0, st,
                   p->d_info, p->d_tab_nhwc, pack, ws);
    else
      VITTA_LAUNCH((moments_nhwc_partial_kernel<false, T>), dim3(p->n_blocks_nhwc), dim3(VITTA_BLOCK), 0, st,
                   p->d_info, p->d_tab_nhwc, pack, ws);
  }
  return VITTA_OK;
}
}  // extern "C++"

int vitta_moments_partials_f32(const vitta_plan* p, const void* const* h_x, void* d_ws, size_t ws_bytes,
                               void* stream) {
  if (!p || !h_x) return VITTA_ERR_INVALID_ARG;
  if (!d_ws || ws_bytes < vitta_plan_workspace_bytes(p)) return VITTA_ERR_WORKSPACE;
  return launch_partials<float>(p, h_x, static_cast<float*>(d_ws), static_cast<hipStream_t>(stream));
}

int vitta_moments_partials_bf16(const vitta_plan* p, const void* const* h_x, void* d_ws, size_t ws_bytes, void* stream) {
  if (!p || !h_x) return VITTA_ERR_INVALID_ARG;
  if (!d_ws || ws_bytes < vitta_plan_workspace_bytes(p)) return VITTA_ERR_WORKSPACE;
  return launch_partials<uint16_t>(p, h_x, static_cast<float*>(d_ws), static_cast<hipStream_t>(stream));
}

int vitta_moments_partials_timed_f32(const vitta_plan* p, const void* const* h_x, void* d_ws, size_t ws_bytes,
                                     void* stream, void* ev_start, void* ev_stop) {
  if (!p || !h_x || !ev_start || !ev_stop) return VITTA_ERR_INVALID_ARG;
  if (!d_ws || ws_bytes < vitta_plan_workspace_bytes(p)) return VITTA_ERR_WORKSPACE;
  return launch_partials<float>(p, h_x, static_cast<float*>(d_ws), static_cast<hipStream_t>(stream),
                                static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop));
}

int vitta_event_create(void** out_event) {
  if (!out_event) return VITTA_ERR_INVALID_ARG;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return VITTA_ERR_LAUNCH;
  *out_event = e;
  return VITTA_OK;
}

void vitta_event_destroy(void* event) {
  if (event) (void)hipEventDestroy(static_cast<hipEvent_t>(event));
}

int vitta_event_elapsed_ms(void* ev_start, void* ev_stop, float* out_ms) {
  if (!ev_start || !ev_stop || !out_ms) return VITTA_ERR_INVALID_ARG;
  if (hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)) != hipSuccess) return VITTA_ERR_LAUNCH;
  return hipEventElapsedTime(out_ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)) == hipSuccess
             ? VITTA_OK : VITTA_ERR_LAUNCH;
}

int vitta_moments_finalize_f32(const vitta_plan* p, const float* d_shift, float* d_cnt, float* d_s1,
                               float* d_s2, const void* d_ws, size_t ws_bytes, void* stream) {
  if (!p || !d_cnt || !d_s1 || !d_s2) return VITTA_ERR_INVALID_ARG;
  if (!d_ws || ws_bytes < vitta_plan_workspace_bytes(p)) return VITTA_ERR_WORKSPACE;
  const int grid = (int)((p->total_channels + VITTA_BLOCK - 1) / VITTA_BLOCK);
  VITTA_LAUNCH(moments_finalize_kernel, dim3(grid), dim3(VITTA_BLOCK), 0, static_cast<hipStream_t>(stream),
                     p->d_info, p->d_chan2layer, p->total_channels, static_cast<const float*>(d_ws), d_shift, 0,
                     d_cnt, d_s1, d_s2);
  return VITTA_OK;
}

}  // extern "C"

template <typename T>
static int moments_batched_t(const vitta_plan* p, const void* const* h_x, const float* d_shift,
                              float* d_cnt, float* d_s1, float* d_s2, void* d_ws, size_t ws_bytes,
                              void* stream) {
  if (!p || !h_x || !d_cnt || !d_s1 || !d_s2) return VITTA_ERR_INVALID_ARG;
  if (!d_ws || ws_bytes < vitta_plan_workspace_bytes(p)) return VITTA_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(d_ws);
  const int rc = launch_partials<T>(p, h_x, ws, st);
  if (rc != VITTA_OK) return rc;
  const int grid = (int)((p->total_channels + VITTA_BLOCK - 1) / VITTA_BLOCK);
  VITTA_LAUNCH(moments_finalize_kernel, dim3(grid), dim3(VITTA_BLOCK), 0, st, p->d_info,
                     p->d_chan2layer, p->total_channels, ws, d_shift, 0, d_cnt, d_s1, d_s2);
  return VITTA_OK;
}

extern "C" {

int vitta_moments_batched_f32(const vitta_plan* p, const void* const* h_x, const float* d_shift,
                              float* d_cnt, float* d_s1, float* d_s2, void* d_ws, size_t ws_bytes,
                              void* stream) {
  return moments_batched_t<float>(p, h_x, d_shift, d_cnt, d_s1, d_s2, d_ws, ws_bytes, stream);
}

int vitta_moments_batched_bf16(const vitta_plan* p, const void* const* h_x, const float* d_shift,
                               float* d_cnt, float* d_s1, float* d_s2, void* d_ws, size_t ws_bytes,
                               void* stream) {
  return moments_batched_t<uint16_t>(p, h_x, d_shift, d_cnt, d_s1, d_s2, d_ws, ws_bytes, stream);
}

int vitta_moments_to_meanvar_f32(const vitta_plan* p, const float* d_shift, const float* d_cnt,
                                 const float* d_s1, const float* d_s2, float* d_mean, float* d_var,
                                 void* stream) {
  if (!p || !d_cnt || !d_s1 || !d_s2 || !d_mean || !d_var) return VITTA_ERR_INVALID_ARG;
  const int grid = (int)((p->total_channels + VITTA_BLOCK - 1) / VITTA_BLOCK);
  VITTA_LAUNCH(moments_to_meanvar_kernel, dim3(grid), dim3(VITTA_BLOCK), 0,
                     static_cast<hipStream_t>(stream), p->d_info, p->d_chan2layer, p->total_channels,
                     d_shift, d_cnt, d_s1, d_s2, d_mean, d_var);
  return VITTA_OK;
}

// ---- single-layer conveniences: no plan object, the tables live in the workspace ----
namespace {
struct SinglePlan {
  LayerInfo info;
  int n_blocks;
  size_t bytes_tab, bytes_info, bytes_c2l, bytes_part;
};
int single_plan(int64_t outer, int32_t C, int64_t inner, int32_t layout, SinglePlan* sp) {
  vitta_layer_shape s{outer, C, inner, layout};
  const int st = fill_layer(s, &sp->info);
  if (st != VITTA_OK) return st;
  LayerInfo& L = sp->info;
  L.chan_off = 0;
  L.ws_off = 0;
  const int target = 2048;
  if (layout == VITTA_LAYOUT_NCHW) {
    const double goal = std::max(4.0, (double)L.nchunks * (double)L.outer / target);
    int64_t ns = (int64_t)((double)L.outer / goal + 0.999);
    L.nsplit = (int32_t)std::max<int64_t>(1, std::min<int64_t>(ns, L.outer));
    sp->bytes_part = sizeof(float) * 3 * (size_t)L.nsplit * L.nchunks * L.slots;
  } else {
    const int TY = VITTA_BLOCK / L.tx;
    const double per_thread = (double)L.outer / TY;
    const double goal = std::max(8.0, (double)L.nchunks * per_thread / target);
    int64_t ns = (int64_t)(per_thread / goal + 0.999);
    L.nsplit = (int32_t)std::max<int64_t>(1, std::min<int64_t>(ns, std::max<int64_t>(1, L.outer / TY)));
    sp->bytes_part = sizeof(float) * 3 * (size_t)L.nsplit * L.C;
  }
  sp->n_blocks = L.nsplit * L.nchunks;
  sp->bytes_info = (sizeof(LayerInfo) + 255) / 256 * 256;
  sp->bytes_tab = (sizeof(BlockEnt) * (size_t)sp->n_blocks + 255) / 256 * 256;
  sp->bytes_c2l = (sizeof(int32_t) * (size_t)C + 255) / 256 * 256;
  sp->bytes_part = (sp->bytes_part + 255) / 256 * 256;
  return VITTA_OK;
}

// Fill the block table / channel map on the device (no host staging, fully async).
__global__ void single_tables_kernel(BlockEnt* tab, int32_t* c2l, LayerInfo* d_info, LayerInfo info,
                                     int nchunks, int nblocks, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *d_info = info;
  if (i < nblocks) tab[i] = BlockEnt{0, i % nchunks, i / nchunks, 0};
  if (i < C) c2l[i] = 0;
}

extern "C++" {
template <typename T>
int moments_single(const T* d_x, int64_t outer, int32_t C, int64_t inner, int32_t layout,
                   float* d_mean, float* d_var, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_x || !d_mean || !d_var) return VITTA_ERR_INVALID_ARG;
  SinglePlan sp;
  const int st = single_plan(outer, C, inner, layout, &sp);
  if (st != VITTA_OK) return st;
  const size_t need = sp.bytes_info + sp.bytes_tab + sp.bytes_c2l + sp.bytes_part;
  if (!d_ws || ws_bytes < need) return VITTA_ERR_WORKSPACE;
  if (sp.info.vec == 4 && (reinterpret_cast<uintptr_t>(d_x) & (4 * sizeof(T) - 1))) return VITTA_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(d_ws);
  LayerInfo* d_info = reinterpret_cast<LayerInfo*>(base);
  BlockEnt* d_tab = reinterpret_cast<BlockEnt*>(base + sp.bytes_info);
  int32_t* d_c2l = reinterpret_cast<int32_t*>(base + sp.bytes_info + sp.bytes_tab);
  float* d_part = reinterpret_cast<float*>(base + sp.bytes_info + sp.bytes_tab + sp.bytes_c2l);
  // tables + the layer record are written by a small kernel (stream ordered, no host staging)
  VITTA_LAUNCH(single_tables_kernel, dim3((std::max(sp.n_blocks, (int)C) + 255) / 256), dim3(256),
                     0, s, d_tab, d_c2l, d_info, sp.info, sp.info.nchunks, sp.n_blocks, (int)C);
  PtrPack pack;
  for (int l = 0; l < VITTA_MAX_LAYERS; ++l) pack.x[l] = nullptr;
  pack.x[0] = d_x;
  if (layout == VITTA_LAYOUT_NCHW)
    VITTA_LAUNCH((moments_nchw_partial_kernel<false, T>), dim3(sp.n_blocks), dim3(VITTA_BLOCK), 0, s, d_info,
                 d_tab, pack, d_part);
  else
    VITTA_LAUNCH((moments_nhwc_partial_kernel<false, T>), dim3(sp.n_blocks), dim3(VITTA_BLOCK), 0, s, d_info,
                       d_tab, pack, d_part);
  VITTA_LAUNCH(moments_finalize_kernel, dim3((C + VITTA_BLOCK - 1) / VITTA_BLOCK), dim3(VITTA_BLOCK),
                     0, s, d_info, d_c2l, (int64_t)C, d_part, (const float*)nullptr, 1, (float*)nullptr,
                     d_mean, d_var);
  return VITTA_OK;
}
}  // extern "C++"
}  // namespace

size_t vitta_moments_workspace_bytes(int64_t outer, int32_t C, int64_t inner, int32_t layout) {
  SinglePlan sp;
  if (single_plan(outer, C, inner, layout, &sp) != VITTA_OK) return 0;
  return sp.bytes_info + sp.bytes_tab + sp.bytes_c2l + sp.bytes_part;
}

int vitta_moments_nchw_f32(const float* d_x, int64_t NT, int32_t C, int64_t HW, float* d_mean,
                           float* d_var, void* d_ws, size_t ws_bytes, void* stream) {
  return moments_single(d_x, NT, C, HW, VITTA_LAYOUT_NCHW, d_mean, d_var, d_ws, ws_bytes, stream);
}

int vitta_moments_nhwc_f32(const float* d_x, int64_t rows, int32_t C, float* d_mean, float* d_var,
                           void* d_ws, size_t ws_bytes, void* stream) {
  return moments_single(d_x, rows, C, 1, VITTA_LAYOUT_NHWC, d_mean, d_var, d_ws, ws_bytes, stream);
}

int vitta_moments_nchw_bf16(const uint16_t* d_x, int64_t NT, int32_t C, int64_t HW, float* d_mean,
                            float* d_var, void* d_ws, size_t ws_bytes, void* stream) {
  return moments_single(d_x, NT, C, HW, VITTA_LAYOUT_NCHW, d_mean, d_var, d_ws, ws_bytes, stream);
}

int vitta_moments_nhwc_bf16(const uint16_t* d_x, int64_t rows, int32_t C, float* d_mean, float* d_var,
                            void* d_ws, size_t ws_bytes, void* stream) {
  return moments_single(d_x, rows, C, 1, VITTA_LAYOUT_NHWC, d_mean, d_var, d_ws, ws_bytes, stream);
}

}  // extern "C"
