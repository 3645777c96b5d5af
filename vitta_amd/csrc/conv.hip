// The 2D convolutions of the TANet trunk as one implicit-GEMM kernel family on v_mfma_f32_32x32x2_f32 (exact fp32).
// Reference call sites: models/tanet_models/temporal_module.py:85-106 (TemporalBottleneck: conv1x1 -> BN -> ReLU -> TAM ->
// conv3x3 -> BN -> ReLU -> conv1x1 -> BN -> +identity -> ReLU) over torchvision's ResNet-50 (tanet.py:125-150).
//
// Data layout: channel-major planes, tensor[c][p] with p = frame * H*W + h * W + w over ALL frames of the clip.
//   D[p][k] = sum_{tap, c} X[c][src(p, tap)] * Wp[tap][c][k]
// Pixels sit on the MFMA ROW axis (A operand), output channels on the COLUMN axis (B operand):
//   * both LDS tiles are [k-slab][pixels | channels] with the fast axis contiguous: an operand read is one ds_read_b32 of 32
//     consecutive floats per half-wave (conflict free), and fp32 MFMA (64 cycles per 32x32x2) leaves the LDS idle anyway;
//   * the 32x32 accumulator layout gives every lane ONE output channel (lane & 31) and 4 CONSECUTIVE pixels per register
//     quad: the epilogue reads / writes 16 bytes per lane, per-channel constants (BatchNorm scale / shift, statistics
//     shift, injection coefficients) are per-lane scalars, and per-channel reductions (hooked-layer moments, d gamma /
//     d beta) are in-register sums + one cross-half shuffle + one atomic per (wave, channel).
// K is walked input-channel slab OUTER, tap INNER: the nine shifted re-reads of a 3x3 convolution's input rows follow
// each other and hit the L1 / L2; each weight slab is read once.  Global -> LDS goes through registers (double-buffered
// LDS, the next slab's loads are in flight under the current slab's MFMAs, one barrier per slab).
// Workgroups are numbered so that the N-tiles of one M-tile run on the same XCD (shared A rows in one L2).
#include "common.h"

using namespace vitta;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PRO_MAX = 2048;  // input channels whose prologue BatchNorm constants are held in LDS

struct ConvK {
  vitta_conv_desc d;
  int64_t xP, yP, rP;  // pixels per channel row of x, y, res
  int Mtot;            // N * Hg * Wg
  int nMt, nNt;        // tiles
  int contig;          // 1: output pixel index == M index (float4 epilogue)
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // dispatcher places workgroup b on XCD b % 8: give every XCD a contiguous range of logical ids (bijective for any nwg)
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int BK, int WM, int WN, bool GATHER>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvK a) {
  constexpr int NTH = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
  constexpr int A4 = BK * BM / 4 / NTH;  // float4 per thread (vector path)
  constexpr int A1 = BK * BM / NTH;      // floats per thread (gather path)
  constexpr int B4 = BK * BN / 4 / NTH;
  constexpr int RSTEP = NTH / BM;        // gather path: slab rows covered per pass
  static_assert(TM % 32 == 0 && TN % 32 == 0 && A4 >= 1 && B4 >= 1 && NTH % BM == 0 && BK % 2 == 0, "tile configuration");
  static_assert((BK * BM / 4) % NTH == 0 && (BK * BN / 4) % NTH == 0, "slab must split evenly over the threads");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                   // [2][BK][BM]
  float* Bs = lds + 2 * BK * BM;     // [2][BK][BN]
  float* cst = Bs + 2 * BK * BN;     // per-output-channel epilogue constants [9][BN]
  float* pro = cst + 9 * BN;         // prologue BN scale / shift [2][C] (only with PRO_BN_RELU)

  const vitta_conv_desc& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lk = lane >> 5;
  const int L = xcd_remap(blockIdx.x, a.nMt * a.nNt);
  const int m0 = (L / a.nNt) * BM, k0 = (L % a.nNt) * BN;
  const int flags = d.flags;
  const bool PRO = flags & VITTA_CONV_PRO_BN_RELU;
  const int C = d.C, K = d.K;

  // ---- per-channel constants ------------------------------------------------------------------------------
  if (tid < BN) {
    const int k = k0 + tid;
    float es = 1.f, et = 0.f, sh = 0.f;
    if (d.epi_bn[0]) {
      es = d.epi_bn[0][k] * rsqrtf(d.epi_bn[3][k] + d.epi_eps);
      et = d.epi_bn[1][k] - d.epi_bn[2][k] * es;
    }
    if (d.st_shift) sh = d.st_shift[k];
    cst[0 * BN + tid] = es;
    cst[1 * BN + tid] = et;
    cst[2 * BN + tid] = sh;
    if (flags & VITTA_CONV_BWD_BN) {
      const float rstd = rsqrtf(d.bwd_bn[3][k] + d.bwd_eps);
      const float bs = d.bwd_bn[0][k] * rstd;
      cst[3 * BN + tid] = bs;
      cst[4 * BN + tid] = d.bwd_bn[1][k] - d.bwd_bn[2][k] * bs;
      cst[5 * BN + tid] = d.bwd_bn[2][k];
      cst[6 * BN + tid] = rstd;
      float gs = 0.f, ia = 0.f, ib = 0.f, mu = 0.f;
      if (d.inj_mu) {
        gs = d.inj_gscale ? d.inj_gscale[0] : 1.f;
        ia = gs * d.inj_a[k];
        ib = gs * d.inj_b[k];
        mu = d.inj_mu[k];
      }
      cst[7 * BN + tid] = ia;
      cst[8 * BN + tid] = ib;
      // mu rides in slot 2 (the forward's statistics shift is unused in backward launches)
      cst[2 * BN + tid] = mu;
    }
  }
  if (PRO) {
    for (int c = tid; c < C; c += NTH) {
      const float s = d.pro_bn[0][c] * rsqrtf(d.pro_bn[3][c] + d.pro_eps);
      pro[c] = s;
      pro[C + c] = d.pro_bn[1][c] - d.pro_bn[2][c] * s;
    }
  }

  // ---- staging geometry -----------------------------------------------------------------------------------
  const int HWs = d.Hs * d.Ws;
  int g_n = 0, g_i = 0, g_j = 0;
  bool g_valid = false;
  if (GATHER) {
    const int m = m0 + (tid % BM);
    g_valid = m < a.Mtot;
    const int hw = d.Hg * d.Wg;
    const int mm = g_valid ? m : 0;
    g_n = mm / hw;
    const int r = mm - g_n * hw;
    g_i = r / d.Wg;
    g_j = r - g_i * d.Wg;
  }

  float4 ra4[GATHER ? 1 : A4];
  float ra1[GATHER ? A1 : 1];
  float4 rb[B4];
  bool ra_valid = false;

  auto load_slab = [&](int q) {
    const int cc = q / d.ntaps, t = q - cc * d.ntaps;
    const int c0 = cc * BK;
    if (GATHER) {
      const int sh = g_i * d.sstride + d.dh[t], sw = g_j * d.sstride + d.dw[t];
      ra_valid = g_valid && (unsigned)sh < (unsigned)d.Hs && (unsigned)sw < (unsigned)d.Ws;
      const int64_t off = (int64_t)g_n * HWs + sh * d.Ws + sw;
      const float* src = d.x + (int64_t)(c0 + tid / BM) * a.xP + off;
#pragma unroll
      for (int u = 0; u < A1; ++u) ra1[u] = ra_valid ? src[(int64_t)u * RSTEP * a.xP] : 0.f;
    } else {
#pragma unroll
      for (int u = 0; u < A4; ++u) {
        const int idx = tid + u * NTH;
        const int kk = idx / (BM / 4), i4 = (idx % (BM / 4)) * 4;
        const int m = m0 + i4;
        ra4[u] = (m < a.Mtot) ? *reinterpret_cast<const float4*>(d.x + (int64_t)(c0 + kk) * a.xP + m)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float* wsrc = d.w + ((int64_t)d.wt[t] * C + c0) * K + k0;
#pragma unroll
    for (int u = 0; u < B4; ++u) {
      const int idx = tid + u * NTH;
      const int kk = idx / (BN / 4), j4 = (idx % (BN / 4)) * 4;
      rb[u] = *reinterpret_cast<const float4*>(wsrc + (int64_t)kk * K + j4);
    }
  };
  auto store_slab = [&](int q, int buf) {
    const int cc = q / d.ntaps;
    const int c0 = cc * BK;
    float* as = As + buf * BK * BM;
    float* bs = Bs + buf * BK * BN;
    if (GATHER) {
#pragma unroll
      for (int u = 0; u < A1; ++u) {
        const int kk = tid / BM + u * RSTEP;
        float v = ra1[u];
        if (PRO) v = ra_valid ? fmaxf(fmaf(v, pro[c0 + kk], pro[C + c0 + kk]), 0.f) : 0.f;
        as[kk * BM + (tid % BM)] = v;
      }
    } else {
#pragma unroll
      for (int u = 0; u < A4; ++u) {
        const int idx = tid + u * NTH;
        const int kk = idx / (BM / 4), i4 = (idx % (BM / 4)) * 4;
        float4 v = ra4[u];
        if (PRO) {
          const float s = pro[c0 + kk], t = pro[C + c0 + kk];
          v.x = fmaxf(fmaf(v.x, s, t), 0.f);
          v.y = fmaxf(fmaf(v.y, s, t), 0.f);
          v.z = fmaxf(fmaf(v.z, s, t), 0.f);
          v.w = fmaxf(fmaf(v.w, s, t), 0.f);
        }
        *reinterpret_cast<float4*>(as + kk * BM + i4) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < B4; ++u) {
      const int idx = tid + u * NTH;
      const int kk = idx / (BN / 4), j4 = (idx % (BN / 4)) * 4;
      *reinterpret_cast<float4*>(bs + kk * BN + j4) = rb[u];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;

  const int nslab = (C / BK) * d.ntaps;
  load_slab(0);
  __syncthreads();  // prologue constants visible before the first store_slab reads them
  store_slab(0, 0);
  __syncthreads();
  for (int q = 0; q < nslab; ++q) {
    const int buf = q & 1;
    if (q + 1 < nslab) load_slab(q + 1);
    const float* as = As + buf * BK * BM + wm * TM + li;
    const float* bs = Bs + buf * BK * BN + wn * TN + li;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float af[MT], bf[NT];
#pragma unroll
      for (int x = 0; x < MT; ++x) af[x] = as[(2 * ks + lk) * BM + 32 * x];
#pragma unroll
      for (int y = 0; y < NT; ++y) bf[y] = bs[(2 * ks + lk) * BN + 32 * y];
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x], bf[y], acc[x][y], 0, 0, 0);
    }
    if (q + 1 < nslab) store_slab(q + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // register v of tile (x, y): pixel row = 32x + 8 (v / 4) + 4 lk + (v % 4), channel column = 32y + li
  const bool BWD = flags & VITTA_CONV_BWD_BN;
  const bool STATS = (flags & VITTA_CONV_STATS) && d.st_s1;
  const bool APPLY = flags & VITTA_CONV_EPI_APPLY;
  const bool RELU = flags & VITTA_CONV_EPI_RELU;
  const bool RES = (flags & VITTA_CONV_RES) && d.res;
  const bool RESH = (flags & VITTA_CONV_RES_HALF) && d.res;
  const bool BRELU = flags & VITTA_CONV_BWD_RELU;
  const int HWy = d.Hy * d.Wy;
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int jl = wn * TN + 32 * y + li;
    const int k = k0 + jl;
    const float es = cst[jl], et = cst[BN + jl], sh = cst[2 * BN + jl];
    float bsc = 0.f, bt = 0.f, brm = 0.f, brs = 0.f, ia = 0.f, ib = 0.f;
    if (BWD) {
      bsc = cst[3 * BN + jl];
      bt = cst[4 * BN + jl];
      brm = cst[5 * BN + jl];
      brs = cst[6 * BN + jl];
      ia = cst[7 * BN + jl];
      ib = cst[8 * BN + jl];
    }
    float r1 = 0.f, r2 = 0.f;  // statistics (forward) or d gamma / d beta (backward) partial sums of this lane
    const int64_t yrow = (int64_t)k * a.yP;
#pragma unroll
    for (int x = 0; x < MT; ++x) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int m = m0 + wm * TM + 32 * x + 8 * qd + 4 * lk;
        if (m >= a.Mtot) continue;
        float v[4] = {acc[x][y][4 * qd], acc[x][y][4 * qd + 1], acc[x][y][4 * qd + 2], acc[x][y][4 * qd + 3]};
        if (a.contig) {
          float* yp = d.y + yrow + m;
          if (RES && BWD) {
            const float4 r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
            if (BWD) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
          }
          if (RESH) {
            const int Hh = (d.Hy + 1) >> 1, Wh = (d.Wy + 1) >> 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int p = m + e;
              const int n = p / HWy, r = p - n * HWy, h = r / d.Wy, w = r - h * d.Wy;
              if (!((h | w) & 1)) v[e] += d.res[(int64_t)k * a.rP + (int64_t)n * Hh * Wh + (h >> 1) * Wh + (w >> 1)];
            }
          }
          if (BWD) {
            const float4 xr = *reinterpret_cast<const float4*>(d.bwd_x + yrow + m);
            const float xv[4] = {xr.x, xr.y, xr.z, xr.w};
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (BRELU && d.bwd_mask) {
              const float4 mr = *reinterpret_cast<const float4*>(d.bwd_mask + yrow + m);
              mk[0] = mr.x > 0.f; mk[1] = mr.y > 0.f; mk[2] = mr.z > 0.f; mk[3] = mr.w > 0.f;
            }
            float o[4], gm[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaf(xv[e], bsc, bt);
              const float mm = (BRELU && !d.bwd_mask) ? (z > 0.f ? 1.f : 0.f) : mk[e];
              gm[e] = v[e] * mm;
              const float dz = gm[e] + fmaf(ib, z - sh, ia);
              r1 += dz * (xv[e] - brm) * brs;
              r2 += dz;
              o[e] = dz * bsc;
            }
            *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
            if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(gm[0], gm[1], gm[2], gm[3]);
          } else {
            if (d.y_raw) *reinterpret_cast<float4*>(d.y_raw + yrow + m) = make_float4(v[0], v[1], v[2], v[3]);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaf(v[e], es, et);
              if (STATS) {
                const float dd = z - sh;
                r1 += dd;
                r2 = fmaf(dd, dd, r2);
              }
              o[e] = APPLY ? z : v[e];
            }
            if (RES) {
              const float4 r = *reinterpret_cast<const float4*>(d.res + (int64_t)k * a.rP + m);
              o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
            }
            if (RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
          }
        } else {
          // scattered destination (data gradient of a stride-2 convolution, one parity class per launch): plain values
          const int hwg = d.Hg * d.Wg;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = m + e;
            const int n = p / hwg, r = p - n * hwg, gi = r / d.Wg, gj = r - gi * d.Wg;
            const int h = gi * d.ostride + d.oa, w = gj * d.ostride + d.ob;
            if (h < d.Hy && w < d.Wy) d.y[yrow + (int64_t)n * HWy + h * d.Wy + w] = v[e];
          }
        }
      }
    }
    if (STATS || BWD) {
      r1 += __shfl_xor(r1, 32, 64);
      r2 += __shfl_xor(r2, 32, 64);
      if (lk == 0) {
        if (BWD) {
          if (d.dgamma) atomicAdd(d.dgamma + k, r1);
          if (d.dbeta) atomicAdd(d.dbeta + k, r2);
        } else {
          atomicAdd(d.st_s1 + k, r1);
          atomicAdd(d.st_s2 + k, r2);
        }
      }
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const ConvK& a, bool gather, hipStream_t st) {
  const size_t lds = sizeof(float) * (2 * BK * BM + 2 * BK * BN + 9 * BN +
                                      ((a.d.flags & VITTA_CONV_PRO_BN_RELU) ? 2 * a.d.C : 0));
  const dim3 grid((unsigned)(a.nMt * a.nNt)), block(WM * WN * 64);
  if (lds > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {  // raise the dynamic-LDS ceiling of this instantiation once (not a stream operation)
    static size_t raised[2] = {0, 0};
    if (lds > raised[gather]) {
      const void* fn = gather ? reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, true>)
                              : reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, WM, WN, false>);
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return VITTA_ERR_LAUNCH;
      raised[gather] = 160 * 1024;
    }
  }
  if (gather) VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, true>), grid, block, lds, st, a);
  else VITTA_LAUNCH((conv_igemm_kernel<BM, BN, BK, WM, WN, false>), grid, block, lds, st, a);
  return VITTA_OK;
}

bool is_vector_geometry(const vitta_conv_desc& d) {
  return d.ntaps == 1 && d.sstride == 1 && d.ostride == 1 && d.dh[0] == 0 && d.dw[0] == 0 && d.Hg == d.Hs && d.Wg == d.Ws &&
         d.Hy == d.Hg && d.Wy == d.Wg;
}

// Tile choice: the largest tile that still gives the launch about two workgroups per CU; small problems fall through to
// 64 x 64 / 64 x 32 (more, shorter workgroups).  BN must divide K.
void choose_tile(const vitta_conv_desc& d, int64_t M, int& bm, int& bn) {
  if (d.tile) {
    bm = d.tile >> 16;
    bn = d.tile & 0xffff;
    return;
  }
  const int cand[][2] = {{128, 128}, {128, 64}, {64, 64}, {64, 32}};
  for (auto& c : cand) {
    if (d.K % c[1]) continue;
    const int64_t blocks = ((M + c[0] - 1) / c[0]) * (d.K / c[1]);
    bm = c[0];
    bn = c[1];
    if (blocks >= 448) return;
  }
}

int fill(const vitta_conv_desc* h, ConvK& a) {
  if (!h || !h->x || !h->w || !h->y) return VITTA_ERR_INVALID_ARG;
  a.d = *h;
  const vitta_conv_desc& d = a.d;
  if (d.C <= 0 || d.K <= 0 || d.N <= 0 || d.ntaps < 1 || d.ntaps > VITTA_CONV_MAX_TAPS || d.sstride < 1 || d.ostride < 1 ||
      d.ostride > 2)
    return VITTA_ERR_INVALID_ARG;
  if (d.C % 16 || d.K % 32) return VITTA_ERR_UNSUPPORTED;
  a.xP = (int64_t)d.N * d.Hs * d.Ws;
  a.yP = (int64_t)d.N * d.Hy * d.Wy;
  const int64_t M = (int64_t)d.N * d.Hg * d.Wg;
  if (M >= (1ll << 31) || a.xP % 4 || a.yP % 4 || M % 4) return VITTA_ERR_UNSUPPORTED;
  a.Mtot = (int)M;
  a.contig = (d.ostride == 1 && d.oa == 0 && d.ob == 0 && d.Hg == d.Hy && d.Wg == d.Wy) ? 1 : 0;
  a.rP = (d.flags & VITTA_CONV_RES_HALF) ? (int64_t)d.N * ((d.Hy + 1) / 2) * ((d.Wy + 1) / 2) : a.yP;
  if (!a.contig && (d.flags & (VITTA_CONV_BWD_BN | VITTA_CONV_STATS | VITTA_CONV_RES | VITTA_CONV_RES_HALF | VITTA_CONV_EPI_APPLY |
                               VITTA_CONV_EPI_RELU)))
    return VITTA_ERR_UNSUPPORTED;
  if ((d.flags & VITTA_CONV_PRO_BN_RELU) && (d.C > PRO_MAX || !d.pro_bn[0] || !d.pro_bn[1] || !d.pro_bn[2] || !d.pro_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & (VITTA_CONV_EPI_APPLY | VITTA_CONV_STATS)) && (!d.epi_bn[0] || !d.epi_bn[1] || !d.epi_bn[2] || !d.epi_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_STATS) && (!d.st_shift || !d.st_s1 || !d.st_s2)) return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_BWD_BN) && (!d.bwd_x || !d.bwd_bn[0] || !d.bwd_bn[1] || !d.bwd_bn[2] || !d.bwd_bn[3]))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & VITTA_CONV_BWD_BN) && (d.flags & (VITTA_CONV_STATS | VITTA_CONV_EPI_APPLY | VITTA_CONV_EPI_RELU)))
    return VITTA_ERR_INVALID_ARG;
  if ((d.flags & (VITTA_CONV_RES | VITTA_CONV_RES_HALF)) && !d.res) return VITTA_ERR_INVALID_ARG;
  int bm = 64, bn = 32;
  choose_tile(d, M, bm, bn);
  if (d.K % bn) return VITTA_ERR_UNSUPPORTED;
  a.nMt = (int)((M + bm - 1) / bm);
  a.nNt = d.K / bn;
  a.d.tile = (bm << 16) | bn;
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_conv_supported(const vitta_conv_desc* h_desc) {
  ConvK a;
  return fill(h_desc, a) == VITTA_OK ? 1 : 0;
}

int64_t vitta_conv_num_blocks(const vitta_conv_desc* h_desc) {
  ConvK a;
  if (fill(h_desc, a) != VITTA_OK) return -1;
  return (int64_t)a.nMt * a.nNt;
}

int vitta_conv_f32(const vitta_conv_desc* h_desc, void* stream) {
  ConvK a;
  const int rc = fill(h_desc, a);
  if (rc != VITTA_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool gather = !is_vector_geometry(a.d);
  const int bm = a.d.tile >> 16, bn = a.d.tile & 0xffff;
  const int bk = (a.d.C % 32 == 0) ? 32 : 16;
#define CFG(M_, N_, K_, WM_, WN_) \
  if (bm == M_ && bn == N_ && bk == K_) return launch_cfg<M_, N_, K_, WM_, WN_>(a, gather, st)
  CFG(128, 128, 32, 2, 2);
  CFG(128, 64, 32, 2, 2);
  CFG(64, 64, 32, 2, 2);
  CFG(64, 32, 32, 2, 1);
  CFG(128, 128, 16, 2, 2);
  CFG(128, 64, 16, 2, 2);
  CFG(64, 64, 16, 2, 2);
  CFG(64, 32, 16, 2, 1);
#undef CFG
  return VITTA_ERR_UNSUPPORTED;
}

}  // extern "C"
