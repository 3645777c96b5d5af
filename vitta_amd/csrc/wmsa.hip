// Fused 3-D shifted-window multi-head self-attention (W-MSA / SW-MSA) for Video Swin.
//
// Reference (models/videoswintransformer_models/swin_transformer.py:138-169), per window b and head h:
//   q = q * scale ; attn = q @ k^T                       [N x N] materialised (315 MB per block at stage 1)
//   attn += relative_position_bias[h] (+ mask[b % nW])    2 more passes
//   attn = softmax(attn) ; x = attn @ v                   2 more passes, + the head permutes
// Here: one kernel keeps the [16 x N] score rows of a wave in registers (N <= 400: 25 MFMA tiles of
// 4 accumulators), nothing of size N x N ever reaches HBM.  fp32 in / fp32 accumulate on
// v_mfma_f32_16x16x4_f32 (exact f32, the reference is fp32 end to end).
//
// Token rows: window b, token n lives at row b*N + n of qkv / out (the reference's partitioned layout), or -- with a
// row map -- at row (b / nWm)*L + rowmap[b % nWm][n] of the NATURAL [B, D*H*W] token order: the cyclic shift and the
// window partition of swin_transformer.py:222-243 (and their inverses) become address arithmetic, the four copies
// per shifted block disappear (qkv / proj are per-token, so they commute with the permutation).
// Layouts: qkv [rows, 3, nH, 32] exactly as the qkv Linear writes it; out [B_, N, nH*32] as the proj
// Linear reads it (the reference's reshape/permute/transpose copies disappear); bias [nH, N, N];
// mask [nW, N, N] or null (window b uses mask[b % nW]); lse [B_, nH, N] = row max + log row sum.
//
// MFMA operand trick: the score tile is computed TRANSPOSED, S^T[key][query] = K Q^T.  Its C/D layout
// (col = lane&15 = query, row = 4*(lane>>4)+r = key) is exactly the A-operand layout the second GEMM
// needs (A[i = lane&15][k = lane>>4] with key = 16t + 4*(lane>>4) + r at k-step r), so P feeds P.V
// straight from the accumulators: no LDS round trip, no shuffles.
#include <cstdlib>

#include "common.h"

using namespace vitta;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32;          // head dim of every Swin variant on the path (C / nH)
constexpr int KPAD = 36;        // LDS row stride in floats: 36 j mod 64 are distinct multiples of 4 -> conflict-free b128
constexpr int NT_MAX = 25;      // 16-token tiles per window: N <= 400 (8*7*7 = 392)
constexpr int WMSA_THREADS = 512;
constexpr int WMSA_WAVES = WMSA_THREADS / 64;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Stage rows [0, N) of a [rows, row_stride] fp32 matrix slice (32 floats per row) into LDS with row stride KPAD;
// rows [N, 16*nt) are zeroed.  The loads of a lane are issued four at a time BEFORE their LDS stores so that
// 4 x 16 B per lane are in flight (a load->store loop serialises on every HBM/L2 round trip; the staging of the
// two 50 KB operands was longer than the MFMA work of a workgroup at the late stages).
__device__ __forceinline__ void stage_rows_dense(float* __restrict__ dst, const float* __restrict__ src_base,
                                                 int64_t row_stride, int N, int nt, const int* __restrict__ rows) {
  const int total = 16 * nt * (HD / 4);
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * WMSA_THREADS) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * WMSA_THREADS;
      const int row = i / (HD / 4), c4 = i % (HD / 4);
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total && row < N) v[u] = *reinterpret_cast<const float4*>(src_base + (int64_t)rows[row] * row_stride + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * WMSA_THREADS;
      if (i < total) *reinterpret_cast<float4*>(dst + (i / (HD / 4)) * KPAD + 4 * (i % (HD / 4))) = v[u];
    }
  }
}

// rows of one of q/k/v (sel) of head h: qkv is [rows, 3, nH, 32]
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ qkv, int h, int sel, int N,
                                           int nH, int nt, const int* __restrict__ rows) {
  const int64_t rs = 3 * (int64_t)nH * HD;  // floats per token
  stage_rows_dense(dst, qkv + (int64_t)sel * nH * HD + (int64_t)h * HD, rs, N, nt, rows);
}

// where the tokens of window b live (see the header): rows[n] for n < 16*nt (clamped past N), then a barrier
struct RowMap {
  const int* map;  // [nWm, N] sample-local rows, or null
  int nWm;
  int64_t L;       // tokens per sample
};
__device__ __forceinline__ void fill_rows(int* __restrict__ rows, const RowMap& rm, int64_t b, int N, int nt) {
  for (int i = threadIdx.x; i < 16 * nt; i += blockDim.x) {
    const int n = i < N ? i : N - 1;
    rows[i] = rm.map ? (int)((b / rm.nWm) * rm.L + rm.map[(b % rm.nWm) * (int64_t)N + n]) : (int)(b * N + n);
  }
  __syncthreads();
}

// 8 contiguous floats of a row (d = 8*kk .. 8*kk+7)
__device__ __forceinline__ void load8(float (&r)[8], const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
}

// Where the additive terms of a score come from.
//  DENSE: bias [nH,N,N] and mask [nW,N,N] in global memory (generic form, what the reference builds).
//  REL  : the relative-position bias is table[code[q] - code[k] + off][h] (the index is linear in the
//         token coordinates, swin_transformer.py:113-124) and the shift mask is -100 where the tokens'
//         pre-shift region ids differ (swin_transformer.py:316-329): table column, codes and region ids
//         sit in LDS, so the N x N terms cost no memory traffic at all.
struct AddTerms {
  // DENSE
  const float* bias_h;   // bias + h*N*N
  const float* mask_b;   // mask + (b % nW)*N*N or null
  // REL
  const float* tab;      // LDS: table[:, h]            [T]
  const int* cr;         // LDS: code[n] | region[n] << 16 of window b, padded to 16*nt entries (region 0: no mask)
  int off;
};

__device__ __forceinline__ int pk_code(int p) { return p & 0xffff; }
__device__ __forceinline__ int pk_region(int p) { return p >> 16; }

// branch-free: the packed entries exist for every padded token, so the lookups never need a guard
__device__ __forceinline__ float rel_term(const AddTerms& a, int pq, int pk) {
  const float v = a.tab[pk_code(pq) - pk_code(pk) + a.off];
  return pk_region(pq) != pk_region(pk) ? v - 100.f : v;
}

template <bool REL>
__device__ __forceinline__ float add_term(const AddTerms& a, int q, int key, int N) {
  if constexpr (REL) {
    return rel_term(a, a.cr[q], a.cr[key]);
  } else {
    float v = a.bias_h[(int64_t)q * N + key];
    if (a.mask_b) v += a.mask_b[(int64_t)q * N + key];
    return v;
  }
}

// S^T tile t for the wave's 16 queries: acc[r] = S[query q][key = 16t + 4*(lane>>4) + r] (scaled, + bias + mask)
// REG: the window carries region ids (a shifted block; else every packed entry is a bare code and the mask arithmetic -- three
// vector instructions per score -- is compiled out).  FULL: all sixteen keys of the tile exist (every tile but a ragged last one).
template <bool REL, bool REG = true, bool FULL = false>
__device__ __forceinline__ f32x4 score_tile(const float* __restrict__ k_lds, const float (&qf)[8], int t, int lane,
                                            const AddTerms& a, int q, int N) {
  const int j = lane & 15, kk = lane >> 4;
  float kf[8];
  load8(kf, k_lds + (16 * t + j) * KPAD + 8 * kk);
  // two accumulator chains: a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles for its predecessor while the pipe could
  // issue every 32 (PMC: SQ_WAIT_INST_ANY was the largest share of the forward kernel's wave cycles)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; s += 2) {
    acc = mfma(kf[s], qf[s], acc);
    acc2 = mfma(kf[s + 1], qf[s + 1], acc2);
  }
  acc += acc2;
  const int key0 = 16 * t + 4 * kk;
  if constexpr (REL) {
    // the four keys of this lane are consecutive: one 16-byte LDS read brings their packed code | region, then four
    // independent table reads -- no branch, no dependent round trip per element (the per-element form cost three
    // serialised LDS latencies each: 36 k cycles per row tile against 12.8 k cycles of MFMA)
    const int4 ck = *reinterpret_cast<const int4*>(a.cr + key0);
    const int pq = a.cr[q];
    const int cq = (REG ? pk_code(pq) : pq) + a.off, rq = pk_region(pq);
    const int kc[4] = {ck.x, ck.y, ck.z, ck.w};
    float tv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = a.tab[cq - (REG ? pk_code(kc[r]) : kc[r])];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = (REG && pk_region(kc[r]) != rq) ? tv[r] - 100.f : tv[r];
      acc[r] = (FULL || key0 + r < N) ? acc[r] + v : -INFINITY;
    }
  } else {
    if (key0 + 3 < N && (N & 3) == 0) {  // 4 consecutive keys of one row: one 16-byte load each
      const float4 bv = *reinterpret_cast<const float4*>(a.bias_h + (int64_t)q * N + key0);
      acc[0] += bv.x; acc[1] += bv.y; acc[2] += bv.z; acc[3] += bv.w;
      if (a.mask_b) {
        const float4 mv = *reinterpret_cast<const float4*>(a.mask_b + (int64_t)q * N + key0);
        acc[0] += mv.x; acc[1] += mv.y; acc[2] += mv.z; acc[3] += mv.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + r;
        if (key < N) acc[r] += add_term<false>(a, q, key, N);
        else acc[r] = -INFINITY;
      }
    }
  }
  return acc;
}

// LDS carve shared by the three kernels: [buf0 | buf1 | extra floats | tab | code | region]
struct Carve {
  float* buf0; float* buf1; float* extra; float* tab; int* code; int* region; int* rows;
};
__device__ __forceinline__ Carve carve(float* smem, int nt, int extra_floats, int T) {
  Carve c;
  c.buf0 = smem;
  c.buf1 = c.buf0 + 16 * nt * KPAD;
  c.extra = c.buf1 + 16 * nt * KPAD;
  c.tab = c.extra + extra_floats;
  c.code = reinterpret_cast<int*>(c.tab + ((T + 3) & ~3));
  c.region = c.code + 16 * nt;
  c.rows = c.region + 16 * nt;
  return c;
}

// fill the REL terms of (b, h) into LDS and return the provider
template <bool REL>
__device__ __forceinline__ AddTerms setup_terms(const Carve& c, const float* bias_or_table, const float* mask,
                                                const int* code_g, const int* region_g, int T, int off, int nW, int N,
                                                int nH, int h, int64_t b, int nt) {
  AddTerms a;
  a.bias_h = nullptr; a.mask_b = nullptr; a.tab = nullptr; a.cr = nullptr; a.off = off;
  if constexpr (REL) {
    // the table column is a strided gather of T (2535) floats: all of a lane's loads in flight together (a plain loop
    // is five dependent L2 round trips at the head of every workgroup)
    for (int i0 = threadIdx.x; i0 < T; i0 += 8 * WMSA_THREADS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = bias_or_table[(int64_t)min(i0 + u * WMSA_THREADS, T - 1) * nH + h];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * WMSA_THREADS < T) c.tab[i0 + u * WMSA_THREADS] = v[u];
    }
    for (int i = threadIdx.x; i < 16 * nt; i += WMSA_THREADS) {
      const int n = i < N ? i : N - 1;
      const int reg = region_g ? region_g[(b % nW) * (int64_t)N + n] : 0;
      c.code[i] = code_g[n] | (reg << 16);
    }
    a.tab = c.tab; a.cr = c.code;
  } else {
    a.bias_h = bias_or_table + (int64_t)h * N * N;
    a.mask_b = mask ? mask + (b % nW) * (int64_t)N * N : nullptr;
  }
  return a;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// NTC: compile-time number of 16-token tiles (25 for the (8,7,7) window of every Swin-B stage) -- the unrolled tile loops
// then carry no per-tile guard and the scheduler can overlap the LDS reads of one tile with the MFMAs of another;
// 0 = run-time count.
template <bool REL, int NTC, bool REG>
__global__ __launch_bounds__(WMSA_THREADS) void wmsa_fwd_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ mask,
                                                                const int* __restrict__ code_g,
                                                                const int* __restrict__ region_g, int T, int off,
                                                                int nW, int N, int nH, float scale, int qsplit,
                                                                RowMap rm, float* __restrict__ out,
                                                                float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt = NTC ? NTC : (N + 15) / 16;
  const Carve cv = carve(smem, nt, 0, T);
  float* k_lds = cv.buf0;                   // [16*nt][KPAD]
  float* v_lds = cv.buf1;                   // [16*nt][KPAD]
  const int* rows = cv.rows;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  fill_rows(cv.rows, rm, b, N, nt);
  stage_rows(k_lds, qkv, h, 1, N, nH, nt, rows);
  stage_rows(v_lds, qkv, h, 2, N, nH, nt, rows);
  const AddTerms terms = setup_terms<REL>(cv, bias, mask, code_g, region_g, T, off, nW, N, nH, h, b, nt);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const float* q_base = qkv + (int64_t)h * HD;
  const int C = nH * HD;

  // row tiles of this workgroup: [rt0, rt1) of the nt tiles, split over `qsplit` workgroups
  const int per = (nt + qsplit - 1) / qsplit;
  const int rt0 = blockIdx.x * per, rt1 = min(nt, rt0 + per);
  for (int rt = rt0 + wave; rt < rt1; rt += WMSA_WAVES) {
    const int q = min(16 * rt + i, N - 1);  // clamped: rows >= N are computed but never stored
    float qf[8];
    load8(qf, q_base + (int64_t)rows[q] * rs + 8 * kk);
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] *= scale;

    f32x4 acc[NT_MAX];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < (NTC ? NTC : NT_MAX); ++t) {
      if (NTC || t < nt) {
        // (t is a compile-time constant of the unrolled loop when NTC is: every tile but the last takes the unguarded form)
        acc[t] = (NTC ? t + 1 < NTC : t + 1 < nt) ? score_tile<REL, REG, true>(k_lds, qf, t, lane, terms, q, N)
                                                  : score_tile<REL, REG, false>(k_lds, qf, t, lane, terms, q, N);
        m = fmaxf(m, fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3])));
        if (NTC) __builtin_amdgcn_sched_barrier(0);  // keep one tile's loads from being hoisted over the previous tiles
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < (NTC ? NTC : NT_MAX); ++t) {
      if (NTC || t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(acc[t][r] - m);
          acc[t][r] = p;
          l += p;
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.f / l;

    // O = P V: two 16-wide halves of the head dim
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (NTC ? NTC : NT_MAX); ++t) {
      if (NTC || t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vrow = v_lds + (16 * t + 4 * kk + r) * KPAD;
          o0 = mfma(acc[t][r], vrow[i], o0);
          o1 = mfma(acc[t][r], vrow[16 + i], o1);
        }
        if (NTC) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // C layout: row = query 4*kk + r, col = d = i; the row's 1/l lives in lane (4*kk + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * kk + r;
      const float il = __shfl(inv_l, 4 * kk + r, 64);
      if (qrow < N) {
        float* o = out + (int64_t)rows[qrow] * C + h * HD;
        o[i] = o0[r] * il;
        o[16 + i] = o1[r] * il;
      }
    }
    if (kk == 0 && 16 * rt + i < N) lse[(b * nH + h) * N + 16 * rt + i] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------------
// backward 1: dQ (query-tile major, K and V staged) + optional dbias via atomics
//   P = exp(S - lse); dP = dO V^T; dS = P o (dP - delta); dQ = scale * dS K
// ------------------------------------------------------------------------------------------------
template <bool REL, bool REG>
__global__ __launch_bounds__(WMSA_THREADS) void wmsa_bwd_dq_kernel(
    const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ mask,
    const int* __restrict__ code_g, const int* __restrict__ region_g, int T, int off, int nW, int N, int nH,
    float scale, int qsplit, RowMap rm, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, float* __restrict__ dbias, int dfix) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt = (N + 15) / 16;
  // REL + dbias: the table gradient of this workgroup accumulates in LDS, one global atomic per table entry at the end.  Round 6: the
  // column is 64-bit FIXED POINT and the binning ds_add_u64 -- a ds_add_f32 instruction occupies the CU's LDS for 192 clocks, a
  // ds_add_u64 for 8 (tools/ubench/lds_atomic_probe.hip).  The power-of-two scale of the workgroup comes from a bound no sum can
  // exceed: |dS| = p |dP - delta| <= 64 max|dO| max|V|, at most N scores per table row (as wmsa_bf16.hip); ldexpf(dS, e) is an integer
  // the conversion represents exactly, so the column is the exact sum of the fp32 dS values, whatever the order of the waves.
  // (dfix = 0: the 64-bit column does not fit LDS beside K and V -- the (16,7,7) table on a clamped window --, float column as before)
  const Carve cv = carve(smem, nt, (REL && dbias) ? (dfix ? 2 : 1) * ((T + 3) & ~3) : 0, T);
  float* k_lds = cv.buf0;
  float* v_lds = cv.buf1;
  unsigned long long* dtab = reinterpret_cast<unsigned long long*>(cv.extra);
  float* dtabf = cv.extra;
  const int* rows = cv.rows;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  fill_rows(cv.rows, rm, b, N, nt);
  stage_rows(k_lds, qkv, h, 1, N, nH, nt, rows);
  stage_rows(v_lds, qkv, h, 2, N, nH, nt, rows);
  const AddTerms terms = setup_terms<REL>(cv, bias, mask, code_g, region_g, T, off, nW, N, nH, h, b, nt);
  if (REL && dbias)
    for (int i = threadIdx.x; i < T; i += WMSA_THREADS) {
      if (dfix) dtab[i] = 0ull;
      else dtabf[i] = 0.f;
    }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  const float* q_base = qkv + (int64_t)h * HD;
  const int per = (nt + qsplit - 1) / qsplit;
  const int rt0 = blockIdx.x * per, rt1 = min(nt, rt0 + per);
  int dexp = 0;
  bool dbad = false;
  if (REL && dbias && dfix) {  // the fixed-point exponent: max |V| from the staged rows, max |dO| over the window's rows (one pass)
    float vmax = 0.f, gmax = 0.f;
    for (int it = threadIdx.x; it < N * 8; it += WMSA_THREADS) {
      const int row = it >> 3, c4 = it & 7;
      const float4 v4 = *reinterpret_cast<const float4*>(v_lds + row * KPAD + 4 * c4);
      const float4 g4 = *reinterpret_cast<const float4*>(dout + (int64_t)rows[row] * C + h * HD + 4 * c4);
      const float vv[8] = {v4.x, v4.y, v4.z, v4.w, g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vmax = fmaxf(vmax, fabsf(vv[u]) + (vv[u] != vv[u] ? INFINITY : 0.f));
        gmax = fmaxf(gmax, fabsf(vv[4 + u]) + (vv[4 + u] != vv[4 + u] ? INFINITY : 0.f));
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    __shared__ float red_max[2 * WMSA_WAVES];
    if (lane == 0) {
      red_max[wave] = gmax;
      red_max[WMSA_WAVES + wave] = vmax;
    }
    __syncthreads();
    gmax = vmax = 0.f;
#pragma unroll
    for (int w = 0; w < WMSA_WAVES; ++w) {
      gmax = fmaxf(gmax, red_max[w]);
      vmax = fmaxf(vmax, red_max[WMSA_WAVES + w]);
    }
    const float B = 64.f * gmax * vmax * (float)N;
    dbad = !(B < INFINITY);
    int ex = 0;
    if (!dbad && B > 0.f) (void)frexpf(B, &ex);
    dexp = dbad ? 0 : min(100, 60 - ex);
  }
  for (int rt = rt0 + wave; rt < rt1; rt += WMSA_WAVES) {
    const bool qvalid = 16 * rt + i < N;
    const int q = min(16 * rt + i, N - 1);
    float qf[8], gf[8], of[8];
    load8(qf, q_base + (int64_t)rows[q] * rs + 8 * kk);
    load8(gf, dout + (int64_t)rows[q] * C + h * HD + 8 * kk);
    load8(of, out + (int64_t)rows[q] * C + h * HD + 8 * kk);
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qf[s] *= scale;
      dl = fmaf(gf[s], of[s], dl);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);  // delta[q] = sum_d dO[q][d] O[q][d]
    const float L = lse[(b * nH + h) * N + q];
    if (kk == 0 && qvalid) delta[(b * nH + h) * N + q] = dl;
    float* dbias_row = (!REL && dbias) ? dbias + ((int64_t)h * N + q) * N : nullptr;

    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      f32x4 s = t + 1 < nt ? score_tile<REL, REG, true>(k_lds, qf, t, lane, terms, q, N)
                           : score_tile<REL, REG, false>(k_lds, qf, t, lane, terms, q, N);
      // dP^T tile = V dO^T (same C layout as S^T)
      float vf[8];
      load8(vf, v_lds + (16 * t + i) * KPAD + 8 * kk);
      f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) dp = mfma(vf[u], gf[u], dp);
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[r] - L);  // exp(-inf) = 0 for padded keys
        ds[r] = p * (dp[r] - dl);
      }
      if (dbias && qvalid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * t + 4 * kk + r;
          if (key < N) {
            if constexpr (REL) {
              const int bin = pk_code(terms.cr[q]) - pk_code(terms.cr[key]) + terms.off;
              if (dfix) atomicAdd(dtab + bin, dbad ? 1ull : (unsigned long long)__float2ll_rn(ldexpf(ds[r], dexp)));
              else atomicAdd(dtabf + bin, ds[r]);
            } else {
              atomicAdd(dbias_row + key, ds[r]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* krow = k_lds + (16 * t + 4 * kk + r) * KPAD;
        dq0 = mfma(ds[r], krow[i], dq0);
        dq1 = mfma(ds[r], krow[16 + i], dq1);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * kk + r;
      if (qrow < N) {
        float* o = dqkv + (int64_t)rows[qrow] * rs + (int64_t)h * HD;  // sel 0 = q
        o[i] = dq0[r] * scale;
        o[16 + i] = dq1[r] * scale;
      }
    }
  }
  if (REL && dbias) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += WMSA_THREADS) {
      float v;
      if (dfix) {
        const long long w = (long long)dtab[t];
        v = dbad ? (w != 0 ? NAN : 0.f) : (float)ldexp((double)w, -dexp);
      } else {
        v = dtabf[t];
      }
      if (v != 0.f) atomicAdd(dbias + (int64_t)t * nH + h, v);  // REL: dbias is the table gradient [T, nH]
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward 2: dK, dV (key-tile major, Q and dO staged)
//   dV = P^T dO ; dK = scale * dS^T Q
// ------------------------------------------------------------------------------------------------
template <bool REL, bool REG>
__global__ __launch_bounds__(WMSA_THREADS) void wmsa_bwd_dkv_kernel(
    const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ mask,
    const int* __restrict__ code_g, const int* __restrict__ region_g, int T, int off, int nW, int N, int nH,
    float scale, int qsplit, RowMap rm, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt = (N + 15) / 16;
  const Carve cv = carve(smem, nt, 2 * 16 * nt, T);
  float* q_lds = cv.buf0;                    // raw (unscaled) Q
  float* g_lds = cv.buf1;                    // dO
  float* l_lds = cv.extra;                   // lse  [16*nt]
  float* d_lds = l_lds + 16 * nt;            // delta [16*nt]
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int C = nH * HD;
  const int* rows = cv.rows;
  fill_rows(cv.rows, rm, b, N, nt);
  stage_rows(q_lds, qkv, h, 0, N, nH, nt, rows);
  stage_rows_dense(g_lds, dout + h * HD, C, N, nt, rows);
  const AddTerms terms = setup_terms<REL>(cv, bias, mask, code_g, region_g, T, off, nW, N, nH, h, b, nt);
  for (int r = threadIdx.x; r < 16 * nt; r += WMSA_THREADS) {
    l_lds[r] = r < N ? lse[(b * nH + h) * N + r] : INFINITY;  // exp(s - inf) = 0 for padded queries
    d_lds[r] = r < N ? delta[(b * nH + h) * N + r] : 0.f;
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const float* k_base = qkv + (int64_t)(nH + h) * HD;
  const float* v_base = qkv + (int64_t)(2 * nH + h) * HD;
  const int per = (nt + qsplit - 1) / qsplit;
  const int kt0 = blockIdx.x * per, kt1 = min(nt, kt0 + per);
  for (int kt = kt0 + wave; kt < kt1; kt += WMSA_WAVES) {
    const int key = min(16 * kt + i, N - 1);
    const bool kvalid = 16 * kt + i < N;
    float kf[8], vf[8];
    const int pkey = REL ? terms.cr[key] : 0;
    load8(kf, k_base + (int64_t)rows[key] * rs + 8 * kk);
    load8(vf, v_base + (int64_t)rows[key] * rs + 8 * kk);
    f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = {0.f, 0.f, 0.f, 0.f}, dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < nt; ++qt) {
      // S tile [query][key]: A = Q tile (LDS), B = K^T (registers); C layout: col = key i, row = query 4kk + r
      float qf[8], gf[8];
      load8(qf, q_lds + (16 * qt + i) * KPAD + 8 * kk);
      load8(gf, g_lds + (16 * qt + i) * KPAD + 8 * kk);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s = mfma(qf[u], kf[u], s);
        dp = mfma(gf[u], vf[u], dp);
      }
      f32x4 p, ds;
      if constexpr (REL) {
        // No validity test per score: a padded QUERY row has lse = +inf in LDS (p = exp(finite - inf) = 0), its packed code and its
        // zero-filled q row exist; a padded KEY lane (clamped to the last key) produces values nobody stores.  The four queries of
        // this lane are consecutive: one 16-byte read each for their packed codes, lse and delta.
        const int q0 = 16 * qt + 4 * kk;
        const int4 cq4 = *reinterpret_cast<const int4*>(terms.cr + q0);
        const float4 l4 = *reinterpret_cast<const float4*>(l_lds + q0), d4 = *reinterpret_cast<const float4*>(d_lds + q0);
        const int cq[4] = {cq4.x, cq4.y, cq4.z, cq4.w};
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
        const int ck = (REG ? pk_code(pkey) : pkey) - terms.off, rk = pk_region(pkey);
        float tv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[r] = terms.tab[(REG ? pk_code(cq[r]) : cq[r]) - ck];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float term = (REG && pk_region(cq[r]) != rk) ? tv[r] - 100.f : tv[r];
          p[r] = __expf(fmaf(s[r], scale, term) - lq[r]);
          ds[r] = p[r] * (dp[r] - dq_[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * kk + r;
          float sv = -INFINITY;
          if (q < N && kvalid) sv = s[r] * scale + add_term<false>(terms, q, key, N);
          p[r] = __expf(sv - l_lds[q]);
          ds[r] = p[r] * (dp[r] - d_lds[q]);
        }
      }
      // dV[key][d] += P^T dO ; dK[key][d] += dS^T Q   (A = C-layout values, B rows = queries 16qt + 4kk + r)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* grow = g_lds + (16 * qt + 4 * kk + r) * KPAD;
        const float* qrow = q_lds + (16 * qt + 4 * kk + r) * KPAD;
        dv0 = mfma(p[r], grow[i], dv0);
        dv1 = mfma(p[r], grow[16 + i], dv1);
        dk0 = mfma(ds[r], qrow[i], dk0);
        dk1 = mfma(ds[r], qrow[16 + i], dk1);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = 16 * kt + 4 * kk + r;
      if (krow < N) {
        float* ok = dqkv + (int64_t)rows[krow] * rs + (int64_t)(nH + h) * HD;
        float* ov = dqkv + (int64_t)rows[krow] * rs + (int64_t)(2 * nH + h) * HD;
        ok[i] = dk0[r] * scale;
        ok[16 + i] = dk1[r] * scale;
        ov[i] = dv0[r];
        ov[16 + i] = dv1[r];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// backward in ONE pass (round 6; relative-position form with a frozen table: the LN-affine adaptation of BASELINE config 3).
// The two kernels above evaluate every score tile twice -- S and dP once query-major for dQ, once key-major for dK / dV: 56
// v_mfma_f32_16x16x4_f32 per (query tile, key tile) and the exp / bias / mask arithmetic twice.  Here a tile is evaluated ONCE (40):
// a wave owns key tiles (K and V fragments in registers, dK / dV in accumulators, as wmsa_bwd_dkv_kernel), the queries are walked in
// chunks whose raw Q and dO rows sit in LDS.  S [query][key] has the query on the accumulator's ROW index, which is the index the next
// product contracts over: dV = P^T dO and dK = dS^T Q take P and dS straight from the accumulators.  dQ = dS K contracts over KEYS: the
// dS tile goes through a wave-private 16 x 20 float LDS turn-around (four ds_write_b32, one ds_read_b128 with the roles of the
// indices exchanged) and meets K rows from LDS.  dQ sums over the waves' key tiles in an fp32 LDS tile: in step s of a chunk wave w
// works on query tile (s + w) mod chunk -- eight waves on eight different tiles --, adds its key tiles' shares in registers and writes
// them with plain read-modify-writes; one barrier per step (the structure of wmsa_bf16.hip's one-pass kernel).
// Reference: swin_transformer.py:138-169 (autograd of the attention).
// ------------------------------------------------------------------------------------------------
constexpr int KT1 = 4;     // key tiles per wave: 8 waves x 4 = 32 tiles = 512 tokens (the (8,7,7) window: 25)
constexpr int TSP1 = 20;   // pitch of the turn-around tile in floats (16-byte aligned rows, quad-bank spread)
struct Carve1 {
  float *k, *q, *g, *dq, *l, *dl, *tab, *scr;
  int *cr, *rows;
};
__host__ __device__ inline size_t bwd1_floats(int nt, int qc, int T) {
  const size_t qrows = 16 * (size_t)qc;
  return (size_t)16 * nt * KPAD + 3 * qrows * KPAD + 2 * qrows + ((T + 3) & ~3) + (size_t)WMSA_WAVES * 16 * TSP1 + 2 * 16 * (size_t)nt;
}
// balanced chunks of >= 8 query tiles that fit LDS; returns the largest chunk (0: none)
inline int bwd1_chunks(int nt, int T, int* nchunks) {
  for (int nc = 1; nc <= nt; ++nc) {
    if (nt / nc < WMSA_WAVES) break;
    const int qc = (nt + nc - 1) / nc;
    if (sizeof(float) * bwd1_floats(nt, qc, T) > 160 * 1024) continue;
    *nchunks = nc;
    return qc;
  }
  return 0;
}
__device__ __forceinline__ Carve1 carve1(float* smem, int nt, int qc, int T) {
  Carve1 c;
  const int qrows = 16 * qc;
  c.k = smem;
  c.q = c.k + 16 * nt * KPAD;
  c.g = c.q + qrows * KPAD;
  c.dq = c.g + qrows * KPAD;
  c.l = c.dq + qrows * KPAD;
  c.dl = c.l + qrows;
  c.tab = c.dl + qrows;
  c.scr = c.tab + ((T + 3) & ~3);
  c.cr = reinterpret_cast<int*>(c.scr + WMSA_WAVES * 16 * TSP1);
  c.rows = c.cr + 16 * nt;
  return c;
}

// the q rows of dqkv [tokens][3][nH][32] to zero (ksplit = 2: both workgroups of a pair add their dQ share)
__global__ __launch_bounds__(256) void wmsa_zero_q_kernel(float* __restrict__ dqkv, int64_t tokens, int C) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;  // over tokens x C floats
  if (e >= tokens * C) return;
  const int64_t tok = e / C;
  *reinterpret_cast<float4*>(dqkv + tok * 3 * C + (e - tok * C)) = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool REG>
__global__ __launch_bounds__(WMSA_THREADS) void wmsa_bwd_fused_kernel(
    const float* __restrict__ qkv, const float* __restrict__ table, const int* __restrict__ code_g, const int* __restrict__ region_g,
    int T, int off, int nW, int N, int nH, float scale, RowMap rm, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, int qc, int nchunks, int ksplit) {
  // ksplit (gridDim.x) = 2: the pair's key tiles are dealt to two workgroups (tile kt belongs to workgroup kt mod 2) where a workgroup
  // per pair would leave half of the CUs idle (stage 2 of Swin-B: 128 (window, head) pairs, 18 of the 24 blocks).  Each workgroup owns
  // dK / dV of its tiles; both ADD their dQ share into the q rows of dqkv, which the launch wrapper zeroed (two addends onto zero:
  // the sum does not depend on who arrives first).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nt = (N + 15) / 16;
  const int kx = blockIdx.x;
  const Carve1 cv = carve1(smem, nt, qc, T);
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  fill_rows(cv.rows, rm, b, N, nt);
  stage_rows(cv.k, qkv, h, 1, N, nH, nt, cv.rows);  // K rows of the whole window: the B operand of dQ
  {  // table column of the head, packed code | region of the window's tokens
    Carve tmp;
    tmp.tab = cv.tab; tmp.code = cv.cr;
    (void)setup_terms<true>(tmp, table, nullptr, code_g, region_g, T, off, nW, N, nH, h, b, nt);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  float* const scr = cv.scr + wave * 16 * TSP1;
  // the wave's key tiles: K / V fragments (B operands of S and dP), the key's packed code, dK / dV accumulators
  float kf[KT1][8], vf[KT1][8];
  int pkey[KT1];
  bool kval[KT1];
  f32x4 dk0[KT1], dk1[KT1], dv0[KT1], dv1[KT1];
  __syncthreads();  // rows / codes in place
#pragma unroll
  for (int j = 0; j < KT1; ++j) {
    dk0[j] = dk1[j] = dv0[j] = dv1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt = min(kx + ksplit * (wave + WMSA_WAVES * j), nt - 1), key = min(16 * kt + i, N - 1);
    kval[j] = kx + ksplit * (wave + WMSA_WAVES * j) < nt && 16 * kt + i < N;
    pkey[j] = cv.cr[key];
    load8(kf[j], qkv + (int64_t)cv.rows[key] * rs + (int64_t)(nH + h) * HD + 8 * kk);
    load8(vf[j], qkv + (int64_t)cv.rows[key] * rs + (int64_t)(2 * nH + h) * HD + 8 * kk);
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    const int qt0 = ch * nt / nchunks, qtn = (ch + 1) * nt / nchunks - qt0, qrows = 16 * qtn;
    __syncthreads();  // (the previous chunk's readers are done)
    // ---- stage the chunk: raw Q rows, dO rows, lse (+inf for padded queries: p = 0), delta = sum_d dO O (also stored), zero dQ ----
    for (int it = threadIdx.x; it < qrows * 8; it += WMSA_THREADS) {
      const int row = it >> 3, c4 = it & 7, q = 16 * qt0 + row;
      const bool in = q < N;
      const int tok = cv.rows[in ? q : N - 1];
      float4 q4 = *reinterpret_cast<const float4*>(qkv + (int64_t)tok * rs + h * HD + 4 * c4);
      float4 g4 = *reinterpret_cast<const float4*>(dout + (int64_t)tok * C + h * HD + 4 * c4);
      const float4 o4 = *reinterpret_cast<const float4*>(out + (int64_t)tok * C + h * HD + 4 * c4);
      if (!in) q4 = g4 = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(cv.q + row * KPAD + 4 * c4) = q4;
      *reinterpret_cast<float4*>(cv.g + row * KPAD + 4 * c4) = g4;
      *reinterpret_cast<float4*>(cv.dq + row * KPAD + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
      float dsum = g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
      dsum += __shfl_xor(dsum, 1, 64);
      dsum += __shfl_xor(dsum, 2, 64);
      dsum += __shfl_xor(dsum, 4, 64);
      if (c4 == 0) {
        cv.dl[row] = in ? dsum : 0.f;
        cv.l[row] = in ? lse[(b * nH + h) * N + q] : INFINITY;
        if (in) delta[(b * nH + h) * N + q] = dsum;
      }
    }
    __syncthreads();
    // ---- step st: wave w on query tile (st + w) mod qtn (qtn >= 8: eight distinct tiles) against all of its key tiles ----
    for (int st = 0; st < qtn; ++st) {
      int qt = st + wave;
      qt = qt >= qtn ? qt - qtn : qt;
      float qf[8], gf[8];
      load8(qf, cv.q + (16 * qt + i) * KPAD + 8 * kk);
      load8(gf, cv.g + (16 * qt + i) * KPAD + 8 * kk);
      const int ql = 16 * qt + 4 * kk;                       // the lane's four queries (accumulator rows), chunk-local
      const int4 cq4 = *reinterpret_cast<const int4*>(cv.cr + 16 * qt0 + ql);
      const float4 l4 = *reinterpret_cast<const float4*>(cv.l + ql), d4 = *reinterpret_cast<const float4*>(cv.dl + ql);
      const int cq[4] = {cq4.x, cq4.y, cq4.z, cq4.w};
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
      f32x4 dqa = {0.f, 0.f, 0.f, 0.f}, dqb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KT1; ++j) {
        const int kt = kx + ksplit * (wave + WMSA_WAVES * j);
        if (kt < nt) {
          // S [query][key] and dP [query][key]: A = Q / dO rows (LDS), B = K^T / V^T (registers); row = query 4 kk + r, col = key i
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s = mfma(qf[u], kf[j][u], s);
            dp = mfma(gf[u], vf[j][u], dp);
          }
          const int ck = (REG ? pk_code(pkey[j]) : pkey[j]) - off, rk = pk_region(pkey[j]);
          float tv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) tv[r] = cv.tab[(REG ? pk_code(cq[r]) : cq[r]) - ck];
          f32x4 p, ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float term = (REG && pk_region(cq[r]) != rk) ? tv[r] - 100.f : tv[r];
            // a padded query has lse = +inf (p = 0); a padded KEY lane must not reach dQ, which sums over keys
            p[r] = kval[j] ? __expf(fmaf(s[r], scale, term) - lq[r]) : 0.f;
            ds[r] = p[r] * (dp[r] - dq_[r]);
          }
          // dV[key][d] += P^T dO ; dK[key][d] += dS^T Q   (A = accumulator-layout values, B rows = queries 16 qt + 4 kk + r)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* grow = cv.g + (ql + r) * KPAD;
            const float* qrow = cv.q + (ql + r) * KPAD;
            dv0[j] = mfma(p[r], grow[i], dv0[j]);
            dv1[j] = mfma(p[r], grow[16 + i], dv1[j]);
            dk0[j] = mfma(ds[r], qrow[i], dk0[j]);
            dk1[j] = mfma(ds[r], qrow[16 + i], dk1[j]);
          }
          // dQ share of this key tile: dS with the QUERY on the operand's row = the tile through the wave's turn-around
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[(4 * kk + r) * TSP1 + i] = ds[r];         // M[query 4 kk + r][key i]
          const float4 x4 = *reinterpret_cast<const float4*>(scr + i * TSP1 + 4 * kk);  // M[query i][keys 4 kk ..]
          const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* krow = cv.k + (16 * kt + 4 * kk + r) * KPAD;
            dqa = mfma(xs[r], krow[i], dqa);
            dqb = mfma(xs[r], krow[16 + i], dqb);
          }
        }
      }
      // this wave is the only one on query tile qt in this step: plain read-modify-write of its rows of the dQ tile
      // (accumulator rows of dqa / dqb = queries 4 kk + r of the tile, column = d = i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cv.dq[(ql + r) * KPAD + i] += dqa[r];
        cv.dq[(ql + r) * KPAD + 16 + i] += dqb[r];
      }
      __syncthreads();
    }
    // ---- the chunk's dQ ----
    for (int it = threadIdx.x; it < qrows * 8; it += WMSA_THREADS) {
      const int row = it >> 3, c4 = it & 7, q = 16 * qt0 + row;
      if (q >= N) continue;
      const float4 v = *reinterpret_cast<const float4*>(cv.dq + row * KPAD + 4 * c4);
      float* o = dqkv + (int64_t)cv.rows[q] * rs + (int64_t)h * HD + 4 * c4;
      if (ksplit == 1) {
        *reinterpret_cast<float4*>(o) = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
      } else {
        atomicAdd(o, v.x * scale);
        atomicAdd(o + 1, v.y * scale);
        atomicAdd(o + 2, v.z * scale);
        atomicAdd(o + 3, v.w * scale);
      }
    }
  }
  // ---- dK, dV of the wave's key tiles ----
#pragma unroll
  for (int j = 0; j < KT1; ++j) {
    const int kt = kx + ksplit * (wave + WMSA_WAVES * j);
    if (kt >= nt) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = 16 * kt + 4 * kk + r;
      if (krow < N) {
        float* ok = dqkv + (int64_t)cv.rows[krow] * rs + (int64_t)(nH + h) * HD;
        float* ov = dqkv + (int64_t)cv.rows[krow] * rs + (int64_t)(2 * nH + h) * HD;
        ok[i] = dk0[j][r] * scale;
        ok[16 + i] = dk1[j][r] * scale;
        ov[i] = dv0[j][r];
        ov[16 + i] = dv1[j][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Windows of 401 .. 800 tokens (the (16,7,7) window of the 32-frame SSv2 recipe: N = 784).  K and V of one (window, head)
// no longer fit LDS together (226 KB), so keys (forward, dQ) / queries (dK dV) are walked in CHUNKS of 25 tiles that are
// re-staged into the same two LDS buffers; a wave keeps the running state of at most TPW = 2 of its tiles in registers
// across the chunks (forward: online softmax m, l, O; backward: plain accumulation, P = exp(S - lse) needs no rescale).
// Relative-position form only (table <= 8192 entries: 5239 for (16,7,7)).
// ------------------------------------------------------------------------------------------------
constexpr int CT = NT_MAX;        // tiles per chunk
constexpr int NTB_MAX = 50;       // N <= 800
constexpr int TPW = 2;            // tiles of state per wave
constexpr int T_MAX_BIG = 8192;

struct CarveC {
  float* buf0; float* buf1; float* extra; float* tab; int* cr; int* rows;
};
__device__ __forceinline__ CarveC carve_c(float* smem, int ntN, int extra_floats, int T) {
  CarveC c;
  c.buf0 = smem;
  c.buf1 = c.buf0 + 16 * CT * KPAD;
  c.extra = c.buf1 + 16 * CT * KPAD;
  c.tab = c.extra + extra_floats;
  c.cr = reinterpret_cast<int*>(c.tab + ((T + 3) & ~3));
  c.rows = c.cr + 16 * ntN;
  return c;
}

// rows / packed codes of ALL tokens of window b, the table column of head h; ends with a barrier
__device__ __forceinline__ AddTerms setup_chunked(const CarveC& c, const float* table, const int* code_g, const int* region_g,
                                                  int T, int off, int nW, int N, int nH, int h, int64_t b, int ntN,
                                                  const RowMap& rm) {
  for (int i0 = threadIdx.x; i0 < T; i0 += 8 * WMSA_THREADS) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = table[(int64_t)min(i0 + u * WMSA_THREADS, T - 1) * nH + h];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * WMSA_THREADS < T) c.tab[i0 + u * WMSA_THREADS] = v[u];
  }
  for (int i = threadIdx.x; i < 16 * ntN; i += WMSA_THREADS) {
    const int n = i < N ? i : N - 1;
    const int reg = region_g ? region_g[(b % nW) * (int64_t)N + n] : 0;
    c.cr[i] = code_g[n] | (reg << 16);
    c.rows[i] = rm.map ? (int)((b / rm.nWm) * rm.L + rm.map[(b % rm.nWm) * (int64_t)N + n]) : (int)(b * N + n);
  }
  __syncthreads();
  AddTerms a;
  a.bias_h = nullptr; a.mask_b = nullptr; a.tab = c.tab; a.cr = c.cr; a.off = off;
  return a;
}

// S^T tile t of a chunk whose first key is `kbase`: keys 16t + 4kk + r of the chunk, query with packed code pq
__device__ __forceinline__ f32x4 score_tile_chunk(const float* __restrict__ k_lds, const float (&qf)[8], int t, int lane,
                                                  const AddTerms& a, int pq, int kbase, int N) {
  const int j = lane & 15, kk = lane >> 4;
  float kf[8];
  load8(kf, k_lds + (16 * t + j) * KPAD + 8 * kk);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; s += 2) {
    acc = mfma(kf[s], qf[s], acc);
    acc2 = mfma(kf[s + 1], qf[s + 1], acc2);
  }
  acc += acc2;
  const int key0 = kbase + 16 * t + 4 * kk;
  const int4 ck = *reinterpret_cast<const int4*>(a.cr + key0);
  const int cq = pk_code(pq) + a.off, rq = pk_region(pq);
  const int kc[4] = {ck.x, ck.y, ck.z, ck.w};
  float tv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tv[r] = a.tab[cq - pk_code(kc[r])];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = pk_region(kc[r]) != rq ? tv[r] - 100.f : tv[r];
    acc[r] = key0 + r < N ? acc[r] + v : -INFINITY;
  }
  return acc;
}

__global__ __launch_bounds__(WMSA_THREADS) void wmsa_fwd_chunked_kernel(
    const float* __restrict__ qkv, const float* __restrict__ table, const int* __restrict__ code_g,
    const int* __restrict__ region_g, int T, int off, int nW, int N, int nH, float scale, int qsplit, RowMap rm,
    float* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntN = (N + 15) / 16;
  const CarveC cv = carve_c(smem, ntN, 0, T);
  float* k_lds = cv.buf0;
  float* v_lds = cv.buf1;
  const int* rows = cv.rows;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const AddTerms terms = setup_chunked(cv, table, code_g, region_g, T, off, nW, N, nH, h, b, ntN, rm);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  const float* q_base = qkv + (int64_t)h * HD;
  const int per = (ntN + qsplit - 1) / qsplit;
  const int rt0 = blockIdx.x * per, rt1 = min(ntN, rt0 + per);

  float qf[TPW][8], m[TPW], l[TPW];
  int pq[TPW];
  bool has[TPW];
  f32x4 o0[TPW], o1[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int rt = rt0 + wave + WMSA_WAVES * j;
    has[j] = rt < rt1;
    const int q = min(16 * rt + i, N - 1);
    load8(qf[j], q_base + (int64_t)rows[q] * rs + 8 * kk);
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[j][s] *= scale;
    pq[j] = cv.cr[q];
    m[j] = -INFINITY;
    l[j] = 0.f;
    o0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    o1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int kbase = 0; kbase < N; kbase += 16 * CT) {
    const int nk = min(N - kbase, 16 * CT), ntk = (nk + 15) / 16;
    __syncthreads();  // every wave is done with the previous chunk
    stage_rows(k_lds, qkv, h, 1, nk, nH, ntk, rows + kbase);
    stage_rows(v_lds, qkv, h, 2, nk, nH, ntk, rows + kbase);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (!has[j]) continue;  // wave-uniform
      f32x4 acc[CT];
      float mc = -INFINITY;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (t < ntk) {
          acc[t] = score_tile_chunk(k_lds, qf[j], t, lane, terms, pq[j], kbase, N);
          mc = fmaxf(mc, fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3])));
        }
      }
      mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
      const float m_new = fmaxf(m[j], mc);
      const float alpha = __expf(m[j] - m_new);  // 0 on the first chunk (m = -inf)
      float ls = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (t < ntk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(acc[t][r] - m_new);
            acc[t][r] = p;
            ls += p;
          }
        }
      }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      l[j] = l[j] * alpha + ls;
      m[j] = m_new;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // O rows are queries 4kk + r: their rescale factor lives in lane 4kk + r
        const float ar = __shfl(alpha, 4 * kk + r, 64);
        o0[j][r] *= ar;
        o1[j][r] *= ar;
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (t < ntk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vrow = v_lds + (16 * t + 4 * kk + r) * KPAD;
            o0[j] = mfma(acc[t][r], vrow[i], o0[j]);
            o1[j] = mfma(acc[t][r], vrow[16 + i], o1[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!has[j]) continue;
    const int rt = rt0 + wave + WMSA_WAVES * j;
    const float inv_l = 1.f / l[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * kk + r;
      const float il = __shfl(inv_l, 4 * kk + r, 64);
      if (qrow < N) {
        float* o = out + (int64_t)rows[qrow] * C + h * HD;
        o[i] = o0[j][r] * il;
        o[16 + i] = o1[j][r] * il;
      }
    }
    if (kk == 0 && 16 * rt + i < N) lse[(b * nH + h) * N + 16 * rt + i] = m[j] + __logf(l[j]);
  }
}

__global__ __launch_bounds__(WMSA_THREADS) void wmsa_bwd_dq_chunked_kernel(
    const float* __restrict__ qkv, const float* __restrict__ table, const int* __restrict__ code_g,
    const int* __restrict__ region_g, int T, int off, int nW, int N, int nH, float scale, int qsplit, RowMap rm,
    const float* __restrict__ out, const float* __restrict__ dout, const float* __restrict__ lse,
    float* __restrict__ delta, float* __restrict__ dqkv, float* __restrict__ dtable) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntN = (N + 15) / 16;
  const CarveC cv = carve_c(smem, ntN, 0, T);
  float* k_lds = cv.buf0;
  float* v_lds = cv.buf1;
  const int* rows = cv.rows;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const AddTerms terms = setup_chunked(cv, table, code_g, region_g, T, off, nW, N, nH, h, b, ntN, rm);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  const float* q_base = qkv + (int64_t)h * HD;
  const int per = (ntN + qsplit - 1) / qsplit;
  const int rt0 = blockIdx.x * per, rt1 = min(ntN, rt0 + per);

  float qf[TPW][8], gf[TPW][8], L[TPW], dl[TPW];
  int pq[TPW], qi[TPW];
  bool has[TPW], qvalid[TPW];
  f32x4 dq0[TPW], dq1[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int rt = rt0 + wave + WMSA_WAVES * j;
    has[j] = rt < rt1;
    qvalid[j] = has[j] && 16 * rt + i < N;
    const int q = min(16 * rt + i, N - 1);
    qi[j] = q;
    float of[8];
    load8(qf[j], q_base + (int64_t)rows[q] * rs + 8 * kk);
    load8(gf[j], dout + (int64_t)rows[q] * C + h * HD + 8 * kk);
    load8(of, out + (int64_t)rows[q] * C + h * HD + 8 * kk);
    float d = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qf[j][s] *= scale;
      d = fmaf(gf[j][s], of[s], d);
    }
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    dl[j] = d;
    L[j] = lse[(b * nH + h) * N + q];
    if (kk == 0 && qvalid[j]) delta[(b * nH + h) * N + q] = d;
    pq[j] = cv.cr[q];
    dq0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    dq1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int kbase = 0; kbase < N; kbase += 16 * CT) {
    const int nk = min(N - kbase, 16 * CT), ntk = (nk + 15) / 16;
    __syncthreads();
    stage_rows(k_lds, qkv, h, 1, nk, nH, ntk, rows + kbase);
    stage_rows(v_lds, qkv, h, 2, nk, nH, ntk, rows + kbase);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (!has[j]) continue;
      for (int t = 0; t < ntk; ++t) {
        const f32x4 s = score_tile_chunk(k_lds, qf[j], t, lane, terms, pq[j], kbase, N);
        float vf[8];
        load8(vf, v_lds + (16 * t + i) * KPAD + 8 * kk);
        f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) dp = mfma(vf[u], gf[j][u], dp);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = __expf(s[r] - L[j]) * (dp[r] - dl[j]);  // exp(-inf) = 0 for padded keys
        if (dtable && qvalid[j]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kbase + 16 * t + 4 * kk + r;
            if (key < N)
              atomicAdd(dtable + (int64_t)(pk_code(pq[j]) - pk_code(terms.cr[key]) + terms.off) * nH + h, ds[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* krow = k_lds + (16 * t + 4 * kk + r) * KPAD;
          dq0[j] = mfma(ds[r], krow[i], dq0[j]);
          dq1[j] = mfma(ds[r], krow[16 + i], dq1[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!has[j]) continue;
    const int rt = rt0 + wave + WMSA_WAVES * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * kk + r;
      if (qrow < N) {
        float* o = dqkv + (int64_t)rows[qrow] * rs + (int64_t)h * HD;
        o[i] = dq0[j][r] * scale;
        o[16 + i] = dq1[j][r] * scale;
      }
    }
  }
}

__global__ __launch_bounds__(WMSA_THREADS) void wmsa_bwd_dkv_chunked_kernel(
    const float* __restrict__ qkv, const float* __restrict__ table, const int* __restrict__ code_g,
    const int* __restrict__ region_g, int T, int off, int nW, int N, int nH, float scale, int qsplit, RowMap rm,
    const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntN = (N + 15) / 16;
  const CarveC cv = carve_c(smem, ntN, 2 * 16 * CT, T);
  float* q_lds = cv.buf0;
  float* g_lds = cv.buf1;
  float* l_lds = cv.extra;
  float* d_lds = l_lds + 16 * CT;
  const int* rows = cv.rows;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int C = nH * HD;
  const AddTerms terms = setup_chunked(cv, table, code_g, region_g, T, off, nW, N, nH, h, b, ntN, rm);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const float* k_base = qkv + (int64_t)(nH + h) * HD;
  const float* v_base = qkv + (int64_t)(2 * nH + h) * HD;
  const int per = (ntN + qsplit - 1) / qsplit;
  const int kt0 = blockIdx.x * per, kt1 = min(ntN, kt0 + per);

  float kf[TPW][8], vf[TPW][8];
  int pkey[TPW];
  bool has[TPW], kvalid[TPW];
  f32x4 dk0[TPW], dk1[TPW], dv0[TPW], dv1[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int kt = kt0 + wave + WMSA_WAVES * j;
    has[j] = kt < kt1;
    const int key = min(16 * kt + i, N - 1);
    kvalid[j] = has[j] && 16 * kt + i < N;
    pkey[j] = cv.cr[key];
    load8(kf[j], k_base + (int64_t)rows[key] * rs + 8 * kk);
    load8(vf[j], v_base + (int64_t)rows[key] * rs + 8 * kk);
    dk0[j] = dk1[j] = dv0[j] = dv1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int qbase = 0; qbase < N; qbase += 16 * CT) {
    const int nq = min(N - qbase, 16 * CT), ntq = (nq + 15) / 16;
    __syncthreads();
    stage_rows(q_lds, qkv, h, 0, nq, nH, ntq, rows + qbase);
    stage_rows_dense(g_lds, dout + h * HD, C, nq, ntq, rows + qbase);
    for (int r = threadIdx.x; r < 16 * ntq; r += WMSA_THREADS) {
      l_lds[r] = r < nq ? lse[(b * nH + h) * N + qbase + r] : INFINITY;  // exp(s - inf) = 0 for padded queries
      d_lds[r] = r < nq ? delta[(b * nH + h) * N + qbase + r] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (!has[j]) continue;
      for (int qt = 0; qt < ntq; ++qt) {
        float qf[8], gf[8];
        load8(qf, q_lds + (16 * qt + i) * KPAD + 8 * kk);
        load8(gf, g_lds + (16 * qt + i) * KPAD + 8 * kk);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s = mfma(qf[u], kf[j][u], s);
          dp = mfma(gf[u], vf[j][u], dp);
        }
        f32x4 p, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 16 * qt + 4 * kk + r, q = qbase + ql;
          const float term = rel_term(terms, terms.cr[q < N ? q : N - 1], pkey[j]);
          const float sv = (q < N && kvalid[j]) ? fmaf(s[r], scale, term) : -INFINITY;
          p[r] = __expf(sv - l_lds[ql]);
          ds[r] = p[r] * (dp[r] - d_lds[ql]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* grow = g_lds + (16 * qt + 4 * kk + r) * KPAD;
          const float* qrow = q_lds + (16 * qt + 4 * kk + r) * KPAD;
          dv0[j] = mfma(p[r], grow[i], dv0[j]);
          dv1[j] = mfma(p[r], grow[16 + i], dv1[j]);
          dk0[j] = mfma(ds[r], qrow[i], dk0[j]);
          dk1[j] = mfma(ds[r], qrow[16 + i], dk1[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!has[j]) continue;
    const int kt = kt0 + wave + WMSA_WAVES * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = 16 * kt + 4 * kk + r;
      if (krow < N) {
        float* ok = dqkv + (int64_t)rows[krow] * rs + (int64_t)(nH + h) * HD;
        float* ov = dqkv + (int64_t)rows[krow] * rs + (int64_t)(2 * nH + h) * HD;
        ok[i] = dk0[j][r] * scale;
        ok[16 + i] = dk1[j][r] * scale;
        ov[i] = dv0[j][r];
        ov[16 + i] = dv1[j][r];
      }
    }
  }
}

inline size_t lds_bytes_chunked(int N, int extra_floats, int T) {
  const int ntN = (N + 15) / 16;
  return sizeof(float) * ((size_t)2 * 16 * CT * KPAD + extra_floats + ((T + 3) & ~3) + 2 * 16 * ntN);
}

inline int pick_qsplit(int64_t pairs, int nt) {
  // One workgroup per CU at a time (the staged operands take ~115 KB of LDS) and every workgroup re-stages
  // K and V: split the query tiles of a (window, head) pair only until every CU has one workgroup (256),
  // and never below one row tile per wave.
  int qs = 1;
  while (pairs * qs < 256 && (nt + qs * 2 - 1) / (qs * 2) >= WMSA_WAVES / 2 && qs < 8) qs *= 2;
  return qs;
}

constexpr int T_MAX = 4096;  // (2wd-1)(2wh-1)(2ww-1): 2535 for the (8,7,7) window

inline size_t lds_bytes(int N, int extra_floats, int T) {
  const int nt = (N + 15) / 16;
  return sizeof(float) * ((size_t)2 * 16 * nt * KPAD + extra_floats + ((T + 3) & ~3) + 3 * 16 * nt);
}

struct WmsaArgs {
  const float* qkv; const float* bias; const float* mask; const int* code; const int* region;
  int T, off, nW; int64_t B_; int N, nH; float scale;
  RowMap rm;
};

template <bool REL>
int launch_fwd(const WmsaArgs& a, float* out, float* lse, hipStream_t st) {
  const int nt = (a.N + 15) / 16;
  const int qs = pick_qsplit(a.B_ * a.nH, nt);
  const size_t lds = lds_bytes(a.N, 0, REL ? a.T : 0);
#define WMSA_FWD(NTC, REG)                                                                                               \
  do {                                                                                                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wmsa_fwd_kernel<REL, NTC, REG>),                             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                       \
      return VITTA_ERR_LAUNCH;                                                                                         \
    VITTA_LAUNCH((wmsa_fwd_kernel<REL, NTC, REG>), dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds, st, a.qkv,   \
                 a.bias, a.mask, a.code, a.region, REL ? a.T : 0, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, out, lse); \
  } while (0)
  // the compile-time variant (NTC = 25) lets the scheduler hoist every tile's loads: 256 VGPRs + 1.4 KB of scratch per
  // lane and 15 % slower end to end (measured); the per-tile guard of the run-time variant keeps live ranges short
  const bool reg = REL && a.region;  // region ids present (a shifted block): the kernels without them carry no mask arithmetic
  if (nt == NT_MAX) {
    if (reg) WMSA_FWD(NT_MAX, true);
    else WMSA_FWD(NT_MAX, false);
  } else {
    if (reg) WMSA_FWD(0, true);
    else WMSA_FWD(0, false);
  }
#undef WMSA_FWD
  return VITTA_OK;
}

template <bool REL>
int launch_bwd(const WmsaArgs& a, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
               float* dbias, hipStream_t st) {
  const int nt = (a.N + 15) / 16;
  const int qs = pick_qsplit(a.B_ * a.nH, nt);
  const int T = REL ? a.T : 0;
  if constexpr (REL) {
    // one pass (frozen table, a workgroup per (window, head) fills the chip): VITTA_WMSA_F32_BWD=two keeps the two-kernel form (A/B)
    const char* form = std::getenv("VITTA_WMSA_F32_BWD");
    int nchunks = 0;
    // a workgroup per pair from 256 pairs on; two workgroups per pair (key tiles dealt out, dQ added) from 128; below: two kernels
    const int64_t pairs = a.B_ * a.nH;
    const bool forced = form && form[0] == 'o';
    const int ksplit = pairs >= 256 ? 1 : 2;
    const int qc = (!dbias && nt <= WMSA_WAVES * KT1 && !(form && form[0] == 't') && (pairs >= 128 || forced))
                       ? bwd1_chunks(nt, T, &nchunks) : 0;
    if (qc > 0) {
      const size_t lf = sizeof(float) * bwd1_floats(nt, qc, T);
      if (ksplit > 1) {
        const int64_t tokens = a.rm.map ? (a.B_ / a.rm.nWm) * a.rm.L : a.B_ * (int64_t)a.N;
        const int Cc = a.nH * HD;
        VITTA_LAUNCH(wmsa_zero_q_kernel, dim3((unsigned)((tokens * Cc / 4 + 255) / 256)), dim3(256), 0, st, dqkv, tokens, Cc);
      }
#define WMSA_BWD1(REG)                                                                                                  \
  do {                                                                                                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wmsa_bwd_fused_kernel<REG>),                                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf) != hipSuccess)                        \
      return VITTA_ERR_LAUNCH;                                                                                         \
    VITTA_LAUNCH((wmsa_bwd_fused_kernel<REG>), dim3(ksplit, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lf, st, a.qkv, a.bias, a.code, \
                 a.region, T, a.off, a.nW, a.N, a.nH, a.scale, a.rm, out, dout, lse, delta, dqkv, qc, nchunks, ksplit);  \
  } while (0)
      if (a.region) WMSA_BWD1(true);
      else WMSA_BWD1(false);
#undef WMSA_BWD1
      return VITTA_OK;
    }
  }
  const int dfix = (REL && dbias && lds_bytes(a.N, 2 * ((T + 3) & ~3), T) <= 160 * 1024) ? 1 : 0;  // the 64-bit table column fits
  const size_t lds1 = lds_bytes(a.N, (REL && dbias) ? (dfix ? 2 : 1) * ((T + 3) & ~3) : 0, T), lds2 = lds_bytes(a.N, 2 * 16 * nt, T);
#define WMSA_BWD(REG)                                                                                                   \
  do {                                                                                                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wmsa_bwd_dq_kernel<REL, REG>),                               \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess ||                    \
        hipFuncSetAttribute(reinterpret_cast<const void*>(wmsa_bwd_dkv_kernel<REL, REG>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)                      \
      return VITTA_ERR_LAUNCH;                                                                                         \
    VITTA_LAUNCH((wmsa_bwd_dq_kernel<REL, REG>), dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds1, st, a.qkv, a.bias, \
                 a.mask, a.code, a.region, T, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, out, dout, lse, delta, dqkv, dbias, dfix); \
    VITTA_LAUNCH((wmsa_bwd_dkv_kernel<REL, REG>), dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds2, st, a.qkv, a.bias, \
                 a.mask, a.code, a.region, T, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, dout, lse, delta, dqkv);       \
  } while (0)
  if (REL && a.region) WMSA_BWD(true);
  else WMSA_BWD(false);
#undef WMSA_BWD
  return VITTA_OK;
}

inline int chunked_qsplit(int N) {
  const int ntN = (N + 15) / 16;
  return (ntN + WMSA_WAVES * TPW - 1) / (WMSA_WAVES * TPW);
}

template <typename K>
inline bool set_dyn_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
         hipSuccess;
}

int launch_fwd_chunked(const WmsaArgs& a, float* out, float* lse, hipStream_t st) {
  const int qs = chunked_qsplit(a.N);
  const size_t lds = lds_bytes_chunked(a.N, 0, a.T);
  if (lds > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  if (!set_dyn_lds(wmsa_fwd_chunked_kernel, lds)) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(wmsa_fwd_chunked_kernel, dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds, st, a.qkv, a.bias, a.code,
               a.region, a.T, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, out, lse);
  return VITTA_OK;
}

int launch_bwd_chunked(const WmsaArgs& a, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                       float* dtable, hipStream_t st) {
  const int qs = chunked_qsplit(a.N);
  const size_t lds1 = lds_bytes_chunked(a.N, 0, a.T), lds2 = lds_bytes_chunked(a.N, 2 * 16 * CT, a.T);
  if (lds2 > 160 * 1024) return VITTA_ERR_UNSUPPORTED;
  if (!set_dyn_lds(wmsa_bwd_dq_chunked_kernel, lds1) || !set_dyn_lds(wmsa_bwd_dkv_chunked_kernel, lds2)) return VITTA_ERR_LAUNCH;
  VITTA_LAUNCH(wmsa_bwd_dq_chunked_kernel, dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds1, st, a.qkv, a.bias, a.code,
               a.region, a.T, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, out, dout, lse, delta, dqkv, dtable);
  VITTA_LAUNCH(wmsa_bwd_dkv_chunked_kernel, dim3(qs, a.nH, (unsigned)a.B_), dim3(WMSA_THREADS), lds2, st, a.qkv, a.bias, a.code,
               a.region, a.T, a.off, a.nW, a.N, a.nH, a.scale, qs, a.rm, dout, lse, delta, dqkv);
  return VITTA_OK;
}

inline bool misaligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15u) != 0;
}

}  // namespace

extern "C" {

int vitta_wmsa_supported(int32_t N, int32_t head_dim) { return (head_dim == HD && N >= 1 && N <= 16 * NT_MAX) ? 1 : 0; }

/* relative-position form: additionally windows of up to 800 tokens (key / query chunks re-staged through LDS) */
int vitta_wmsa_rel_supported(int32_t N, int32_t head_dim) {
  return (head_dim == HD && N >= 1 && N <= 16 * NTB_MAX) ? 1 : 0;
}

int vitta_wmsa_fwd_f32(const float* d_qkv, const float* d_bias, const float* d_mask, int32_t nW, int64_t B_, int32_t N,
                       int32_t nH, int32_t head_dim, float scale, float* d_out, float* d_lse, void* stream) {
  if (!d_qkv || !d_bias || !d_out || !d_lse || B_ <= 0 || nH <= 0) return VITTA_ERR_INVALID_ARG;
  if (!vitta_wmsa_supported(N, head_dim)) return VITTA_ERR_UNSUPPORTED;
  if (d_mask && (nW <= 0 || B_ % nW)) return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_qkv, d_out, d_bias, d_mask)) return VITTA_ERR_INVALID_ARG;
  const WmsaArgs a{d_qkv, d_bias, d_mask, nullptr, nullptr, 0, 0, d_mask ? nW : 1, B_, N, nH, scale, RowMap{nullptr, 1, 0}};
  return launch_fwd<false>(a, d_out, d_lse, static_cast<hipStream_t>(stream));
}

int vitta_wmsa_bwd_f32(const float* d_qkv, const float* d_bias, const float* d_mask, int32_t nW, int64_t B_, int32_t N,
                       int32_t nH, int32_t head_dim, float scale, const float* d_out, const float* d_dout,
                       const float* d_lse, float* d_delta, float* d_dqkv, float* d_dbias, void* stream) {
  if (!d_qkv || !d_bias || !d_out || !d_dout || !d_lse || !d_delta || !d_dqkv || B_ <= 0 || nH <= 0)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_wmsa_supported(N, head_dim)) return VITTA_ERR_UNSUPPORTED;
  if (d_mask && (nW <= 0 || B_ % nW)) return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_qkv, d_out, d_dout, d_dqkv) || misaligned(d_bias, d_mask)) return VITTA_ERR_INVALID_ARG;
  const WmsaArgs a{d_qkv, d_bias, d_mask, nullptr, nullptr, 0, 0, d_mask ? nW : 1, B_, N, nH, scale, RowMap{nullptr, 1, 0}};
  return launch_bwd<false>(a, d_out, d_dout, d_lse, d_delta, d_dqkv, d_dbias, static_cast<hipStream_t>(stream));
}

int vitta_wmsa_rel_fwd_f32(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                           const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                           float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                           float* d_out, float* d_lse, void* stream) {
  if (!d_qkv || !d_table || !d_code || !d_out || !d_lse || B_ <= 0 || nH <= 0 || T <= 0 || T > T_MAX_BIG)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_wmsa_rel_supported(N, head_dim)) return VITTA_ERR_UNSUPPORTED;
  const bool chunked = N > 16 * NT_MAX;
  if (!chunked && T > T_MAX) return VITTA_ERR_UNSUPPORTED;
  if (d_region && (nW <= 0 || B_ % nW)) return VITTA_ERR_INVALID_ARG;
  if (d_rowmap && (map_windows <= 0 || B_ % map_windows || tokens_per_sample != (int64_t)map_windows * N))
    return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_qkv, d_out)) return VITTA_ERR_INVALID_ARG;
  const WmsaArgs a{d_qkv, d_table, nullptr, d_code, d_region, T, code_off, d_region ? nW : 1, B_, N, nH, scale,
                   RowMap{d_rowmap, d_rowmap ? map_windows : 1, tokens_per_sample}};
  if (chunked) return launch_fwd_chunked(a, d_out, d_lse, static_cast<hipStream_t>(stream));
  return launch_fwd<true>(a, d_out, d_lse, static_cast<hipStream_t>(stream));
}

int vitta_wmsa_rel_bwd_f32(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                           const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                           float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                           const float* d_out, const float* d_dout, const float* d_lse, float* d_delta,
                           float* d_dqkv, float* d_dtable, void* stream) {
  if (!d_qkv || !d_table || !d_code || !d_out || !d_dout || !d_lse || !d_delta || !d_dqkv || B_ <= 0 || nH <= 0 ||
      T <= 0 || T > T_MAX_BIG)
    return VITTA_ERR_INVALID_ARG;
  if (!vitta_wmsa_rel_supported(N, head_dim)) return VITTA_ERR_UNSUPPORTED;
  const bool chunked = N > 16 * NT_MAX;
  if (!chunked && T > T_MAX) return VITTA_ERR_UNSUPPORTED;
  if (d_region && (nW <= 0 || B_ % nW)) return VITTA_ERR_INVALID_ARG;
  if (d_rowmap && (map_windows <= 0 || B_ % map_windows || tokens_per_sample != (int64_t)map_windows * N))
    return VITTA_ERR_INVALID_ARG;
  if (misaligned(d_qkv, d_out, d_dout, d_dqkv)) return VITTA_ERR_INVALID_ARG;
  const WmsaArgs a{d_qkv, d_table, nullptr, d_code, d_region, T, code_off, d_region ? nW : 1, B_, N, nH, scale,
                   RowMap{d_rowmap, d_rowmap ? map_windows : 1, tokens_per_sample}};
  if (chunked)
    return launch_bwd_chunked(a, d_out, d_dout, d_lse, d_delta, d_dqkv, d_dtable, static_cast<hipStream_t>(stream));
  return launch_bwd<true>(a, d_out, d_dout, d_lse, d_delta, d_dqkv, d_dtable, static_cast<hipStream_t>(stream));
}

}  // extern "C"
