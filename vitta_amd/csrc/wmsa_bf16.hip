// bfloat16-operand form of the fused 3-D shifted-window attention of wmsa.hip (BASELINE config 5: Video Swin-B on SSv2-C,
// 32 frames x 4 views, window (16,7,7) = 784 tokens; reference models/videoswintransformer_models/swin_transformer.py:
// 138-169, recipe recognizer3d.py:36-40).  q, k, v, dO are converted to bf16 while they are staged (round to nearest even),
// the two GEMMs of every direction run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation, the softmax, the relative-
// position bias / shift mask (generated on chip from the code table exactly as in wmsa.hip), lse, delta and every output
// stay fp32.  At 2 bytes per element K and V of a 784-token window fit LDS together (115 KB), so windows up to 800 tokens
// need no chunking (the fp32 kernels re-stage keys / queries in chunks of 400 there).
//
// MFMA operand layouts (16x16x16: lane (i = lane & 15, g = lane >> 4) supplies 4 consecutive k of row / column i):
//   * contraction over the head dim (S = K Q^T, dP = V dO^T): d = 8 g + 4 s + e, s = 0, 1 -> one 16-byte read of a
//     row-major [token][d] bf16 row per lane and tile;
//   * contraction over keys (O = P V, dQ = dS K) / over queries (dV = P^T dO, dK = dS^T Q): the A operand IS the C layout
//     of the score tile (rows 4 g + r of column i), packed to bf16; the B operand needs 4 consecutive tokens of one d:
//     the forward stages V TRANSPOSED ([d][token], one 8-byte read), the backward kernels gather the four 2-byte values
//     from the row-major copy they hold anyway (a transposed second copy would not fit LDS at 784 tokens).
// Relative-position form only; the bias table is an input (its gradient -- SGD over all parameters -- stays on the fp32 kernels).
#include <cstdlib>

#include "common.h"

using namespace vitta;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HD = 32;
constexpr int RP = 40;        // row pitch of a [token][32] bf16 tile in elements (80 bytes: conflict-free 16-byte reads)
constexpr int NTB = 50;       // 16-token tiles per window: N <= 800
constexpr int TH_FWD_S = 512; // forward: 8 waves, 25 score tiles of 16 queries in registers at a time (one or two key chunks)
constexpr int TH_BWD = 1024;  // backward: 16 waves (the tile loops are rolled: ~56 registers per lane)
constexpr int T_MAX = 8192;

// fp32 -> bf16, round to nearest even, on the hardware conversion (v_cvt_pk_bf16_f32: one instruction per two values; the integer
// form used until round 3 was five per value, ~20 of the ~60 vector instructions a 16 x 16 score tile costs)
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int hu32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  const hu32x2 r = {pack2(a, b), pack2(c, d)};
  return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_bf16 (gfx950): the contraction of a head's 32 channels -- or of TWO 16-key tiles -- in one instruction; lane
// (i, g) holds elements 8 g .. 8 g + 7 of the contraction on both operands, the C layout is that of the 16x16x16 form
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hbf16x8, a), __builtin_bit_cast(hbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
constexpr float LOG2E = 1.4426950408889634f;
// exp(x - m) as ONE multiply-add and the hardware's 2^x (v_exp_f32), mL = m log2(e): __expf(x - m) is a subtraction, a multiplication
// and the exponential -- a quarter of the score tile's vector arithmetic sits on this line
__device__ __forceinline__ float exp_sub(float x, float mL) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, LOG2E, -mL)); }

struct RowMap {
  const int* map;
  int nWm;
  int64_t L;
};

struct Args {
  const float* qkv; const float* table; const int* code; const int* region;
  int T, off, nW; int64_t B_; int N, nH; float scale;
  RowMap rm;
  int io16;  // qkv, out, d out, d qkv are bfloat16 in memory (the bf16 data flow: a dense product on either side), else float32
};

// element e of a tensor that is fp32 or (io16) bfloat16: address and converting accessors
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }


__device__ __forceinline__ int pk_code(int p) { return p & 0xffff; }
__device__ __forceinline__ int pk_region(int p) { return p >> 16; }

// pitch of the transposed [32][tokens] bf16 tile: (pitch / 2) % 16 == 2 spreads the 16 rows a half-wave reads over the banks
__host__ __device__ inline int tpitch(int nt) { return 16 * nt + ((nt & 1) ? 20 : 4); }

struct Carve {
  unsigned short* a0;  // [16 nt][RP]
  unsigned short* a1;  // [16 nt][RP] or [32][tpitch]
  float* extra; float* tab; int* cr; int* rows;
};
__device__ __forceinline__ Carve carve(unsigned char* smem, int nt, size_t a1_elems, int extra_floats, int T) {
  Carve c;
  c.a0 = reinterpret_cast<unsigned short*>(smem);
  c.a1 = c.a0 + 16 * nt * RP;
  c.extra = reinterpret_cast<float*>(c.a1 + ((a1_elems + 7) & ~(size_t)7));
  c.tab = c.extra + extra_floats;
  c.cr = reinterpret_cast<int*>(c.tab + ((T + 3) & ~3));
  c.rows = c.cr + 16 * nt;
  return c;
}
inline size_t lds_bytes(int nt, size_t a1_elems, int extra_floats, int T) {
  return 2 * ((size_t)16 * nt * RP + ((a1_elems + 7) & ~(size_t)7)) + 4 * ((size_t)extra_floats + ((T + 3) & ~3) + 2 * 16 * nt);
}

__device__ __forceinline__ void fill_rows(int* rows, const RowMap& rm, int64_t b, int N, int nt) {
  for (int i = threadIdx.x; i < 16 * nt; i += blockDim.x) {
    const int n = i < N ? i : N - 1;
    rows[i] = rm.map ? (int)((b / rm.nWm) * rm.L + rm.map[(b % rm.nWm) * (int64_t)N + n]) : (int)(b * N + n);
  }
  __syncthreads();
}

// table column of head h + packed code | region of the window's tokens
// SH: the codes are stored << SH (2: byte offsets into the table column -- the forward's table reads need no address arithmetic)
template <int SH = 0>
__device__ __forceinline__ void setup_terms(const Carve& c, const Args& a, int h, int64_t b, int nt) {
  const int TH = blockDim.x;
  for (int i0 = threadIdx.x; i0 < a.T; i0 += 8 * TH) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a.table[(int64_t)min(i0 + u * TH, a.T - 1) * a.nH + h];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * TH < a.T) c.tab[i0 + u * TH] = v[u];
  }
  for (int i = threadIdx.x; i < 16 * nt; i += TH) {
    const int n = i < a.N ? i : a.N - 1;
    const int reg = a.region ? a.region[(b % a.nW) * (int64_t)a.N + n] : 0;
    c.cr[i] = (a.code[n] << SH) | (reg << 16);
  }
}

// rows [0, N) of a [rows, row_stride] slice (32 elements per row; fp32, or bfloat16 with io16) -> bf16 [16 nt][RP] (x mul); rows
// >= N are zero
__device__ __forceinline__ void stage_rows(unsigned short* dst, const float* src, int64_t row_stride, int N, int nt,
                                           const int* rows, float mul, int io16) {
  const int TH = blockDim.x;
  const int total = 16 * nt * 8;
  if (io16) {
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * TH) {
      ushort4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * TH, row = i >> 3, c4 = i & 7;
        v[u] = make_ushort4(0, 0, 0, 0);
        if (i < total && row < N) v[u] = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(src) + (int64_t)rows[row] * row_stride + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * TH;
        if (i < total) {
          bf16x4 o;
          if (mul == 1.f) {
            o[0] = (short)v[u].x; o[1] = (short)v[u].y; o[2] = (short)v[u].z; o[3] = (short)v[u].w;
          } else {
            o = pack4(bf2f(v[u].x) * mul, bf2f(v[u].y) * mul, bf2f(v[u].z) * mul, bf2f(v[u].w) * mul);
          }
          *reinterpret_cast<bf16x4*>(dst + (i >> 3) * RP + 4 * (i & 7)) = o;
        }
      }
    }
    return;
  }
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * TH) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TH, row = i >> 3, c4 = i & 7;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total && row < N) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)rows[row] * row_stride + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TH;
      if (i < total)
        *reinterpret_cast<bf16x4*>(dst + (i >> 3) * RP + 4 * (i & 7)) = pack4(v[u].x * mul, v[u].y * mul, v[u].z * mul, v[u].w * mul);
    }
  }
}

// 8 contiguous elements (element offset e of an fp32 / bfloat16 tensor) -> two bf16x4 (d = 8 g .. 8 g + 3 | 8 g + 4 .. 8 g + 7),
// scaled; f[8]: the values as floats
__device__ __forceinline__ void load_frag(const float* base, int64_t e, int io16, float mul, bf16x4& lo, bf16x4& hi, float (&f)[8]) {
  if (io16) {
    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + e);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = __uint_as_float(w[j] << 16);
      f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>(base + e);
    const float4 b = *reinterpret_cast<const float4*>(base + e + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  lo = pack4(f[0] * mul, f[1] * mul, f[2] * mul, f[3] * mul);
  hi = pack4(f[4] * mul, f[5] * mul, f[6] * mul, f[7] * mul);
}
__device__ __forceinline__ const float* qkv_at(const Args& a, int64_t e) {
  return a.io16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.qkv) + e) : a.qkv + e;
}
__device__ __forceinline__ void store_el(float* base, int64_t e, int io16, float v) {
  if (io16) reinterpret_cast<unsigned short*>(base)[e] = f2bf(v);
  else base[e] = v;
}

// additive terms of the S^T tile t (rows = keys 16 t + 4 g + r, column = query q): bias + shift mask, -inf past N.
// REG: the launch has region ids (shifted windows); TAIL: N is not a multiple of 16.  Without them (every unshifted block of a
// 784-token window) the tile costs four table reads, four subtractions and four additions -- the score tiles are VECTOR-ALU bound
// (head dim 32: 128 matrix flops per element against ~8 vector instructions), so every instruction here is forward time.
template <bool REG, bool TAIL>
__device__ __forceinline__ void add_terms_t(f32x4& acc, const float* tab, const int* cr, int off, int t, int g, int pq, int N) {
  const int key0 = 16 * t + 4 * g;
  const int4 ck = *reinterpret_cast<const int4*>(cr + key0);
  const int cq = (REG ? pk_code(pq) : pq) + off, rq = pk_region(pq);
  const int kc[4] = {ck.x, ck.y, ck.z, ck.w};
  float tv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tv[r] = tab[cq - (REG ? pk_code(kc[r]) : kc[r])];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = (REG && pk_region(kc[r]) != rq) ? tv[r] - 100.f : tv[r];
    acc[r] = (!TAIL || key0 + r < N) ? acc[r] + v : -INFINITY;
  }
}

// The same on codes stored as BYTE offsets (setup_terms<2>): cqb = LDS address of the table column + 4 (code[q] + off) (+ the region of
// q in the high half), so a table read is one subtraction and the load (the index form: subtraction, shift-add, load).
typedef __attribute__((address_space(3))) const float lds_cfloat;
// The terms are returned as the score product's INITIAL accumulator (S = bias + K Q^T costs no addition); TAIL: mask_tail afterwards.
template <bool REG>
__device__ __forceinline__ f32x4 terms_b(unsigned cqb, int rq, const int* cr, int t, int g) {
  const int key0 = 16 * t + 4 * g;
  const int4 ck = *reinterpret_cast<const int4*>(cr + key0);
  const int kc[4] = {ck.x, ck.y, ck.z, ck.w};
  f32x4 tv;
#pragma unroll
  for (int r = 0; r < 4; ++r) tv[r] = *(lds_cfloat*)(cqb - (unsigned)(REG ? pk_code(kc[r]) : kc[r]));
  if (REG) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = pk_region(kc[r]) != rq ? tv[r] - 100.f : tv[r];
  }
  return tv;
}
__device__ __forceinline__ void mask_tail(f32x4& acc, int t, int g, int N) {
  const int key0 = 16 * t + 4 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = key0 + r < N ? acc[r] : -INFINITY;
}

// B operand of a contraction over tokens from a ROW-major tile: the four tokens 16 t + 4 g + e of column d = 16 half + i, by the
// LDS transpose read of gfx950 (ds_read_b64_tr_b16): lane k of a 16-lane group hands in the address of a quarter row -- token
// 4 g + k / 4, columns 4 (k % 4) .. + 3 of the [4 tokens][16 columns] block -- and receives column k of the block (verified element
// by element: tools/ubench/tr16_probe.hip).  ONE instruction where rounds 2-3 issued four 2-byte reads per operand (16 per score
// tile in the dK / dV kernel).
__device__ __forceinline__ bf16x4 gather4(const unsigned short* tile, int t, int g, int i, int half) {
  const unsigned short* p = tile + (16 * t + 4 * g + (i >> 2)) * RP + 16 * half + 4 * (i & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

// ------------------------------------------------------------------------------------------------
// forward: K row-major, V transposed; a wave keeps all score tiles of its 16 queries in registers
// ------------------------------------------------------------------------------------------------
// CH key chunks of NTM tiles each, online softmax across them (running maximum m, running sum l, the output rescaled by
// exp(m_old - m_new) when a chunk raises the maximum): a 784-token window is two chunks of 25 tiles = 100 score registers per lane
// and EIGHT waves per workgroup, where holding all 50 tiles (200 registers, round 2-3) left one wave per SIMD with nothing to
// overlap its LDS gathers and MFMA latencies with (config-5 forward: 206 us per launch).
template <int NTM, int THREADS, int CH, bool REG, bool TAIL>
__global__ __launch_bounds__(THREADS) void wmsa_bf16_fwd_kernel(const Args a, float* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a.N, nH = a.nH, nt = (N + 15) / 16;
  const Carve cv = carve(smem, nt, (size_t)16 * nt * RP, 0, a.T);
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t rs = 3 * (int64_t)nH * HD;
  fill_rows(cv.rows, a.rm, b, N, nt);
  stage_rows(cv.a0, qkv_at(a, (int64_t)(nH + h) * HD), rs, N, nt, cv.rows, 1.f, a.io16);
  stage_rows(cv.a1, qkv_at(a, (int64_t)(2 * nH + h) * HD), rs, N, nt, cv.rows, 1.f, a.io16);  // V row-major: the PV operand comes by transpose read
  setup_terms<2>(cv, a, h, b, nt);
  __syncthreads();
  const unsigned short* krow = cv.a0;
  const unsigned short* vrow = cv.a1;
  const unsigned tab_b = (unsigned)(size_t)(lds_cfloat*)cv.tab + 4u * (unsigned)a.off;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, WV = blockDim.x >> 6;
  const int C = nH * HD;
  for (int rt = blockIdx.x * WV + wave; rt < nt; rt += gridDim.x * WV) {
    const int q = min(16 * rt + i, N - 1);
    bf16x4 qa, qb;
    float qf_[8];
    load_frag(a.qkv, (int64_t)h * HD + (int64_t)cv.rows[q] * rs + 8 * g, a.io16, a.scale, qa, qb, qf_);
    const int pq = cv.cr[q];
    const unsigned cqb = tab_b + (unsigned)(REG ? pk_code(pq) : pq);
    const int rq = pk_region(pq);
    const bf16x8 q8 = cat8(qa, qb);
    float m = -INFINITY;  // of query i (this lane's column of the S^T tiles)
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};  // rows = queries 4 g + r, column d = i (| 16 + i)
    // the softmax denominators as a THIRD product against a column of ones (the matrix pipe is a tenth busy, the vector ALU is the
    // bound): l[r] = sum over the keys of the bf16 probabilities the numerator uses, in the rows' layout the normalisation needs
    f32x4 ol = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int tb = c * NTM;
      if (tb >= nt) break;
      f32x4 acc[NTM];
      float mc = -INFINITY;
#pragma unroll
      for (int t = 0; t < NTM; ++t) {
        if (tb + t < nt) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (16 * (tb + t) + i) * RP + 8 * g);
          f32x4 s = terms_b<REG>(cqb, rq, cv.cr, tb + t, g);  // bias (+ shift mask): the accumulator's initial value
          s = mfma32(kf, q8, s);                              // the head's 32 channels in one product
          if (TAIL) mask_tail(s, tb + t, g, N);
          acc[t] = s;
          mc = fmaxf(mc, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        } else {
          acc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // past the window: probability zero in the paired product below
        }
      }
      mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
      const float mn = fmaxf(m, mc);
      if (CH > 1) {
        const float corr = __expf(m - mn);  // (first chunk: exp(-inf) = 0 on zero sums)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float cr_ = __shfl(corr, 4 * g + r, 64);  // the factor of query 4 g + r lives in the lanes whose column it is
          o0[r] *= cr_;
          o1[r] *= cr_;
          ol[r] *= cr_;
        }
      }
      m = mn;
      const float mL = m * LOG2E;
      // P V over PAIRS of key tiles: contraction element 8 g + j of the 32-wide product = key 4 g + j of tile t (j < 4) | of tile t + 1;
      // a chunk's odd last tile pairs with zero probabilities
#pragma unroll
      for (int t = 0; t < NTM; t += 2) {
        if (tb + t < nt) {
          const int t1 = (t + 1 < NTM && tb + t + 1 < nt) ? t + 1 : t;  // the V rows read for a missing partner: any tile of the window
          float p[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = exp_sub(acc[t][r], mL);
            p[4 + r] = t + 1 < NTM ? exp_sub(acc[t + 1 < NTM ? t + 1 : t][r], mL) : 0.f;  // (-inf scores past the window: 0)
          }
          const bf16x8 pa = cat8(pack4(p[0], p[1], p[2], p[3]), pack4(p[4], p[5], p[6], p[7]));
          const bf16x8 v0 = cat8(gather4(vrow, tb + t, g, i, 0), gather4(vrow, tb + t1, g, i, 0));
          const bf16x8 v1 = cat8(gather4(vrow, tb + t, g, i, 1), gather4(vrow, tb + t1, g, i, 1));
          o0 = mfma32(pa, v0, o0);
          o1 = mfma32(pa, v1, o1);
          ol = mfma32(pa, ones, ol);
        }
      }
    }
    // C layout: row = query 4 g + r, column = d = i (every column of ol holds the row's denominator)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * g + r;
      const float il = 1.f / ol[r];
      const float mr = __shfl(m, 4 * g + r, 64);  // the maximum of query 4 g + r lives in the lanes whose column it is
      if (qrow < N) {
        const int64_t o = (int64_t)cv.rows[qrow] * C + h * HD;
        store_el(out, o + i, a.io16, o0[r] * il);
        store_el(out, o + 16 + i, a.io16, o1[r] * il);
        if (i == 0) lse[(b * nH + h) * N + qrow] = mr + __logf(ol[r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward 1: dQ (query-tile major; K and V row-major); also writes delta[q] = sum_d dO O
//   P = exp(S - lse); dP = dO V^T; dS = P o (dP - delta); dQ = scale * dS K
// ------------------------------------------------------------------------------------------------
template <bool REG, bool TAIL>
__global__ __launch_bounds__(TH_BWD) void wmsa_bf16_bwd_dq_kernel(const Args a, const float* __restrict__ out,
                                                              const float* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ delta, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a.N, nH = a.nH, nt = (N + 15) / 16;
  const Carve cv = carve(smem, nt, (size_t)16 * nt * RP, 0, a.T);
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t rs = 3 * (int64_t)nH * HD;
  fill_rows(cv.rows, a.rm, b, N, nt);
  stage_rows(cv.a0, qkv_at(a, (int64_t)(nH + h) * HD), rs, N, nt, cv.rows, 1.f, a.io16);
  stage_rows(cv.a1, qkv_at(a, (int64_t)(2 * nH + h) * HD), rs, N, nt, cv.rows, 1.f, a.io16);
  setup_terms(cv, a, h, b, nt);
  __syncthreads();
  const unsigned short* krow = cv.a0;
  const unsigned short* vrow = cv.a1;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, WV = blockDim.x >> 6;
  const int C = nH * HD;
  for (int rt = blockIdx.x * WV + wave; rt < nt; rt += gridDim.x * WV) {
    const bool qvalid = 16 * rt + i < N;
    const int q = min(16 * rt + i, N - 1);
    bf16x4 qa, qb, ga, gb, oa_, ob_;
    float qf_[8], gf_[8], of_[8];
    load_frag(a.qkv, (int64_t)h * HD + (int64_t)cv.rows[q] * rs + 8 * g, a.io16, a.scale, qa, qb, qf_);
    load_frag(dout, (int64_t)cv.rows[q] * C + h * HD + 8 * g, a.io16, 1.f, ga, gb, gf_);
    load_frag(out, (int64_t)cv.rows[q] * C + h * HD + 8 * g, a.io16, 1.f, oa_, ob_, of_);
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) dl = fmaf(gf_[s], of_[s], dl);
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float L = lse[(b * nH + h) * N + q];
    if (g == 0 && qvalid) delta[(b * nH + h) * N + q] = dl;
    const int pq = cv.cr[q];
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (16 * t + i) * RP + 8 * g);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + (16 * t + i) * RP + 8 * g);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = mfma(__builtin_shufflevector(kf, kf, 0, 1, 2, 3), qa, s);
      s = mfma(__builtin_shufflevector(kf, kf, 4, 5, 6, 7), qb, s);
      dp = mfma(__builtin_shufflevector(vf, vf, 0, 1, 2, 3), ga, dp);
      dp = mfma(__builtin_shufflevector(vf, vf, 4, 5, 6, 7), gb, dp);
      add_terms_t<REG, TAIL>(s, cv.tab, cv.cr, a.off, t, g, pq, N);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = __expf(s[r] - L) * (dp[r] - dl);  // exp(-inf) = 0 for padded keys
      const bf16x4 da = pack4(ds[0], ds[1], ds[2], ds[3]);
      dq0 = mfma(da, gather4(krow, t, g, i, 0), dq0);
      dq1 = mfma(da, gather4(krow, t, g, i, 1), dq1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * rt + 4 * g + r;
      if (qrow < N) {
        const int64_t o = (int64_t)cv.rows[qrow] * rs + (int64_t)h * HD;
        store_el(dqkv, o + i, a.io16, dq0[r] * a.scale);
        store_el(dqkv, o + 16 + i, a.io16, dq1[r] * a.scale);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward 2: dK, dV (key-tile major; Q (pre-scaled) and dO row-major, lse / delta in LDS)
//   dV = P^T dO ; dK = dS^T (scale Q)
// ------------------------------------------------------------------------------------------------
template <bool REG, bool TAIL>
__global__ __launch_bounds__(TH_BWD) void wmsa_bf16_bwd_dkv_kernel(const Args a, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a.N, nH = a.nH, nt = (N + 15) / 16;
  const Carve cv = carve(smem, nt, (size_t)16 * nt * RP, 2 * 16 * nt, a.T);
  float* l_lds = cv.extra;
  float* d_lds = l_lds + 16 * nt;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  fill_rows(cv.rows, a.rm, b, N, nt);
  stage_rows(cv.a0, qkv_at(a, (int64_t)h * HD), rs, N, nt, cv.rows, a.scale, a.io16);
  stage_rows(cv.a1, a.io16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(dout) + h * HD) : dout + h * HD, C, N, nt,
             cv.rows, 1.f, a.io16);
  setup_terms(cv, a, h, b, nt);
  for (int r = threadIdx.x; r < 16 * nt; r += blockDim.x) {
    l_lds[r] = r < N ? lse[(b * nH + h) * N + r] : INFINITY;  // exp(s - inf) = 0 for padded queries
    d_lds[r] = r < N ? delta[(b * nH + h) * N + r] : 0.f;
  }
  __syncthreads();
  const unsigned short* qrow_l = cv.a0;
  const unsigned short* grow_l = cv.a1;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, WV = blockDim.x >> 6;
  for (int kt = blockIdx.x * WV + wave; kt < nt; kt += gridDim.x * WV) {
    const int key = min(16 * kt + i, N - 1);
    const bool kvalid = 16 * kt + i < N;
    bf16x4 ka, kb, va, vb;
    float kf_[8], vf_[8];
    load_frag(a.qkv, (int64_t)cv.rows[key] * rs + (int64_t)(nH + h) * HD + 8 * g, a.io16, 1.f, ka, kb, kf_);
    load_frag(a.qkv, (int64_t)cv.rows[key] * rs + (int64_t)(2 * nH + h) * HD + 8 * g, a.io16, 1.f, va, vb, vf_);
    const int pkey = cv.cr[key];
    const int ckey = REG ? pk_code(pkey) : pkey, rkey = pk_region(pkey);
    f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = {0.f, 0.f, 0.f, 0.f}, dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < nt; ++qt) {
      // S tile [query][key]: A = Q rows (LDS), B = K (registers); C layout: column = key i, row = query 4 g + r
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qrow_l + (16 * qt + i) * RP + 8 * g);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(grow_l + (16 * qt + i) * RP + 8 * g);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = mfma(__builtin_shufflevector(qf, qf, 0, 1, 2, 3), ka, s);
      s = mfma(__builtin_shufflevector(qf, qf, 4, 5, 6, 7), kb, s);
      dp = mfma(__builtin_shufflevector(gf, gf, 0, 1, 2, 3), va, dp);
      dp = mfma(__builtin_shufflevector(gf, gf, 4, 5, 6, 7), vb, dp);
      const int q0 = 16 * qt + 4 * g;
      const int4 cq = *reinterpret_cast<const int4*>(cv.cr + q0);
      const float4 lq = *reinterpret_cast<const float4*>(l_lds + q0);
      const float4 dq = *reinterpret_cast<const float4*>(d_lds + q0);
      const int qc[4] = {cq.x, cq.y, cq.z, cq.w};
      const float lv[4] = {lq.x, lq.y, lq.z, lq.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w};
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float term = cv.tab[(REG ? pk_code(qc[r]) : qc[r]) - ckey + a.off];
        if (REG && pk_region(qc[r]) != rkey) term -= 100.f;
        const float sv = (!TAIL || (q0 + r < N && kvalid)) ? s[r] + term : -INFINITY;
        p[r] = __expf(sv - lv[r]);
        ds[r] = p[r] * (dp[r] - dv[r]);
      }
      const bf16x4 pa = pack4(p[0], p[1], p[2], p[3]), da = pack4(ds[0], ds[1], ds[2], ds[3]);
      dv0 = mfma(pa, gather4(grow_l, qt, g, i, 0), dv0);
      dv1 = mfma(pa, gather4(grow_l, qt, g, i, 1), dv1);
      dk0 = mfma(da, gather4(qrow_l, qt, g, i, 0), dk0);
      dk1 = mfma(da, gather4(qrow_l, qt, g, i, 1), dk1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = 16 * kt + 4 * g + r;
      if (krow < N) {
        const int64_t ok = (int64_t)cv.rows[krow] * rs + (int64_t)(nH + h) * HD, ov = (int64_t)cv.rows[krow] * rs + (int64_t)(2 * nH + h) * HD;
        store_el(dqkv, ok + i, a.io16, dk0[r]);
        store_el(dqkv, ok + 16 + i, a.io16, dk1[r]);
        store_el(dqkv, ov + i, a.io16, dv0[r]);
        store_el(dqkv, ov + 16 + i, a.io16, dv1[r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, ONE pass (round 5): dQ, dK, dV of a (window, head) from a single evaluation of every score tile.
// The two kernels above evaluate S = K Q^T, dP = dO V^T and -- the expensive part: the tiles are vector-ALU bound (bias / mask terms,
// exp, dS: ~8 vector instructions per element against 128 matrix flops) -- P and dS TWICE, once query-major for dQ and once key-major
// for dK / dV (config 5: 111-150 us each per stage-0 launch against 80 us forward).  Here a wave owns key tiles (dK / dV accumulate in
// its registers, as in the key-major kernel) and the queries are walked in CHUNKS of <= QC tiles whose scaled Q and dO rows sit in LDS
// beside the whole K tile.  dQ: in step s of a chunk wave w works on query tile (s + w) mod chunk -- eight waves, eight DIFFERENT query
// tiles --, sums the shares of all its key tiles in registers and adds them to the chunk's fp32 dQ tile in LDS with plain
// read-modify-writes; one barrier per step keeps the waves on distinct tiles.  (LDS float atomics instead -- every wave adding every
// pair's share -- measured 6.7 ms per launch against 0.75 ms without the accumulation: tools/bench_wmsa.py.)  The one operand the
// key-major tile layout does not give -- dS with the QUERY on the operand's row, for dQ = dS K -- comes from a 640-byte per-wave LDS
// turn-around: the packed tile is written key-major and read back with the transpose read (ds_read_b64_tr_b16).
// Per tile: 10 MFMAs, ONE pass of the vector arithmetic.  Eight waves (a wave owns up to 7 key tiles: 112 accumulator registers);
// workgroup = one (window, head) pair (launches with fewer pairs than CUs keep the two kernels above, which split a pair over several
// workgroups).  Chunks are balanced and must hold >= 8 tiles each (bwd1_chunks; otherwise the two kernels above).
// ------------------------------------------------------------------------------------------------
constexpr int NW1 = 8;  // waves of the one-pass backward (twelve -- 168 registers, 45 of them spilled, chunks of >= 12 tiles -- measured 757 vs 652 us)
constexpr int TH_BWD1 = 64 * NW1;
constexpr int KT_MAX = (NTB + NW1 - 1) / NW1;  // key tiles a wave can own: NW1 x KT_MAX >= 50
constexpr int DQP = 36;        // pitch of the fp32 dQ tile in floats: the four row groups of a tile land 16 banks apart
constexpr int TSP = 20;        // pitch of the per-wave turn-around tile in bf16 elements

struct Carve1 {
  unsigned short *kb, *qb, *gb, *tscr;
  float *dq, *l, *dl, *tab, *dtab;
  int *cr, *rows;
};
// qc = query tiles of the largest chunk; dtab: room for the table gradient of the (window, head) pair (T floats)
// grads = false: the table-gradient pass of its own (no dQ tile, no turn-around scratch)
__host__ __device__ inline size_t bwd1_lds_bytes(int nt, int qc, int T, bool dtab, bool grads = true) {
  const size_t qrows = 16 * (size_t)qc, tp = (size_t)((T + 3) & ~3);
  return 2 * ((size_t)16 * (nt + 1) * RP + 2 * qrows * RP + (grads ? NW1 * 16 * TSP : 0)) +   // (K: the window + one tile of zeros)
         4 * ((grads ? qrows * DQP : 0) + 2 * qrows + tp * (dtab ? 3 : 1) + 2 * 16 * (size_t)nt) + 64;
}
// chunks of query tiles: the fewest that fit LDS, BALANCED (chunk c = tiles [c nt / nc, (c + 1) nt / nc)), every chunk >= 8 tiles (the
// step-synchronous rotation puts the eight waves on eight distinct tiles); returns the largest chunk, 0: no admissible chunking
inline int bwd1_chunks(int nt, int T, bool dtab, int* nchunks, bool grads = true) {
  for (int nc = (nt + 12) / 13; nc <= nt; ++nc) {
    if (nt / nc < NW1) break;
    const int qc = (nt + nc - 1) / nc;
    if (bwd1_lds_bytes(nt, qc, T, dtab, grads) > 160 * 1024) continue;
    *nchunks = nc;
    return qc;
  }
  return 0;
}
__device__ __forceinline__ Carve1 carve1(unsigned char* smem, int nt, int qc, int T, bool dtab, bool grads = true) {
  Carve1 c;
  const int qrows = 16 * qc, tp = (T + 3) & ~3;
  c.kb = reinterpret_cast<unsigned short*>(smem);
  c.qb = c.kb + 16 * (nt + 1) * RP;
  c.gb = c.qb + qrows * RP;
  c.tscr = c.gb + qrows * RP;
  c.dq = reinterpret_cast<float*>(c.tscr + (grads ? NW1 * 16 * TSP : 0));
  c.l = c.dq + (grads ? qrows * DQP : 0);
  c.dl = c.l + qrows;
  c.tab = c.dl + qrows;
  c.dtab = c.tab + tp;
  c.cr = reinterpret_cast<int*>(c.dtab + (dtab ? 2 * tp : 0));  // (the table-gradient column holds 64-bit fixed-point words)
  c.rows = c.cr + 16 * nt;
  return c;
}

// DTAB: the gradient of the relative-position table (swin_transformer.py:110-151, trainable under SGD over all parameters): d bias =
// dS, binned by code[q] - code[k] + off into an LDS column of the (window, head) pair.
// Round 6: the column is 64-bit FIXED POINT and the binning is ds_add_u64.  tools/ubench/lds_atomic_probe.hip, eight waves on one
// column: a ds_add_f32 instruction occupies the CU's LDS for 192 clocks, a ds_add_u32 for 7.5, a ds_add_u64 for 8.0 (a plain
// read-modify-write for 10.6) -- float atomics in LDS run at a twenty-fifth of the integer rate, which was 4.3 of a stage-0 launch's
// 4.9 ms in round 5 (and what a pre-reduction of dS along the score diagonals, DPP + ds_bpermute, could only shave: 4.9 -> 4.5 ms).
// The scale is a power of two per pair, chosen so that no sum can overflow: |dS| = p |dP - delta| <= |dO_q . V_k| + |dO_q . O_q| <=
// 64 max|dO| max|V| = B (O is a convex combination of V rows), at most N scores per table row, so 2^e B N < 2^62; ldexpf(dS, e) is then
// an integer of <= 24 significant bits that the conversion represents exactly: the column is the EXACT sum of the fp32 dS values
// (integer addition is associative: the table gradient of a pair no longer depends on the order in which the waves arrive).  A
// non-finite dO / V (B not finite) marks the whole column NaN.
template <bool REG, bool TAIL, bool DTAB, bool GRADS = true, int KT = KT_MAX>
__global__ __launch_bounds__(TH_BWD1) void wmsa_bf16_bwd_fused_kernel(const Args a, const float* __restrict__ out,
                                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                                      float* __restrict__ delta, float* __restrict__ dqkv, int qc, int nchunks,
                                                                      float* __restrict__ dtable, float* __restrict__ dtable_ws) {
  // DTAB + GRADS: the pair's table gradient rides in the launch that produces dQ / dK / dV (windows whose 64-bit column still fits LDS
  // beside the dQ tile: N <= 400).  DTAB alone: a pass of its own behind the plain launch -- S, P, dP, dS and the binning only, no dK /
  // dV / dQ products, no accumulator registers, no dQ tile -- for the 784-token windows, where the 42 KB column does not fit otherwise
  // (with float atomics that pass measured 3.06 ms on 1024 pairs and lost to the single kernel; with ds_add_u64 it is the score
  // arithmetic of a forward).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a.N, nH = a.nH, nt = (N + 15) / 16;
  const Carve1 cv = carve1(smem, nt, qc, a.T, DTAB, GRADS);
  unsigned long long* const dcol = reinterpret_cast<unsigned long long*>(cv.dtab);
  if (DTAB)
    for (int t = threadIdx.x; t < a.T; t += TH_BWD1) dcol[t] = 0ull;
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t rs = 3 * (int64_t)nH * HD;
  const int C = nH * HD;
  fill_rows(cv.rows, a.rm, b, N, nt);
  stage_rows(cv.kb, qkv_at(a, (int64_t)(nH + h) * HD), rs, N, nt + 1, cv.rows, 1.f, a.io16);  // K, row-major, the whole window + one tile of zeros
  {  // table column of the head + packed code | region of the window's tokens (setup_terms on this carve)
    Carve tmp;
    tmp.tab = cv.tab; tmp.cr = cv.cr;
    setup_terms(tmp, a, h, b, nt);
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;  // (scalar: the tiles' offsets are SALU work)
  unsigned short* const tscr = cv.tscr + wave * 16 * TSP;
  // key tiles of this wave: wave, wave + 8, ...: V fragments, code / region of the lane's key, dK / dV accumulators -- in registers
  bf16x4 va[KT], vb[KT];
  f32x4 dk0[KT], dk1[KT], dv0[KT], dv1[KT];
  float vmax = 0.f;
  __syncthreads();  // rows / code table in place
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    dk0[j] = dk1[j] = dv0[j] = dv1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt = min(wave + NW1 * j, nt - 1), key = min(16 * kt + i, N - 1);
    float vf_[8];
    load_frag(a.qkv, (int64_t)cv.rows[key] * rs + (int64_t)(2 * nH + h) * HD + 8 * g, a.io16, 1.f, va[j], vb[j], vf_);
    if (DTAB) {
#pragma unroll
      for (int u = 0; u < 8; ++u) vmax = fmaxf(vmax, fabsf(vf_[u]) + (vf_[u] != vf_[u] ? INFINITY : 0.f));
    }
  }
  int dexp = 0;        // fixed-point exponent of the pair's table-gradient column
  bool dbad = false;   // non-finite operands: the column leaves as NaN
  if (DTAB) {
    float gmax = 0.f;  // max |dO| over the pair's rows (one pass over 16 nt x 32 values)
    for (int it = threadIdx.x; it < N * 4; it += TH_BWD1) {
      const int row = it >> 2, c8 = it & 3;
      const int64_t e = (int64_t)cv.rows[row] * C + h * HD + 8 * c8;
      bf16x4 lo_, hi_;
      float f[8];
      load_frag(dout, e, a.io16, 1.f, lo_, hi_, f);
#pragma unroll
      for (int u = 0; u < 8; ++u) gmax = fmaxf(gmax, fabsf(f[u]) + (f[u] != f[u] ? INFINITY : 0.f));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    float* const red = cv.l;   // (free until the first chunk is staged)
    if (lane == 0) {
      red[wave] = gmax;
      red[16 + wave] = vmax;
    }
    __syncthreads();
    gmax = vmax = 0.f;
#pragma unroll
    for (int w = 0; w < TH_BWD1 / 64; ++w) {
      gmax = fmaxf(gmax, red[w]);
      vmax = fmaxf(vmax, red[16 + w]);
    }
    const float B = 64.f * gmax * vmax * (float)N;
    dbad = !(B < INFINITY);
    int ex = 0;
    if (!dbad && B > 0.f) (void)frexpf(B, &ex);  // B < 2^ex
    dexp = dbad ? 0 : min(100, 60 - ex);         // 2^dexp B < 2^60 (B holds the factor N; one bit of slack for the bf16 roundings); dexp <= 100
  }
  const bool kvalid_all = !TAIL;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int qt0 = ch * nt / nchunks, qtn = (ch + 1) * nt / nchunks - qt0, qrows = 16 * qtn;
    __syncthreads();  // (the previous chunk's readers are done)
    // ---- stage the chunk: scaled Q rows, dO rows (bf16), lse, delta = sum_d dO O (also stored for the record), zero the dQ tile ----
    for (int it = threadIdx.x; it < qrows * 8; it += TH_BWD1) {
      const int row = it >> 3, c4 = it & 7, q = 16 * qt0 + row;
      const bool in = q < N;
      const int tok = cv.rows[in ? q : N - 1];
      float qv[4], gv[4], ov[4];
      if (a.io16) {
        const ushort4 q4 = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(a.qkv) + (int64_t)tok * rs + h * HD + 4 * c4);
        const ushort4 g4 = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(dout) + (int64_t)tok * C + h * HD + 4 * c4);
        const ushort4 o4 = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(out) + (int64_t)tok * C + h * HD + 4 * c4);
        qv[0] = bf2f(q4.x); qv[1] = bf2f(q4.y); qv[2] = bf2f(q4.z); qv[3] = bf2f(q4.w);
        gv[0] = bf2f(g4.x); gv[1] = bf2f(g4.y); gv[2] = bf2f(g4.z); gv[3] = bf2f(g4.w);
        ov[0] = bf2f(o4.x); ov[1] = bf2f(o4.y); ov[2] = bf2f(o4.z); ov[3] = bf2f(o4.w);
      } else {
        const float4 q4 = *reinterpret_cast<const float4*>(a.qkv + (int64_t)tok * rs + h * HD + 4 * c4);
        const float4 g4 = *reinterpret_cast<const float4*>(dout + (int64_t)tok * C + h * HD + 4 * c4);
        const float4 o4 = *reinterpret_cast<const float4*>(out + (int64_t)tok * C + h * HD + 4 * c4);
        qv[0] = q4.x; qv[1] = q4.y; qv[2] = q4.z; qv[3] = q4.w;
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        ov[0] = o4.x; ov[1] = o4.y; ov[2] = o4.z; ov[3] = o4.w;
      }
      if (!in) {
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] = gv[e] = ov[e] = 0.f;
      }
      *reinterpret_cast<bf16x4*>(cv.qb + row * RP + 4 * c4) = pack4(qv[0] * a.scale, qv[1] * a.scale, qv[2] * a.scale, qv[3] * a.scale);
      *reinterpret_cast<bf16x4*>(cv.gb + row * RP + 4 * c4) = pack4(gv[0], gv[1], gv[2], gv[3]);
      float dsum = gv[0] * ov[0] + gv[1] * ov[1] + gv[2] * ov[2] + gv[3] * ov[3];
      dsum += __shfl_xor(dsum, 1, 64);
      dsum += __shfl_xor(dsum, 2, 64);
      dsum += __shfl_xor(dsum, 4, 64);
      if (c4 == 0) {
        cv.dl[row] = in ? dsum : 0.f;
        cv.l[row] = in ? lse[(b * nH + h) * N + q] : INFINITY;  // exp(s - inf) = 0 for padded queries
        if (GRADS && in && delta) delta[(b * nH + h) * N + q] = dsum;
      }
    }
    if (GRADS)
      for (int it = threadIdx.x; it < qrows * DQP; it += TH_BWD1) cv.dq[it] = 0.f;
    __syncthreads();
    // ---- step s: wave w on query tile (s + w) mod qtn (qtn >= 8: eight distinct tiles), against all of its key tiles ----
    for (int st = 0; st < qtn; ++st) {
      int qt = st + wave;
      qt = qt >= qtn ? qt - qtn : qt;
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(cv.qb + (16 * qt + i) * RP + 8 * g);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(cv.gb + (16 * qt + i) * RP + 8 * g);
      const bf16x4 gq0 = gather4(cv.gb, qt, g, i, 0), gq1 = gather4(cv.gb, qt, g, i, 1);
      const bf16x4 qq0 = gather4(cv.qb, qt, g, i, 0), qq1 = gather4(cv.qb, qt, g, i, 1);
      const int ql = 16 * qt + 4 * g, q0 = 16 * qt0 + ql;
      const int4 cq = *reinterpret_cast<const int4*>(cv.cr + q0);
      const float4 lq = *reinterpret_cast<const float4*>(cv.l + ql);
      const float4 dq4 = *reinterpret_cast<const float4*>(cv.dl + ql);
      const int qcd[4] = {REG ? pk_code(cq.x) : cq.x, REG ? pk_code(cq.y) : cq.y, REG ? pk_code(cq.z) : cq.z, REG ? pk_code(cq.w) : cq.w};
      const int qrg[4] = {pk_region(cq.x), pk_region(cq.y), pk_region(cq.z), pk_region(cq.w)};
      const float lvL[4] = {lq.x * LOG2E, lq.y * LOG2E, lq.z * LOG2E, lq.w * LOG2E}, dv[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
      f32x4 dqa = {0.f, 0.f, 0.f, 0.f}, dqb = {0.f, 0.f, 0.f, 0.f};
      auto tile = [&](const int j) {
        const int kt = wave + NW1 * j;
        {
          const int kte = min(kt, nt);  // (a tile past the window reads the zero tile: S = bias only, and its dS meets K = 0 in the dQ product)
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(cv.kb + (16 * kte + i) * RP + 8 * g);
          const bool kvalid = kvalid_all || 16 * kt + i < N;
          const int pkey = cv.cr[min(16 * kt + i, N - 1)];
          const int ckey_j = REG ? pk_code(pkey) : pkey, rkey_j = pk_region(pkey);
          float p[4], ds[4];
          int bin[4];  // table row of the lane's four scores; -1: padded query / key (TAIL)
          // the additive terms are the products' INITIAL accumulators: S = (bias + mask) + Q K^T, dP - delta = -delta + dO V^T
          f32x4 sacc, dp = {-dv[0], -dv[1], -dv[2], -dv[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bin[r] = qcd[r] - ckey_j + a.off;
            float term = cv.tab[bin[r]];
            if (REG && qrg[r] != rkey_j) term -= 100.f;
            sacc[r] = term;
          }
          sacc = mfma32(qf, kf, sacc);  // the head's 32 channels in one product each
          dp = mfma32(gf, cat8(va[j], vb[j]), dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool live = !TAIL || (q0 + r < N && kvalid);
            const float sv = live ? sacc[r] : -INFINITY;
            p[r] = exp_sub(sv, lvL[r]);
            ds[r] = p[r] * dp[r];
            if (DTAB && !(live && kt < nt)) bin[r] = -1;
          }
          if (DTAB) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (bin[r] >= 0) atomicAdd(dcol + bin[r], dbad ? 1ull : (unsigned long long)__float2ll_rn(ldexpf(ds[r], dexp)));
          }
          if (GRADS) {
            const bf16x4 pa = pack4(p[0], p[1], p[2], p[3]), da = pack4(ds[0], ds[1], ds[2], ds[3]);
            dv0[j] = mfma(pa, gq0, dv0[j]);
            dv1[j] = mfma(pa, gq1, dv1[j]);
            dk0[j] = mfma(da, qq0, dk0[j]);
            dk1[j] = mfma(da, qq1, dk1[j]);
            // dQ share of this key tile: dS with the query on the operand's row = the tile written key-major and read transposed
            *reinterpret_cast<bf16x4*>(tscr + i * TSP + 4 * g) = da;  // M[key i][queries 4 g ..]
            const bf16x4 dst = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) bf16x4*)(tscr + (4 * g + (i >> 2)) * TSP + 4 * (i & 3)));  // M[keys 4 g ..][query i]
            dqa = mfma(dst, gather4(cv.kb, kte, g, i, 0), dqa);
            dqb = mfma(dst, gather4(cv.kb, kte, g, i, 1), dqb);
          }
        }
      };
      // The first KT - 1 tiles of a wave run as ONE straight-line block, valid or not (a tile past the window works on the zero tile
      // and its dK / dV are never stored), the last one behind a branch: with every tile behind its own branch (round 5) the compiler
      // could not interleave them, and a tile is a chain of LDS reads, matrix products, a table read, exp, conversions, an LDS
      // turn-around and more products -- ~1300 clocks per tile and wave at two waves per SIMD.  KT is picked per window size on the host
      // (784 tokens: seven -- six for every wave and the 49th tile for wave 0).
#pragma unroll
      for (int j = 0; j < KT - 1; ++j) tile(j);
      if (wave + NW1 * (KT - 1) < nt) tile(KT - 1);
      // this wave is the only one on query tile qt in this step: plain read-modify-write of its rows of the dQ tile
      if (GRADS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cv.dq[(ql + r) * DQP + i] += dqa[r];
          cv.dq[(ql + r) * DQP + 16 + i] += dqb[r];
        }
        __syncthreads();
      }
    }
    // ---- the chunk's dQ ----
    for (int it = threadIdx.x; GRADS && it < qrows * 8; it += TH_BWD1) {
      const int row = it >> 3, c4 = it & 7, q = 16 * qt0 + row;
      if (q >= N) continue;
      const float4 v = *reinterpret_cast<const float4*>(cv.dq + row * DQP + 4 * c4);
      const int64_t o = (int64_t)cv.rows[q] * rs + (int64_t)h * HD + 4 * c4;
      if (a.io16) {
        *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(dqkv) + o) = pack4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
      } else {
        *reinterpret_cast<float4*>(dqkv + o) = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
      }
    }
  }
  if (DTAB) {
    __syncthreads();  // every wave's ds_add of the last chunk (the table pass has no per-step barrier)
    if (dtable_ws) {
      // round 6: the pair's column leaves as PLAIN stores into the caller's workspace [window][head][T]; dtable_reduce_kernel adds the
      // windows.  (One global atomic per entry and pair was 5.4 M device-scope atomics onto 21 K addresses per stage-0 launch -- every
      // window of a head adds to the same T words --: 3.8 of the launch's 4.5 ms once the LDS binning was pre-reduced.)
      float* col = dtable_ws + ((int64_t)b * nH + h) * a.T;
      for (int t = threadIdx.x; t < a.T; t += TH_BWD1) {
        const long long w = (long long)dcol[t];
        col[t] = dbad ? (w != 0 ? NAN : 0.f) : (float)ldexp((double)w, -dexp);
      }
    } else {
      for (int t = threadIdx.x; t < a.T; t += TH_BWD1) {
        const long long w = (long long)dcol[t];
        const float v = dbad ? (w != 0 ? NAN : 0.f) : (float)ldexp((double)w, -dexp);
        if (v != 0.f) atomicAdd(dtable + (int64_t)t * nH + h, v);
      }
    }
  }
  // ---- dK, dV of the wave's key tiles ----
#pragma unroll
  for (int j = 0; GRADS && j < KT; ++j) {
    const int kt = wave + NW1 * j;
    if (kt >= nt) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = 16 * kt + 4 * g + r;
      if (krow < N) {
        const int64_t ok = (int64_t)cv.rows[krow] * rs + (int64_t)(nH + h) * HD, ov = (int64_t)cv.rows[krow] * rs + (int64_t)(2 * nH + h) * HD;
        store_el(dqkv, ok + i, a.io16, dk0[j][r]);
        store_el(dqkv, ok + 16 + i, a.io16, dk1[j][r]);
        store_el(dqkv, ov + i, a.io16, dv0[j][r]);
        store_el(dqkv, ov + 16 + i, a.io16, dv1[j][r]);
      }
    }
  }
}

// dtable[t][h] += sum over the windows of a segment of ws[window][h][t]; grid (ceil(nH T / 256), segments): consecutive threads walk
// consecutive t of one head (coalesced), a thread sums its segment's windows in window order and adds once
__global__ __launch_bounds__(256) void dtable_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dtable, int64_t B_, int nH, int T,
                                                            int seg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // h * T + t
  if (e >= (int64_t)nH * T) return;
  const int h = (int)(e / T), t = (int)(e - (int64_t)h * T);
  const int64_t b0 = (int64_t)blockIdx.y * seg, b1 = b0 + seg < B_ ? b0 + seg : B_;
  float acc = 0.f;
  const int64_t stride = (int64_t)nH * T;
  const float* p = ws + b0 * stride + e;
  int64_t b = b0;
  for (; b + 8 <= b1; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    p += 8 * stride;
  }
  for (; b < b1; ++b, p += stride) acc += *p;
  if (acc != 0.f) atomicAdd(dtable + (int64_t)t * nH + h, acc);
}

inline int pick_split(int64_t pairs, int nt, int waves) {
  int qs = 1;  // one workgroup per CU (LDS): split the tiles of a (window, head) pair only while CUs would sit idle
  while (pairs * qs < 256 && (nt + qs * waves - 1) / (qs * waves) >= 2 && qs < 8) qs *= 2;
  return qs;
}

template <typename K>
inline bool set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

inline bool misaligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15u) != 0;
}

inline int check(const Args& a, int head_dim, const int32_t* rowmap, int map_windows, int64_t tokens) {
  if (!a.qkv || !a.table || !a.code || a.B_ <= 0 || a.nH <= 0 || a.T <= 0) return VITTA_ERR_INVALID_ARG;
  if (!vitta_wmsa_bf16_supported(a.N, head_dim, a.T)) return VITTA_ERR_UNSUPPORTED;
  if (a.region && (a.nW <= 0 || a.B_ % a.nW)) return VITTA_ERR_INVALID_ARG;
  if (rowmap && (map_windows <= 0 || a.B_ % map_windows || tokens != (int64_t)map_windows * a.N)) return VITTA_ERR_INVALID_ARG;
  return VITTA_OK;
}

}  // namespace

extern "C" {

int vitta_wmsa_bf16_supported(int32_t N, int32_t head_dim, int32_t table_rows) {
  if (head_dim != HD || N < 1 || N > 16 * NTB || table_rows < 1 || table_rows > T_MAX) return 0;
  const int nt = (N + 15) / 16;
  const size_t f = lds_bytes(nt, (size_t)16 * nt * RP, 0, table_rows), b1 = lds_bytes(nt, (size_t)16 * nt * RP, 0, table_rows),
               b2 = lds_bytes(nt, (size_t)16 * nt * RP, 2 * 16 * nt, table_rows);
  return (f <= 160 * 1024 && b1 <= 160 * 1024 && b2 <= 160 * 1024) ? 1 : 0;
}

size_t vitta_wmsa_bf16_dtable_workspace_bytes(int64_t B_, int32_t nH, int32_t table_rows) {
  if (B_ <= 0 || nH <= 0 || table_rows <= 0) return 0;
  return (size_t)B_ * (size_t)nH * (size_t)table_rows * sizeof(float);
}

int vitta_wmsa_bf16_dtable_supported(int32_t N, int32_t head_dim, int32_t table_rows) {
  if (!vitta_wmsa_bf16_supported(N, head_dim, table_rows)) return 0;
  const int nt = (N + 15) / 16;
  int nc = 0;
  if (nt > NW1 * KT_MAX) return 0;
  if (bwd1_chunks(nt, table_rows, true, &nc) > 0) return 1;                                                     // in the gradients' launch
  return (bwd1_chunks(nt, table_rows, false, &nc) > 0 && bwd1_chunks(nt, table_rows, true, &nc, false) > 0) ? 1 : 0;  // as a pass of its own
}

int vitta_wmsa_rel_fwd_bf16(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                            const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                            float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                            float* d_out, float* d_lse, void* stream) {
  return vitta_wmsa_rel_fwd_bf16_io(d_qkv, d_table, T, d_code, code_off, d_region, nW, B_, N, nH, head_dim, scale, d_rowmap, map_windows,
                                    tokens_per_sample, d_out, d_lse, 0, stream);
}

int vitta_wmsa_rel_fwd_bf16_io(const void* d_qkv_, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                               const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                               float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                               void* d_out_, float* d_lse, int32_t io_bf16, void* stream) {
  const float* d_qkv = static_cast<const float*>(d_qkv_);
  float* d_out = static_cast<float*>(d_out_);
  const Args a{d_qkv, d_table, d_code, d_region, T, code_off, d_region ? nW : 1, B_, N, nH, scale,
               RowMap{d_rowmap, d_rowmap ? map_windows : 1, tokens_per_sample}, io_bf16 ? 1 : 0};
  if (!d_out || !d_lse) return VITTA_ERR_INVALID_ARG;
  const int rc = check(a, head_dim, d_rowmap, map_windows, tokens_per_sample);
  if (rc != VITTA_OK) return rc;
  if (misaligned(d_qkv, d_out)) return VITTA_ERR_INVALID_ARG;
  const int nt = (N + 15) / 16, qs = pick_split(B_ * nH, nt, TH_FWD_S / 64);
  const size_t lds = lds_bytes(nt, (size_t)16 * nt * RP, 0, T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool reg = a.region != nullptr, tail = (N % 16) != 0;
#define WMSA_FWD(CHV, R, TL)                                                                                                          \
  do {                                                                                                                                \
    if (!set_lds(wmsa_bf16_fwd_kernel<25, TH_FWD_S, CHV, R, TL>, lds)) return VITTA_ERR_LAUNCH;                                         \
    VITTA_LAUNCH((wmsa_bf16_fwd_kernel<25, TH_FWD_S, CHV, R, TL>), dim3(qs, nH, (unsigned)B_), dim3(TH_FWD_S), lds, st, a, d_out, d_lse); \
  } while (0)
  // <= 400 tokens: one key chunk; 401 .. 800: two chunks of 25 tiles, online softmax; eight waves either way
  // round 6: 401 .. 800 tokens run SIXTEEN waves on four chunks of 13 tiles (128 registers, four waves per SIMD): with the score
  // products on the 32-wide MFMA, exp on one multiply-add + v_exp_f32 and the table reads on byte offsets the kernel is latency bound
  // (vector ALU busy 34 % of a wave's lifetime at two waves per SIMD), and the second pair of waves buys 398 -> 339 us on 1024 shifted
  // pairs (round 4 measured no gain from the same split, before those cuts).  VITTA_WMSA_FWD16=0: the eight-wave form.
  static const bool fwd16 = [] { const char* e = std::getenv("VITTA_WMSA_FWD16"); return !(e && e[0] == '0'); }();
  if (fwd16 && nt > 25) {
#define WMSA_FWD16(R, TL)                                                                                                             \
  do {                                                                                                                                \
    if (!set_lds(wmsa_bf16_fwd_kernel<13, 1024, 4, R, TL>, lds)) return VITTA_ERR_LAUNCH;                                               \
    VITTA_LAUNCH((wmsa_bf16_fwd_kernel<13, 1024, 4, R, TL>), dim3(qs, nH, (unsigned)B_), dim3(1024), lds, st, a, d_out, d_lse);         \
  } while (0)
    if (reg) { if (tail) WMSA_FWD16(true, true); else WMSA_FWD16(true, false); }
    else { if (tail) WMSA_FWD16(false, true); else WMSA_FWD16(false, false); }
#undef WMSA_FWD16
    return VITTA_OK;
  }
  if (nt <= 25) {
    if (reg) { if (tail) WMSA_FWD(1, true, true); else WMSA_FWD(1, true, false); }
    else { if (tail) WMSA_FWD(1, false, true); else WMSA_FWD(1, false, false); }
  } else {
    if (reg) { if (tail) WMSA_FWD(2, true, true); else WMSA_FWD(2, true, false); }
    else { if (tail) WMSA_FWD(2, false, true); else WMSA_FWD(2, false, false); }
  }
#undef WMSA_FWD
  return VITTA_OK;
}

int vitta_wmsa_rel_bwd_bf16(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                            const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                            float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                            const float* d_out, const float* d_dout, const float* d_lse, float* d_delta, float* d_dqkv,
                            void* stream) {
  return vitta_wmsa_rel_bwd_bf16_io(d_qkv, d_table, T, d_code, code_off, d_region, nW, B_, N, nH, head_dim, scale, d_rowmap, map_windows,
                                    tokens_per_sample, d_out, d_dout, d_lse, d_delta, d_dqkv, nullptr, nullptr, 0, 0, stream);
}

int vitta_wmsa_rel_bwd_bf16_io(const void* d_qkv_, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                               const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                               float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                               const void* d_out_, const void* d_dout_, const float* d_lse, float* d_delta, void* d_dqkv_,
                               float* d_dtable, float* d_dtable_ws, int64_t dtable_ws_bytes, int32_t io_bf16, void* stream) {
  const float* d_qkv = static_cast<const float*>(d_qkv_);
  const float* d_out = static_cast<const float*>(d_out_);
  const float* d_dout = static_cast<const float*>(d_dout_);
  float* d_dqkv = static_cast<float*>(d_dqkv_);
  const Args a{d_qkv, d_table, d_code, d_region, T, code_off, d_region ? nW : 1, B_, N, nH, scale,
               RowMap{d_rowmap, d_rowmap ? map_windows : 1, tokens_per_sample}, io_bf16 ? 1 : 0};
  if (!d_out || !d_dout || !d_lse || !d_delta || !d_dqkv) return VITTA_ERR_INVALID_ARG;
  const int rc = check(a, head_dim, d_rowmap, map_windows, tokens_per_sample);
  if (rc != VITTA_OK) return rc;
  if (misaligned(d_qkv, d_out, d_dout, d_dqkv)) return VITTA_ERR_INVALID_ARG;
  const int nt = (N + 15) / 16, qs = pick_split(B_ * nH, nt, TH_BWD / 64);
  const size_t l1 = lds_bytes(nt, (size_t)16 * nt * RP, 0, T), l2 = lds_bytes(nt, (size_t)16 * nt * RP, 2 * 16 * nt, T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool reg = a.region != nullptr, tail = (N % 16) != 0;
  // VITTA_WMSA_BF16_BWD: "one" / "two" force a form (tests, A/B); default: one pass wherever a workgroup per pair fills the chip
  const char* form = std::getenv("VITTA_WMSA_BF16_BWD");
  const bool force_one = form && form[0] == 'o', force_two = form && form[0] == 't';
  int nchunks1 = 0;
  const bool dtab = d_dtable != nullptr;  // the table gradient exists in the one-pass kernel only
  // with a workspace of B_ x nH x T floats the pairs' columns leave as plain stores and ONE reduce launch adds them (else: global atomics)
  float* const ws = (dtab && d_dtable_ws && dtable_ws_bytes >= (int64_t)vitta_wmsa_bf16_dtable_workspace_bytes(B_, nH, T)) ? d_dtable_ws : nullptr;
  // chunkings: plain (gradients only), fused (gradients + the 64-bit table column in one launch), table pass alone (no dQ tile)
  int nc_plain = 0, nc_fused = 0, nc_tab = 0;
  const int qc_plain = bwd1_chunks(nt, T, false, &nc_plain);
  const int qc_fused = dtab ? bwd1_chunks(nt, T, true, &nc_fused) : 0;
  const int qc_tab = (dtab && qc_fused <= 0) ? bwd1_chunks(nt, T, true, &nc_tab, false) : 0;
  (void)nchunks1;
  if (dtab && ((qc_fused <= 0 && (qc_tab <= 0 || qc_plain <= 0)) || nt > NW1 * KT_MAX)) return VITTA_ERR_UNSUPPORTED;  // (vitta_wmsa_bf16_dtable_supported)
  if ((dtab || (!force_two && (qs == 1 || force_one))) && qc_plain > 0 && nt <= NW1 * KT_MAX) {  // one workgroup per (window, head): the one-pass kernel
#define WMSA_BWD1_KT(R, TL, DT, GR, KTV, QC, NC, LF, DTP, WSP)                                                                        \
  do {                                                                                                                                \
    if (!set_lds(wmsa_bf16_bwd_fused_kernel<R, TL, DT, GR, KTV>, LF)) return VITTA_ERR_LAUNCH;                                          \
    VITTA_LAUNCH((wmsa_bf16_bwd_fused_kernel<R, TL, DT, GR, KTV>), dim3(1, nH, (unsigned)B_), dim3(TH_BWD1), LF, st, a, d_out, d_dout,  \
                 d_lse, d_delta, d_dqkv, QC, NC, DTP, WSP);                                                                           \
  } while (0)
  // key tiles per wave (compile time: the first KT - 1 run unconditionally): <= 32 tiles four, else seven
#define WMSA_BWD1(R, TL, DT, GR, QC, NC, LF, DTP, WSP)                                                                                \
  do {                                                                                                                                \
    if (nt <= NW1 * 4) WMSA_BWD1_KT(R, TL, DT, GR, 4, QC, NC, LF, DTP, WSP);                                                            \
    else WMSA_BWD1_KT(R, TL, DT, GR, KT_MAX, QC, NC, LF, DTP, WSP);                                                                     \
  } while (0)
#define WMSA_BWD1_ALL(DT, GR, QC, NC, LF, DTP, WSP)                                                                                   \
  do {                                                                                                                                \
    if (reg) { if (tail) WMSA_BWD1(true, true, DT, GR, QC, NC, LF, DTP, WSP); else WMSA_BWD1(true, false, DT, GR, QC, NC, LF, DTP, WSP); } \
    else { if (tail) WMSA_BWD1(false, true, DT, GR, QC, NC, LF, DTP, WSP); else WMSA_BWD1(false, false, DT, GR, QC, NC, LF, DTP, WSP); } \
  } while (0)
    if (dtab && qc_fused > 0) {  // gradients and table column in one launch
      const size_t lf = bwd1_lds_bytes(nt, qc_fused, T, true);
      WMSA_BWD1_ALL(true, true, qc_fused, nc_fused, lf, d_dtable, ws);
    } else {
      const size_t lf0 = bwd1_lds_bytes(nt, qc_plain, T, false);
      WMSA_BWD1_ALL(false, true, qc_plain, nc_plain, lf0, nullptr, nullptr);
      if (dtab) {  // ... and the table pass of every pair behind it (784-token windows)
        const size_t lft = bwd1_lds_bytes(nt, qc_tab, T, true, false);
        WMSA_BWD1_ALL(true, false, qc_tab, nc_tab, lft, d_dtable, ws);
      }
    }
#undef WMSA_BWD1_ALL
#undef WMSA_BWD1
#undef WMSA_BWD1_KT
    if (ws) {
      const int segs = (int)(B_ >= 64 ? 8 : B_ >= 8 ? 2 : 1), seg = (int)((B_ + segs - 1) / segs);
      VITTA_LAUNCH(dtable_reduce_kernel, dim3((unsigned)(((int64_t)nH * T + 255) / 256), (unsigned)segs), dim3(256), 0, st, ws, d_dtable, B_,
                   nH, T, seg);
    }
    return VITTA_OK;
  }
#define WMSA_BWD(R, TL)                                                                                                               \
  do {                                                                                                                                \
    if (!set_lds(wmsa_bf16_bwd_dq_kernel<R, TL>, l1) || !set_lds(wmsa_bf16_bwd_dkv_kernel<R, TL>, l2)) return VITTA_ERR_LAUNCH;         \
    VITTA_LAUNCH((wmsa_bf16_bwd_dq_kernel<R, TL>), dim3(qs, nH, (unsigned)B_), dim3(TH_BWD), l1, st, a, d_out, d_dout, d_lse, d_delta,  \
                 d_dqkv);                                                                                                             \
    VITTA_LAUNCH((wmsa_bf16_bwd_dkv_kernel<R, TL>), dim3(qs, nH, (unsigned)B_), dim3(TH_BWD), l2, st, a, d_dout, d_lse, d_delta, d_dqkv); \
  } while (0)
  if (reg) { if (tail) WMSA_BWD(true, true); else WMSA_BWD(true, false); }
  else { if (tail) WMSA_BWD(false, true); else WMSA_BWD(false, false); }
#undef WMSA_BWD
  return VITTA_OK;
}

}  // extern "C"
