// Weight gradients of the trunk's convolutions at fp32 accuracy on the bf16 matrix pipe (SGD over all parameters, the reference's
// DEFAULT optimizer: corpus/basics.py:547-560; call sites as conv_wgrad.hip):
//
//   dW[k][c][tap] += sum_p dY[k][p] * A[c][src(p, tap)]          p over the N * Hg * Wg output positions
//
// conv_wgrad.hip does this on v_mfma_f32_32x32x2_f32 at 65 TF in the step.  Here BOTH operands are split into three bf16 terms in
// registers, per MFMA fragment, and a product is the six v_mfma_f32_32x32x16_bf16 of conv_b3.hip (same arithmetic, same error class:
// tests/test_gpu_conv.py::test_weight_gradient_matches_fp64_autograd holds both forms to one bound).  What makes it pay:
//   * a split costs 44 vector instructions per 32 x 16 fragment whoever consumes it, so a wave owns a 64 x 64 block of dW (2 x 2
//     accumulators): four fragment splits (176 instructions) feed 24 MFMAs (768 matrix-pipe cycles) per 16 pixels -- with 32 x 32
//     per wave the vector ALU, not the matrix pipe, would set the pace;
//   * both operands are channel-major planes, i.e. the REDUCTION axis (pixels) is the contiguous one: a fragment (one row, eight
//     consecutive pixels per lane) is 32 contiguous bytes of a row.  A stage = [rows][32 pixels] fp32 per operand, filled by LDS-DMA
//     (16 bytes per lane: 8 rows per instruction; the 16-byte chunks of a row are XOR-ed with (row / 2) % 8 on the SOURCE side so
//     that the 16 rows a ds_read_b128 serves per cycle fall into 16 different bank groups); gathered taps (3x3, stride 2): 4 bytes
//     per lane, two rows per instruction, the source pixel from reciprocal-multiply index arithmetic (no tables);
//   * workgroup = 4 waves as 2 x 2 (128 x 128 of dW), 1 x 4 (K = 64) or 4 x 1 (C = 64); two stages (64 KB): two workgroups per CU;
//     the pipeline is conv_b3.hip's pointwise form (one barrier per 32-pixel slab, in its middle);
//   * work = (tap, tile) x Z pixel ranges; a workgroup leaves its partial tile (row-major [rows k][columns c]) in the workspace and
//     wgrad_b3_reduce_kernel adds the Z partials to dW in the parameter's own [K][C][kh][kw] layout, in a fixed order.
#include <hip/hip_runtime.h>

#include "conv_common.h"

using namespace vitta;
using namespace vitta_conv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int WR, int WC, bool GATHER>
__global__ __launch_bounds__(256, 2) void wgrad_b3_kernel(const WgB3Plan a) {
  constexpr int TM = 64 * WR, TN = 64 * WC, BP = 32, NB = 2;
  constexpr int A_BYTES = TM * BP * 4, B_BYTES = TN * BP * 4, STAGE = A_BYTES + B_BYTES;
  constexpr int NA_I = TM / 8;                      // LDS-DMA instructions per slab for dY (8 rows each)
  constexpr int NB_I = GATHER ? TN / 2 : TN / 8;    // ... for the activations (gathered: 2 rows each)
  constexpr int PER = (NA_I + NB_I) / 4;            // per wave
  static_assert(WR * WC == 4 && NA_I % 8 == 0 && NB_I % 4 == 0, "four waves; dY instructions in groups of eight (swizzle parity per wave)");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  // ---- which (tap, tile) and which pixel range ------------------------------------------------------------------------------
  const int b = xcd_remap(blockIdx.x, a.nwg);
  const int tile = fdiv(b, a.d_Z), z = b - tile * a.Z;
  const int ct = tile % a.nCt, r_ = tile / a.nCt, kt = r_ % a.nKt, tap = r_ / a.nKt;
  const int k0 = kt * TM, c0 = ct * TN;
  const int s0 = (int)(((int64_t)a.nslab * z) / a.Z), s1 = (int)(((int64_t)a.nslab * (z + 1)) / a.Z);
  const int S = s1 - s0;
  const int P = a.P;
  const int yrow = P * 4, xrow = (int)(a.xP * 4);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.K * yrow, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.C * xrow, 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- LDS-DMA side ------------------------------------------------------------------------------------------------------
  // 16-byte form: instruction i of an operand = rows 8 i .. 8 i + 7; lane -> row 8 i + lane / 8, LDS chunk slot lane % 8, which
  // holds the row's global chunk slot ^ ((row / 2) % 8) = slot ^ ((4 i + lane / 16) % 8): two lane offsets, for even and odd i
  const int rin = lane >> 3, slot = lane & 7;
  int chunk16[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) chunk16[par] = slot ^ ((4 * par + (rin >> 1)) & 7);
  // gathered form (activations only): instruction i = rows 2 i, 2 i + 1; lane -> row 2 i + lane / 32, pixel slot q = lane % 32 of the
  // LDS row, which holds pixel q ^ (4 (i % 8)) of the slab: the lane's own pixel is q, the others come by lane exchange
  const int q = lane & 31;
  const int tdh = a.dh[tap], tdw = a.dw[tap];
  // instruction j = 4 u + wave of a slab belongs to wave `wave` (u = 0 .. PER - 1): j < NA_I (dY) <=> u < NA_I / 4, and since NA_I is
  // a multiple of 8 the swizzle of a wave's gathered instructions takes two values only, f = wave (u even) and wave + 4 (u odd)
  int gvoff[2] = {0, 0};  // byte offset of the source pixel under those two swizzles (or out of range)
  auto gather_offsets = [&](int p0) __attribute__((always_inline)) {
    if constexpr (GATHER) {
      const int p = p0 + q;
      const int pp = p < P ? p : 0;
      const int n = fdiv(pp, a.d_hw), r = pp - n * a.HgWg, gi = fdiv(r, a.d_w), gj = r - gi * a.Wg;
      const int sh = gi * a.sstride + tdh, sw = gj * a.sstride + tdw;
      const bool ok = p < P && (unsigned)sh < (unsigned)a.Hs && (unsigned)sw < (unsigned)a.Ws;
      const int own = ok ? ((n * a.Hs + sh) * a.Ws + sw) * 4 + (lane >> 5) * xrow : OOB;
      // (the row part (lane / 32) x xrow travels with the exchange: partner lanes sit in the same half)
      gvoff[0] = __shfl_xor(own, 4 * wave, 64);
      gvoff[1] = __shfl_xor(own, 4 * (wave + 4), 64);
    }
  };
  int q_s = s0;  // next slab to request (clamped to the range's last: the tail re-requests it into a stage nobody reads again)
  auto request = [&](int stage) __attribute__((always_inline)) {
    const int p0 = q_s * BP;
    const bool tail = p0 + BP > P;  // wave-uniform: chunks past the row's end must read zero, not the next row
    unsigned char* base = lds + stage * STAGE;
    if constexpr (GATHER) gather_offsets(p0);
    const int ch = chunk16[wave & 1];  // (4 u + wave) % 2
    const bool dead = tail && p0 + 4 * ch >= P;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = 4 * u + wave;
      if (u < NA_I / 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr)(base + j * 1024), 16, dead ? OOB : rin * yrow + ch * 16,
                                                 (k0 + 8 * j) * yrow + p0 * 4, 0, 0);
      } else if constexpr (GATHER) {
        const int i = j - NA_I;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(base + A_BYTES + i * 256), 4, gvoff[u & 1], (c0 + 2 * i) * xrow, 0, 0);
      } else {
        const int i = j - NA_I;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(base + A_BYTES + i * 1024), 16, dead ? OOB : rin * xrow + ch * 16,
                                                 (c0 + 8 * i) * xrow + p0 * 4, 0, 0);
      }
    }
    q_s += (q_s + 1 < s1) ? 1 : 0;
  };

  // ---- multiplying side ----------------------------------------------------------------------------------------------------
  // fragment of row `row`, 16-pixel k-step kk: the lane's eight pixels 16 kk + 8 lk .. + 7 = chunks 4 kk + 2 lk, + 1
  int a_off[2][2], b_off[2][2];  // [block][k-step]: byte offset of the first chunk; the second is its XOR partner (chunk ^ 1)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ra = 64 * wr + 32 * x + li, rb = 64 * wc + 32 * x + li;
      a_off[x][kk] = ra * 128 + (((4 * kk + 2 * lk) ^ ((ra >> 1) & 7)) << 4);
      b_off[x][kk] = A_BYTES + rb * 128 + (((4 * kk + 2 * lk) ^ ((rb >> 1) & 7)) << 4);
    }
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;

  struct Raw {
    f32x4 a[2][2], b[2][2];  // [block][half of the eight pixels]
  };
  auto read_raw = [&](const unsigned char* st_, int kk, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      r.a[x][0] = *reinterpret_cast<const f32x4*>(st_ + a_off[x][kk]);
      r.a[x][1] = *reinterpret_cast<const f32x4*>(st_ + (a_off[x][kk] ^ 16));
      r.b[x][0] = *reinterpret_cast<const f32x4*>(st_ + b_off[x][kk]);
      r.b[x][1] = *reinterpret_cast<const f32x4*>(st_ + (b_off[x][kk] ^ 16));
    }
  };
  struct Frag {
    bf16x8 a[2][3], b[2][3];
  };
  auto split8 = [&](const f32x4& lo4, const f32x4& hi4, bf16x8 (&out)[3]) __attribute__((always_inline)) {
    u32x4 sp[3];
    const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h_, m_, l_;
      split2(v[2 * j], v[2 * j + 1], h_, m_, l_);
      sp[0][j] = h_;
      sp[1][j] = m_;
      sp[2][j] = l_;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) out[p] = __builtin_bit_cast(bf16x8, sp[p]);
  };
  auto split_all = [&](const Raw& r, Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      split8(r.a[x][0], r.a[x][1], f.a[x]);
      split8(r.b[x][0], r.b[x][1], f.b[x]);
    }
  };
  // six products per accumulator, small terms first (conv_b3.hip)
  auto mfma24 = [&](const Frag& f) __attribute__((always_inline)) {
#define WG_MFMA(PA, PB)                            \
  _Pragma("unroll") for (int x = 0; x < 2; ++x)    \
  _Pragma("unroll") for (int y = 0; y < 2; ++y)    \
    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[x][PA], f.b[y][PB], acc[x][y], 0, 0, 0)
    WG_MFMA(2, 0);
    WG_MFMA(0, 2);
    WG_MFMA(1, 1);
    WG_MFMA(1, 0);
    WG_MFMA(0, 1);
    WG_MFMA(0, 0);
#undef WG_MFMA
  };
  // 24 MFMAs with the NEXT k-step's eight operand reads in front and its splits (176 vector instructions) between them
  auto interleave = [&]() __attribute__((always_inline)) {
    SGB(0x100, 8);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      SGB(0x008, 1);
      SGB(0x002, 8);
    }
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_all = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- pipeline: slab s in stage s % 2; one barrier per slab, between its two k-steps (conv_b3.hip, pointwise form) -----------
  request(0);
  request(1);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");  // slab s0 has landed
  barrier();
  Raw raw;
  Frag f0, f1;
  read_raw(lds, 0, raw);
  split_all(raw, f0);
  int st = 0;
  for (int s = 0; s + 1 < S; ++s) {
    const unsigned char* cur = lds + st * STAGE;
    const unsigned char* nxt = lds + (st ^ 1) * STAGE;
    read_raw(cur, 1, raw);
    split_all(raw, f1);
    mfma24(f0);
    interleave();
    wait_all();   // the other stage (slab s + 1) has landed
    barrier();    // ... for every wave, and every wave has the rest of this stage in registers
    request(st);  // slab s + 2 into the stage slab s has left
    read_raw(nxt, 0, raw);
    split_all(raw, f0);
    mfma24(f1);
    interleave();
    st ^= 1;
  }
  {
    read_raw(lds + st * STAGE, 1, raw);
    split_all(raw, f1);
    mfma24(f0);
    interleave();
    mfma24(f1);
  }
  wait_all();  // the tail's surplus requests: nothing may land in LDS that the next workgroup of this CU owns

  // ---- partial tile, row-major [TM][TN]: register v of block (x, y) = row 8 (v / 4) + 4 lk + v % 4, column li -----------------
  float* part = a.partials + ((int64_t)tile * a.Z + z) * (TM * TN);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int col = 64 * wc + 32 * y + li;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = 64 * wr + 32 * x + 8 * (v >> 2) + 4 * lk + (v & 3);
        part[row * TN + col] = acc[x][y][v];
      }
    }
}
#undef SGB

// dW[k][c][wt[tap]] += sum_z partial[(tile, z)][k - k0][c - c0]: one thread per (tap, k, c), lanes along c.  Where a tile has many
// partials (the early layers: one or two tiles, hundreds of pixel ranges) the z list is cut into gridDim.y chunks, each adding its
// share with one atomic per weight -- a thread walking 256 partials alone is 256 dependent loads
struct WgB3Reduce {
  WgB3Plan p[4];
  int n;
  int block_end[4];  // running count of 256-thread blocks
  int zc[4];         // z chunks of entry i (<= gridDim.y)
};
__global__ __launch_bounds__(256) void wgrad_b3_reduce_kernel(const WgB3Reduce m) {
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.block_end[i]) ++i;
  const WgB3Plan& a = m.p[i];
  const int zc = m.zc[i], zy = (int)blockIdx.y;
  if (zy >= zc) return;
  const int64_t e = (int64_t)((int)blockIdx.x - (i ? m.block_end[i - 1] : 0)) * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.ntaps * a.K * a.C;
  if (e >= total) return;
  const int c = (int)(e % a.C);
  const int64_t r = e / a.C;
  const int k = (int)(r % a.K), tap = (int)(r / a.K);
  const int kt = k / a.TM, ct = c / a.TN;
  const int tile = (tap * a.nKt + kt) * a.nCt + ct;
  const float* src = a.partials + (int64_t)tile * a.Z * (a.TM * a.TN) + (k - kt * a.TM) * a.TN + (c - ct * a.TN);
  const int z0 = (int)((int64_t)a.Z * zy / zc), z1 = (int)((int64_t)a.Z * (zy + 1) / zc);
  float s = 0.f;
  for (int z = z0; z < z1; ++z) s += src[(int64_t)z * (a.TM * a.TN)];
  float* dst = a.grad_w + ((int64_t)k * a.C + c) * a.wtaps + a.wt[tap];
  if (zc > 1) atomicAdd(dst, s);
  else *dst += s;
}

template <int WR, int WC, bool GATHER>
int launch_shape(const WgB3Plan& p, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (64 * WR + 64 * WC) * 32 * 4;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_b3_kernel<WR, WC, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return VITTA_ERR_LAUNCH;
    raised = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL((wgrad_b3_kernel<WR, WC, GATHER>), dim3((unsigned)p.nwg), dim3(256), lds, st, p);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace

namespace vitta_conv {

int wgrad_b3_plan(const vitta_wgrad_desc& d, int64_t xP, int P, WgB3Plan& p) {
  if (!(d.flags & VITTA_WGRAD_B3) || (d.flags & VITTA_CONV_PRO_BN_RELU) || !d.workspace) return VITTA_ERR_UNSUPPORTED;
  if (d.C % 64 || d.K % 64) return VITTA_ERR_UNSUPPORTED;
  int wr, wc;
  if (d.K % 128 == 0 && d.C % 128 == 0) wr = 2, wc = 2;
  else if (d.K % 64 == 0 && d.K % 128 != 0 && d.C % 256 == 0) wr = 1, wc = 4;
  else if (d.C % 64 == 0 && d.C % 128 != 0 && d.K % 256 == 0) wr = 4, wc = 1;
  else return VITTA_ERR_UNSUPPORTED;  // (64 x 64 and the like: conv_wgrad.hip)
  p.x = d.x;
  p.dy = d.dy;
  p.grad_w = d.grad_w;
  p.C = d.C;
  p.K = d.K;
  p.P = P;
  p.xP = xP;
  p.TM = 64 * wr;
  p.TN = 64 * wc;
  p.nKt = d.K / p.TM;
  p.nCt = d.C / p.TN;
  p.ntaps = d.ntaps;
  p.wtaps = d.wtaps;
  p.nslab = (P + 31) / 32;
  p.Hs = d.Hs;
  p.Ws = d.Ws;
  p.Wg = d.Wg;
  p.HgWg = d.Hg * d.Wg;
  p.sstride = d.sstride;
  p.d_hw = make_fastdiv(p.HgWg);
  p.d_w = make_fastdiv(d.Wg);
  for (int t = 0; t < VITTA_CONV_MAX_TAPS; ++t) {
    p.dh[t] = t < d.ntaps ? d.dh[t] : 0;
    p.dw[t] = t < d.ntaps ? d.dw[t] : 0;
    p.wt[t] = t < d.ntaps ? d.wt[t] : 0;
  }
  p.gather = !(d.ntaps == 1 && d.sstride == 1 && d.dh[0] == 0 && d.dw[0] == 0 && d.Hg == d.Hs && d.Wg == d.Ws);
  const int64_t tiles = (int64_t)d.ntaps * p.nKt * p.nCt;
  const int64_t tile_bytes = (int64_t)p.TM * p.TN * 4;
  // pixel ranges: ONE round of at most two workgroups per CU (a second, thin round doubles the launch), at least four slabs per
  // range (a range pays a prologue and a 64 KB partial tile whatever its length), whatever the workspace holds
  int64_t Z = 512 / tiles;
  if (Z > p.nslab / 4) Z = p.nslab / 4;
  if (Z < 1 && tiles <= 2048) Z = 1;
  if (Z > d.workspace_bytes / (tiles * tile_bytes)) Z = d.workspace_bytes / (tiles * tile_bytes);
  if (Z < 1) return VITTA_ERR_UNSUPPORTED;
  p.Z = (int)Z;
  p.d_Z = make_fastdiv(p.Z);
  p.nwg = (int)(tiles * Z);
  p.partials = static_cast<float*>(d.workspace);
  p.wr = wr;
  p.wc = wc;
  return VITTA_OK;
}

int wgrad_b3_launch(const WgB3Plan& p, hipStream_t st) {
  if (p.wr == 2) return p.gather ? launch_shape<2, 2, true>(p, st) : launch_shape<2, 2, false>(p, st);
  if (p.wr == 1) return p.gather ? launch_shape<1, 4, true>(p, st) : launch_shape<1, 4, false>(p, st);
  return p.gather ? launch_shape<4, 1, true>(p, st) : launch_shape<4, 1, false>(p, st);
}

int wgrad_b3_reduce(const WgB3Plan* plans, int n, hipStream_t st) {
  if (n < 1 || n > 4) return VITTA_ERR_INVALID_ARG;
  WgB3Reduce m;
  m.n = n;
  int blocks = 0, zcmax = 1;
  for (int i = 0; i < 4; ++i) {
    m.zc[i] = 0;
    if (i < n) {
      m.p[i] = plans[i];
      blocks += (int)(((int64_t)plans[i].ntaps * plans[i].K * plans[i].C + 255) / 256);
      const int zc = (plans[i].Z + 7) / 8;  // at most eight partials per thread
      m.zc[i] = zc > 64 ? 64 : zc;
      zcmax = m.zc[i] > zcmax ? m.zc[i] : zcmax;
    }
    m.block_end[i] = blocks;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(wgrad_b3_reduce_kernel, dim3((unsigned)blocks, (unsigned)zcmax), dim3(256), 0, st, m);
  return hipGetLastError() == hipSuccess ? VITTA_OK : VITTA_ERR_LAUNCH;
}

}  // namespace vitta_conv
