// What costs matrix-pipe time in the convolution's slab loop?  The loop is rebuilt piece by piece around a dependent
// v_mfma_f32_32x32x2_f32 chain (3 workgroups per CU through a 49 KB dynamic LDS allocation, like conv.hip):
//   bit 0: LDS operand reads through a 4-deep register ring (two ds_read_b32 per MFMA)
//   bit 1: sched_barrier(0) on both sides of every MFMA
//   bit 2: one s_barrier per 16 MFMAs
//   bit 3: one global_load_dword per k-step (10 of 16), results stored to LDS after the barrier
//   bit 4: a few integer VALU instructions per k-step
//   bit 5: a uniform (scalar) branch per k-step
// hipcc --offload-arch=gfx950 -O3 mfma_loop_probe.hip -o mfma_loop_probe && ./mfma_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int slabs, int flagv) {
  extern __shared__ float sm[];  // [3][32][64] A + [3][32][64] B
  for (int i = threadIdx.x; i < 3 * 32 * 64 * 2; i += 256) sm[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5, wm = wave >> 1, wn = wave & 1;
  float af[4] = {1.f + lane, 1.f, 1.f, 1.f}, bf[4] = {2.f - lane, 1.f, 1.f, 1.f};
  const float* As = sm;
  const float* Bs = sm + 3 * 32 * 64;
  int r0 = 0, r1 = 1, r2 = 2;
  float stage[10];
  for (int u = 0; u < 10; ++u) stage[u] = 0.f;
  const float* gp = src + (size_t)blockIdx.x * 4096 + tid;
  int vjunk = tid;
  for (int s = 0; s < slabs; ++s) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (MODE & 1) {
        const int buf = (ks + 3 < 16) ? r0 : r1, kq = (ks + 3) & 15;
        af[(ks + 3) & 3] = As[buf * 2048 + (2 * kq + lk) * 64 + wm * 32 + li];
        bf[(ks + 3) & 3] = Bs[buf * 2048 + (2 * kq + lk) * 64 + wn * 32 + li];
      }
      if ((MODE & 4) && ks == 10) __syncthreads();
      if (MODE & 8) {
        if (ks < 10) stage[ks] = gp[(size_t)(s & 7) * 256 + ks * 32768];
        else if (ks - 10 < 6) {
          sm[r2 * 2048 + ((ks - 10) * 4 + (tid >> 6)) * 64 + (tid & 63)] = stage[ks - 10];
          if (ks - 10 < 4) sm[6144 + r2 * 2048 + ((ks - 10) * 4 + (tid >> 6)) * 64 + (tid & 63)] = stage[6 + ks - 10];
        }
      }
      if (MODE & 16) {
        vjunk = vjunk * 3 + ks;
        vjunk ^= (vjunk >> 3);
        vjunk += s;
      }
      if (MODE & 32) {
        if (flagv & (1 << ks)) vjunk += 7;
      }
      if (MODE & 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 3], bf[ks & 3], acc[a], 0, 0, 0);
      if (MODE & 2) __builtin_amdgcn_sched_barrier(0);
    }
    const int t = r0;
    r0 = r1;
    r1 = r2;
    r2 = t;
  }
  float sacc = (float)vjunk;
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) sacc += acc[a][v];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int MODE, int NACC>
void run(int blocks, int slabs, const float* src) {
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = 3 * 32 * 64 * 2 * 4 + 2048;
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), lds, 0, src, out, 4, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), lds, 0, src, out, slabs, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double fl = 2.0 * 32 * 32 * 2 * 16.0 * NACC * slabs * 4.0 * blocks;
  printf("mode %2d nacc %d blocks %5d slabs %d: %8.3f ms  %7.1f TF\n", MODE, NACC, blocks, slabs, ms, fl / ms / 1e9);
  hipFree(out);
}

// Variant: ring slots are compile-time (loop unrolled by 3: every LDS address is lane base + immediate), the global loads are
// buffer loads whose per-slab part travels in the SCALAR offset (no per-load VALU), optional s_setprio around the MFMA.
template <int MODE, int NACC, int PRIO>
__global__ __launch_bounds__(256) void k2(const float* __restrict__ src, float* out, int slabs, int flagv) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < 3 * 32 * 64 * 2; i += 256) sm[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5, wm = wave >> 1, wn = wave & 1;
  float af[4] = {1.f + lane, 1.f, 1.f, 1.f}, bf[4] = {2.f - lane, 1.f, 1.f, 1.f};
  const float* Al = sm + lk * 64 + wm * 32 + li;             // + buf * 2048 + 2 kq * 64
  const float* Bl = sm + 6144 + lk * 64 + wn * 32 + li;
  float* Sl = sm + (tid >> 6) * 64 + (tid & 63);              // store base
  float stage[10];
  for (int u = 0; u < 10; ++u) stage[u] = 0.f;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64 << 20, 0x00020000);
  const int voff = (blockIdx.x * 4096 + tid) * 4;
  int vjunk = tid;
#define SLAB2(R0, R1, R2)                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 16; ++ks) {                                                          \
    if (MODE & 1) {                                                                                            \
      const int buf = (ks + 3 < 16) ? R0 : R1, kq = (ks + 3) & 15;                                             \
      af[(ks + 3) & 3] = Al[buf * 2048 + 2 * kq * 64];                                                         \
      bf[(ks + 3) & 3] = Bl[buf * 2048 + 2 * kq * 64];                                                         \
    }                                                                                                          \
    if ((MODE & 4) && ks == 10) __syncthreads();                                                               \
    if (MODE & 8) {                                                                                            \
      if (ks < 10) stage[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + ks * 131072, 0)); \
      else if (ks - 10 < 6) {                                                                                  \
        Sl[R2 * 2048 + (ks - 10) * 256] = stage[ks - 10];                                                      \
        if (ks - 10 < 4) Sl[6144 + R2 * 2048 + (ks - 10) * 256] = stage[6 + ks - 10];                          \
      }                                                                                                        \
    }                                                                                                          \
    if (MODE & 16) {                                                                                           \
      vjunk = vjunk * 3 + ks;                                                                                  \
      vjunk ^= (vjunk >> 3);                                                                                   \
      vjunk += s;                                                                                              \
    }                                                                                                          \
    if (MODE & 2) __builtin_amdgcn_sched_barrier(0);                                                           \
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);                                                                \
    _Pragma("unroll") for (int a = 0; a < NACC; ++a)                                                            \
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 3], bf[ks & 3], acc[a], 0, 0, 0);                   \
    if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                   \
    if (MODE & 2) __builtin_amdgcn_sched_barrier(0);                                                           \
  }
  for (int s = 0; s < slabs; s += 3) {
    int soff = (s & 7) * 1024;
    SLAB2(0, 1, 2)
    soff += 1024;
    SLAB2(1, 2, 0)
    soff += 1024;
    SLAB2(2, 0, 1)
  }
  float sacc = (float)vjunk;
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) sacc += acc[a][v];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int MODE, int NACC, int PRIO>
void run2(int blocks, int slabs, const float* src) {
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k2<MODE, NACC, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = 3 * 32 * 64 * 2 * 4 + 2048;
  hipLaunchKernelGGL((k2<MODE, NACC, PRIO>), dim3(blocks), dim3(256), lds, 0, src, out, 6, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k2<MODE, NACC, PRIO>), dim3(blocks), dim3(256), lds, 0, src, out, slabs, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double fl = 2.0 * 32 * 32 * 2 * 16.0 * NACC * slabs * 4.0 * blocks;
  printf("k2 mode %2d nacc %d prio %d blocks %5d slabs %d: %8.3f ms  %7.1f TF\n", MODE, NACC, PRIO, blocks, slabs, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  float* src;
  hipMalloc(&src, (size_t)64 << 20);
  hipMemset(src, 0, (size_t)64 << 20);
  run<0, 1>(768, 2000, src);
  run<1, 1>(768, 2000, src);
  run<2, 1>(768, 2000, src);
  run<3, 1>(768, 2000, src);
  run<7, 1>(768, 2000, src);
  run<15, 1>(768, 2000, src);
  run<31, 1>(768, 2000, src);
  run<63, 1>(768, 2000, src);
  run<16, 1>(768, 2000, src);
  run<32, 1>(768, 2000, src);
  run<8, 1>(768, 2000, src);
  run<11, 1>(768, 2000, src);
  run<9, 1>(768, 2000, src);
  run<15, 2>(768, 1000, src);
  run<63, 2>(768, 1000, src);
  run2<7, 1, 0>(768, 1998, src);
  run2<15, 1, 0>(768, 1998, src);
  run2<15, 1, 1>(768, 1998, src);
  run2<15, 1, 3>(768, 1998, src);
  run2<31, 1, 0>(768, 1998, src);
  run2<31, 1, 1>(768, 1998, src);
  run2<31, 1, 3>(768, 1998, src);
  run2<15, 2, 0>(768, 999, src);
  run2<15, 2, 1>(768, 999, src);
  run2<13, 1, 0>(768, 1998, src);
  run2<15, 1, 0>(1024, 1998, src);
  return 0;
}
