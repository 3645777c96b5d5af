// Sustained v_mfma_f32_32x32x2_f32 rate on this box: pure MFMA loop (NACC accumulators per wave), optionally with the
// LDS operand reads of a GEMM k-loop.  hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_f32_peak && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float sm[2 * 32 * 128];
  for (int i = threadIdx.x; i < 2 * 32 * 128; i += 256) { unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; sm[i] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000) : 1.0f + i * 1e-6f; }
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
  const int lane = threadIdx.x & 63;
  float av = RANDOM ? sm[lane * 7] : 1.0f + lane, bv = RANDOM ? sm[lane * 11 + 3] : 2.0f - lane;
  const float* p = sm + (lane & 31) + (lane >> 5) * 128;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (LDS) {
        av = p[ks * 256];
        bv = p[ks * 256 + 64];
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int v = 0; v < 16; ++v) s += acc[a][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS, bool RANDOM>
void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k<NACC, LDS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<NACC, LDS, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double fl = 2.0 * 32 * 32 * 2 * 16.0 * NACC * iters * 4.0 * blocks;
  printf("random %d NACC %d LDS %d blocks %5d iters %d: %8.3f ms  %7.1f TF\n", (int)RANDOM, NACC, (int)LDS, blocks, iters, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<1, true, false>(768, 4000);
    run<1, true, true>(768, 4000);
    run<1, false, true>(768, 4000);
    run<4, true, true>(256, 4000);
    run<2, true, true>(512, 4000);
    run<1, true, true>(1024, 4000);
    run<1, true, true>(256, 16000);
  }
  return 0;
}
