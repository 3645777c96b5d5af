// What a cold instruction-cache line costs a wave on gfx950, and whether the instruction cache survives from one launch of a
// kernel to the next: a block of straight-line code (REPT x 8-byte VALU instructions = REPT / 8 cache lines of 64 bytes) is run
// twice inside one launch by wave 0 of every workgroup (pass 0 cold or not, pass 1 warm), timed with s_memtime; the launch is
// repeated back to back.  hipcc --offload-arch=gfx950 -O3 tools/ubench/icache_probe.hip -o tools/ubench/icache_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#ifndef REPT
#define REPT 512  // 4 KB of code
#endif

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int passes) {
  float v = threadIdx.x;
  unsigned long long t[4] = {0, 0, 0, 0};
  for (int p = 0; p < passes; ++p) {
    const unsigned long long c0 = __builtin_readcyclecounter();
    asm volatile(".rept %1\n\tv_add_f32_e64 %0, %0, 1.0\n\t.endr" : "+v"(v) : "n"(REPT));
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (p < 4) t[p] = c1 - c0;
  }
  if (threadIdx.x == 0) {
    for (int p = 0; p < 4; ++p) out[blockIdx.x * 4 + p] = t[p];
  }
  if (v == -1.f) out[0] = 0;
}

int main() {
  const int grid = 512;
  unsigned long long* d;
  hipMalloc(&d, grid * 4 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(grid * 4);
  for (int launch = 0; launch < 4; ++launch) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 3);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0};
    unsigned long long mn[3] = {~0ull, ~0ull, ~0ull}, mx[3] = {0, 0, 0};
    for (int b = 0; b < grid; ++b)
      for (int p = 0; p < 3; ++p) {
        s[p] += h[b * 4 + p];
        mn[p] = h[b * 4 + p] < mn[p] ? h[b * 4 + p] : mn[p];
        mx[p] = h[b * 4 + p] > mx[p] ? h[b * 4 + p] : mx[p];
      }
    printf("launch %d: %d instructions (%d bytes, %d lines): pass 0 mean %.0f (min %llu max %llu)  pass 1 mean %.0f (min %llu)  pass 2 mean %.0f cycles\n",
           launch, REPT, REPT * 8, REPT / 8, s[0] / grid, mn[0], mx[0], s[1] / grid, mn[1], s[2] / grid);
  }
  // back to back without a host round trip in between (the step's launches follow each other on the stream)
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 3);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s0 = 0, s1 = 0;
  for (int b = 0; b < grid; ++b) {
    s0 += h[b * 4];
    s1 += h[b * 4 + 1];
  }
  printf("third of three back-to-back launches: pass 0 mean %.0f, pass 1 mean %.0f cycles\n", s0 / grid, s1 / grid);
  return 0;
}
