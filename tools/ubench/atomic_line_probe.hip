// What do the pooled-sum atomics of a convolution epilogue cost?  G workgroups x 4 waves, every wave adds 64 values (two 32-lane
// halves = two channel blocks) into a [frames][64 channels] or a [64 channels][frames] float table; hit frame = workgroup / (G / frames).
//   layout 0: pool[frame][channel]  (a half-wave's 32 lanes = 128 contiguous bytes)
//   layout 1: pool[channel][frame]  (32 lanes 4 * frames bytes apart)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_line_probe.hip -o variants/atomic_line_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(float* pool, int frames, int per, int layout, int reps) {
  const int lane = threadIdx.x & 63, ch = lane & 31, half = lane >> 5;
  const int f = blockIdx.x / per;
  float v = 1.f + threadIdx.x * 1e-3f;
  for (int r = 0; r < reps; ++r) {
    const int c = 32 * half + ch;
    if (layout == 2) {  // 64-bit fixed-point sums (order-independent), frame-major
      atomicAdd(reinterpret_cast<unsigned long long*>(pool) + f * 64 + c, (unsigned long long)__float2ll_rn(v * 4294967296.f));
    } else {
      float* p = layout == 0 ? pool + f * 64 + c : pool + c * frames + f;
      atomicAdd(p, v);
    }
  }
}
__global__ void empty(float*) {}

int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  hipMemset(d, 0, 1 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int layout = 0; layout < 3; ++layout)
    for (int G : {392, 1568}) {
      const int frames = 16, per = (G + frames - 1) / frames;
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, frames, per, layout, 1);
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, frames, per, layout, 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(empty, dim3(G), dim3(256), 0, 0, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms0;
      hipEventElapsedTime(&ms0, e0, e1);
      printf("layout %d (%s), %d workgroups: %.2f us per launch (empty launch %.2f us)\n", layout, layout == 2 ? "[frame][channel] u64 fixed point" : layout ? "[channel][frame]" : "[frame][channel]", G,
             ms * 50.f, ms0 * 50.f);
    }
  return 0;
}
