// LDS atomic rate on gfx950 (round 6): ds_add_f32 against ds_add_u32 / ds_add_u64 / plain read-modify-write, 8 waves per workgroup hammering one
// 21 KB column (the table-gradient binning of wmsa_bf16.hip), addresses with the locality of a score tile (16 consecutive words per
// 16-lane group, groups offset by ~4) or uniformly random.  hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int T = 5239, ITERS = 2000;

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(512) void probe(float* out, long long* cycles) {
  __shared__ float colf[T + 64];
  __shared__ unsigned long long coll[(MODE == 2) ? T + 64 : 1];
  unsigned* colu = reinterpret_cast<unsigned*>(colf);
  for (int i = threadIdx.x; i < T + 64; i += 512) {
    colf[i] = 0.f;
    if (MODE == 2) coll[i] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  unsigned rng = threadIdx.x * 2654435761u + 12345u;
  const long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    int idx;
    if (RANDOM) {
      rng = rng * 1664525u + 1013904223u;
      idx = (rng >> 8) % T;
    } else {
      const int base = ((it * 37 + wave * 611) % (T - 64));
      idx = base + (15 - i) + 4 * g;  // four groups of 16 consecutive words, offset by 4: up to four lanes per word
    }
    const float v = 1e-3f * (lane + 1);
    if (MODE == 0) atomicAdd(colf + idx, v);
    else if (MODE == 1) atomicAdd(colu + idx, (unsigned)(lane + 1));
    else if (MODE == 2) atomicAdd(coll + idx, (unsigned long long)(lane + 1));
    else colf[idx] += v;  // (racy: rate reference only)
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int k = threadIdx.x; k < T; k += 512) s += MODE == 2 ? (float)coll[k] : colf[k];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, bool RANDOM>
void run(const char* name) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE, RANDOM><<<256, 512>>>(out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, RANDOM><<<256, 512>>>(out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
  // 8 waves x ITERS instructions per workgroup (one workgroup per CU)
  printf("%-34s %8.1f us   %7.1f clocks per wave-instruction (8 waves sharing the LDS: x8 per CU = %7.1f)\n", name, ms * 1e3,
         (double)h[0] / ITERS, (double)h[0] / ITERS / 8.0);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, false>("ds_add_f32, tile-like addresses");
  run<1, false>("ds_add_u32, tile-like addresses");
  run<2, false>("ds_add_u64, tile-like addresses");
  run<3, false>("plain RMW,  tile-like addresses");
  run<0, true>("ds_add_f32, random addresses");
  run<1, true>("ds_add_u32, random addresses");
  run<2, true>("ds_add_u64, random addresses");
  run<3, true>("plain RMW,  random addresses");
  return 0;
}
