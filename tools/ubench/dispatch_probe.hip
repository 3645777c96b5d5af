// Where does the dispatcher put workgroup i of a launch whose workgroups all fit at TWO per CU (64 KB of LDS each)?
// Every workgroup records (XCC, SE, SH/CU) from the hardware id registers and the shader clock at its start, then spins ~30 us
// so that all of them are resident together.  Output: for i < n / 2, whether workgroups i and i + n / 2 share a CU; the order in
// which the slots of one CU were filled.  hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_probe.hip -o /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
  lds[threadIdx.x] = acc;
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)(t0 & 0xffffffffu);
    out[4 * blockIdx.x + 3] = (unsigned)lds[0];
  }
}

int main() {
  for (int n : {512, 392, 784}) {
    unsigned* d;
    hipMalloc(&d, 16 * n);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 64 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(4 * n);
    hipMemcpy(h.data(), d, 16 * n, hipMemcpyDeviceToHost);
    // CU key: xcc (4 bits) | se_id | sh_id | cu_id of HW_ID
    std::map<unsigned, std::vector<int>> cus;
    for (int i = 0; i < n; ++i) {
      const unsigned hw = h[4 * i], xcc = h[4 * i + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    int pairs_half = 0, pairs_adjacent = 0, singles = 0, more = 0;
    for (auto& kv : cus) {
      auto& v = kv.second;
      if (v.size() == 1) ++singles;
      if (v.size() > 2) ++more;
      if (v.size() == 2) {
        const int dlt = v[1] - v[0];
        if (dlt == 256 || dlt == n / 2) ++pairs_half;
        if (dlt <= 8) ++pairs_adjacent;
      }
    }
    printf("n = %d: %zu distinct CUs, %d with one workgroup, %d with more than two; of the pairs: %d are (i, i + 256 or n / 2), %d are <= 8 apart\n", n,
           cus.size(), singles, more, pairs_half, pairs_adjacent);
    int shown = 0;
    for (auto& kv : cus) {
      if (shown++ >= 6) break;
      printf("   cu %05x:", kv.first);
      for (int i : kv.second) printf(" %d", i);
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}
