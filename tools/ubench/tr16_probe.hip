#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane k of a 16-lane group: row k / 4, columns 4 (k % 4) of a [4][16] block with row pitch 40
  const int kq = l & 15, g = l >> 4;
  const short* p = buf + (4 * g + (kq >> 2)) * 40 + 4 * (kq & 3);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  // expectation: lane (i, g) gets buf[(4 g + e) * 40 + i], e = 0..3
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int want = (4 * (l >> 4) + e) * 40 + (l & 15); if (h[l * 4 + e] != want) ++bad; }
  printf("mismatches %d; lane 1: %d %d %d %d (want 1 41 81 121); lane 17: %d %d %d %d (want 161 201 241 281)\n", bad, h[4], h[5], h[6], h[7], h[68], h[69], h[70], h[71]);
  return 0;
}
