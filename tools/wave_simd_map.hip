// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits 5:4 = SIMD_ID on gfx9-family parts.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/wave_simd_map tools/wave_simd_map.hip && tools/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(int* out) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6;
  const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (int)hw;
}
int main() {
  int* d;
  const int nb = 512;
  hipMalloc(&d, nb * 8 * sizeof(int));
  for (int lds : {0, 131072}) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    probe<<<nb, 512, lds>>>(d);
    static int h[512 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    int same_w4 = 0, same_w1 = 0;
    for (int b = 0; b < nb; ++b) {
      int simd[8];
      for (int w = 0; w < 8; ++w) { simd[w] = (h[b * 8 + w] >> 4) & 3; hist[w][simd[w]]++; }
      bool a = true, c = true;
      for (int w = 0; w < 4; ++w) a &= simd[w] == simd[w + 4];
      for (int w = 0; w < 8; w += 2) c &= simd[w] == simd[w + 1];
      same_w4 += a; same_w1 += c;
    }
    printf("dynamic LDS %d: blocks with simd(w)==simd(w+4) for all w: %d / %d;  simd(2i)==simd(2i+1): %d / %d\n", lds, same_w4, nb, same_w1, nb);
    for (int b = 0; b < 4; ++b) {
      printf("  block %d:", b);
      for (int w = 0; w < 8; ++w) printf(" w%d->simd%d(hw=%04x)", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 0xffff);
      printf("\n");
    }
  }
  return 0;
}
