// Register-only MFMA rate of one MI355X with the chip's clock under load: what "peak" means for the GEMM kernels' roofline.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_ceiling tools/mfma_ceiling.hip && tools/mfma_ceiling
// Variants: v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16, 1 or 2 waves per SIMD, zero or random operands (the power
// a real kernel draws depends on the operand bits: MI355X_MICROARCH.md "DVFS give-back").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint4 va = src[(tid * 8 + i) & 65535], vb = src[(tid * 8 + 4 + i) & 65535];
    a[i] = __builtin_bit_cast(bf16x8, va);
    b[i] = __builtin_bit_cast(bf16x8, vb);
  }
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) r += acc[i][e];
  }
  if (r == 123.456f) out[tid] = r;
}

int main() {
  const int n = 65536;
  std::vector<unsigned> hz(n * 4, 0u), hr(n * 4);
  srand(1);
  for (auto& v : hr) {
    // two random bf16 in [-2, 2): sign, exponent 0x3F-0x40 range, random mantissa
    unsigned lo = ((rand() & 1) << 15) | ((0x7E + (rand() & 3)) << 7) | (rand() & 127);
    unsigned hi = ((rand() & 1) << 15) | ((0x7E + (rand() & 3)) << 7) | (rand() & 127);
    v = lo | (hi << 16);
  }
  uint4 *dz, *dr;
  float* out;
  hipMalloc(&dz, n * 16);
  hipMalloc(&dr, n * 16);
  hipMalloc(&out, 1 << 22);
  hipMemcpy(dz, hz.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(dr, hr.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  printf("%-10s %-8s %-7s %10s %10s\n", "mfma", "waves/SIMD", "data", "ms", "TFLOP/s");
  for (int shape : {16, 32})
    for (int wps : {1, 2})
      for (int rnd : {0, 1}) {
        const int blocks = 256 * wps;   // 256 threads = 4 waves = one wave per SIMD of a CU
        const uint4* src = rnd ? dr : dz;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (shape == 16) mfma_loop<16><<<blocks, 256>>>(src, out, iters);
          else mfma_loop<32><<<blocks, 256>>>(src, out, iters);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop_per_iter_wave = shape == 16 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16;
        const double tf = flop_per_iter_wave * iters * blocks * 4 / (ms * 1e-3) / 1e12;
        printf("%-10s %-8d %-7s %10.3f %10.1f\n", shape == 16 ? "16x16x32" : "32x32x16", wps, rnd ? "random" : "zero", ms, tf);
      }
  return 0;
}
