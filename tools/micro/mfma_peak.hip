// Sustained v_mfma_f32_16x16x32_bf16 rate and shader clock with operands in registers (no memory traffic): the practical MFMA
// ceiling of this device under its power limit. hipcc --offload-arch=gfx950 -O3 -w mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_fp8_loop(float* sink, long* clocks, int iters, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  i32x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {     // random e4m3 codes with the exponent kept below NaN/overflow range
      x = x * 1664525u + 1013904223u; a[i][e] = (int)(x & 0x6f6f6f6fu);
      x = x * 1664525u + 1013904223u; b[i][e] = (int)(x & 0x6f6f6f6fu);
    }
  f32x4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
  const long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  const long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
#endif
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* sink, long* clocks, int iters, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
  // random-ish bf16 operands (data toggling matters for power)
  unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      x = x * 1664525u + 1013904223u; a[i][e] = (__bf16)(((int)(x >> 16) & 0xff) / 128.0f - 1.0f);
      x = x * 1664525u + 1013904223u; b[i][e] = (__bf16)(((int)(x >> 16) & 0xff) / 128.0f - 1.0f);
    }
  f32x4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
  const long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  const long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
#endif
}

int main() {
  float* sink; long* clocks;
  hipMalloc(&sink, 64); hipMalloc(&clocks, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int waves_per_simd : {1, 2}) {
      const int threads = waves_per_simd * 256, iters = 40000, wgs = 256;
      constexpr int NACC = 16;
      hipLaunchKernelGGL((mfma_loop<NACC>), dim3(wgs), dim3(threads), 0, 0, sink, clocks, 100, 1u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL((mfma_loop<NACC>), dim3(wgs), dim3(threads), 0, 0, sink, clocks, iters, 7u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long h[2]; hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
      const double flops = (double)wgs * (threads / 64) * iters * NACC * (16.0 * 16 * 32 * 2);
      printf("waves/SIMD %d: %7.1f TFLOP/s over %.2f ms; shader clock %.0f MHz (clock64 / wall_clock64 @100 MHz)\n", waves_per_simd,
             flops / ms / 1e9, ms, (double)h[0] / (double)h[1] * 100.0);
    }
  for (int rep = 0; rep < 3; ++rep)
    for (int waves_per_simd : {1, 2}) {
      const int threads = waves_per_simd * 256, iters = 20000, wgs = 256;
      constexpr int NACC = 16;
      hipLaunchKernelGGL((mfma_fp8_loop<NACC>), dim3(wgs), dim3(threads), 0, 0, sink, clocks, 100, 1u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL((mfma_fp8_loop<NACC>), dim3(wgs), dim3(threads), 0, 0, sink, clocks, iters, 7u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long h[2]; hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
      const double flops = (double)wgs * (threads / 64) * iters * NACC * (16.0 * 16 * 128 * 2);
      printf("fp8 16x16x128, waves/SIMD %d: %7.1f TFLOP/s over %.2f ms; shader clock %.0f MHz\n", waves_per_simd,
             flops / ms / 1e9, ms, (double)h[0] / (double)h[1] * 100.0);
    }
  return 0;
}
