// What does an LDS-DMA piece (and a ds_read_b128) cost when it is issued inside a stream of independent MFMAs by the one
// wave a SIMD runs? hipcc --offload-arch=gfx950 -O3 -w mfma_dma_mix.hip -o mfma_dma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// per iteration: 64 MFMAs; DMA pieces and ds_reads spread evenly among them
template <int NDMA, int NREAD, int NW>
__global__ __launch_bounds__(NW * 64) void mix(const char* src, float* sink, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (blockIdx.x % 64) * 65536), 0, 65536u, 0x00020000);
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.01f * (lane + e + i)); b[i][e] = (__bf16)(0.02f * (lane - e + i)); }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
  const unsigned voff = lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (NDMA > 0 && m % (64 / (NDMA > 0 ? NDMA : 1)) == 0) {
        const int pc = m / (64 / (NDMA > 0 ? NDMA : 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + ((pc & 7) * NW + wave) * 1024), 16, voff, ((it * 8 + pc) & 63) * 1024, 0, 0);
      }
      if (NREAD > 0 && m % (64 / (NREAD > 0 ? NREAD : 1)) == 1) {
        const int rd = m / (64 / (NREAD > 0 ? NREAD : 1));
        a[rd & 3] = *(const bf16x8_t*)(smem + ((rd & 7) * NW + wave) * 1024 + lane * 16);
      }
      acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[(m >> 2) & 3], acc[m & 15], 0, 0, 0);
    }
    if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 1234.5f) sink[0] = s;
#endif
}

template <int NDMA, int NREAD, int NW>
void run(const char* src, float* sink) {
  const int iters = 4000, wgs = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<NDMA, NREAD, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NW * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mix<NDMA, NREAD, NW>), dim3(wgs), dim3(NW * 64), 8 * NW * 1024, 0, src, sink, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix<NDMA, NREAD, NW>), dim3(wgs), dim3(NW * 64), 8 * NW * 1024, 0, src, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * NW * iters * 64 * (16.0 * 16 * 32 * 2);
  printf("waves/CU %d, per 64 MFMAs: %2d LDS-DMA pieces, %2d ds_read_b128: %7.1f TFLOP/s\n", NW, NDMA, NREAD, flops / ms / 1e9);
}

int main() {
  char* src; float* sink;
  hipMalloc(&src, 64 * 65536); hipMalloc(&sink, 64); hipMemset(src, 1, 64 * 65536);
  run<0, 0, 4>(src, sink); run<4, 0, 4>(src, sink); run<8, 0, 4>(src, sink); run<16, 0, 4>(src, sink);
  run<0, 16, 4>(src, sink); run<8, 16, 4>(src, sink); run<16, 16, 4>(src, sink);
  run<0, 0, 8>(src, sink); run<8, 0, 8>(src, sink); run<8, 16, 8>(src, sink); run<8, 32, 8>(src, sink);
  return 0;
}
