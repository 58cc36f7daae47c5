// Per-CU global->LDS (LDS-DMA) vs global->VGPR ingest rate on L2-resident data. hipcc --offload-arch=gfx950 -O3 ingest.hip -o ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE, int NW>   // MODE 0: LDS-DMA, 1: VGPR loads, 2: half/half
__global__ __launch_bounds__(NW * 64) void ingest(const char* src, uint32_t* sink, int iters, int bytes_per_wg) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (long)(blockIdx.x % 64) * bytes_per_wg;      // 64 distinct regions: L2-resident
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)bytes_per_wg, 0x00020000);
  u32x4_t accv = {0, 0, 0, 0};
  const int pieces = bytes_per_wg / 1024 / NW;       // pieces per wave per sweep
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int pc = 0; pc < pieces; ++pc) {
      const unsigned off = (unsigned)((pc * NW + wave) * 1024 + lane * 16);
      const bool dma = MODE == 0 || (MODE == 2 && (pc & 1));
      if (dma) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + ((pc & 7) * NW + wave) * 1024), 16, off, 0, 0, 0);
      else {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        accv ^= v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (accv[0] == 0x12345678 && accv[1] == 77) sink[0] = accv[2] + smem[lane];
#endif
}

template <int MODE, int NW>
void run(const char* name, const char* src, uint32_t* sink, int wgs) {
  const int bytes_per_wg = 64 * 1024, iters = 400;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NW * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest<MODE, NW>), dim3(wgs), dim3(NW * 64), 8 * NW * 1024, 0, src, sink, 10, bytes_per_wg);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((ingest<MODE, NW>), dim3(wgs), dim3(NW * 64), 8 * NW * 1024, 0, src, sink, iters, bytes_per_wg);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * bytes_per_wg * iters;
  printf("%-28s wgs=%4d waves/wg=%d: %8.1f GB/s total, %6.1f GB/s per WG\n", name, wgs, NW, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
}

int main() {
  char* src; uint32_t* sink;
  hipMalloc(&src, 64 * 64 * 1024); hipMalloc(&sink, 64);
  hipMemset(src, 1, 64 * 64 * 1024);
  for (int wgs : {256, 512}) {
    run<0, 4>("LDS-DMA b128", src, sink, wgs);
    run<0, 8>("LDS-DMA b128", src, sink, wgs);
    run<1, 4>("VGPR buffer_load b128", src, sink, wgs);
    run<1, 8>("VGPR buffer_load b128", src, sink, wgs);
    run<2, 8>("half DMA / half VGPR", src, sink, wgs);
  }
  return 0;
}
