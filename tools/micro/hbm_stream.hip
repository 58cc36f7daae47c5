// How fast can ONE compute unit pull a stream out of HBM, and what does the chip reach with W workgroups?  Every wave
// streams its own contiguous run of a 4 GiB buffer (>> 256 MB Infinity Cache, nothing is re-read) with D KiB of loads in
// flight, through non-temporal HBM->VGPR loads (the weight-streaming kernels' path) or LDS-DMA.
// hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int D, int NW, bool DMA>   // D = 1-KiB pieces per wave in the two-buffer ring (D/2 … D KiB in flight per wave)
__global__ __launch_bounds__(NW * 64) void stream(const u32x4_t* src, uint32_t* sink, int pieces_per_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = D / 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4_t* p = src + ((long)(blockIdx.x * NW + wave) * pieces_per_wave) * 64 + lane;
  u32x4_t acc = {0, 0, 0, 0};
  if constexpr (!DMA) {
    u32x4_t buf[2][H];
#pragma unroll
    for (int j = 0; j < H; ++j) buf[0][j] = __builtin_nontemporal_load(p + j * 64);
    for (int i = 0; i < pieces_per_wave; i += D) {
#pragma unroll
      for (int j = 0; j < H; ++j) buf[1][j] = __builtin_nontemporal_load(p + (long)min(i + H + j, pieces_per_wave - 1) * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < H; ++j) acc ^= buf[0][j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < H; ++j) buf[0][j] = __builtin_nontemporal_load(p + (long)min(i + D + j, pieces_per_wave - 1) * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < H; ++j) acc ^= buf[1][j];
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + ((long)(blockIdx.x * NW + wave) * pieces_per_wave) * 64), 0,
                                                                        (unsigned)pieces_per_wave * 1024u, 0x00020000);
    char* my = smem + wave * D * 1024;
#pragma unroll
    for (int j = 0; j < H; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(my + j * 1024), 16, lane * 16, j * 1024, 0, 0);
    for (int i = 0; i < pieces_per_wave; i += D) {
#pragma unroll
      for (int j = 0; j < H; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(my + (H + j) * 1024), 16, lane * 16, min(i + H + j, pieces_per_wave - 1) * 1024, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H) : "memory");
      acc[0] ^= *(const uint32_t*)(my + lane * 4);
#pragma unroll
      for (int j = 0; j < H; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(my + j * 1024), 16, lane * 16, min(i + D + j, pieces_per_wave - 1) * 1024, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H) : "memory");
      acc[1] ^= *(const uint32_t*)(my + H * 1024 + lane * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc[0] == 0x12345678 && acc[1] == 77 && acc[2] == 5 && acc[3] == 9) sink[0] = acc[2];
#endif
}

template <int D, int NW, bool DMA>
void run(const u32x4_t* src, uint32_t* sink, int wgs, long total_bytes) {
  const long per_wave = total_bytes / ((long)wgs * NW);
  const int pieces = (int)(per_wave / 1024 / D) * D;
  const int lds = DMA ? NW * D * 1024 : 0;
  if (DMA) hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<D, NW, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream<D, NW, DMA>), dim3(wgs), dim3(NW * 64), lds, 0, src, sink, pieces);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((stream<D, NW, DMA>), dim3(wgs), dim3(NW * 64), lds, 0, src, sink, pieces);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * NW * pieces * 1024.0;
  printf("%-8s wgs=%4d waves/wg=%d ring=%2d KiB/wave (%3d KiB/WG): %7.1f GB/s total, %6.1f GB/s per WG\n", DMA ? "LDS-DMA" : "VGPR nt", wgs, NW, D,
         D * NW, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
  fflush(stdout);
}

int main() {
  const long total = 4L << 30;
  u32x4_t* src; uint32_t* sink;
  if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, total);
  hipDeviceSynchronize();
  for (int wgs : {32, 128, 256, 512}) {
    const long bytes = wgs >= 128 ? (2L << 30) : (512L << 20);
    run<4, 8, false>(src, sink, wgs, bytes);
    run<8, 8, false>(src, sink, wgs, bytes);
    run<16, 8, false>(src, sink, wgs, bytes);
    run<32, 8, false>(src, sink, wgs, bytes);
    run<16, 4, false>(src, sink, wgs, bytes);
    run<32, 4, false>(src, sink, wgs, bytes);
    run<8, 8, true>(src, sink, wgs, bytes);
    run<16, 8, true>(src, sink, wgs, bytes);
  }
  return 0;
}
