// How fast does a one-round GEMM's C tile leave the chip, by store pattern? 256 workgroups x 512 threads each write a
// 288 x 256 bf16 tile of a 4608 x 4096 matrix (37.7 MB):
//   A  the tile epilogue's pattern after the 16-byte widening: a wave instruction = 16 rows x 64 contiguous bytes
//   B  row-contiguous: a wave instruction = 2 rows x 512 contiguous bytes (what an LDS-staged epilogue would issue)
//   C  the per-store form of rounds 1-3: a wave instruction = 16 rows x 32 bytes (8 bytes per lane)
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

template <int MODE>
__global__ __launch_bounds__(512) void store_tile(uint16_t* c, int ldc, int tiles_n) {
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3;
  const int lin = xcd * q + (blockIdx.x >> 3);
  const int tm = lin / tiles_n, tn = lin % tiles_n;
  const int m0 = tm * 288, n0 = tn * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, l15 = lane & 15, lg = lane >> 4;
  const u32x4_t v = {(uint32_t)tid, (uint32_t)blockIdx.x, 3u, 4u};
  if (MODE == 0) {
    const int dc = (lg & 1) * 16 + (lg >> 1) * 8;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
        *(u32x4_t*)(c + (long)(m0 + wm * 144 + j * 16 + l15) * ldc + n0 + wn * 64 + pr * 32 + dc) = v;
  } else if (MODE == 1) {
#pragma unroll
    for (int it = 0; it < 18; ++it) {
      const int row = (it * 8 + wave) * 2 + (lane >> 5);
      *(u32x4_t*)(c + (long)(m0 + row) * ldc + n0 + (lane & 31) * 8) = v;
    }
  } else {
    const u32x2_t v2 = {v[0], v[1]};
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(u32x2_t*)(c + (long)(m0 + wm * 144 + j * 16 + l15) * ldc + n0 + wn * 64 + i * 16 + lg * 4) = v2;
  }
}

template <int MODE>
float run(uint16_t* c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_tile<MODE>), dim3(256), dim3(512), 0, 0, c, 4096, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((store_tile<MODE>), dim3(256), dim3(512), 0, 0, c, 4096, 16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 50 * 1e3f;
}

int main() {
  uint16_t* c; hipMalloc(&c, 4608L * 4096 * 2);
  for (int rep = 0; rep < 2; ++rep) {
    const float a = run<0>(c), b = run<1>(c), d = run<2>(c);
    printf("37.7 MB tile store: 16 rows x 64 B per instruction %.1f us (%.2f TB/s) | 2 rows x 512 B %.1f us (%.2f TB/s) | 16 rows x 32 B %.1f us (%.2f TB/s)\n",
           a, 37.75 / a, b, 37.75 / b, d, 37.75 / d);
  }
  return 0;
}
