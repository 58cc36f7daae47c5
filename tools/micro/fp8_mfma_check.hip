// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales: operand lane map check with exact integer data.
// hipcc --offload-arch=gfx950 -O3 -w fp8_mfma_check.hip -o fp8_mfma_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// MODE 0: lane (r = l&15, g = l>>4) holds k = 32g .. 32g+31 of row r (natural). MODE 1: k = {16g..16g+15} ∪ {64+16g..}
template <int MODE>
__global__ void check(const uint8_t* A, const uint8_t* B, float* C, int scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8_t a, b;
  for (int w = 0; w < 8; ++w) {
    int av = 0, bv = 0;
    for (int e = 0; e < 4; ++e) {
      const int idx = w * 4 + e;                       // byte index within the lane's 32 bytes
      const int k = MODE == 0 ? 32 * g + idx : (idx < 16 ? 16 * g + idx : 64 + 16 * g + (idx - 16));
      av |= (int)A[r * 128 + k] << (8 * e);            // A[row r][k]
      bv |= (int)B[r * 128 + k] << (8 * e);            // B stored as [col r][k]
    }
    a[w] = av; b[w] = bv;
  }
  f32x4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];   // C/D: col = lane&15, row = (lane>>4)*4 + reg
#endif
}

static uint8_t enc_e4m3(int v) {   // small integers -8..8 exactly representable in OCP e4m3 (bias 7)
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0; int m = abs(v);
  int e = 0; while ((1 << (e + 1)) <= m) ++e;            // m in [2^e, 2^(e+1))
  int frac = ((m << 3) >> e) & 7;                         // 3 mantissa bits (exact for m <= 15 when low bits are 0)
  return s | ((e + 7) << 3) | frac;
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128]; int iA[16 * 128], iB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 17 - 8; iB[i] = rand() % 13 - 6; hA[i] = enc_e4m3(iA[i]); hB[i] = enc_e4m3(iB[i]); }
  uint8_t *dA, *dB; float* dC; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 1024);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode)
    for (int scale : {(int)0x7F7F7F7F, 0, (int)0x80808080}) {
      if (mode == 0) hipLaunchKernelGGL(check<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, scale);
      else hipLaunchKernelGGL(check<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, scale);
      float hC[256]; hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
      int bad = 0; double ratio = 0; int nz = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        long ref = 0; for (int k = 0; k < 128; ++k) ref += iA[i * 128 + k] * iB[j * 128 + k];
        if (hC[i * 16 + j] != (float)ref) ++bad;
        if (ref != 0) { ratio += hC[i * 16 + j] / (double)ref; ++nz; }
      }
      printf("mode %d scale 0x%08x: %d / 256 mismatches, mean got/ref = %g\n", mode, (unsigned)scale, bad, ratio / nz);
    }
  return 0;
}
