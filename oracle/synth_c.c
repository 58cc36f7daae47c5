/* TEST INFRASTRUCTURE — C restatement of the synthetic-tensor generator (same arithmetic as oracle/synth.py and
 * `synth_value` in bridgelang_amd/csrc/glue.hip), used only to build full-size (7.5 B element) oracle checkpoints in
 * minutes instead of half an hour. Integer hash + one fp32 multiply + one fp32 add (compiled with -ffp-contract=off: no
 * FMA) + round-to-nearest-even to bf16, so the bits equal the numpy and the device versions.
 * No reference counterpart (the reference loads real checkpoints; none exists offline — SURVEY.md §8c). */
#include <stdint.h>
#include <string.h>

static inline uint32_t mix32(uint32_t x) {          /* lowbias32 */
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

static inline uint16_t f2bf_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   /* NaN stays NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

/* out[i] = bf16(mean + (irwin_hall(start + i) - 131070) * scale), i < n; index arithmetic is mod 2^32 as on the device */
void bl_oracle_synth_bf16(uint16_t* out, int64_t n, uint32_t seed32, float mean, float scale, int64_t start) {
  const uint32_t sm = mix32(seed32);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t idx = (uint32_t)(start + i);
    const uint32_t h1 = mix32(idx ^ sm);
    const uint32_t h2 = mix32(h1 + 0x9E3779B9u);
    const int32_t s = (int32_t)((h1 & 0xFFFFu) + (h1 >> 16) + (h2 & 0xFFFFu) + (h2 >> 16)) - 131070;
    const float prod = (float)s * scale;
    out[i] = f2bf_rne(mean + prod);
  }
}
