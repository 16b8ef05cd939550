// Counter-based RNG for dropout masks and the RandomWeightedAverage alpha.
// Mirror of oracle/rng.py (same constants, same bit-exact definition): the reference draws
// these from TensorFlow's unseeded global RNG (gan_train_cwgangp_pixelnorm.py:223,289-301),
// so no bit pattern is pinned by the reference; both sides of the parity tests use this one.
#pragma once
#include <stdint.h>

#define RD_STREAM_D1 1u
#define RD_STREAM_ALPHA 5u
#define RD_DROP_THRESHOLD 0x400000u   // 0.25 * 2^24

#if defined(__HIPCC__)
#define RD_HD __host__ __device__
#else
#define RD_HD
#endif

RD_HD static inline uint32_t rd_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
RD_HD static inline uint32_t rd_make_key(uint64_t seed, uint32_t stream) {
  uint32_t lo = (uint32_t)(seed & 0xFFFFFFFFull), hi = (uint32_t)(seed >> 32);
  return rd_mix32(lo ^ rd_mix32(hi ^ 0x9E3779B9u)) + stream * 0x85EBCA6Bu;
}
RD_HD static inline uint32_t rd_bits(uint32_t key, uint32_t idx) { return rd_mix32(rd_mix32(idx) ^ key); }
RD_HD static inline float rd_uniform(uint32_t key, uint32_t idx) { return (float)(rd_bits(key, idx) >> 8) * (1.0f / 16777216.0f); }
// inverted dropout, rate 0.25: 0 or 1/0.75
RD_HD static inline float rd_drop_scale(uint32_t key, uint32_t idx) {
  return ((rd_bits(key, idx) >> 8) >= RD_DROP_THRESHOLD) ? (1.0f / 0.75f) : 0.0f;
}
