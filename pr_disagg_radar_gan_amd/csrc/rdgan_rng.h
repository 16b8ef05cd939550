// Counter-based RNG for dropout masks and the RandomWeightedAverage alpha.
// Mirror of oracle/rng.py (same constants, same bit-exact definition): the reference draws
// these from TensorFlow's unseeded global RNG (gan_train_cwgangp_pixelnorm.py:223,289-301),
// so no bit pattern is pinned by the reference; both sides of the parity tests use this one.
#pragma once
#include <stdint.h>

#define RD_STREAM_D1 1u
#define RD_STREAM_ALPHA 5u
#define RD_DROP_THRESHOLD 64u          // 0.25 * 2^8: a byte of the hash word decides one element

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
// inverted dropout, rate 0.25: 0 or 1/0.75.  ONE hash word decides the four elements idx & ~3 ... idx | 3 (a byte each: keep iff
// byte >= 64, P = 192/256): the epilogues handle four consecutive channels per lane, so the two rd_mix32 -- four quarter-rate
// v_mul_lo_u32 -- are paid once per four elements instead of once per element (round 3; the 24-bit-per-element form cost the
// first critic layer's forward 0.18 ms of pure VALU time at 6144 samples).
RD_HD static inline uint32_t rd_drop_word(uint32_t key, uint32_t idx) { return rd_bits(key, idx >> 2); }
RD_HD static inline int rd_drop_keep(uint32_t word, uint32_t e) { return ((word >> (8u * (e & 3u))) & 0xFFu) >= RD_DROP_THRESHOLD; }
RD_HD static inline float rd_drop_scale(uint32_t key, uint32_t idx) {
  return rd_drop_keep(rd_drop_word(key, idx), idx) ? (1.0f / 0.75f) : 0.0f;
}
