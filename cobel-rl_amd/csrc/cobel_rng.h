// Counter-based random streams shared by every kernel (and the host-side tests).
// Philox-4x32-10 (Salmon et al., SC'11).  Layout, also restated in oracle/philox.py:
//   key = (seed lo, seed hi)   ctr = (block, sub, instance, stream)
// A stream is consumed through a draw counter c (one per instance and stream):
//   bounded integer  c : word (c & 3) of block (c >> 2)          -> mulhi32(word, n)
//   uniform double   c : words 2(c & 1), 2(c & 1) + 1 of block (c >> 1)
// so one Philox evaluation serves four integer draws / two doubles of consecutive counters and
// the kernels evaluate it once every few steps.  `sub` distinguishes the elements of a vector
// draw (the j-th index of a replay batch).  A draw is a pure function of (seed, global instance
// id, stream, counter, sub): results do not depend on how instances are sharded over GPUs or
// chunked over launches.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define COBEL_HD __host__ __device__ __forceinline__
#else
#define COBEL_HD static inline
#endif

struct cobel_u4 {
  uint32_t x, y, z, w;
};

COBEL_HD uint32_t cobel_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

COBEL_HD cobel_u4 cobel_philox(uint32_t index, uint32_t sub, uint32_t instance, uint32_t stream,
                               uint64_t seed) {
  uint32_t c0 = index, c1 = sub, c2 = instance, c3 = stream;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32), not a mul_hi / mul_lo pair
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  cobel_u4 o = {c0, c1, c2, c3};
  return o;
}

COBEL_HD uint32_t cobel_word(const cobel_u4& b, uint32_t k) {
  // two-level select (no comparisons against constants: with a wave-uniform k the compiler
  // otherwise builds a chain of scalar branches)
  const uint32_t lo = (k & 1u) ? b.y : b.x, hi = (k & 1u) ? b.w : b.z;
  return (k & 2u) ? hi : lo;
}

// k in [0, n): Lemire multiply-shift without rejection (bias <= n / 2^32).
COBEL_HD uint32_t cobel_bounded(uint32_t x, uint32_t n) { return cobel_mulhi32(x, n); }

// 53-bit double in [0, 1) from two words, NumPy's recipe.
COBEL_HD double cobel_u01(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// Whole draws by counter (slow path: one Philox evaluation per call).
COBEL_HD uint32_t cobel_draw_bounded(uint32_t counter, uint32_t sub, uint32_t instance,
                                     uint32_t stream, uint64_t seed, uint32_t n) {
  const cobel_u4 b = cobel_philox(counter >> 2, sub, instance, stream, seed);
  return cobel_bounded(cobel_word(b, counter & 3u), n);
}
COBEL_HD double cobel_draw_u01(uint32_t counter, uint32_t sub, uint32_t instance, uint32_t stream,
                               uint64_t seed) {
  const cobel_u4 b = cobel_philox(counter >> 1, sub, instance, stream, seed);
  return (counter & 1u) ? cobel_u01(b.z, b.w) : cobel_u01(b.x, b.y);
}
// The 53-bit integer K with u = K * 2^-53 (for integer threshold comparisons).
COBEL_HD uint64_t cobel_u53(uint32_t a, uint32_t b) {
  return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}
