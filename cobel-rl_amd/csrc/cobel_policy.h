// Epsilon-greedy action selection, restating policy/greedy.py:40-88 of the reference plus the
// numpy Generator.choice draw it ends in:
//     probs[allowed] = eps / n_allowed
//     probs[allowed] += (1 - eps) * ties / n_ties        ties = (max(v[allowed]) == v[allowed])
//     action = searchsorted(cumsum(probs) / cumsum(probs)[-1], u, side='right')
// The probabilities are float64 whatever the table dtype; ties are detected with exact equality
// in the table dtype (float32 here).  eps / n and (1 - eps) / n_ties are formed on the host in
// float64 (cobel_eps_consts) so the device only adds, divides (IEEE, correctly rounded) and
// compares.
#pragma once
#include <math.h>
#include <stdint.h>

struct cobel_eps_consts {
  double base[5];  // base[n]  = eps / n                 n = 1..4
  double bonus[5]; // bonus[t] = ((1 - eps) * 1.0) / t   t = 1..4
  // Unmasked case, per tie pattern t (bit a set = action a attains the maximum): the three
  // thresholds of the normalised CDF as integers, thr[t][k] = ceil(cdf_k * 2^53).  With
  // u = K * 2^-53 (K the 53-bit integer the draw is built from) cdf_k <= u  <=>  thr <= K, so
  // the selection needs no floating point at all.  Scaling by 2^53 is exact and K is an integer,
  // hence the ceil.
  uint64_t thr[16][3];
};

// The two tables the float64 (masked) selection needs, for kernels that take them from a
// per-instance parameter set instead of the launch-wide constants.
struct cobel_eps_bb {
  double base[5];
  double bonus[5];
};

// Constant lookup written as selects: a runtime index into a by-value kernel argument would send
// the struct to scratch memory.
#if defined(__HIPCC__)
// (and written as a one-hot SUM rather than a select chain: the optimizer turns a chain of selects
// over t[1..4] back into t[n], i.e. a table in scratch memory with a dynamic load on the critical
// path of every action selection.  Exactly one term is non-zero and all are >= 0, so the sum is
// exact.)
__device__ __forceinline__ double cobel_pick(const double* t, int n) {
  return (((n <= 1 ? t[1] : 0.0) + (n == 2 ? t[2] : 0.0)) + (n == 3 ? t[3] : 0.0)) +
         (n >= 4 ? t[4] : 0.0);
}
#endif

static inline cobel_eps_consts cobel_make_eps_consts(double eps) {
  cobel_eps_consts c;
  c.base[0] = c.bonus[0] = 0.0;
  for (int n = 1; n <= 4; ++n) {
    c.base[n] = eps / (double)n;
    c.bonus[n] = ((1.0 - eps) * 1.0) / (double)n;
  }
  for (int t = 0; t < 16; ++t) {
    int nt = 0;
    for (int a = 0; a < 4; ++a) nt += (t >> a) & 1;
    double cum[4], run = 0.0;
    for (int a = 0; a < 4; ++a) {
      // probs[a] = eps / 4; probs[a] += (1 - eps) * tie_a / n_ties      (greedy.py:83-86)
      double p = eps / 4.0;
      if (nt) p += ((1.0 - eps) * (((t >> a) & 1) ? 1.0 : 0.0)) / (double)nt;
      run = (a == 0) ? p : run + p;
      cum[a] = run;
    }
    for (int k = 0; k < 3; ++k) {
      const double cdf = nt ? cum[k] / cum[3] : 2.0;   // t = 0 cannot occur; never selected
      const double scaled = ceil(ldexp(cdf, 53));
      c.thr[t][k] = scaled >= 18446744073709551615.0 ? ~0ull : (uint64_t)scaled;
    }
  }
  return c;
}

#if defined(__HIPCC__)
// All arguments wave-uniform or per-lane alike; every lane returns the same answer it would get
// alone.  mask: 4-bit set of allowed actions (non-zero).  probs (optional): 4 doubles out.
template <typename V, typename K>
__device__ __forceinline__ int cobel_eps_greedy_select(V v0, V v1, V v2, V v3, uint32_t mask,
                                                       double u, const K& k,
                                                       double* probs = nullptr) {
  const V ninf = -(V)__builtin_huge_valf();
  const bool a0 = mask & 1u, a1 = mask & 2u, a2 = mask & 4u, a3 = mask & 8u;
  V m = ninf;
  m = (a0 && v0 > m) ? v0 : m;
  m = (a1 && v1 > m) ? v1 : m;
  m = (a2 && v2 > m) ? v2 : m;
  m = (a3 && v3 > m) ? v3 : m;
  const bool t0 = a0 && v0 == m, t1 = a1 && v1 == m, t2 = a2 && v2 == m, t3 = a3 && v3 == m;
  const int n = __popc(mask & 15u);
  const int nt = (int)t0 + (int)t1 + (int)t2 + (int)t3;
  const double base = cobel_pick(k.base, n), bonus = cobel_pick(k.bonus, nt);
  const double p0 = a0 ? base + (t0 ? bonus : 0.0) : 0.0;
  const double p1 = a1 ? base + (t1 ? bonus : 0.0) : 0.0;
  const double p2 = a2 ? base + (t2 ? bonus : 0.0) : 0.0;
  const double p3 = a3 ? base + (t3 ? bonus : 0.0) : 0.0;
  if (probs) {
    probs[0] = p0;
    probs[1] = p1;
    probs[2] = p2;
    probs[3] = p3;
  }
  const double c0 = p0, c1 = c0 + p1, c2 = c1 + p2, c3 = c2 + p3;
  // c3 / c3 == 1 > u always, so at most three thresholds can be passed.
  return (int)(c0 / c3 <= u) + (int)(c1 / c3 <= u) + (int)(c2 / c3 <= u);
}

// Same selection when the whole wave holds identical arguments: lanes 0..2 each take one of the
// three float64 divisions, a ballot counts the thresholds passed.  Returns a wave-uniform value.
template <typename K>
__device__ __forceinline__ int cobel_eps_greedy_select_wave(float v0, float v1, float v2,
                                                            float v3, uint32_t mask, double u,
                                                            const K& k, int lane) {
  const float ninf = -__builtin_huge_valf();
  const bool a0 = mask & 1u, a1 = mask & 2u, a2 = mask & 4u, a3 = mask & 8u;
  float m = ninf;
  m = a0 ? fmaxf(m, v0) : m;
  m = a1 ? fmaxf(m, v1) : m;
  m = a2 ? fmaxf(m, v2) : m;
  m = a3 ? fmaxf(m, v3) : m;
  const bool t0 = a0 && v0 == m, t1 = a1 && v1 == m, t2 = a2 && v2 == m, t3 = a3 && v3 == m;
  const int n = __popc(mask & 15u);
  const int nt = (int)t0 + (int)t1 + (int)t2 + (int)t3;
  const double base = cobel_pick(k.base, n), bonus = cobel_pick(k.bonus, nt);
  const double p0 = a0 ? base + (t0 ? bonus : 0.0) : 0.0;
  const double p1 = a1 ? base + (t1 ? bonus : 0.0) : 0.0;
  const double p2 = a2 ? base + (t2 ? bonus : 0.0) : 0.0;
  const double p3 = a3 ? base + (t3 ? bonus : 0.0) : 0.0;
  const double c0 = p0, c1 = c0 + p1, c2 = c1 + p2, c3 = c2 + p3;
  const double mine = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
  const bool pass = lane < 3 && (mine / c3 <= u);
  return __popcll(__ballot(pass));
}

// Unmasked fast path: thresholds from the table (staged in LDS as thr[t * 3 + k]), K = 53-bit draw.
__device__ __forceinline__ int cobel_eps_greedy_select_thr(float v0, float v1, float v2, float v3,
                                                           uint64_t K, const uint64_t* thr_lds,
                                                           int lane) {
  const float m = fmaxf(fmaxf(fmaxf(v0, v1), v2), v3);
  const int t = (int)(v0 == m) | ((int)(v1 == m) << 1) | ((int)(v2 == m) << 2) |
                ((int)(v3 == m) << 3);
  // (an LDS-typed pointer and the row's address by a 24-bit multiply-add written out: the compiler's choice for
  //  t * 24 + lane * 8 + table is v_mad_u64_u32, whose addend — the table's address plus the lane's
  //  offset, hoisted out of the caller's step loop — is a register PAIR that gets spilled and is
  //  reloaded in every step)
  typedef const __attribute__((address_space(3))) unsigned char* lds_bytes;
  typedef const __attribute__((address_space(3))) uint64_t* lds_u64;
  const uint32_t lane_addr = (uint32_t)(uintptr_t)((lds_bytes)thr_lds + ((uint32_t)lane << 3));
  uint32_t addr;
  asm("v_mad_u32_u24 %0, %1, 24, %2" : "=v"(addr) : "v"(t), "v"(lane_addr));
  const uint64_t T = lane < 3 ? *(lds_u64)(uintptr_t)addr : ~0ull;
  return __popcll(__ballot(lane < 3 && T <= K));
}
// policy/greedy.py:77-86 + Generator.choice for n <= AMAX values: float64 probabilities, sequential
// cumulative sum, normalisation by the last entry, searchsorted(side='right').
template <typename V, int AMAX = 8>
__device__ __forceinline__ int cobel_eps_greedy_select_n(const V* v, int A, uint32_t mask,
                                                         double u, double eps, double* probs) {
  const uint32_t allowed = mask & (A >= 32 ? 0xffffffffu : ((1u << A) - 1u));
  const int n = __popc(allowed);
  V m = -(V)__builtin_huge_valf();
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A && ((allowed >> a) & 1u) && v[a] > m) m = v[a];
  int nt = 0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a) nt += (a < A && ((allowed >> a) & 1u) && v[a] == m) ? 1 : 0;
  const double base = eps / (double)n;
  const double bonus = ((1.0 - eps) * 1.0) / (double)nt;
  double cum[AMAX];
  double run = 0.0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a) {
    double p = 0.0;
    if (a < A && ((allowed >> a) & 1u)) p = base + ((v[a] == m) ? bonus : 0.0);
    if (probs && a < A) probs[a] = p;
    run = (a == 0) ? p : run + p;
    cum[a] = run;
  }
  double total = cum[0];
#pragma unroll
  for (int a = 1; a < AMAX; ++a) total = (a == A - 1) ? cum[a] : total;
  int pick = 0;
#pragma unroll
  for (int a = 0; a < AMAX - 1; ++a) pick += (a < A - 1 && cum[a] / total <= u) ? 1 : 0;
  return pick;
}
#endif
