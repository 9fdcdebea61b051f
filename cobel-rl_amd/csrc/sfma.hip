// SFMA agent — one wavefront per agent–env instance (gfx950).
//
// SFMA is Dyna-Q whose replay picks experiences by priority R = C * D * (1 - I) [* T]: strength x
// similarity to the experience replayed last x (1 - inhibition) [x recency].  The reference
// evaluates R over all 4S experiences with NumPy for every single reactivation and draws one
// from softmax(R); that O(4S) scan + draw is the hot part (32 per trial by default).
//
// Here an instance's tables live in LDS for the whole call: Q (float32 [S][4]), strengths C
// (float64 [4S]), inhibition I (float64 [S]), the model's successor table NS (u16 [4S], experience
// order j = a * S + s, flag bit 15), its reward estimates R (float32 [4S]) and one scratch vector
// of 4S priorities.  Lane l owns the experiences
// [l * chunk, (l + 1) * chunk), so the cumulative sum behind the draw is one wave scan of lane
// totals plus a short in-lane running sum — element order as in the reference's cumsum.  The rows
// D[cur], D[next] of the similarity matrix (shared by all instances of a world, L2 resident) are
// staged in LDS once per reactivation.  The packed model records in HBM are written through on
// every store and never read back during the call; recency T is kept as a store stamp per experience and a
// table of decay powers instead of a vector that is rescaled on every store.
//
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   agent/sfma.py:233-334 (train), :336-396 (test), :398-458 (replay, update_q)
//   memory/sfma.py:195-236 (store), :238-347 (replay), :349-372 (softmax), :374-416 (random batch)
#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct sfma_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  int32_t chunk;  // experiences per lane: ceil(4S / 64)
  cobel_sfma_run_t r;
  cobel_eps_bb eps;
  uint64_t eps_thr[16][3];   // integer CDF thresholds of the unmasked selection (cobel_policy.h)
  float alpha_f, gamma_f, model_lr_f;
  // transition rows that are distributions (cobel_world_set_transitions), else NULL: SFMA.train
  // steps the interface (agent/sfma.py:262-264), whose step() then DRAWS the successor
  // (interface/gridworld.py:119-123) — one double of the env stream per step
  const uint32_t* succ_off;
  const uint16_t* succ_state;
  const double* succ_cdf;
};

struct sfma_lds {
  float4* Q;     // [S]
  double* C;     // [4S] strengths
  double* P;     // [4S] priorities / draw weights of the current reactivation
  double* I;     // [S]  inhibition
  double* Dc;    // [S]  similarity row of the current state
  double* Dn;    // [S]  similarity row of the next state
  float* R;      // [4S] model reward estimate of experience j
  uint16_t* NS;  // [4S] model successor of experience j | nonterminal flag << 15
  double* red;   // [4][8] scratch of the cross-wave reductions (several waves per instance)
  uint64_t* thr; // [48] epsilon-greedy thresholds, entry t * 3 + k
  double* epsc;  // [16] masked selection: base[1..4], bonus[1..4] (cobel_policy.h); then blend,
                 //      interp_fwd, interp_rev, decay_inhibition, i_step, alpha, gamma, beta
};

constexpr int kFastStates = 32;   // 4 S <= 128 experiences, two per lane

__host__ __device__ __forceinline__ size_t sfma_lds_bytes(int S) {
  return (((size_t)S * (16 + 32 + 32 + 8 + 8 + 8 + 16 + 8) + 15) & ~(size_t)15) + 256 + 384 + 128;
}

__device__ __forceinline__ sfma_lds carve(unsigned char* base, int S) {
  sfma_lds L;
  size_t off = 0;
  L.Q = reinterpret_cast<float4*>(base + off);
  off += (size_t)S * 16;
  L.C = reinterpret_cast<double*>(base + off);
  off += (size_t)S * 32;
  L.P = reinterpret_cast<double*>(base + off);
  off += (size_t)S * 32;
  L.I = reinterpret_cast<double*>(base + off);
  off += (size_t)S * 8;
  L.Dc = reinterpret_cast<double*>(base + off);
  off += (size_t)S * 8;
  L.Dn = reinterpret_cast<double*>(base + off);
  off += (size_t)S * 8;
  L.R = reinterpret_cast<float*>(base + off);
  off += (size_t)S * 16;
  L.NS = reinterpret_cast<uint16_t*>(base + off);
  off = (off + (size_t)S * 8 + 15) & ~(size_t)15;
  L.red = reinterpret_cast<double*>(base + off);
  off += 256;
  L.thr = reinterpret_cast<uint64_t*>(base + off);
  off += 384;
  L.epsc = reinterpret_cast<double*>(base + off);
  return L;
}

// Orders this wave's LDS traffic across lanes (one wave per workgroup: no s_barrier needed).
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}
// Cross-lane data movement on the VALU (DPP) instead of ds_bpermute through the LDS crossbar: a
// reactivation is a chain of five dependent wave-wide reductions / scans, so their latency is the
// critical path.  Controls: quad_perm 0x00-0xff, row_shr:n 0x110+n, wave_shr:1 0x138, row_mirror
// 0x140, row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143.
// (ZERO_FILL: lanes whose source lane does not exist read 0 — bound_ctrl — instead of keeping
//  `old`: with every row enabled the destination needs no initialisation, two moves less per use)
template <int CTRL, int ROW_MASK = 0xf, bool ZERO_FILL = false>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v), o = __builtin_bit_cast(uint64_t, old);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)b,
                                                            CTRL, ROW_MASK, 0xf, ZERO_FILL);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32),
                                                            (int)(uint32_t)(b >> 32), CTRL,
                                                            ROW_MASK, 0xf, ZERO_FILL);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
// max over the wave, returned in every lane (scalar registers)
__device__ __forceinline__ double wave_max_f64(double v) {
  v = fmax(v, dpp_f64<0xB1>(v, v));         // quad_perm [1,0,3,2]
  v = fmax(v, dpp_f64<0x4E>(v, v));         // quad_perm [2,3,0,1]
  v = fmax(v, dpp_f64<0x141>(v, v));        // row_half_mirror
  v = fmax(v, dpp_f64<0x140>(v, v));        // row_mirror: every lane holds its row's max
  v = fmax(v, dpp_f64<0x142, 0xa>(v, v));   // row_bcast:15 into rows 1, 3
  v = fmax(v, dpp_f64<0x143, 0xc>(v, v));   // row_bcast:31 into rows 2, 3
  return readlane_f64(v, 63);
}
// max over the wave of doubles that are >= +0 and not NaN: their order is the order of their bit
// patterns, so the maximum is the largest high word and, among its holders, the largest low word —
// twelve 32-bit DPP maxima instead of six float64 maxima with two DPP moves each.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  // (the compiler keeps DPP move and maximum apart — three instructions per stage; the fused form
  //  needs two wait states after the write of its DPP operand, which inline assembly must supply)
  asm volatile(
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_nonneg_f64(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t hi = (uint32_t)(b >> 32), lo = (uint32_t)b;
  const uint32_t mh = wave_max_u32(hi);
  const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
  return __builtin_bit_cast(double, ((uint64_t)mh << 32) | (uint64_t)ml);
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
// inclusive prefix sum over the lanes
__device__ __forceinline__ double wave_scan_f64(double v) {
  v = v + dpp_f64<0x111, 0xf, true>(0.0, v);   // row_shr:1
  v = v + dpp_f64<0x112, 0xf, true>(0.0, v);   // row_shr:2
  v = v + dpp_f64<0x114, 0xf, true>(0.0, v);   // row_shr:4
  v = v + dpp_f64<0x118, 0xf, true>(0.0, v);   // row_shr:8: prefix within each row of 16
  v = v + dpp_f64<0x142, 0xa>(0.0, v);      // rows 1, 3 += last lane of the row before
  v = v + dpp_f64<0x143, 0xc>(0.0, v);      // rows 2, 3 += lane 31
  return v;
}
__device__ __forceinline__ float max4_masked(const float4 q, uint32_t mask) {
  float m = -__builtin_huge_valf();
  if (mask & 1u) m = fmaxf(m, q.x);
  if (mask & 2u) m = fmaxf(m, q.y);
  if (mask & 4u) m = fmaxf(m, q.z);
  if (mask & 8u) m = fmaxf(m, q.w);
  return m;
}

// First experience whose weight equals vmax (np.argmax), wave-uniform.
__device__ __forceinline__ int wave_first_equal(const double* P, int n4, int chunk, int lane,
                                                double vmax) {
  const int j0 = lane * chunk;
  int first = 0x7fffffff;
  for (int k = chunk - 1; k >= 0; --k) {
    const int j = j0 + k;
    if (j < n4 && P[j] == vmax) first = j;
  }
  return wave_min_i32(first);
}

// CH > 0: the common switches (no recency, no C / D normalisation, R normalisation on, softmax
// draw) with exactly CH experiences per lane, which then live in registers from the priority
// rating to the draw.  CH = 0: every switch, any number of experiences per lane, through LDS.
// NW: waves per instance.  1 for the small worlds the reference's demos use; 4 (with CH = 0) for
// worlds of several hundred states, whose 4S experiences would otherwise sit 16-64 deep in each
// lane of a single wave.  Every wave carries the scalar state of the instance redundantly; thread
// 0 does the single-cell writes; the wave-wide reductions are completed across waves through a
// few LDS words and one workgroup barrier each.
// FAST (with CH > 0): the plain training case — learning on, one replay per trial, no start /
// random / dynamic replays, no strength modulation or decay, no per-step host log, occupancy,
// replay trace or per-instance latency trace — with those run-time switches fixed at compile time,
// so that the flags and pointers behind them do not have to stay live across the step loop.
// a / b for many a and one b, given y = 1 / b correctly rounded (one division per reactivation
// instead of one per experience): q0 = RN(a y), r = a - b q0 (exact in an fma), RN(q0 + r y) is the
// correctly rounded quotient (Markstein 1990) — the bits of a / b unless b's significand is all
// ones or the residual leaves the normal range, which the priorities never do.
__device__ __forceinline__ double quotient_by(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-q0, b, a);
  return __builtin_fma(r, y, q0);
}

// exp(x) for 0 <= x <= 700 as the device library evaluates it (argument reduction by ln 2 in two
// parts, its degree-11 polynomial, ldexp) without the overflow / underflow selections: the same
// bits for these arguments, six vector instructions fewer per experience.
__device__ __forceinline__ double exp_in_range(double x) {
  const double t = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(t, -0x1.62e42fefa39efp-1, x);
  r = __builtin_fma(t, -0x1.abc9e3b39803fp-56, r);
  double p = __builtin_fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = __builtin_fma(r, p, 0x1.71dee623fde64p-19);
  p = __builtin_fma(r, p, 0x1.a01997c89e6b0p-16);
  p = __builtin_fma(r, p, 0x1.a01a014761f6ep-13);
  p = __builtin_fma(r, p, 0x1.6c16c1852b7b0p-10);
  p = __builtin_fma(r, p, 0x1.1111111122322p-7);
  p = __builtin_fma(r, p, 0x1.55555555502a1p-5);
  p = __builtin_fma(r, p, 0x1.5555555555511p-3);
  p = __builtin_fma(r, p, 0x1.000000000000bp-1);
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  return __builtin_ldexp(p, (int)t);
}

template <int CH, int NW, bool FAST = false>
__device__ __forceinline__ void sfma_body(const sfma_args A) {
  static_assert(CH == 0 || NW == 1, "the register path is one wave per instance");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NT = 64 * NW;
  const int S = A.S, n4 = 4 * A.S, chunk = A.chunk;
  // (FAST: two experiences per lane means at most kFastStates states — the layout of that many, so
  //  that every LDS address is a compile-time offset instead of ten scalar registers)
  const sfma_lds L = FAST ? carve(lds_raw, kFastStates) : carve(lds_raw, S);
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = (int)blockIdx.x;
  int slot = 0;   // rotating scratch slot: one barrier per cross-wave reduction
  auto bsync = [&]() {
    if (NW == 1) wsync();
    else __syncthreads();
  };
  auto block_max = [&](double v) -> double {
    v = wave_max_f64(v);
    if (NW == 1) return v;
    double* const r = L.red + (slot++ & 3) * 8;
    if (lane == 0) r[wave] = v;
    __syncthreads();
    double m = r[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmax(m, r[w]);
    return m;
  };
  auto block_min_i32 = [&](int v) -> int {
    v = wave_min_i32(v);
    if (NW == 1) return v;
    int* const r = reinterpret_cast<int*>(L.red + (slot++ & 3) * 8);
    if (lane == 0) r[wave] = v;
    __syncthreads();
    int m = r[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = min(m, r[w]);
    return m;
  };
  auto block_sum_i32 = [&](int v) -> int {   // v: one value per wave
    if (NW == 1) return v;
    int* const r = reinterpret_cast<int*>(L.red + (slot++ & 3) * 8);
    if (lane == 0) r[wave] = v;
    __syncthreads();
    int m = r[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m += r[w];
    return m;
  };
  // np.argmax over L.P: the first experience whose weight equals vmax
  auto first_equal = [&](double vmax) -> int {
    const int j0 = t * chunk;
    int first = 0x7fffffff;
    for (int k = chunk - 1; k >= 0; --k) {
      const int j = j0 + k;
      if (j < n4 && L.P[j] == vmax) first = j;
    }
    return block_min_i32(first);
  };
  // Generator.choice(arange(n4), p = w / sum(w)) for the weights w >= 0 in L.P, driven by the
  // uniform u: the number of experiences whose cumulative weight is <= u * total.
  auto choice = [&](double u, double wmax) -> int {
    const int j0 = t * chunk;
    double loc = 0.0;
    for (int k = 0; k < chunk; ++k)
      if (j0 + k < n4) loc = loc + L.P[j0 + k];
    const double incl = wave_scan_f64(loc);
    double excl = dpp_f64<0x138, 0xf, true>(0.0, incl);   // wave_shr:1, lane 0 reads 0
    if (NW > 1) {   // add the totals of the waves before this one
      double* const r = L.red + (slot++ & 3) * 8;
      if (lane == 63) r[wave] = incl;
      __syncthreads();
      double off = 0.0;
      for (int w = 0; w < wave; ++w) off = off + r[w];
      excl = off + excl;
    }
    // the cumulative weight at the last experience; threads behind it hold nothing
    double total;
    if (NW == 1) {
      total = readlane_f64(excl + loc, (n4 - 1) / chunk);
    } else {
      double* const r = L.red + (slot++ & 3) * 8;
      if (t == (n4 - 1) / chunk) r[0] = excl + loc;
      __syncthreads();
      total = r[0];
    }
    const double thr = u * total;
    int idx = 0;
    double run = 0.0;
    for (int k = 0; k < chunk; ++k) {
      const bool in = j0 + k < n4;
      if (in) run = run + L.P[j0 + k];
      idx += __popcll(__ballot(in && (excl + run <= thr)));
    }
    idx = block_sum_i32(idx);
    idx = idx < n4 ? idx : n4 - 1;
    // an experience of weight zero has probability zero; rounding at a lane boundary of the scan
    // (or of u * total at u -> 1) is the only way to land on one
    if (!(L.P[idx] > 0.0)) idx = first_equal(wmax);
    return idx;
  };
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  const double* const Dm = A.r.metric + (size_t)world * S * S;
  float* const Qg = A.r.q + (size_t)i * n4;
  uint64_t* const Mg = A.r.model + (size_t)i * n4;
  double* const Cg = A.r.strength + (size_t)i * n4;
  uint32_t* const stamp = A.r.stamp + (size_t)i * n4;

  for (int e = t; e < S; e += NT) {
    L.Q[e] = reinterpret_cast<const float4*>(Qg)[e];
    L.I[e] = 0.0;
  }
  if (t < 48) L.thr[t] = A.eps_thr[t / 3][t % 3];
  if (t < 8) L.epsc[t] = t < 4 ? A.eps.base[t + 1] : A.eps.bonus[t - 3];
  if (t == 8) {
    L.epsc[8] = A.r.blend;
    L.epsc[9] = A.r.interp_fwd;
    L.epsc[10] = A.r.interp_rev;
    L.epsc[11] = A.r.decay_inhibition;
    L.epsc[12] = A.r.i_step;
    L.epsc[13] = A.r.alpha;
    L.epsc[14] = A.r.gamma;
    L.epsc[15] = A.r.beta;
  }
  for (int e = t; e < n4; e += NT) {
    L.C[e] = Cg[e];
    const uint64_t rec = Mg[e];
    const int j = (e & 3) * S + (e >> 2);
    L.R[j] = __builtin_bit_cast(float, (uint32_t)rec);
    L.NS[j] = (uint16_t)(((rec >> 32) & 0x7fffu) | (((rec >> 48) & 1u) << 15));
  }
  bsync();

  int32_t* const inst = A.r.inst + (size_t)i * COBEL_I_WORDS;
  int32_t* const sinst = A.r.sfma_inst + (size_t)i * COBEL_SI_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t cm = (uint32_t)inst[COBEL_I_CTR_MEMORY];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);
  unsigned long long nsteps = *reinterpret_cast<const unsigned long long*>(inst + COBEL_I_STEPS_LO);
  uint32_t clock = (uint32_t)sinst[COBEL_SI_CLOCK];
  uint32_t epoch = (uint32_t)sinst[COBEL_SI_EPOCH];
  int mode = sinst[COBEL_SI_MODE];
  uint32_t sflags = (uint32_t)sinst[COBEL_SI_FLAGS];
  double td_acc = *reinterpret_cast<const double*>(sinst + COBEL_SI_TD_LO);
  uint32_t ca = (uint32_t)sinst[COBEL_SI_CTR_AGENT];
  int tpos = A.r.trace_len ? A.r.trace_len[i] : 0;

  const uint32_t flags = A.r.flags, sf = A.r.sfma_flags;
  const bool learn = FAST || (flags & COBEL_F_LEARN);
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (flags & COBEL_F_MASK_ACTIONS) ? A.r.action_mask : nullptr;
  const uint64_t seed = A.r.seed;
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  // Everything in this kernel is wave-uniform, so the compiler wants it all in scalar registers
  // and then spills (1 300 of 3 000 vector instructions were v_readlane / v_writelane).  Constants
  // that only feed vector arithmetic are pinned to vector registers instead.
  double r_thr = A.r.r_threshold;
  float alpha_f = A.alpha_f, gamma_f = A.gamma_f, mlr_f = A.model_lr_f;
  asm volatile("" : "+v"(r_thr), "+v"(alpha_f), "+v"(gamma_f), "+v"(mlr_f));
  // (the constants used once per reactivation or only by some replay modes are read from LDS where
  //  they are used: blend, interp_fwd / _rev, decay_inhibition, i_step, alpha, gamma, beta — L.epsc[8 ..])
  // (the eight constants of the masked action selection sit in LDS: 16 vector registers that decide
  //  between five and six waves per SIMD for the two-experiences-per-lane kernels)
  // cobel_eps_greedy_select_wave (cobel_policy.h) on those scalars: lanes 0..2 take one float64
  // division each, a ballot counts the thresholds of the normalised CDF that u has passed
  auto select_action = [&](const float4 v, uint32_t mask, double u) -> int {
    const float ninf = -__builtin_huge_valf();
    const bool a0 = mask & 1u, a1 = mask & 2u, a2 = mask & 4u, a3 = mask & 8u;
    float m = ninf;
    m = a0 ? fmaxf(m, v.x) : m;
    m = a1 ? fmaxf(m, v.y) : m;
    m = a2 ? fmaxf(m, v.z) : m;
    m = a3 ? fmaxf(m, v.w) : m;
    const bool t0 = a0 && v.x == m, t1 = a1 && v.y == m, t2 = a2 && v.z == m, t3 = a3 && v.w == m;
    const int n = __popc(mask & 15u);
    const int nt = (int)t0 + (int)t1 + (int)t2 + (int)t3;
    const double base = L.epsc[(n <= 1 ? 1 : n) - 1];
    const double bonus = L.epsc[4 + (nt <= 1 ? 1 : nt) - 1];
    const double p0 = a0 ? base + (t0 ? bonus : 0.0) : 0.0;
    const double p1 = a1 ? base + (t1 ? bonus : 0.0) : 0.0;
    const double p2 = a2 ? base + (t2 ? bonus : 0.0) : 0.0;
    const double p3 = a3 ? base + (t3 ? bonus : 0.0) : 0.0;
    const double c0 = p0, c1 = c0 + p1, c2 = c1 + p2, c3 = c2 + p3;
    const double mine = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
    return __popcll(__ballot(lane < 3 && (mine / c3 <= u)));
  };

  // CH > 0: this lane's experiences j = lane * CH + k, their states and whether they exist
  constexpr int CHN = CH > 0 ? CH : 1;
  int jj[CHN], sid[CHN];
  bool inb[CHN];
#pragma unroll
  for (int k = 0; k < CHN; ++k) {
    const int j = t * CHN + k;
    inb[k] = j < n4;
    jj[k] = inb[k] ? j : n4 - 1;
    sid[k] = jj[k] % S;
  }

  cobel_u4 pblk = {0, 0, 0, 0}, mblk = {0, 0, 0, 0};
  uint32_t pb_idx = ~0u, mb_idx = ~0u;
  // scalar double draw number cm of the memory stream (sub 1): one block serves two counters
  auto mem_u01 = [&]() -> double {
    if ((cm >> 1) != mb_idx) {
      mb_idx = cm >> 1;
      mblk = cobel_philox(mb_idx, COBEL_SUB_DOUBLE, g, COBEL_STREAM_MEMORY, seed);
    }
    const double u = (cm & 1u) ? cobel_u01(mblk.z, mblk.w) : cobel_u01(mblk.x, mblk.y);
    cm += 1u;
    return u;
  };
  unsigned long long executed = 0, replayed = 0;
  int budget = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;

  // agent.update_q for a replayed experience (agent/sfma.py:437-455 with M.rewards float32 and
  // M.terminals int64: float64 arithmetic, one rounding into the float32 table)
  auto replay_td = [&](int s, int a, int ns, float R, uint32_t nt) -> double {
    const float4 nrow = L.Q[ns];
    const float m = max4_masked(nrow, amask ? (uint32_t)amask[ns] & 15u : 15u);
    const float q = reinterpret_cast<const float*>(L.Q)[s * 4 + a];
    const double gnt = L.epsc[14] * (double)nt;
    double td = (double)R + gnt * (double)m;
    td = td - (double)q;
    bsync();
    if (t == 0) reinterpret_cast<float*>(L.Q)[s * 4 + a] = (float)((double)q + L.epsc[13] * td);
    bsync();
    td_acc = ((sflags & 1u) ? (double)(float)td_acc : td_acc) + fabs(td);
    sflags &= ~1u;
    return td;
  };
  auto record = [&](int s, int a, int ns, float R, uint32_t nt, int kind, int tr, double td) {
    if (!FAST && A.r.replay_trace && t == 0 && tpos < A.r.trace_cap) {
      cobel_sfma_event_t ev;
      ev.sa = (uint32_t)s | ((uint32_t)a << 16) | (nt << 24) | ((uint32_t)kind << 25);
      ev.next = (uint32_t)ns;
      ev.reward = R;
      ev.trial = tr;
      ev.td = td;
      A.r.replay_trace[(size_t)i * A.r.trace_cap + tpos] = ev;
    }
    tpos += 1;
  };

  // SFMAMemory.replay (memory/sfma.py:238-347) [+ the TD updates of SFMA.replay when `update`]
  // (the kernels with experiences in registers are launched without the normalisation switches)
  const bool c_norm = CH == 0 && (sf & COBEL_SF_C_NORMALIZE);
  const bool d_norm = CH == 0 && (sf & COBEL_SF_D_NORMALIZE);
  auto sfma_replay = [&](int start_state, bool update, int kind, int tr) {
    int action = (int)cobel_draw_bounded(cm, 0u, g, COBEL_STREAM_MEMORY, seed, 4u);
    cm += 1u;
    int cur = start_state;
    const int j0 = t * chunk;
    if (cur < 0) {
      // no terminal state was reached: start from an experience drawn by strength (:262-270)
      double wmax = 0.0;
      for (int k = 0; k < chunk; ++k)
        if (j0 + k < n4) {
          const double c = L.C[j0 + k];
          const double w = c < 0.0 ? 0.0 : c;
          L.P[j0 + k] = w;
          wmax = fmax(wmax, w);
        }
      wmax = block_max(wmax);
      bsync();
      const double u = mem_u01();
      const int pick = choice(u, wmax);
      action = pick / S;
      cur = pick - action * S;
      bsync();
    }
    int nxt = (int)(L.NS[action * S + cur] & 0x7fffu);
    for (int e = t; e < S; e += NT) L.I[e] = 0.0;
    double cmax = 1.0;
    if (c_norm) {
      double m = -__builtin_huge_val();
      for (int e = t; e < n4; e += NT) m = fmax(m, L.C[e]);
      cmax = block_max(m);
    }
    bsync();
    const bool need_next = mode == COBEL_SFMA_FORWARD || mode == COBEL_SFMA_BLEND_FORWARD ||
                           mode == COBEL_SFMA_INTERPOLATE || mode == COBEL_SFMA_SWEEPING;
    for (int it = 0; it < A.r.batch; ++it) {
      // similarity rows (:284-287); D_normalize divides the row of the current state only
      {
        const double* const rc = Dm + (size_t)cur * S;
        const double* const rn = Dm + (size_t)nxt * S;
        double dmax = 1.0;
        if (d_norm) {
          double m = -__builtin_huge_val();
          for (int e = t; e < S; e += NT) m = fmax(m, rc[e]);
          dmax = block_max(m);
        }
        for (int e = t; e < S; e += NT) {
          const double d = rc[e];
          L.Dc[e] = d_norm ? d / dmax : d;
          if (need_next) L.Dn[e] = rn[e];
        }
      }
      bsync();
      int pick;
      if (CH > 0) {
        // similarity of every experience to the one replayed last, by mode (:284-307)
        double d[CHN];
        uint32_t nsv[CHN];
#pragma unroll
        for (int k = 0; k < CHN; ++k) nsv[k] = L.NS[jj[k]] & 0x7fffu;
        switch (mode) {
          case COBEL_SFMA_DEFAULT:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dc[sid[k]];
            break;
          case COBEL_SFMA_FORWARD:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dn[sid[k]];
            break;
          case COBEL_SFMA_REVERSE:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dc[nsv[k]];
            break;
          case COBEL_SFMA_BLEND_FORWARD:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dc[sid[k]] + L.epsc[8] * L.Dn[sid[k]];
            break;
          case COBEL_SFMA_BLEND_REVERSE:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dc[sid[k]] + L.epsc[8] * L.Dc[nsv[k]];
            break;
          case COBEL_SFMA_INTERPOLATE:
#pragma unroll
            for (int k = 0; k < CHN; ++k)
              d[k] = L.epsc[9] * L.Dn[sid[k]] + L.epsc[10] * L.Dc[nsv[k]];
            break;
          default:
#pragma unroll
            for (int k = 0; k < CHN; ++k) d[k] = L.Dn[nsv[k]];
            break;
        }
        // priority ratings (:308-318)
        double p[CHN];
        double rmax = 0.0;
#pragma unroll
        for (int k = 0; k < CHN; ++k) {
          double R = L.C[jj[k]] * d[k];
          R = R * (1.0 - L.I[sid[k]]);
          if (R < r_thr) R = 0.0;
          p[k] = inb[k] ? R : 0.0;
          rmax = fmax(rmax, p[k]);
        }
        // (FAST: r_threshold >= 0 and strengths, similarities and 1 - I are >= 0 — checked on the
        //  host, resp. true of what the fast launch admits — so the ratings are >= +0)
        rmax = FAST ? wave_max_nonneg_f64(rmax) : block_max(rmax);
        if (!(rmax > 0.0)) break;
        // softmax weights exp(beta R / max R) - 1 (:319-327, :349-372)
        bool some = false;
        const double inv_rmax = 1.0 / rmax;
#pragma unroll
        for (int k = 0; k < CHN; ++k) {
          // (FAST: 0 <= R / max R * beta <= 700 is checked on the host)
          const double x = quotient_by(p[k], rmax, inv_rmax) * L.epsc[15];
          const double w = (FAST ? exp_in_range(x) : exp(x)) + -1.0;
          p[k] = inb[k] ? w : 0.0;
          some = some || p[k] > 0.0;
        }
        if (!__ballot(some)) {  // np.sum(exp) == 0 -> exp.fill(1)
#pragma unroll
          for (int k = 0; k < CHN; ++k) p[k] = inb[k] ? 1.0 : 0.0;
        }
        // the draw: experiences whose cumulative weight is <= u * total
        const double u = mem_u01();
        double loc = p[0];
#pragma unroll
        for (int k = 1; k < CHN; ++k) loc = loc + p[k];
        const double incl = wave_scan_f64(loc);
        const double excl = dpp_f64<0x138, 0xf, true>(0.0, incl);
        const double total = readlane_f64(excl + loc, (n4 - 1) / CHN);
        const double thr = u * total;
        int idx = 0;
        double run = 0.0;
#pragma unroll
        for (int k = 0; k < CHN; ++k) {
          run = run + p[k];
          idx += __popcll(__ballot(inb[k] && (excl + run <= thr)));
        }
        idx = idx < n4 ? idx : n4 - 1;
        bool ok = false;
#pragma unroll
        for (int k = 0; k < CHN; ++k) ok = ok || (inb[k] && jj[k] == idx && p[k] > 0.0);
        if (!__ballot(ok)) {
          // rounding put the draw on an experience of weight zero: take the argmax instead
          double wmax = 0.0;
#pragma unroll
          for (int k = 0; k < CHN; ++k) {
            if (inb[k]) L.P[jj[k]] = p[k];
            wmax = fmax(wmax, p[k]);
          }
          wmax = block_max(wmax);
          bsync();
          idx = wave_first_equal(L.P, n4, CHN, lane, wmax);
        }
        pick = idx;
      } else {
      // priority ratings (:288-316)
      double rmax = 0.0;
      {
        int s = j0 % S;
        for (int k = 0; k < chunk; ++k) {
          const int j = j0 + k;
          if (j < n4) {
            double c = L.C[j];
            if (c_norm) c = c / cmax;
            double d;
            if (mode == COBEL_SFMA_DEFAULT) d = L.Dc[s];
            else if (mode == COBEL_SFMA_FORWARD) d = L.Dn[s];
            else if (mode == COBEL_SFMA_REVERSE) d = L.Dc[L.NS[j] & 0x7fffu];
            else if (mode == COBEL_SFMA_BLEND_FORWARD) d = L.Dc[s] + L.epsc[8] * L.Dn[s];
            else if (mode == COBEL_SFMA_BLEND_REVERSE) d = L.Dc[s] + L.epsc[8] * L.Dc[L.NS[j] & 0x7fffu];
            else if (mode == COBEL_SFMA_INTERPOLATE)
              d = L.epsc[9] * L.Dn[s] + L.epsc[10] * L.Dc[L.NS[j] & 0x7fffu];
            else d = L.Dn[L.NS[j] & 0x7fffu];
            double R = c * d;
            R = R * (1.0 - L.I[s]);
            if (sf & COBEL_SF_RECENCY) {
              const uint32_t st = stamp[j];
              double t = 0.0;
              if (st > epoch) {
                const uint32_t age = clock - st;
                t = A.r.recency_tab[age < (uint32_t)A.r.recency_len ? age
                                                                      : (uint32_t)A.r.recency_len - 1u];
              }
              R = R * t;
            }
            if (R < r_thr) R = 0.0;
            L.P[j] = R;
            rmax = fmax(rmax, R);
          }
          s += 1;
          if (s == S) s = 0;
        }
      }
      rmax = block_max(rmax);
      if (!(rmax > 0.0)) break;  // np.sum(R) == 0: nothing left to reactivate (:317-318)
      bsync();
      if (sf & COBEL_SF_DETERMINISTIC) {
        pick = first_equal(rmax);
      } else {
        // softmax(R, offset -1, beta) = exp(beta R) - 1 (:349-372), then the draw
        double wmax = 0.0;
        const double inv_rmax = 1.0 / rmax;
        for (int k = 0; k < chunk; ++k)
          if (j0 + k < n4) {
            double R = L.P[j0 + k];
            if (sf & COBEL_SF_R_NORMALIZE) R = quotient_by(R, rmax, inv_rmax);
            const double w = exp(R * L.epsc[15]) + -1.0;
            L.P[j0 + k] = w;
            wmax = fmax(wmax, w);
          }
        wmax = block_max(wmax);
        if (!(wmax > 0.0)) {  // np.sum(exp) == 0 -> exp.fill(1)
          for (int k = 0; k < chunk; ++k)
            if (j0 + k < n4) L.P[j0 + k] = 1.0;
          wmax = 1.0;
        }
        bsync();
        const double u = mem_u01();
        pick = choice(u, wmax);
      }
      }
      action = (int)(pick >= S) + (int)(pick >= 2 * S) + (int)(pick >= 3 * S);   // pick / S
      cur = pick - action * S;
      const uint32_t nrec = L.NS[pick];
      const float R = L.R[pick];
      nxt = (int)(nrec & 0x7fffu);
      const uint32_t nt = nrec >> 15;
      bsync();
      // inhibition (:336-337)
      {
        const double dec_inh = L.epsc[11];
        for (int e = t; e < S; e += NT) L.I[e] = L.I[e] * dec_inh;
      }
      bsync();
      if (t == 0) L.I[cur] = fmin(L.I[cur] + L.epsc[12], 1.0);
      // the reactivated experience
      double td = __builtin_nan("");
      if (update) td = replay_td(cur, action, nxt, R, nt);
      record(cur, action, nxt, R, nt, kind, tr, td);
      replayed += 1ull;
      bsync();
    }
  };

  // SFMAMemory.retrieve_random_batch (:374-416) + the TD updates
  auto random_replay = [&](int tr) {
    const int j0 = t * chunk;
    for (int b = 0; b < A.r.batch; ++b) {
      const double u = cobel_draw_u01(cm, COBEL_SUB_DOUBLE + (uint32_t)b, g, COBEL_STREAM_MEMORY,
                                      seed);
      int idx = 0;
      for (int k = 0; k < chunk; ++k)
        idx += __popcll(__ballot(j0 + k < n4 && A.r.random_cdf[j0 + k] <= u));
      idx = block_sum_i32(idx);
      idx = idx < n4 ? idx : n4 - 1;
      const int a = idx / S, s = idx - a * S;   // unravel_index(order='F')
      const uint32_t nrec = L.NS[idx];
      const int ns = (int)(nrec & 0x7fffu);
      const float R = L.R[idx];
      const uint32_t nt = nrec >> 15;
      const double td = replay_td(s, a, ns, R, nt);
      record(s, a, ns, R, nt, 0, tr, td);
      replayed += 1ull;
    }
    cm += 1u;   // one vector draw per batch
  };

  // Replays are requested (trial start: one without TD updates; trial end: nb_replays with) and
  // served at ONE place at the top of the loop, so the reactivation code exists once.
  // (Round 4: the replays and the online steps of a trial are inner loops of their own.  As ONE loop
  //  with `continue`s every scalar of either phase was live across every iteration: the online step
  //  reloaded ~140 spilled scalars — v_readlane, a vector instruction on a kernel bound by vector
  //  issue.)
  int req_count = 0, req_start = -1, req_kind = 0, req_trial = 0;
  while (true) {
    while (req_count > 0) {
      req_count -= 1;
      if (!FAST && req_kind == 0 && (sf & COBEL_SF_RANDOM)) random_replay(req_trial);
      else sfma_replay(req_start, req_kind == 0, req_kind, req_trial);
      if (req_count == 0 && req_kind == 0) epoch = clock;  // M.T.fill(0) after a trial's replays
    }
    if (!(iflags & 1u)) {
      if (trial >= A.r.trials_target) break;
      if (budget == 0) break;
      state = (int)A.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                               start_cnt)];
      ce += 1u;
      step = 0;
      trew = 0.0;
      iflags |= 1u;
      if (!FAST && learn && (sf & COBEL_SF_START_REPLAY)) {
        req_count = 1;
        req_start = state;
        req_kind = 1;
        req_trial = trial;
        continue;
      }
    }
    // ---- the online steps of the running trial ---------------------------------------------------
    bool out_of_budget = false;
    int ns = state;
    uint32_t end = 0u;
    for (;;) {
    if (budget == 0) {
      out_of_budget = true;
      break;
    }
    budget -= 1;

    // ---- select + env.step ---------------------------------------------------------------------
    const float4 q = L.Q[state];
    const uint32_t mask_cur = amask ? (uint32_t)amask[state] & 15u : 15u;
    if ((cp >> 1) != pb_idx) {
      pb_idx = cp >> 1;
      pblk = cobel_philox(pb_idx, 0u, g, pol_stream, seed);
    }
    const uint32_t w0 = (cp & 1u) ? pblk.z : pblk.x, w1 = (cp & 1u) ? pblk.w : pblk.y;
    cp += 1u;
    // all actions allowed (no mask, or the reference's default all-true mask): the draw is compared
    // with integer thresholds of the tie pattern's CDF, no floating point (cobel_policy.h) — the
    // float64 selection with its three divisions was 30 % of an online step
    const int a = mask_cur == 15u
                      ? (int)rfl((uint32_t)cobel_eps_greedy_select_thr(q.x, q.y, q.z, q.w,
                                                                        cobel_u53(w0, w1), L.thr, lane))
                      : (int)rfl((uint32_t)select_action(q, mask_cur, cobel_u01(w0, w1)));
    if (!FAST && A.succ_off) {
      const double ue = cobel_draw_u01(ce, COBEL_SUB_DOUBLE, g, COBEL_STREAM_ENV, seed);
      ce += 1u;
      ns = (int)rfl((uint32_t)cobel_draw_successor(
          A.succ_off, A.succ_state, A.succ_cdf, ((size_t)world * S + (size_t)state) * 4 + a, ue));
    } else {
      const uint4 wc = W4[state];
      ns = (int)next_of(rfl(wc.x), rfl(wc.y), a);
    }
    const uint4 wn = W4[ns];
    const float r = __builtin_bit_cast(float, rfl(wn.z));
    end = rfl(wn.w);
    const uint32_t nt = 1u - end;
    float td_online = 0.0f;

    if (learn) {
      const int sa = state * 4 + a, j = a * S + state;
      // M.store (memory/sfma.py:204-236)
      const float Rold = L.R[j];
      const float d = r - Rold;
      const float Rnew = Rold + mlr_f * d;
      if (t == 0) {
        Mg[sa] = cobel_model_pack(Rnew, (uint32_t)ns, nt);   // written through
        L.R[j] = Rnew;
        L.NS[j] = (uint16_t)((uint32_t)ns | (nt << 15));
      }
      if (!FAST && A.r.decay_strength != 1.0) {
        for (int e = t; e < n4; e += NT) L.C[e] = L.C[e] * A.r.decay_strength;
        bsync();
      }
      clock += 1u;
      if (t == 0) {
        double c = L.C[j] + A.r.c_step;
        if (!FAST && (sf & COBEL_SF_REWARD_MOD_LOCAL)) c = c + (double)r * A.r.reward_modulation;
        L.C[j] = c;
        stamp[j] = clock;
      }
      if (!FAST && (sf & COBEL_SF_REWARD_MOD)) {
        bsync();
        const double* const row = Dm + (size_t)state * S;
        for (int e = t; e < n4; e += NT) {
          const int s2 = e % S;
          L.C[e] = L.C[e] + ((double)r * row[s2]) * A.r.reward_modulation;
        }
      }
      if (!FAST && (sf & COBEL_SF_STATE_MOD)) {
        bsync();
        if (t < 4) L.C[t * S + state] = L.C[t * S + state] + 1.0;
      }
      // agent.update_q online (agent/sfma.py:437-455), float32
      const float4 nrow = L.Q[ns];
      const float m = max4_masked(nrow, amask ? (uint32_t)amask[ns] & 15u : 15u);
      const float qsa = (a & 2) ? ((a & 1) ? q.w : q.z) : ((a & 1) ? q.y : q.x);
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * m;
      td = td - qsa;
      bsync();
      if (t == 0) reinterpret_cast<float*>(L.Q)[sa] = qsa + alpha_f * td;
      bsync();
      td_online = td;
      if (sflags & 1u) td_acc = (double)((float)td_acc + fabsf(td));
      else td_acc = td_acc + (double)fabsf(td);
    }

    if (!FAST && A.r.last_exp && t == 0) {
      int32_t* const e = A.r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = __builtin_bit_cast(int32_t, td_online);
    }
    trew += (double)r;
    nsteps += 1ull;
    executed += 1ull;
    if (!FAST && A.r.occupancy && t == 0) atomicAdd(A.r.occupancy + (size_t)world * S + ns, 1ull);
    state = ns;
    if (end || (step + 1 >= A.r.steps_per_trial)) break;
    step += 1;
    }
    if (out_of_budget) break;
    {
      if (t == 0 && trial >= 0 && trial < A.r.trial_cap) {
        const size_t m = cobel_mon_offset(A.r.mon_stripes, A.r.trial_cap) + (size_t)trial;
        if (A.r.lat_sum) atomicAdd(A.r.lat_sum + m, (unsigned long long)step);
        if (A.r.lat_cnt) atomicAdd(A.r.lat_cnt + m, 1ull);
        if (A.r.reward_sum) atomicAdd(A.r.reward_sum + m, trew);
        if (A.r.resp_cnt && trew > 0.0) atomicAdd(A.r.resp_cnt + m, 1ull);
        if (!FAST && A.r.lat_trace) A.r.lat_trace[(size_t)i * A.r.trial_cap + trial] = step;
      }
      const int tr = trial;
      trial += 1;
      iflags &= ~1u;
      if (FAST || (learn && !(flags & COBEL_F_NO_REPLAY))) {
        if (!FAST && (sf & COBEL_SF_DYNAMIC)) {
          // agent/sfma.py:308-316: p(reverse) = 1 / (1 + exp(-(5 td - 2))), in the type the
          // |TD| sum has at this point
          double p0, p1;
          if (sflags & 1u) {
            const float x = (float)td_acc * 5.0f - 2.0f;
            const float p = 1.0f / (1.0f + expf(-x));
            p0 = (double)p;
            p1 = (double)(1.0f - p);
          } else {
            const double x = td_acc * 5.0 - 2.0;
            p0 = 1.0 / (1.0 + exp(-x));
            p1 = 1.0 - p0;
          }
          const double uu = cobel_draw_u01(ca, 0u, g, COBEL_STREAM_AGENT, seed);
          ca += 1u;
          const double c1 = p0 + p1;
          mode = (p0 / c1 <= uu) ? COBEL_SFMA_DEFAULT : COBEL_SFMA_REVERSE;
          td_acc = 0.0;
          sflags |= 1u;
        }
        req_count = FAST ? 1 : A.r.nb_replays;
        req_start = end ? ns : -1;
        req_kind = 0;
        req_trial = tr;
        if (req_count == 0) epoch = clock;  // M.T.fill(0)
      }
    }
  }

  bsync();
  for (int e = t; e < S; e += NT) reinterpret_cast<float4*>(Qg)[e] = L.Q[e];
  for (int e = t; e < n4; e += NT) Cg[e] = L.C[e];
  if (t == 0) {
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_CTR_MEMORY] = (int32_t)cm;
    inst[COBEL_I_FLAGS] = (int32_t)iflags;
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) = nsteps;
    sinst[COBEL_SI_CLOCK] = (int32_t)clock;
    sinst[COBEL_SI_EPOCH] = (int32_t)epoch;
    sinst[COBEL_SI_MODE] = mode;
    sinst[COBEL_SI_FLAGS] = (int32_t)sflags;
    *reinterpret_cast<double*>(sinst + COBEL_SI_TD_LO) = td_acc;
    sinst[COBEL_SI_CTR_AGENT] = (int32_t)ca;
    if (A.r.trace_len) A.r.trace_len[i] = tpos;
    if (A.r.steps_done && executed) atomicAdd(A.r.steps_done, executed);
    if (A.r.replays_done && replayed) atomicAdd(A.r.replays_done, replayed);
  }
}

template <int CH>
__global__ __launch_bounds__(64) void k_sfma(const sfma_args A) {
  sfma_body<CH, 1>(A);
}
// Two experiences per lane (worlds up to 32 states, the reference's demos): five waves per SIMD
// instead of the four the register allocation settles on by itself — +12 % on C6 —, six for the
// plain-training instantiation since the constants of the masked action selection and the ones a
// reactivation uses once (L.epsc) are read from LDS instead of being pinned to vector registers
// (80 registers, no scratch: +5 %; seven waves measure the same, eight 5 % less; 69 registers —
// seven waves resident — since the LDS layout is fixed at compile time).  The same hint
// costs the wider variants 20-25 % (spills into scratch), so they keep the default.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_sfma_2(
    const sfma_args A) {
  sfma_body<2, 1>(A);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_sfma_2_fast(
    const sfma_args A) {
  sfma_body<2, 1, true>(A);
}
// Four waves per instance, all switches: worlds of several hundred states.
__global__ __launch_bounds__(256) void k_sfma_wg(const sfma_args A) {
  sfma_body<0, 4>(A);
}

template <int CH>
int launch_sfma(const sfma_args& A, size_t lds, hipStream_t st) {
  const void* fn = CH == 2 ? reinterpret_cast<const void*>(&k_sfma_2)
                           : reinterpret_cast<const void*>(&k_sfma<CH>);
  if (lds > 64 * 1024)
    COBEL_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (CH == 2) hipLaunchKernelGGL(k_sfma_2, dim3(A.r.n), dim3(64), lds, st, A);
  else hipLaunchKernelGGL((k_sfma<CH>), dim3(A.r.n), dim3(64), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
int launch_sfma_wg(const sfma_args& A, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sfma_wg),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_sfma_wg, dim3(A.r.n), dim3(256), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

}  // namespace

static const size_t kSfmaLdsLimit = 160 * 1024;

extern "C" int cobel_sfma_query(int32_t n_states, int32_t* lds_bytes) {
  COBEL_REQUIRE(n_states > 0, COBEL_E_RANGE, "cobel_sfma_query: %d states", n_states);
  const size_t lds = sfma_lds_bytes(n_states);
  if (lds_bytes) *lds_bytes = (int32_t)lds;
  COBEL_REQUIRE(lds <= kSfmaLdsLimit && n_states <= 16383, COBEL_E_UNSUPPORTED,
                "cobel_sfma_query: %d states need %zu B of LDS per instance (limit %zu)", n_states,
                lds, kSfmaLdsLimit);
  return COBEL_OK;
}

namespace {
__global__ void k_exp_check(const double* x, double* a, double* b, const double* d, double* qa,
                            double* qb, int n) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e < n) {
    a[e] = exp_in_range(x[e]);
    b[e] = exp(x[e]);
    if (d) {
      qa[e] = quotient_by(x[e], d[e], 1.0 / d[e]);
      qb[e] = x[e] / d[e];
    }
  }
}
}  // namespace

extern "C" int cobel_sfma_exp_check(const double* x, double* in_range, double* library,
                                    const double* divisor, double* quotient_by_reciprocal,
                                    double* quotient, int32_t n, void* stream) {
  COBEL_REQUIRE(x && in_range && library && n >= 0 &&
                    (!divisor || (quotient_by_reciprocal && quotient)),
                COBEL_E_ARG, "cobel_sfma_exp_check: bad arguments");
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_exp_check, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x,
                     in_range, library, divisor, quotient_by_reciprocal, quotient, n);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_sfma_run(const cobel_world_t* world, const cobel_sfma_run_t* run,
                              void* stream) {
  if (int rc = cobel_world_check(world, "cobel_sfma_run")) return rc;
  COBEL_REQUIRE(world->n_actions == 4, COBEL_E_UNSUPPORTED,
                "cobel_sfma_run: the world has %d actions, this entry point serves four-action worlds",
                world->n_actions);
  COBEL_REQUIRE(world && run, COBEL_E_ARG, "cobel_sfma_run: NULL world/run");
  const cobel_sfma_run_t& r = *run;
  COBEL_REQUIRE(r.q && r.model && r.strength && r.stamp && r.inst && r.sfma_inst && r.metric,
                COBEL_E_ARG,
                "cobel_sfma_run: q, model, strength, stamp, inst, sfma_inst and metric are required");
  COBEL_REQUIRE(((uintptr_t)r.q & 15u) == 0 && ((uintptr_t)r.inst & 7u) == 0 &&
                    ((uintptr_t)r.sfma_inst & 7u) == 0 && ((uintptr_t)r.replay_trace & 7u) == 0,
                COBEL_E_ARG, "cobel_sfma_run: q must be 16-byte, inst / sfma_inst / trace 8-byte aligned");
  COBEL_REQUIRE(r.n >= 0, COBEL_E_RANGE, "cobel_sfma_run: n = %d", r.n);
  COBEL_REQUIRE(r.steps_per_trial > 0, COBEL_E_RANGE, "cobel_sfma_run: steps_per_trial = %d",
                r.steps_per_trial);
  COBEL_REQUIRE(r.batch >= 0 && r.nb_replays >= 0, COBEL_E_RANGE,
                "cobel_sfma_run: batch = %d, nb_replays = %d", r.batch, r.nb_replays);
  COBEL_REQUIRE(r.epsilon >= 0.0 && r.epsilon <= 1.0, COBEL_E_ARG,
                "cobel_sfma_run: epsilon %g outside [0, 1]", r.epsilon);
  COBEL_REQUIRE(!(r.flags & COBEL_F_MASK_ACTIONS) || r.action_mask, COBEL_E_ARG,
                "cobel_sfma_run: mask_actions set without an action mask");
  COBEL_REQUIRE(!(r.sfma_flags & COBEL_SF_RANDOM) || r.random_cdf, COBEL_E_ARG,
                "cobel_sfma_run: random replay needs random_cdf");
  COBEL_REQUIRE(!(r.sfma_flags & COBEL_SF_RECENCY) || (r.recency_tab && r.recency_len > 0),
                COBEL_E_ARG, "cobel_sfma_run: recency needs recency_tab");
  COBEL_REQUIRE(!r.replay_trace || (r.trace_len && r.trace_cap > 0), COBEL_E_ARG,
                "cobel_sfma_run: replay_trace needs trace_len and trace_cap");
  const int S = world->n_states;
  int32_t lds = 0;
  const int rc = cobel_sfma_query(S, &lds);
  if (rc != COBEL_OK) return rc;
  if (r.n == 0) return COBEL_OK;
  sfma_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = S;
  A.n_worlds = world->n_worlds;
  A.chunk = (4 * S + 63) / 64;
  A.r = r;
  const cobel_eps_consts ec = cobel_make_eps_consts(r.epsilon);
  for (int k = 0; k < 5; ++k) {
    A.eps.base[k] = ec.base[k];
    A.eps.bonus[k] = ec.bonus[k];
  }
  for (int k = 0; k < 16; ++k)
    for (int j = 0; j < 3; ++j) A.eps_thr[k][j] = ec.thr[k][j];
  A.alpha_f = (float)r.alpha;
  A.gamma_f = (float)r.gamma;
  A.model_lr_f = (float)r.model_lr;
  A.succ_off = world->succ_off;
  A.succ_state = world->succ_state;
  A.succ_cdf = world->succ_cdf;
  const uint32_t special = COBEL_SF_RECENCY | COBEL_SF_C_NORMALIZE | COBEL_SF_D_NORMALIZE |
                           COBEL_SF_DETERMINISTIC;
  const bool plain = !(r.sfma_flags & special) && (r.sfma_flags & COBEL_SF_R_NORMALIZE) &&
                     !(r.flags & COBEL_F_FORCE_WAVE);
  hipStream_t st = (hipStream_t)stream;
  if (plain && A.chunk <= 2) {
    A.chunk = 2;
    const uint32_t slow_sf = COBEL_SF_RANDOM | COBEL_SF_DYNAMIC | COBEL_SF_START_REPLAY |
                             COBEL_SF_REWARD_MOD_LOCAL | COBEL_SF_REWARD_MOD | COBEL_SF_STATE_MOD;
    const bool fast = (r.flags & COBEL_F_LEARN) &&
                      !(r.flags & (COBEL_F_NO_REPLAY | COBEL_F_TEST_STREAM)) &&
                      !(r.sfma_flags & slow_sf) && r.nb_replays == 1 && r.decay_strength == 1.0 &&
                      r.beta >= 0.0 && r.beta <= 700.0 && r.r_threshold >= 0.0 &&
                      !r.last_exp && !r.occupancy && !r.replay_trace && !r.lat_trace &&
                      !world->succ_off;
    if (fast) {
      hipLaunchKernelGGL(k_sfma_2_fast, dim3(A.r.n), dim3(64), sfma_lds_bytes(kFastStates), st, A);
      COBEL_HIP_TRY(hipGetLastError());
      return COBEL_OK;
    }
    return launch_sfma<2>(A, (size_t)lds, st);
  }
  if (plain && A.chunk <= 4) {
    A.chunk = 4;
    return launch_sfma<4>(A, (size_t)lds, st);
  }
  if (plain && A.chunk <= 8) {
    A.chunk = 8;
    return launch_sfma<8>(A, (size_t)lds, st);
  }
  // the general kernel: one wave per instance up to 800 experiences, four waves beyond
  // (measured: 14x14 = 784 experiences 9.3e7 vs 7.9e7 reactivations/s in favour of one wave,
  // 20x20 = 1 600 experiences 6.4e6 vs 5.2e7 in favour of four)
  // (COBEL_F_NO_PREFETCH, otherwise unused here, pins the one-wave form for tests)
  if (4 * S > 800 && !(r.flags & COBEL_F_NO_PREFETCH)) {
    A.chunk = (4 * S + 255) / 256;
    return launch_sfma_wg(A, (size_t)lds, st);
  }
  return launch_sfma<0>(A, (size_t)lds, st);
}
