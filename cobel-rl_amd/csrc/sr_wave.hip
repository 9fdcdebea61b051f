// Successor-representation agent, sparse-reward form — ONE WAVEFRONT per agent-env instance.
//
// retrieve_q (sr.py:302-306) evaluates V[j] = sum_k SR[j][k] * R[k] for the four rows
// j = T[s][a].  R[k], the agent's reward estimate, is non-zero only for states the agent has
// been rewarded in, and a product with a zero factor is +-0 and leaves every partial sum of
// NumPy's pairwise summation unchanged.  With at most two non-zero estimates
//     V[j] = SR[j][e0] * R[e0]  (+ SR[j][e1] * R[e1])            one product, at most one addition
// is therefore bit for bit what the reference's full row sum returns (up to the sign of an exact
// zero, which no comparison of the epsilon-greedy selection can see), and a step needs FOUR OR
// EIGHT FLOATS of the four value rows instead of 4 x S.  What remains is the row update
// (sr.py:276-284): read SR[ns], write SR[s] — and SR[s] is the row the previous step read as
// SR[ns], so it is carried in registers (S / 64 floats per lane).  Per step: one row read, one row
// written, a handful of 4-byte gathers.  No LDS staging, no workgroup barriers; eight instances
// per SIMD hide the latency of the one row read each of them waits for.
//
// Instances whose reward estimate holds more than two non-zeros (possible only if the caller
// edited `rewards`, since cobel_sr_run sends worlds with more than two rewarded states to the
// row-streaming kernel k_sr in sr.hip) evaluate the full pairwise sum straight from memory —
// slow, exact, and tested.
//
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   agent/sr.py:155-197 (train loop), :267-284 (update), :302-308 (retrieve_q)
// Numerics as in sr.hip: row TD error in float64 rounded once on store, gamma * SR[ns] in
// float32, reward estimate in float32.
#include <stdlib.h>

#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct srw_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  cobel_sr_run_t r;
  cobel_eps_consts eps;
  float alpha_f, gamma_f;
  const cobel_rw_info* rw;   // [n_worlds] rewarded states + NumPy's combine order (KX kernels)
  // NumPy's pairwise sum over S elements (sizes without the 128-element leaf layout; the rare
  // path with more than two non-zero reward estimates): its leaves and the order they combine in
  uint32_t leaf[16];         // lo | len << 16
  uint8_t comb_dst[16], comb_src[16];
  int32_t n_leaves;
  // transition rows that are distributions (cobel_world_set_transitions), STOCH kernels only
  const uint32_t* succ_off;
  const uint16_t* succ_state;
  const double* succ_cdf;
};

__device__ __forceinline__ int t_of(uint64_t row, int k) { return (int)((row >> (16 * k)) & 0xffffu); }
__device__ __forceinline__ uint64_t t_set(uint64_t row, int k, uint32_t v) {
  return (row & ~(0xffffull << (16 * k))) | ((uint64_t)v << (16 * k));
}
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float rlf(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float rflf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}
// A load that is served by L2, never by this CU's L1: single elements of rows that OTHER lanes
// of the wave wrote in an earlier step.
__device__ __forceinline__ float ld_l2(const float* p) {
  return __builtin_bit_cast(
      float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Lane l holds elements (j * 64 + l) * 4 .. + 3 of a row in c[j]: every load / store instruction
// of a row moves 1 KiB of consecutive addresses, and an element is owned by the same lane in
// every row — a row load after a row store is ordered by the program order of one thread.
template <int NV>
struct row_regs {
  float4 c[NV];
};

// (quads: float4 groups a row really has — S / 4; NV * 64 for the sizes the layout fills exactly)
// (odd: the state count is not a multiple of four — rows are then not 16-byte aligned: they are
//  moved one element per lane and instruction, 64 consecutive floats at a time, and live in the
//  registers lane-strided: component c of c[j] = element (4 j + c) * 64 + lane)
template <int NV, bool ANY_S>
__device__ __forceinline__ void load_row(row_regs<NV>& d, const float* __restrict__ src, int lane,
                                         int quads, int S, bool odd) {
  if (ANY_S && odd) {
    // rows that are not 16-byte aligned: component c of c[j] holds element (4 j + c) * 64 + lane, so
    // that every load instruction reads 64 consecutive floats (one pass over each cache line)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      // (unsigned 32-bit element offsets from the wave-uniform row address: nothing 64-bit and
      //  per-lane for the compiler to hoist out of the step loop and spill)
      const uint32_t e = (uint32_t)((4 * j) * 64 + lane), n = (uint32_t)S;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < n) v.x = src[e];
      if (e + 64u < n) v.y = src[e + 64u];
      if (e + 128u < n) v.z = src[e + 128u];
      if (e + 192u < n) v.w = src[e + 192u];
      d.c[j] = v;
    }
    return;
  }
  // (a wave-uniform base and a 32-bit lane offset, as above: `+ lane` on the pointer is a 64-bit
  //  per-lane value that gets hoisted out of the step loop, spilled, and reloaded in every step)
  const float4* const p = reinterpret_cast<const float4*>(src);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (!ANY_S || j * 64 + lane < quads) d.c[j] = p[(uint32_t)(j * 64 + lane)];
    else d.c[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// The leaves of np.sum over n float32 values and the order their sums combine in (numpy/core/src/
// umath/loops_utils.h: ranges of more than 128 elements are split at n / 2 rounded down to a
// multiple of 8; a leaf runs eight accumulators over its multiple-of-eight part, combines them
// ((0+1)+(2+3))+((4+5)+(6+7)) and adds the remaining elements one by one).  Leaves of a split range
// are at least 64 long: at most 16 of them up to 1 024 elements.
struct leaf_plan {
  srw_args* A;
  int build(int lo, int len) {
    if (len <= 128) {
      const int l = A->n_leaves++;
      A->leaf[l & 15] = (uint32_t)lo | ((uint32_t)len << 16);
      return l;
    }
    int n2 = len / 2;
    n2 -= n2 % 8;
    const int left = build(lo, n2);
    const int right = build(lo + n2, len - n2);
    const int t = n_comb++;
    A->comb_dst[t & 15] = (uint8_t)left;
    A->comb_src[t & 15] = (uint8_t)right;
    return left;
  }
  int n_comb = 0;
};

// Element e of a row held in registers, as a wave-uniform value (e wave-uniform).
template <int NV>
__device__ __forceinline__ float row_element(const row_regs<NV>& r, int e) {
  const int j = e >> 8, comp = e & 3, owner = (e >> 2) & 63;
  float v = 0.0f;
#pragma unroll
  for (int jj = 0; jj < NV; ++jj) {
    const float4 c = r.c[jj];
    const float x = comp == 0 ? c.x : (comp == 1 ? c.y : (comp == 2 ? c.z : c.w));
    v = jj == j ? x : v;
  }
  return rlf(v, owner);
}

// The kernel arguments as they lie in the kernarg segment, through a pointer the optimizer cannot
// see through: what is read through it is loaded where it is used (trial ends, masked selection,
// the final save) instead of being held in scalar registers across the step loop, which is short
// of them (about sixty wave-uniform values are live in it).
// (in the CONSTANT address space — round 4: what is read through it is a scalar load and
//  wave-uniform to the compiler; through a generic pointer the loads were flat loads into vector
//  registers, per-lane as far as the compiler knows)
typedef const __attribute__((address_space(4))) srw_args* srw_kargs;
__device__ __forceinline__ srw_kargs rare_args() {
#if defined(__HIP_DEVICE_COMPILE__)
  srw_kargs p = (srw_kargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
#else
  return nullptr;
#endif
}

// NV = S / 256 float4 per lane (S = 256, 512, 1024).  OCC: visit counts in LDS.  PSETS: per-
// instance hyper-parameters.  Six waves per SIMD (80 registers): 24 rows of 4 KiB in flight per
// CU, more than the ~64 KB per CU that 8 TB/s at 2 us of latency take.
// ANY_S: any state count up to NV * 256: a multiple of four (rows are float4 streams; the register
// layout is the same, lanes past the end of a row hold zeros and neither load nor store) or, ODD,
// any other (rows move by element, see load_row).
// KX: worlds with three to eight rewarded states.  The agent's reward estimate can only be non-zero
// at those states, so V[j] = sum_k SR[j][k] R[k] is a sum of at most eight products — added in the
// grouping NumPy's pairwise summation gives exactly these positions (cobel_rw_info, built on the
// host per world; products whose estimate is still zero are added like the others: x + 0 = x).
// Lane 8a + n gathers element pos[n] of value row a, holds R[pos[n]] and its product; the k - 1
// additions run as shuffles inside the groups of eight lanes.  The values of the row a step
// rewrites are read back from an LDS copy of the new row.
// Registers: 80 (six waves per SIMD) everywhere except the instantiation that spilled most at that
// cap — rows of 769 .. 1 023 states (ANY_S) with the three-to-eight-rewards form: 44 B of scratch,
// 18 scratch accesses inside the step loop.  At 96 registers it has none; measured on 31 x 31 and
// 30 x 31 worlds with three rewarded states 3.22 -> 2.88 and 3.15 -> 2.81 ms per launch (16 384
// instances x 128 steps).  The 32-slot form and the narrower rows measured 4-7 % SLOWER at five
// waves (scripts/experiments/exp_sr_small.py) and keep six.
// STOCH: the world's transition rows are distributions — the successor is drawn in the step
// (one float64 of the env stream per step, as interface/gridworld.py:119-123 draws it) and its
// world record fetched behind the draw; nothing else of the step knows the difference.
template <int NV, bool OCC, bool PSETS, bool ANY_S, int KX, bool ODD, bool STOCH = false>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu((NV == 4 && ANY_S && KX == 8) ? 5 : 6, 8))) void k_sr_wave(
    const srw_args A) {
  __shared__ uint64_t thr[48];
  // The row written in the previous step, kept on chip (4 KiB at 32 x 32): a step straight back — a
  // quarter of a random walk's moves — reads SR[ns] from here instead of HBM, single elements of
  // that row (value gathers) come from here instead of waiting for the store to reach L2, and the
  // values of the row a step rewrites are read back from it.
  __shared__ __attribute__((aligned(16))) float frow[NV * 256];
  __shared__ uint8_t sched32[64];   // KX == 32: dst[31] | src[31] of the world's combine order
  extern __shared__ __attribute__((aligned(16))) uint32_t occ[];   // [S] if OCC
  const int S = ANY_S ? A.S : NV * 256;
  const int quads = S >> 2;
  constexpr bool odd = ANY_S && ODD;   // (a state count that is not a multiple of four)
  // where element e of the row kept in `frow` (an image of the row registers) sits
  auto fpos = [&](int e) -> int {
    return odd ? ((((e >> 8) * 64 + (e & 63)) * 4) + ((e >> 6) & 3)) : e;
  };
  constexpr int NL = NV * 2;   // !ANY_S: leaves of NumPy's pairwise sum, all 128 long
  const int lane = (int)threadIdx.x;
  const int i = (int)blockIdx.x;
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  float* const SRg = A.r.sr + (size_t)i * S * S;
  uint16_t* const Tg = A.r.trans + (size_t)i * S * 4;
  const uint64_t* const T8 = reinterpret_cast<const uint64_t*>(Tg);
  float* const Rg = A.r.rewards + (size_t)i * S;

  const cobel_param_set_t* P = nullptr;
  if (PSETS) {
    const int k = (int)A.r.param_index[i];
    P = A.r.param_sets + (k < A.r.n_param_sets ? k : A.r.n_param_sets - 1);
  }
  if (lane < 48) thr[lane] = PSETS ? P->eps_thr[lane / 3][lane % 3] : A.eps.thr[lane / 3][lane % 3];
  if (OCC)
    for (int e = lane; e < S; e += 64) occ[e] = 0u;

  // ---- the non-zero reward estimates of this instance --------------------------------------
  int nz = 0, e0 = 0, e1 = 0;
  float r0 = 0.0f, r1 = 0.0f;
  bool dense = false;
  // KX == 8: slot n = lane & 7 of each group of eight lanes (lane 8a + n: row a)
  // KX == 32 (nine to 32 rewarded states): groups of sixteen lanes, lane 16a + n holds slots n and
  // n + 16 of row a; the combine order (up to 31 additions) is read from LDS step by step
  int Kw = 0, kroot = 0;
  uint32_t sched_d = 0u, sched_s = 0u;   // KX == 8: 3 bits per step
  uint32_t elane = 0u, elane_hi = 0u;
  float rlane = 0.0f, rlane_hi = 0.0f;
  float gv_hi = 0.0f;                    // KX == 32: the gathered value of the lane's second slot
  int nzc = 0;
  constexpr int GW = KX == 32 ? 16 : 8;  // lanes per row group
  if (KX) {
    const cobel_rw_info* const I = A.rw + world;
    Kw = (int)I->k;
    kroot = (int)I->root;
    if (KX == 32) {
      if (lane < 31) {
        sched32[lane] = I->dst[lane];
        sched32[32 + lane] = I->src[lane];
      }
      if ((lane & 15) + 16 < Kw) elane_hi = (uint32_t)I->pos[(lane & 15) + 16];
    } else {
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        sched_d |= (uint32_t)I->dst[t] << (3 * t);
        sched_s |= (uint32_t)I->src[t] << (3 * t);
      }
    }
    if ((lane & (GW - 1)) < Kw) elane = (uint32_t)I->pos[lane & (GW - 1)];
  }
  {
    const float4* const R4 = reinterpret_cast<const float4*>(Rg) + lane;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (odd) {
        const int e = (j * 64 + lane) * 4;
        if (e + 0 < S) v.x = Rg[e + 0];
        if (e + 1 < S) v.y = Rg[e + 1];
        if (e + 2 < S) v.z = Rg[e + 2];
        if (e + 3 < S) v.w = Rg[e + 3];
      } else if (!ANY_S || j * 64 + lane < quads) {
        v = R4[j * 64];
      }
#pragma unroll
      for (int comp = 0; comp < 4; ++comp) {
        const float x = comp == 0 ? v.x : (comp == 1 ? v.y : (comp == 2 ? v.z : v.w));
        unsigned long long m = __ballot(x != 0.0f);
        if (KX) {
          nzc += __popcll(m);
          m = 0ull;
        }
        while (m) {
          const int l = __builtin_ctzll(m);
          m &= m - 1;
          const int e = (j * 64 + l) * 4 + comp;
          const float val = rlf(x, l);
          if (nz == 0) { e0 = e; r0 = val; }
          else if (nz == 1) { e1 = e; r1 = val; }
          else dense = true;
          nz += 1;
        }
      }
    }
    if (dense) nz = 2;
    if (KX) {
      // every non-zero estimate must sit at one of the world's rewarded states (only a caller's
      // edit of `rewards` can break that): otherwise the full sums from memory
      if ((lane & (GW - 1)) < Kw) rlane = Rg[elane];
      int listed = __popcll(__ballot(lane < GW && rlane != 0.0f));
      if (KX == 32) {
        if ((lane & 15) + 16 < Kw) rlane_hi = Rg[elane_hi];
        listed += __popcll(__ballot(lane < 16 && rlane_hi != 0.0f));
      }
      dense = nzc != listed;
    }
  }
  __syncthreads();   // (one wave: orders the LDS writes above)

  int32_t* const inst = A.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = (int)rfl((uint32_t)inst[COBEL_I_STATE]);
  int step = (int)rfl((uint32_t)inst[COBEL_I_STEP]);
  int trial = (int)rfl((uint32_t)inst[COBEL_I_TRIAL]);
  uint32_t ce = rfl((uint32_t)inst[COBEL_I_CTR_ENV]);
  uint32_t cp = rfl((uint32_t)inst[COBEL_I_CTR_POLICY]);
  uint32_t iflags = rfl((uint32_t)inst[COBEL_I_FLAGS]);
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);

  const uint32_t flags = A.r.flags;
  const bool learn = flags & COBEL_F_LEARN;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (flags & COBEL_F_MASK_ACTIONS) ? A.r.action_mask : nullptr;
  const uint64_t seed = A.r.seed;
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  const double alpha = PSETS ? P->alpha : A.r.alpha, gamma = PSETS ? P->gamma : A.r.gamma;
  const float alpha_f = PSETS ? P->alpha_f : A.alpha_f, gamma_f = PSETS ? P->gamma_f : A.gamma_f;

  uint32_t cw0 = 0, cw1 = 0;
  uint4 cand = {0, 0, 0, 0};
  uint32_t mask_cur = 15u;
  uint64_t tcur = 0;     // T[state][0..3], carried from step to step
  uint64_t tleft = 0;    // the row of the state just left, after that step's write
  int left_state = -1;
  int lds_row = -1;      // the state whose row, as written in the last step, is in `frow`
  row_regs<NV> cur;      // SR[state] (learning runs)
#pragma unroll
  for (int j = 0; j < NV; ++j) cur.c[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;   // V[T[state][a]]
  cobel_u4 pblk = {0, 0, 0, 0};
  uint32_t pb_idx = ~0u;
  uint32_t rows_read = 0, gathers = 0;

  auto load_trow = [&](int s) -> uint64_t {
    const uint64_t v = __hip_atomic_load(T8 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | (uint64_t)rfl((uint32_t)v);
  };

  // The value gathers of the rows tq = T[.][0..3]: lane 2a + n reads element e_n of row tq[a],
  // unless that row is `fresh` (being rewritten in this step: its values come from registers).
  auto issue_gathers = [&](uint64_t tq, int fresh) -> float {
    float gv = 0.0f;
    if (lane < 8) {
      const int n = lane & 1;
      const int j = t_of(tq, lane >> 1);
      // (the one row whose store may still be on its way to L2 is the one in `frow`)
      if (n < nz && j != fresh) {
        if (j == lds_row) gv = frow[fpos(n ? e1 : e0)];
        else gv = ld_l2(SRg + ((uint32_t)j * (uint32_t)S + (uint32_t)(n ? e1 : e0)));
      }
    }
    gathers += (uint32_t)(4 * nz);
    return gv;
  };
  auto assemble = [&](uint64_t tq, int fresh, float f0, float f1, float gv) {
    float qv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bool is_fresh = t_of(tq, a) == fresh;
      const float s0 = is_fresh ? f0 : rlf(gv, 2 * a);
      const float s1 = is_fresh ? f1 : rlf(gv, 2 * a + 1);
      const float p0 = s0 * r0;
      const float p1 = s1 * r1;
      const float two = p0 + p1;
      qv[a] = nz == 0 ? 0.0f : (nz == 1 ? p0 : two);
    }
    q0 = qv[0]; q1 = qv[1]; q2 = qv[2]; q3 = qv[3];
  };
  // KX: lane 8a + n reads element pos[n] of row tq[a]
  auto issue_gathers_x = [&](uint64_t tq, int fresh) -> float {
    float gv = 0.0f;
    if (KX == 32) {
      const int j = t_of(tq, lane >> 4);
      gv_hi = 0.0f;
      if (j != fresh) {
        if ((lane & 15) < Kw)
          gv = j == lds_row ? frow[fpos((int)elane)]
                            : ld_l2(SRg + ((uint32_t)j * (uint32_t)S + elane));
        if ((lane & 15) + 16 < Kw)
          gv_hi = j == lds_row ? frow[fpos((int)elane_hi)]
                               : ld_l2(SRg + ((uint32_t)j * (uint32_t)S + elane_hi));
      }
      gathers += (uint32_t)(4 * Kw);
      return gv;
    }
    const int j = t_of(tq, (lane >> 3) & 3);
    if (lane < 32 && (lane & 7) < Kw && j != fresh) {
      if (j == lds_row) gv = frow[fpos((int)elane)];
      // (a 32-bit element offset from the wave-uniform table address: a 64-bit per-lane address
      //  SRg + elane is loop-invariant, was hoisted, spilled at the 80-register cap and reloaded
      //  here behind s_waitcnt vmcnt(0) — which made every gather wait for the row load before it)
      else gv = ld_l2(SRg + ((uint32_t)j * (uint32_t)S + elane));
    }
    gathers += (uint32_t)(4 * Kw);
    return gv;
  };
  auto assemble_x = [&](uint64_t tq, int fresh, float gv) {
    if (KX == 32) {
      const bool is_fresh = t_of(tq, lane >> 4) == fresh;
      float f0 = 0.0f, f1 = 0.0f;
      if (is_fresh) {
        if ((lane & 15) < Kw) f0 = frow[fpos((int)elane)];
        if ((lane & 15) + 16 < Kw) f1 = frow[fpos((int)elane_hi)];
      }
      float p0 = (is_fresh ? f0 : gv) * rlane;
      float p1 = (is_fresh ? f1 : gv_hi) * rlane_hi;
      for (int t = 0; t + 1 < Kw; ++t) {
        const int dst = (int)rfl((uint32_t)sched32[t]), src = (int)rfl((uint32_t)sched32[32 + t]);
        const int from = (lane & ~15) | (src & 15);
        const float o0 = __shfl(p0, from), o1 = __shfl(p1, from);
        const float other = (src & 16) ? o1 : o0;
        if ((lane & 15) == (dst & 15)) {
          if (dst & 16) p1 = p1 + other;
          else p0 = p0 + other;
        }
      }
      const float pr = (kroot & 16) ? p1 : p0;
      q0 = rlf(pr, kroot & 15);
      q1 = rlf(pr, 16 + (kroot & 15));
      q2 = rlf(pr, 32 + (kroot & 15));
      q3 = rlf(pr, 48 + (kroot & 15));
      return;
    }
    const bool is_fresh = t_of(tq, (lane >> 3) & 3) == fresh;
    float fv = 0.0f;
    if (is_fresh && lane < 32 && (lane & 7) < Kw) fv = frow[fpos((int)elane)];
    float p = (is_fresh ? fv : gv) * rlane;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      if (t < Kw - 1) {
        const int dst = (int)((sched_d >> (3 * t)) & 7u), src = (int)((sched_s >> (3 * t)) & 7u);
        const float other = __shfl(p, (lane & ~7) | src);
        if ((lane & 7) == dst) p = p + other;
      }
    }
    q0 = rlf(p, kroot);
    q1 = rlf(p, 8 + kroot);
    q2 = rlf(p, 16 + kroot);
    q3 = rlf(p, 24 + kroot);
  };
  // Full pairwise sums from memory (more than two non-zero reward estimates): lane (l, k) runs
  // accumulator k of leaf l, then the butterflies of NumPy's combine order (as k_sr does in LDS).
  auto dense_values = [&](uint64_t tq) {
    wait_vm0();
    if constexpr (ANY_S) {
      // Leaf L (of at most 16) belongs to the eight lanes 8 (L & 7) .. + 7 in pass L >> 3, one
      // accumulator each; its sum is then kept by lane 8 (L & 7) + (L >> 3), and the sums combine
      // in the order of the plan (the result ends up with leaf 0 = lane 0).
      const srw_kargs R = rare_args();
      const int nl = R->n_leaves;
      const int k = lane & 7;
      for (int a = 0; a < 4; ++a) {
        const float* const row = SRg + (size_t)t_of(tq, a) * S;
        float p = 0.0f;
        for (int half = 0; half < 2; ++half) {
          const int L = half * 8 + (lane >> 3);
          int lo = 0, len = 0;
          if (L < nl) {
            const uint32_t w = R->leaf[L];
            lo = (int)(w & 0xffffu);
            len = (int)(w >> 16);
          }
          const int len8 = len & ~7;
          float acc = 0.0f;
          if (len8) {
            acc = ld_l2(row + lo + k) * ld_l2(Rg + lo + k);
            for (int e = 8 + k; e < len8; e += 8) {
              const float prod = ld_l2(row + lo + e) * ld_l2(Rg + lo + e);
              acc = acc + prod;
            }
          }
          acc = acc + __shfl_xor(acc, 1);
          acc = acc + __shfl_xor(acc, 2);
          acc = acc + __shfl_xor(acc, 4);
          float res = len8 ? acc : 0.0f;
          for (int e = len8; e < len; ++e) {
            const float prod = ld_l2(row + lo + e) * ld_l2(Rg + lo + e);
            res = res + prod;
          }
          if (k == half) p = res;
        }
        for (int t = 0; t + 1 < nl; ++t) {
          const int d = (int)R->comb_dst[t], sidx = (int)R->comb_src[t];
          const float other = __shfl(p, 8 * (sidx & 7) + (sidx >> 3));
          if (lane == 8 * (d & 7) + (d >> 3)) p = p + other;
        }
        const float v = rlf(p, 0);
        q0 = a == 0 ? v : q0;
        q1 = a == 1 ? v : q1;
        q2 = a == 2 ? v : q2;
        q3 = a == 3 ? v : q3;
      }
      rows_read += 4u;
      return;
    }
    float qv[4];
    const int l = lane >> 3, k = lane & 7;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* const row = SRg + (size_t)t_of(tq, a) * S;
      float acc = 0.0f;
      if (l < NL) {
        const int base = 128 * l + k;
        acc = ld_l2(row + base) * ld_l2(Rg + base);
        for (int o = 8; o < 128; o += 8) {
          const float prod = ld_l2(row + base + o) * ld_l2(Rg + base + o);
          acc = acc + prod;
        }
      }
      acc = acc + __shfl_xor(acc, 1);
      acc = acc + __shfl_xor(acc, 2);
      acc = acc + __shfl_xor(acc, 4);
#pragma unroll
      for (int o = 8; o < 8 * NL; o <<= 1) acc = acc + __shfl_xor(acc, o);
      qv[a] = rflf(acc);
    }
    rows_read += 4u;
    q0 = qv[0]; q1 = qv[1]; q2 = qv[2]; q3 = qv[3];
  };
  auto enter_state = [&](int s) {
    const uint4 c = W4[s];
    cw0 = rfl(c.x);
    cw1 = rfl(c.y);
    if (!STOCH && lane < 4) cand = W4[next_of(cw0, cw1, lane)];
    mask_cur = amask ? (uint32_t)amask[s] & 15u : 15u;
    tcur = load_trow(s);
    left_state = -1;
    if (learn) {
      if (s == lds_row) {
#pragma unroll
        for (int j = 0; j < NV; ++j) cur.c[j] = reinterpret_cast<const float4*>(frow)[j * 64 + lane];
      } else {
        load_row<NV, ANY_S>(cur, SRg + (size_t)s * S, lane, quads, S, odd);
        rows_read += 1u;
      }
    }
    if (dense) {
      dense_values(tcur);
    } else if (KX) {
      const float gv = issue_gathers_x(tcur, -1);
      assemble_x(tcur, -1, gv);
    } else {
      const float gv = issue_gathers(tcur, -1);
      assemble(tcur, -1, 0.0f, 0.0f, gv);
    }
  };
  if (iflags & 1u) enter_state(state);

  int budget = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;
  uint32_t executed = 0;

  while (true) {
    if (!(iflags & 1u)) {
      if (trial >= A.r.trials_target) break;
      state = (int)A.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                               start_cnt)];
      state = (int)rfl((uint32_t)state);
      ce += 1u;
      step = 0;
      trew = 0.0;
      iflags |= 1u;
      enter_state(state);
    }
    if (budget == 0) break;
    budget -= 1;

    // ---- select (greedy.py:40-88) + env.step (gridworld.py:115-126) -------------------------
    if ((cp >> 1) != pb_idx) {
      pb_idx = cp >> 1;
      pblk = cobel_philox(pb_idx, 0u, g, pol_stream, seed);
    }
    const uint32_t w0 = (cp & 1u) ? pblk.z : pblk.x;
    const uint32_t w1 = (cp & 1u) ? pblk.w : pblk.y;
    cp += 1u;
    int a;
    if (mask_cur == 15u) {
      a = (int)rfl((uint32_t)cobel_eps_greedy_select_thr(q0, q1, q2, q3, cobel_u53(w0, w1), thr,
                                                          lane));
    } else {
      cobel_eps_bb ebb;
      const srw_kargs R = rare_args();
      ebb.base[0] = ebb.bonus[0] = 0.0;
#pragma unroll
      for (int n = 1; n <= 4; ++n) {
        ebb.base[n] = PSETS ? P->eps_base[n] : R->eps.base[n];
        ebb.bonus[n] = PSETS ? P->eps_bonus[n] : R->eps.bonus[n];
      }
      a = (int)rfl((uint32_t)cobel_eps_greedy_select_wave(q0, q1, q2, q3, mask_cur,
                                                           cobel_u01(w0, w1), ebb, lane));
    }
    int ns;
    uint32_t nw0, nw1, end;
    float r;
    if (STOCH) {
      const double ue = cobel_draw_u01(ce, COBEL_SUB_DOUBLE, g, COBEL_STREAM_ENV, seed);
      ce += 1u;
      ns = (int)rfl((uint32_t)cobel_draw_successor(
          A.succ_off, A.succ_state, A.succ_cdf, ((size_t)world * S + (size_t)state) * 4 + a, ue));
      const uint4 c = W4[ns];
      nw0 = rfl(c.x);
      nw1 = rfl(c.y);
      r = __builtin_bit_cast(float, rfl(c.z));
      end = rfl(c.w);
    } else {
      ns = (int)next_of(cw0, cw1, a);
      nw0 = rl(cand.x, a);
      nw1 = rl(cand.y, a);
      r = __builtin_bit_cast(float, rl(cand.z, a));
      end = rl(cand.w, a);
    }
    const uint32_t nt = 1u - end;
    const bool trial_over = end || (step + 1 >= A.r.steps_per_trial);
    // T[ns][.] as it stands before this step's write
    // (the 8-byte load is issued first and waited for only after the row of ns has been requested
    //  behind it: loads return in order, so its latency hides in the row's)
    uint64_t tnxt = tcur, traw = 0;
    const bool t_load = !trial_over && ns != state && ns != left_state;
    if (!trial_over && ns == left_state && ns != state) tnxt = tleft;
    if (t_load) traw = __hip_atomic_load(T8 + ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // SR[ns] (sr.py:276-281): loaded, or — after a bump (ns == state) — the row in hand
    row_regs<NV> nxt = cur;
    if (learn && nt != 0u && ns != state) {
      if (ns == lds_row) {   // a step straight back: the row written in the last step
#pragma unroll
        for (int j = 0; j < NV; ++j) nxt.c[j] = reinterpret_cast<const float4*>(frow)[j * 64 + lane];
      } else {
        load_row<NV, ANY_S>(nxt, SRg + (size_t)ns * S, lane, quads, S, odd);
        rows_read += 1u;
      }
    }
    if (t_load) tnxt = ((uint64_t)rfl((uint32_t)(traw >> 32)) << 32) | (uint64_t)rfl((uint32_t)traw);

    float f0 = 0.0f, f1 = 0.0f, gv = 0.0f;
    uint64_t tq = tnxt;   // the rows whose values the next step needs
    if (learn) {
      // sr.py:272-274 (float32): rewards[ns] += (r - rewards[ns]) * lr; transitions[s][a] = ns
      float old;
      unsigned long long slot_m = 0ull, slot_hi = 0ull;
      if (dense) old = rflf(ld_l2(Rg + ns));
      else if (KX == 32) {
        slot_m = __ballot(lane < 16 && lane < Kw && elane == (uint32_t)ns);
        slot_hi = __ballot(lane < 16 && lane + 16 < Kw && elane_hi == (uint32_t)ns);
        old = slot_m ? rlf(rlane, __builtin_ctzll(slot_m))
                     : (slot_hi ? rlf(rlane_hi, __builtin_ctzll(slot_hi)) : 0.0f);
      }
      else if (KX) {
        slot_m = __ballot((lane & 7) < Kw && elane == (uint32_t)ns);
        old = slot_m ? rlf(rlane, __builtin_ctzll(slot_m)) : 0.0f;
      }
      else old = (nz > 0 && ns == e0) ? r0 : ((nz > 1 && ns == e1) ? r1 : 0.0f);
      const float d = r - old;
      const float upd = old + d * alpha_f;
      if (lane == 0) {
        Rg[ns] = upd;
        Tg[state * 4 + a] = (uint16_t)ns;
      }
      if (KX) {
        if (!dense) {
          if (KX == 32 && (slot_m || slot_hi)) {
            if ((lane & 15) < Kw && elane == (uint32_t)ns) rlane = upd;
            if ((lane & 15) + 16 < Kw && elane_hi == (uint32_t)ns) rlane_hi = upd;
          } else if (KX != 32 && slot_m) {
            if ((lane & 7) < Kw && elane == (uint32_t)ns) rlane = upd;
          } else if (upd != 0.0f) {
            dense = true;   // (a non-zero estimate outside the world's rewarded states)
          }
        }
      } else if (!dense) {
        if (nz > 0 && ns == e0) r0 = upd;
        else if (nz > 1 && ns == e1) r1 = upd;
        else if (upd != 0.0f) {
          if (nz == 0) { e0 = ns; r0 = upd; nz = 1; }
          else if (nz == 1) { e1 = ns; r1 = upd; nz = 2; }
          else dense = true;
        }
      }
      tcur = t_set(tcur, a, (uint32_t)ns);
      if (ns == state) tq = tcur;
    }
    const int fresh = learn ? state : -1;
    if (!trial_over && !dense) gv = KX ? issue_gathers_x(tq, fresh) : issue_gathers(tq, fresh);

    if (learn) {
      // sr.py:276-284: td = e_s + gamma * (SR[ns] | e_ns) - SR[s];  SR[s] += lr * td
      const bool need_ns = nt != 0u;
      // Everything issued before this point has landed: the gathers, the row of ns, and the row
      // store of the previous step.
      wait_vm0();
      float4* const out = reinterpret_cast<float4*>(SRg + (size_t)state * S);   // (wave-uniform)
      const bool want_fresh =
          !trial_over && !dense && (t_of(tq, 0) == state || t_of(tq, 1) == state ||
                                    t_of(tq, 2) == state || t_of(tq, 3) == state);
      auto upd1 = [&](int e, float cs, float cn) -> float {
        double td = (e == state) ? 1.0 : 0.0;
        if (need_ns) {
          const float gs = gamma_f * cn;
          td = td + (double)gs;
        } else {
          td = td + gamma * ((e == ns) ? 1.0 : 0.0);
        }
        td = td - (double)cs;
        return (float)((double)cs + alpha * td);
      };
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 cs4 = cur.c[j];
        const float4 cn4 = nxt.c[j];
        // elements of the four components: consecutive, or (odd sizes) 64 apart
        const int e = odd ? (4 * j) * 64 + lane : (j * 64 + lane) * 4;
        const int de = odd ? 64 : 1;
        float4 o4;
        o4.x = upd1(e, cs4.x, cn4.x);
        o4.y = upd1(e + de, cs4.y, cn4.y);
        o4.z = upd1(e + 2 * de, cs4.z, cn4.z);
        o4.w = upd1(e + 3 * de, cs4.w, cn4.w);
        if (odd) {
          float* const o1 = SRg + (size_t)state * S;   // (wave-uniform; 32-bit offsets, as load_row)
          const uint32_t ue = (uint32_t)e, n = (uint32_t)S;
          if (ue < n) o1[ue] = o4.x;
          if (ue + 64u < n) o1[ue + 64u] = o4.y;
          if (ue + 128u < n) o1[ue + 128u] = o4.z;
          if (ue + 192u < n) o1[ue + 192u] = o4.w;
        } else if (!ANY_S || j * 64 + lane < quads) {
          out[(uint32_t)(j * 64 + lane)] = o4;
        }
        reinterpret_cast<float4*>(frow)[j * 64 + lane] = o4;
        // the row the next step starts from: SR[ns], or the row just written after a bump
        // (component by component: a choice between two float4 objects is compiled into a choice
        //  between two addresses in scratch memory)
        const bool bump = ns == state;
        cur.c[j] = make_float4(bump ? o4.x : cn4.x, bump ? o4.y : cn4.y, bump ? o4.z : cn4.z,
                               bump ? o4.w : cn4.w);
      }
      lds_row = state;
      if (!KX && want_fresh) {   // the new row's elements e0 / e1, wave-uniform
        if (nz > 0) f0 = rflf(frow[fpos(e0)]);
        if (nz > 1) f1 = rflf(frow[fpos(e1)]);
      }
    }

    if (A.r.step_budget == 1 && rare_args()->r.last_exp && lane == 0) {
      int32_t* const e = rare_args()->r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = 0;
    }
    trew += (double)r;
    executed += 1u;
    if (OCC && lane == 0) occ[ns] += 1u;
    if (ns != state) {   // (a bumping move stays in the row just updated)
      tleft = tcur;
      left_state = state;
      tcur = tnxt;
    }
    state = ns;
    cw0 = nw0;
    cw1 = nw1;
    if (!trial_over) {
      if (!STOCH && lane < 4) cand = W4[next_of(cw0, cw1, lane)];
      mask_cur = amask ? (uint32_t)amask[state] & 15u : 15u;
      step += 1;
      if (dense) dense_values(tq);
      else if (KX) assemble_x(tq, fresh, gv);
      else assemble(tq, fresh, f0, f1, gv);
    } else {
      const srw_kargs RR = rare_args();
#define rr (RR->r)
      if (lane == 0 && trial >= 0 && trial < rr.trial_cap) {
        const size_t m = cobel_mon_offset(rr.mon_stripes, rr.trial_cap) + (size_t)trial;
        if (rr.lat_sum) atomicAdd(rr.lat_sum + m, (unsigned long long)step);
        if (rr.lat_cnt) atomicAdd(rr.lat_cnt + m, 1ull);
        if (rr.reward_sum) atomicAdd(rr.reward_sum + m, trew);
        if (rr.resp_cnt && trew > 0.0) atomicAdd(rr.resp_cnt + m, 1ull);
        if (rr.lat_trace) rr.lat_trace[(size_t)i * rr.trial_cap + trial] = step;
      }
#undef rr
      trial += 1;
      iflags &= ~1u;
    }
  }

  const srw_kargs RR = rare_args();
#define rr (RR->r)
  if (OCC) {
    __syncthreads();
    for (int e = lane; e < S; e += 64) {
      const uint32_t c = occ[e];
      if (c && rr.occupancy) atomicAdd(rr.occupancy + (size_t)world * S + e, (unsigned long long)c);
    }
  }
  if (lane == 0) {
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_FLAGS] = (int32_t)iflags;
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) += (unsigned long long)executed;
    if (rr.steps_done && executed) atomicAdd(rr.steps_done, (unsigned long long)executed);
    if (rr.traffic) {
      atomicAdd(rr.traffic + 0, (unsigned long long)rows_read);
      if (learn) atomicAdd(rr.traffic + 1, (unsigned long long)executed);   // one row per step
      atomicAdd(rr.traffic + 2, (unsigned long long)gathers);
      if (dense) atomicAdd(rr.traffic + 3, 1ull);
    }
  }
}

#undef rr
template <int NV, bool OCC, bool PSETS, bool ANY_S, int KX, bool ODD, bool STOCH = false>
int launch(const srw_args& A, hipStream_t st) {
  size_t lds = OCC ? (size_t)A.S * 4 : 0;
  if (const size_t pad = cobel_debug_lds_pad(lds + 8 * 1024, 160 * 1024)) {   // (occupancy experiments)
    lds += pad;
    if (lds > 48 * 1024)
      COBEL_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&k_sr_wave<NV, OCC, PSETS, ANY_S, KX, ODD, STOCH>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_sr_wave<NV, OCC, PSETS, ANY_S, KX, ODD, STOCH>), dim3(A.r.n), dim3(64), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

template <int NV, bool ANY_S, bool ODD = false>
int launch_nv(const srw_args& A, bool occ, bool psets, int kx, hipStream_t st) {
  if (A.succ_off) {   // (drawn successors: without occupancy counters and parameter sets — covers())
    if (kx == 32) return launch<NV, false, false, ANY_S, 32, ODD, true>(A, st);
    if (kx) return launch<NV, false, false, ANY_S, 8, ODD, true>(A, st);
    return launch<NV, false, false, ANY_S, 0, ODD, true>(A, st);
  }
  if (kx == 32)   // (nine to 32 rewarded states; launch-wide hyper-parameters only)
    return occ ? launch<NV, true, false, ANY_S, 32, ODD>(A, st)
               : launch<NV, false, false, ANY_S, 32, ODD>(A, st);
  if (kx)   // (three to eight rewarded states; launch-wide hyper-parameters only)
    return occ ? launch<NV, true, false, ANY_S, 8, ODD>(A, st)
               : launch<NV, false, false, ANY_S, 8, ODD>(A, st);
  if (psets)
    return occ ? launch<NV, true, true, ANY_S, 0, ODD>(A, st)
               : launch<NV, false, true, ANY_S, 0, ODD>(A, st);
  return occ ? launch<NV, true, false, ANY_S, 0, ODD>(A, st)
             : launch<NV, false, false, ANY_S, 0, ODD>(A, st);
}

}  // namespace

bool cobel_sr_wave_covers(const cobel_world* world, const cobel_sr_run_t& r) {
  const int S = world->n_states;
  // rows are streamed as float4 groups: any multiple of four up to 1 024 states (256 / 512 /
  // 1 024 fill the register layout exactly and take instantiations without bounds checks)
  // (state counts that are not multiples of four: rows move one element per lane and instruction,
  //  the ODD instantiations: 17x17 1.26e9 env-steps/s, 25x25 8.7e8, 31x31 7.2e8 against 3.0 / 2.8 /
  //  2.5e8 of the row-streaming kernel — scripts/experiments/exp_sr_sizes.py)
  // at most two rewarded states (every maze / open field builder of the reference), or up to eight
  // with launch-wide hyper-parameters (the KX kernels)
  const bool rewards_ok = world->max_rewarded_states <= 2 ||
                          (world->max_rewarded_states <= 32 && world->rw && !r.param_index);
  // worlds whose transition rows are distributions: the plain instantiations only
  const bool draws_ok = !world->succ_off || (!r.occupancy && !r.param_index);
  return S >= 2 && S <= 1024 && rewards_ok && draws_ok &&
         !(r.flags & COBEL_F_SR_STREAM_ROWS);
}

// Arguments already checked by cobel_sr_run.
int cobel_sr_wave_launch(const cobel_world* world, const cobel_sr_run_t& r, hipStream_t st) {
  srw_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = world->n_states;
  A.n_worlds = world->n_worlds;
  A.r = r;
  A.eps = cobel_make_eps_consts(r.epsilon);
  A.alpha_f = (float)r.alpha;
  A.gamma_f = (float)r.gamma;
  A.rw = world->rw;
  A.succ_off = world->succ_off;
  A.succ_state = world->succ_state;
  A.succ_cdf = world->succ_cdf;
  A.n_leaves = 0;
  for (int t = 0; t < 16; ++t) A.leaf[t] = 0u, A.comb_dst[t] = A.comb_src[t] = 0;
  leaf_plan plan{&A};
  plan.build(0, world->n_states);
  COBEL_REQUIRE(A.n_leaves <= 16 && plan.n_comb == A.n_leaves - 1, COBEL_E_ARG,
                "cobel_sr_run: pairwise plan out of range");
  const bool occ = r.occupancy != nullptr, psets = r.param_index != nullptr;
  const int kx = world->max_rewarded_states > 8 ? 32 : (world->max_rewarded_states > 2 ? 8 : 0);
  const int S = world->n_states;
  if (S == 256) return launch_nv<1, false>(A, occ, psets, kx, st);
  if (S == 512) return launch_nv<2, false>(A, occ, psets, kx, st);
  if (S == 1024) return launch_nv<4, false>(A, occ, psets, kx, st);
  if (S & 3) {   // rows that are not float4 streams: instantiations of their own
    if (S <= 256) return launch_nv<1, true, true>(A, occ, psets, kx, st);
    if (S <= 512) return launch_nv<2, true, true>(A, occ, psets, kx, st);
    if (S <= 768) return launch_nv<3, true, true>(A, occ, psets, kx, st);
    return launch_nv<4, true, true>(A, occ, psets, kx, st);
  }
  if (S <= 256) return launch_nv<1, true>(A, occ, psets, kx, st);
  if (S <= 512) return launch_nv<2, true>(A, occ, psets, kx, st);
  if (S <= 768) return launch_nv<3, true>(A, occ, psets, kx, st);
  return launch_nv<4, true>(A, occ, psets, kx, st);
}
