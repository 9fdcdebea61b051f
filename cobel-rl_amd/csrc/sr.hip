// Successor-representation agent — one 256-thread workgroup per agent–env instance (gfx950).
//
// The SR matrix (S x S float32, 4 MiB at 32x32) stays in HBM; a step streams whole rows:
//   * wave a (a = 0..3) loads row T[s][a] with coalesced 16-byte loads into LDS and reduces
//     V_a = sum_k SR[T[s][a]][k] * R[k].  The reduction reproduces NumPy's pairwise summation
//     (np.sum(SR * rewards, axis=1), sr.py:302) exactly: per-element float32 products, leaves of
//     <= 128 elements with 8 strided accumulators (8 lanes per leaf), then the halving tree — so
//     the epsilon-greedy tie pattern matches the reference bit for bit for any reward layout;
//   * the rows needed by the TD update (SR[s], SR[ns]) are taken from those four LDS copies when
//     T[s][.] already points at them (always, once (s, a) has been visited) and fetched only
//     otherwise; the updated row is written back with coalesced 16-byte stores;
//   * the reward estimate R (4 B per state) lives in LDS for the whole call and is written through;
//     the agent's transition table T (8 B per state) stays in HBM / L2: the row of the current state
//     travels in registers from step to step and the row of the state being entered is one 8-byte
//     load.  Keeping T, a sixth row buffer and 58 unused leaf sums out of LDS takes an instance from
//     39.1 to 26 KiB: six workgroups per CU instead of four, and the step is latency-bound
//     (measured: 1 / 2 / 3 / 4 workgroups per CU -> 0.65 / 1.23 / 1.71 / 2.10e8 steps/s).
//
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   agent/sr.py:155-197 (train loop), :267-284 (update), :302-308 (retrieve_q)
// Row TD error in float64, one rounding on store (np.eye is float64, sr.py:276-284); the
// gamma * SR[ns] product is float32 (weak Python scalar times a float32 row), gamma * e_ns is
// float64; the reward estimate is updated in float32.
#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

constexpr int kMaxLeaves = 64;
constexpr int kRows = 5;  // four value rows + one spare row (the new SR[s], or SR[s] fetched)

struct sr_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  int32_t leaves;   // leaves of the pairwise-sum tree over S elements (leaf sums kept per wave)
  cobel_sr_run_t r;
  cobel_eps_consts eps;
  float alpha_f, gamma_f;
  // transition rows that are distributions (cobel_world_set_transitions), STOCH kernels only
  const uint32_t* succ_off;
  const uint16_t* succ_state;
  const double* succ_cdf;
};

// Padded LDS position of row element e: 8 floats of padding per 128 keep the four leaves a
// half-wave reads at once on different banks.
__device__ __forceinline__ int phys(int e) { return e + ((e >> 7) << 3); }
__host__ __device__ __forceinline__ int padded(int S) {
  return (S + (((S + 127) >> 7) << 3) + 3) & ~3;
}

template <int MAXL, typename IDX>
struct sr_plan_t {
  uint16_t leaf_start[MAXL];
  uint8_t leaf_len[MAXL];
  IDX op_dst[MAXL], op_src[MAXL];
  typedef IDX idx_t;
  int n_leaves, n_ops;
  int balanced;  // 1: n_leaves in {1, 2, 4, 8}, all leaves 128 long -> butterfly combine
};
typedef sr_plan_t<kMaxLeaves, uint8_t> sr_plan;
// Worlds whose six rows do not fit one workgroup's LDS (more than 6 336 states; BIG kernels): the
// rows are read where they lie, the plan and the leaf sums are all the LDS holds — up to 65 535
// states (the 16-bit state ids of a world handle).
constexpr int kMaxLeavesBig = 1024;
typedef sr_plan_t<kMaxLeavesBig, uint16_t> sr_plan_big;

// NumPy pairwise_sum recursion (n <= 128: leaf; else split at n/2 rounded down to a multiple
// of 8), flattened: leaves in address order plus the post-order list of "dst += src" combines.
template <typename PLAN>
__device__ void build_plan(PLAN* p, int S, int* stack /* 64 ints of LDS scratch */) {
  int* const st_start = stack;
  int* const st_n = stack + 16;
  int* const st_phase = stack + 32;
  int* const st_left = stack + 48;
  int top = 0, nl = 0, no = 0, ret = 0;
  st_start[0] = 0;
  st_n[0] = S;
  st_phase[0] = 0;
  st_left[0] = 0;
  while (top >= 0) {
    const int start = st_start[top], n = st_n[top];
    if (n <= 128) {
      p->leaf_start[nl] = (uint16_t)start;
      p->leaf_len[nl] = (uint8_t)n;
      ret = nl++;
      --top;
      continue;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    if (st_phase[top] == 0) {
      st_phase[top] = 1;
      ++top;
      st_start[top] = start;
      st_n[top] = n2;
      st_phase[top] = 0;
    } else if (st_phase[top] == 1) {
      st_left[top] = ret;
      st_phase[top] = 2;
      ++top;
      st_start[top] = start + n2;
      st_n[top] = n - n2;
      st_phase[top] = 0;
    } else {
      p->op_dst[no] = (typename PLAN::idx_t)st_left[top];
      p->op_src[no] = (typename PLAN::idx_t)ret;
      ++no;
      ret = st_left[top];
      --top;
    }
  }
  p->n_leaves = nl;
  p->n_ops = no;
  p->balanced = (nl == 1 || nl == 2 || nl == 4 || nl == 8) && S == 128 * nl;
}

// V = pairwise_sum_k(row[k] * rw[k]) by one wave; row and rw are in padded LDS layout.
// Returns the value in every lane.
// (PADDED: row and rw in the padded LDS layout; else plain arrays, e.g. in global memory)
template <bool PADDED = true, typename PLAN = sr_plan>
__device__ __forceinline__ float wave_pairwise_dot(const float* row, const float* rw,
                                                   const PLAN* plan, float* leafsum, int lane) {
  auto phys = [](int e) -> int { return PADDED ? e + ((e >> 7) << 3) : e; };
  const int k = lane & 7;
  const int nl = plan->n_leaves;
  for (int l0 = 0; l0 < nl; l0 += 8) {
    const int l = l0 + (lane >> 3);
    float res = 0.0f;
    const bool live = l < nl;
    const int start = live ? plan->leaf_start[l] : 0;
    const int m = live ? plan->leaf_len[l] : 0;
    if (m >= 8) {
      const int body = m - (m & 7);
      int e = start + k;
      float acc = row[phys(e)] * rw[phys(e)];
      for (int o = 8; o < body; o += 8) {
        e = start + o + k;
        const float prod = row[phys(e)] * rw[phys(e)];
        acc = acc + prod;
      }
      res = acc;
    }
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): butterflies over the 8-lane group
    res = res + __shfl_xor(res, 1);
    res = res + __shfl_xor(res, 2);
    res = res + __shfl_xor(res, 4);
    if (live && k == 0) {
      if (m < 8) {
        res = 0.0f;
        for (int e = start; e < start + m; ++e) res = res + row[phys(e)] * rw[phys(e)];
      } else {
        for (int e = start + (m - (m & 7)); e < start + m; ++e)
          res = res + row[phys(e)] * rw[phys(e)];
      }
      leafsum[l] = res;
    }
  }
  __builtin_amdgcn_wave_barrier();
  float v = 0.0f;
  if (plan->balanced) {
    // 2^k equal leaves (S = 128 * 2^k, k <= 3): the halving tree is a butterfly over the leaf
    // index, ((L0+L1)+(L2+L3))+((L4+L5)+(L6+L7)), and every leaf sum is already in a register
    v = leafsum[lane >> 3 < nl ? lane >> 3 : 0];
    for (int o = 8; o < 8 * nl; o <<= 1) v = v + __shfl_xor(v, o);
    return __shfl(v, 0);
  }
  if (lane == 0) {
    for (int o = 0; o < plan->n_ops; ++o)
      leafsum[plan->op_dst[o]] = leafsum[plan->op_dst[o]] + leafsum[plan->op_src[o]];
    v = leafsum[0];
  }
  return __shfl(v, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), which would wait for the row prefetch that is meant to stay in
// flight across the update phase.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool VEC>
__device__ __forceinline__ void load_row(float* dst, const float* __restrict__ src, int S, int t,
                                         int nthreads) {
  if (VEC) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int e = t * 4; e < S; e += nthreads * 4)
      *reinterpret_cast<float4*>(dst + phys(e)) = s4[e >> 2];
  } else {
    for (int e = t; e < S; e += nthreads) dst[phys(e)] = src[e];
  }
}

struct sr_lds {
  float* rw;
  float* rows;
  sr_plan* plan;
  sr_plan_big* plan_big;   // BIG kernels (no rows, no reward vector in LDS)
  float* leafsum;  // [4][LS], LS = leaf_stride(leaves)
  float* V;        // [4]
  uint64_t* thr;   // [48] epsilon-greedy thresholds, entry t * 3 + k (cobel_policy.h)
  uint32_t* occ;   // [S] visit counts (OCC only)
  int PS, LS;
};

// Leaves of NumPy's pairwise sum over n elements, and the per-wave stride of the leaf sums.
int count_leaves(int n) {
  if (n <= 128) return 1;
  int n2 = n / 2;
  n2 -= n2 % 8;
  return count_leaves(n2) + count_leaves(n - n2);
}
__host__ __device__ __forceinline__ int leaf_stride(int leaves) { return (leaves + 3) & ~3; }

__device__ __forceinline__ sr_lds carve(unsigned char* base, int S, int leaves) {
  sr_lds L;
  L.PS = padded(S);
  L.LS = leaf_stride(leaves);
  size_t off = 0;
  L.rows = reinterpret_cast<float*>(base + off);
  off += (size_t)kRows * L.PS * 4;
  L.rw = reinterpret_cast<float*>(base + off);
  off += (size_t)L.PS * 4;
  L.leafsum = reinterpret_cast<float*>(base + off);
  off += (size_t)4 * L.LS * 4;
  L.V = reinterpret_cast<float*>(base + off);   // [4] values + the two words of the action draw
  off += 32;
  L.plan = reinterpret_cast<sr_plan*>(base + off);
  off += (sizeof(sr_plan) + 15) & ~(size_t)15;
  L.thr = reinterpret_cast<uint64_t*>(base + off);
  off += 384;
  L.occ = reinterpret_cast<uint32_t*>(base + off);
  return L;
}

// BIG kernels: leaf sums, values, thresholds, the plan and 64 ints for building it.
__device__ __forceinline__ sr_lds carve_big(unsigned char* base, int S, int leaves) {
  sr_lds L;
  L.PS = S;
  L.LS = leaf_stride(leaves);
  L.rows = nullptr;
  L.rw = nullptr;
  L.plan = nullptr;
  L.occ = nullptr;
  size_t off = 0;
  L.leafsum = reinterpret_cast<float*>(base + off);
  off += (size_t)4 * L.LS * 4;
  L.V = reinterpret_cast<float*>(base + off);
  off += 32;
  L.thr = reinterpret_cast<uint64_t*>(base + off);
  off += 384;
  L.plan_big = reinterpret_cast<sr_plan_big*>(base + off);
  return L;
}
size_t sr_lds_bytes_big(int leaves) {
  return (size_t)4 * leaf_stride(leaves) * 4 + 32 + 384 + ((sizeof(sr_plan_big) + 15) & ~(size_t)15) + 256;
}

size_t sr_lds_bytes(int S, int leaves, bool occ) {
  size_t b = (size_t)(kRows + 1) * padded(S) * 4 + (size_t)4 * leaf_stride(leaves) * 4 + 32 +
             ((sizeof(sr_plan) + 15) & ~(size_t)15) + 384;
  if (occ) b += (size_t)S * 4;
  return (b + 15) & ~(size_t)15;
}

// The four 16-bit successors of one state in the agent's transition table, as one 64-bit word.
__device__ __forceinline__ int t_of(uint64_t row, int k) { return (int)((row >> (16 * k)) & 0xffffu); }
__device__ __forceinline__ uint64_t t_set(uint64_t row, int k, uint32_t v) {
  return (row & ~(0xffffull << (16 * k))) | ((uint64_t)v << (16 * k));
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}

// PRE (VEC and S <= 1024): as soon as the action is known, wave a starts loading the value row it
// will need in the NEXT step (row T[ns][a]) into registers, so the HBM latency of the four value
// rows overlaps the row update of this step.  The one row that cannot be prefetched — SR[s], which
// this step rewrites — is taken from the LDS copy the update leaves behind.
// PSETS: hyper-parameters from per-instance parameter sets (run.param_index).
// Five waves per SIMD = five workgroups per CU (96 registers; the allocator's own choice of 98
// stops at four, and six — 80 registers — spills: 10.0 / 8.9 / 9.3 ms per launch on C4).
// STOCH: the world's transition rows are distributions — SR.train simply calls interface.step
// (agent/sr.py:170-182), and Gridworld.step draws the successor from the row (interface/
// gridworld.py:119-123): one double of the env stream per step (sub-stream 1 of the counter the
// trial starts share, as cobel_env_step_draw and the general tabular kernel), first successor whose
// cumulative probability exceeds it; the record of the state entered is then fetched, not taken
// from the prefetched four.
// BIG: more states than six rows of them fit the LDS (VEC and PRE off): every row is read where
// it lies — the value rows by the dot products, SR[s] and SR[ns] by the update, the reward
// estimate too —, same arithmetic, same order of the pairwise sums.
template <bool VEC, bool OCC, bool PRE, bool PSETS, bool STOCH = false, bool BIG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_sr(
    const sr_args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  static_assert(!BIG || (!VEC && !PRE), "BIG kernels stream rows element by element");
  const int S = A.S;
  const sr_lds L = BIG ? carve_big(lds_raw, S, A.leaves) : carve(lds_raw, S, A.leaves);
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = (int)blockIdx.x;
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  float* const SRg = A.r.sr + (size_t)i * S * S;
  uint16_t* const Tg = A.r.trans + (size_t)i * S * 4;
  const uint64_t* const T8 = reinterpret_cast<const uint64_t*>(Tg);   // one row per load
  float* const Rg = A.r.rewards + (size_t)i * S;
  uint32_t* const occ = L.occ;  // only if OCC

  if (BIG) {
    if (t == 0)
      build_plan(L.plan_big, S, reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(L.plan_big) +
                                                       ((sizeof(sr_plan_big) + 15) & ~(size_t)15)));
  } else {
    for (int e = t; e < S; e += 256) {
      L.rw[phys(e)] = Rg[e];
      if (OCC) occ[e] = 0u;
    }
    if (t == 0) build_plan(L.plan, S, reinterpret_cast<int*>(L.rows));   // rows: free scratch here
  }
  __syncthreads();

  int32_t* const inst = A.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);
  unsigned long long nsteps = *reinterpret_cast<const unsigned long long*>(inst + COBEL_I_STEPS_LO);

  const uint32_t flags = A.r.flags;
  const bool learn = flags & COBEL_F_LEARN;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (flags & COBEL_F_MASK_ACTIONS) ? A.r.action_mask : nullptr;
  const uint64_t seed = A.r.seed;
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  const cobel_param_set_t* P = nullptr;   // this instance's parameter set, if any
  if (PSETS) {
    const int k = (int)A.r.param_index[i];
    P = A.r.param_sets + (k < A.r.n_param_sets ? k : A.r.n_param_sets - 1);
  }
  const double alpha = PSETS ? P->alpha : A.r.alpha, gamma = PSETS ? P->gamma : A.r.gamma;
  const float alpha_f = PSETS ? P->alpha_f : A.alpha_f, gamma_f = PSETS ? P->gamma_f : A.gamma_f;
  cobel_eps_bb ebb;
  ebb.base[0] = ebb.bonus[0] = 0.0;
#pragma unroll
  for (int n = 1; n <= 4; ++n) {
    ebb.base[n] = PSETS ? P->eps_base[n] : A.eps.base[n];
    ebb.bonus[n] = PSETS ? P->eps_bonus[n] : A.eps.bonus[n];
  }

  if (t < 48) L.thr[t] = PSETS ? P->eps_thr[t / 3][t % 3] : A.eps.thr[t / 3][t % 3];
  __syncthreads();
  uint32_t cw0 = 0, cw1 = 0;
  uint4 cand = {0, 0, 0, 0};
  uint32_t mask_cur = 15u;
  uint64_t tcur = 0;   // T[state][0..3]: carried from step to step, loaded only at trial starts
  // The row of the state just left, after that step's write: a step straight back reads it from
  // here instead of racing the 2-byte store through the cache.  Older writes are read from L2
  // (agent-scope loads bypass the CU's L1), at least a whole step after they were issued.
  uint64_t tleft = 0;
  int left_state = -1;
  auto load_trow = [&](int s) -> uint64_t {
    return __hip_atomic_load(T8 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  cobel_u4 pblk = {0, 0, 0, 0};
  uint32_t pb_idx = ~0u;
  auto enter_state = [&](int s) {
    const uint4 c = W4[s];
    cw0 = rfl(c.x);
    cw1 = rfl(c.y);
    if (!STOCH && lane < 4) cand = W4[next_of(cw0, cw1, lane)];
    mask_cur = amask ? (uint32_t)amask[s] & 15u : 15u;
    tcur = load_trow(s);
    left_state = -1;
  };
  if (iflags & 1u) enter_state(state);

  int budget = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;
  unsigned long long executed = 0;
  // PRE: this wave's value row of the coming step, elements lane*4 + 256*j in pre_j (named
  // registers on purpose: an array carried around the loop is demoted to scratch memory)
  float4 pre0 = {0, 0, 0, 0}, pre1 = pre0, pre2 = pre0, pre3 = pre0;
  int pre_mode = 0;     // 0 load at the top of the step, 1 in `pre`, 2 in the LDS spare row 4
  float* const spare = BIG ? nullptr : L.rows + (size_t)4 * L.PS;

  while (true) {
    if (!(iflags & 1u)) {
      if (trial >= A.r.trials_target) break;
      state = (int)A.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                               start_cnt)];
      ce += 1u;
      step = 0;
      trew = 0.0;
      iflags |= 1u;
      pre_mode = 0;
      enter_state(state);
    }
    if (budget == 0) break;
    budget -= 1;

    // ---- retrieve_q (sr.py:302-306): wave a evaluates V[T[s][a]] ----------------------------
    const int my_row = t_of(tcur, wave);
    float* const my_buf = BIG ? SRg + (size_t)my_row * S : L.rows + (size_t)wave * L.PS;
    const int e0 = lane * 4;
    if (BIG) {
      // (read in place by the dot product below)
    } else if (PRE && pre_mode == 1) {
      if (e0 < S) *reinterpret_cast<float4*>(my_buf + phys(e0)) = pre0;
      if (e0 + 256 < S) *reinterpret_cast<float4*>(my_buf + phys(e0 + 256)) = pre1;
      if (e0 + 512 < S) *reinterpret_cast<float4*>(my_buf + phys(e0 + 512)) = pre2;
      if (e0 + 768 < S) *reinterpret_cast<float4*>(my_buf + phys(e0 + 768)) = pre3;
    } else if (PRE && pre_mode == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + 256 * j;
        if (e < S)
          *reinterpret_cast<float4*>(my_buf + phys(e)) =
              *reinterpret_cast<const float4*>(spare + phys(e));
      }
    } else {
      load_row<VEC>(my_buf, SRg + (size_t)my_row * S, S, lane, 64);
    }
    __builtin_amdgcn_wave_barrier();
    // The action draw is the same for the whole workgroup: only wave 0 evaluates Philox (one block
    // per two draws) and hands the two words over through LDS with the values.  A wave pays for
    // an instruction whatever the number of live lanes, and the four waves sit on four SIMDs that
    // they share with the other workgroups, so evaluating it in all four cost 16 % of a step.
    if (wave == 0) {
      if ((cp >> 1) != pb_idx) {
        pb_idx = cp >> 1;
        pblk = cobel_philox(pb_idx, 0u, g, pol_stream, seed);
      }
      if (lane == 0) {
        uint32_t* const dw = reinterpret_cast<uint32_t*>(L.V + 4);
        dw[0] = (cp & 1u) ? pblk.z : pblk.x;
        dw[1] = (cp & 1u) ? pblk.w : pblk.y;
      }
    }
    const float v = BIG ? wave_pairwise_dot<false>(my_buf, Rg, L.plan_big, L.leafsum + wave * L.LS, lane)
                        : wave_pairwise_dot(my_buf, L.rw, L.plan, L.leafsum + wave * L.LS, lane);
    if (lane == 0) L.V[wave] = v;
    lds_barrier();
    const float4 q = *reinterpret_cast<const float4*>(L.V);

    // ---- select + env.step -------------------------------------------------------------------
    const uint32_t w0 = rfl(reinterpret_cast<const uint32_t*>(L.V + 4)[0]);
    const uint32_t w1 = rfl(reinterpret_cast<const uint32_t*>(L.V + 4)[1]);
    cp += 1u;
    // all actions allowed: integer thresholds of the tie pattern's CDF instead of the float64
    // selection with its three divisions (cobel_policy.h; same result, a third of the latency)
    const int a = mask_cur == 15u
                      ? (int)rfl((uint32_t)cobel_eps_greedy_select_thr(q.x, q.y, q.z, q.w,
                                                                        cobel_u53(w0, w1), L.thr, lane))
                      : (int)rfl((uint32_t)cobel_eps_greedy_select_wave(
                            q.x, q.y, q.z, q.w, mask_cur, cobel_u01(w0, w1), ebb, lane));
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
    // T[ns][.] as it stands before this step's write (one 8-byte load, the same for all threads)
    uint64_t tnxt = tcur;
    if (!trial_over && ns != state) tnxt = ns == left_state ? tleft : load_trow(ns);
    if (PRE) {
      pre_mode = 0;
      if (!trial_over) {
        // T[state][a] becomes ns in this step; every other entry of T[ns][.] is already final
        const int nrow = (learn && ns == state && wave == a) ? ns : t_of(tnxt, wave);
        if (learn && nrow == state) {
          pre_mode = 2;
        } else {
          const float4* const src4 =
              reinterpret_cast<const float4*>(SRg + (size_t)nrow * S) + lane;
          if (e0 < S) pre0 = src4[0];
          if (e0 + 256 < S) pre1 = src4[64];
          if (e0 + 512 < S) pre2 = src4[128];
          if (e0 + 768 < S) pre3 = src4[192];
          pre_mode = 1;
        }
      }
    }

    if (learn) {
      // rows already in LDS? (T[s][.] as it was when the value rows were loaded)
      int src_s = -1, src_ns = -1;
      const bool need_ns = nt != 0u;
      if (!BIG) {
#pragma unroll
        for (int k2 = 3; k2 >= 0; --k2) {
          const int rowk = t_of(tcur, k2);
          if (rowk == state) src_s = k2;
          if (rowk == ns) src_ns = k2;
        }
        if (ns == state && src_ns < 0) src_ns = 4;  // shares the spare row with SR[s]
        // (every wave is past its dot product — the barrier behind L.V — so the value rows that are
        //  not a source of this update are free, and the spare row is no longer being copied from)
        if (src_s < 0) {
          load_row<VEC>(L.rows + (size_t)4 * L.PS, SRg + (size_t)state * S, S, t, 256);
          src_s = 4;
        }
        if (need_ns && src_ns < 0) {
          // first visit of (s, a): SR[ns] is not among the value rows yet; it goes into one of them
          // that this update does not read
          src_ns = src_s == 0 ? 1 : 0;
          load_row<VEC>(L.rows + (size_t)src_ns * L.PS, SRg + (size_t)ns * S, S, t, 256);
        }
      }
      if (t == 0) {
        // sr.py:272-274 (float32): rewards[ns] += (r - rewards[ns]) * lr; transitions[s][a] = ns
        const float old = BIG ? Rg[ns] : L.rw[phys(ns)];
        const float d = r - old;
        const float upd = old + d * alpha_f;
        if (!BIG) L.rw[phys(ns)] = upd;
        Rg[ns] = upd;
        Tg[state * 4 + a] = (uint16_t)ns;
      }
      tcur = t_set(tcur, a, (uint32_t)ns);
      lds_barrier();
      // sr.py:276-284: td = e_s + gamma * (SR[ns] | e_ns) - SR[s];  SR[s] += lr * td
      // (BIG: both rows where they lie; an element of SR[s] is read and written by the same thread)
      const float* const row_s = BIG ? SRg + (size_t)state * S : L.rows + (size_t)src_s * L.PS;
      const float* const row_n = BIG ? SRg + (size_t)(need_ns ? ns : state) * S
                                     : L.rows + (size_t)(need_ns ? src_ns : src_s) * L.PS;
      float* const out = SRg + (size_t)state * S;
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
      if (VEC) {
        for (int e = t * 4; e < S; e += 1024) {
          const float4 cs = *reinterpret_cast<const float4*>(row_s + phys(e));
          const float4 cn = *reinterpret_cast<const float4*>(row_n + phys(e));
          float4 o;
          o.x = upd1(e + 0, cs.x, cn.x);
          o.y = upd1(e + 1, cs.y, cn.y);
          o.z = upd1(e + 2, cs.z, cn.z);
          o.w = upd1(e + 3, cs.w, cn.w);
          reinterpret_cast<float4*>(out)[e >> 2] = o;
          // keep the new row in LDS for the waves whose next value row it is (the thread that
          // read elements e..e+3 is the only one that writes them, also when spare is a source)
          if (PRE) *reinterpret_cast<float4*>(spare + phys(e)) = o;
        }
      } else {
        for (int e = t; e < S; e += 256) {
          const int pe = BIG ? e : phys(e);
          out[e] = upd1(e, row_s[pe], row_n[pe]);
        }
      }
    }

    if (A.r.last_exp && t == 0) {
      int32_t* const e = A.r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = 0;
    }
    trew += (double)r;
    nsteps += 1ull;
    executed += 1ull;
    if (OCC && t == 0) {
      if (BIG) atomicAdd(A.r.occupancy + (size_t)world * S + ns, 1ull);
      else occ[ns] += 1u;
    }
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
    } else {
      if (t == 0 && trial >= 0 && trial < A.r.trial_cap) {
        const size_t m = cobel_mon_offset(A.r.mon_stripes, A.r.trial_cap) + (size_t)trial;
        if (A.r.lat_sum) atomicAdd(A.r.lat_sum + m, (unsigned long long)step);
        if (A.r.lat_cnt) atomicAdd(A.r.lat_cnt + m, 1ull);
        if (A.r.reward_sum) atomicAdd(A.r.reward_sum + m, trew);
        if (A.r.resp_cnt && trew > 0.0) atomicAdd(A.r.resp_cnt + m, 1ull);
        if (A.r.lat_trace) A.r.lat_trace[(size_t)i * A.r.trial_cap + trial] = step;
      }
      trial += 1;
      iflags &= ~1u;
    }
    // The LDS rows this step computed from must not be overwritten before every thread is done
    // with them.  Without the prefetch (and at trial ends) the row just stored must also be
    // visible to the global loads of the next step: full barrier.
    if (PRE && !trial_over) lds_barrier();
    else __syncthreads();
  }

  if (OCC && !BIG) {
    __syncthreads();
    for (int e = t; e < S; e += 256) {
      const uint32_t c = occ[e];
      if (c && A.r.occupancy) atomicAdd(A.r.occupancy + (size_t)world * S + e, (unsigned long long)c);
    }
  }
  if (t == 0) {
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_FLAGS] = (int32_t)iflags;
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) = nsteps;
    if (A.r.steps_done && executed) atomicAdd(A.r.steps_done, executed);
  }
}

// q[i][a] = V[T[s_i][a]] for given states — predict_on_batch (sr.py:310-324).
template <bool VEC, bool BIG = false>
__global__ __launch_bounds__(256) void k_sr_q(const float* __restrict__ sr,
                                              const uint16_t* __restrict__ trans,
                                              const float* __restrict__ rewards,
                                              const int32_t* __restrict__ states,
                                              float* __restrict__ q_out, int S, int leaves) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const sr_lds L = BIG ? carve_big(lds_raw, S, leaves) : carve(lds_raw, S, leaves);
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = (int)blockIdx.x;
  const float* const SRg = sr + (size_t)i * S * S;
  if (BIG) {
    if (t == 0)
      build_plan(L.plan_big, S, reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(L.plan_big) +
                                                       ((sizeof(sr_plan_big) + 15) & ~(size_t)15)));
    __syncthreads();
    const int s = states[i];
    const int row = (int)trans[((size_t)i * S + s) * 4 + wave];
    const float v = wave_pairwise_dot<false>(SRg + (size_t)row * S, rewards + (size_t)i * S,
                                             L.plan_big, L.leafsum + wave * L.LS, lane);
    if (lane == 0) q_out[(size_t)i * 4 + wave] = v;
    return;
  }
  for (int e = t; e < S; e += 256) L.rw[phys(e)] = rewards[(size_t)i * S + e];
  if (t == 0) build_plan(L.plan, S, reinterpret_cast<int*>(L.rows + (size_t)4 * L.PS));
  __syncthreads();
  const int s = states[i];
  const int row = (int)trans[((size_t)i * S + s) * 4 + wave];
  float* const buf = L.rows + (size_t)wave * L.PS;
  load_row<VEC>(buf, SRg + (size_t)row * S, S, lane, 64);
  __builtin_amdgcn_wave_barrier();
  const float v = wave_pairwise_dot(buf, L.rw, L.plan, L.leafsum + wave * L.LS, lane);
  if (lane == 0) q_out[(size_t)i * 4 + wave] = v;
}

__global__ __launch_bounds__(256) void k_sr_init(float* __restrict__ sr,
                                                 uint16_t* __restrict__ trans,
                                                 float* __restrict__ rewards, size_t n, int S) {
  const size_t SS = (size_t)S * S;
  // one index space for the three tables: per instance max(S * S, 4 * S) slots (worlds of fewer
  // than four states have more transition entries than SR elements)
  const size_t span = SS > (size_t)S * 4 ? SS : (size_t)S * 4;
  const size_t total = n * span;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const size_t inst = e / span, within = e % span;
    if (within < SS) sr[inst * SS + within] = (within / S == within % S) ? 1.0f : 0.0f;
    if (within < (size_t)S * 4) trans[inst * S * 4 + within] = (uint16_t)(within >> 2);
    if (within < (size_t)S) rewards[inst * S + within] = 0.0f;
  }
}

template <bool VEC, bool OCC, bool PRE, bool PSETS, bool STOCH = false, bool BIG = false>
int launch_sr(const sr_args& A, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    COBEL_HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_sr<VEC, OCC, PRE, PSETS, STOCH, BIG>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_sr<VEC, OCC, PRE, PSETS, STOCH, BIG>), dim3(A.r.n), dim3(256), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

}  // namespace

static const size_t kLdsLimit = 160 * 1024;

extern "C" int cobel_sr_init(float* sr, uint16_t* trans, float* rewards, int32_t n,
                             int32_t n_states, void* stream) {
  COBEL_REQUIRE(sr && trans && rewards, COBEL_E_ARG, "cobel_sr_init: NULL table");
  COBEL_REQUIRE(n >= 0 && n_states > 0 && n_states <= 65535, COBEL_E_RANGE,
                "cobel_sr_init: bad sizes n=%d S=%d", n, n_states);
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_sr_init, dim3(8192), dim3(256), 0, (hipStream_t)stream, sr, trans, rewards,
                     (size_t)n, n_states);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_sr_run(const cobel_world_t* world, const cobel_sr_run_t* run, void* stream) {
  if (int rc = cobel_world_check(world, "cobel_sr_run")) return rc;
  COBEL_REQUIRE(world->n_actions == 4, COBEL_E_UNSUPPORTED,
                "cobel_sr_run: the world has %d actions, this entry point serves four-action worlds",
                world->n_actions);
  COBEL_REQUIRE(world && run, COBEL_E_ARG, "cobel_sr_run: NULL world/run");
  const cobel_sr_run_t& r = *run;
  COBEL_REQUIRE(r.sr && r.trans && r.rewards && r.inst, COBEL_E_ARG,
                "cobel_sr_run: sr, trans, rewards and inst are required");
  COBEL_REQUIRE(((uintptr_t)r.sr & 15u) == 0 && ((uintptr_t)r.inst & 7u) == 0, COBEL_E_ARG,
                "cobel_sr_run: sr must be 16-byte and inst 8-byte aligned");
  COBEL_REQUIRE(r.n >= 0, COBEL_E_RANGE, "cobel_sr_run: n = %d", r.n);
  COBEL_REQUIRE(r.steps_per_trial > 0, COBEL_E_RANGE, "cobel_sr_run: steps_per_trial = %d",
                r.steps_per_trial);
  COBEL_REQUIRE(r.epsilon >= 0.0 && r.epsilon <= 1.0, COBEL_E_ARG,
                "cobel_sr_run: epsilon %g outside [0, 1]", r.epsilon);
  COBEL_REQUIRE(!r.param_index || (r.param_sets && r.n_param_sets > 0), COBEL_E_ARG,
                "cobel_sr_run: param_index given without parameter sets");
  COBEL_REQUIRE(!(r.flags & COBEL_F_MASK_ACTIONS) || r.action_mask, COBEL_E_ARG,
                "cobel_sr_run: mask_actions set without an action mask");
  const int S = world->n_states;
  const bool occ = r.occupancy != nullptr;
  const int leaves = count_leaves(S);
  size_t lds = sr_lds_bytes(S, leaves, occ);
  lds += cobel_debug_lds_pad(lds, kLdsLimit);   // (occupancy experiments only)
  // More states than six rows of them fit one workgroup's LDS (6 336, e.g. 79 x 79): the BIG
  // kernels, rows read where they lie — the reference's loop has no size limit (agent/sr.py:
  // 109-140; its own tensors are 2.6 GB per agent there).
  const bool big = leaves > kMaxLeaves || lds > kLdsLimit;
  if (big) {
    COBEL_REQUIRE(leaves <= kMaxLeavesBig, COBEL_E_RANGE, "cobel_sr_run: %d states", S);
    lds = sr_lds_bytes_big(leaves);
  }
  if (r.n == 0) return COBEL_OK;
  // Worlds with at most eight rewarded states and up to 1 024 states (every builder of the reference):
  // the value rows collapse to a few elements each, see sr_wave.hip.
  if (((uintptr_t)r.rewards & 15u) == 0 && cobel_sr_wave_covers(world, r))
    return cobel_sr_wave_launch(world, r, (hipStream_t)stream);
  sr_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = S;
  A.n_worlds = world->n_worlds;
  A.leaves = leaves;
  A.r = r;
  A.eps = cobel_make_eps_consts(r.epsilon);
  A.alpha_f = (float)r.alpha;
  A.gamma_f = (float)r.gamma;
  A.succ_off = world->succ_off;
  A.succ_state = world->succ_state;
  A.succ_cdf = world->succ_cdf;
  hipStream_t st = (hipStream_t)stream;
  if (big) {
#define COBEL_SR_BIG(OC, PS)                                                        \
  return world->succ_off ? launch_sr<false, OC, false, PS, true, true>(A, lds, st)  \
                         : launch_sr<false, OC, false, PS, false, true>(A, lds, st)
    if (occ && r.param_index) COBEL_SR_BIG(true, true);
    if (occ) COBEL_SR_BIG(true, false);
    if (r.param_index) COBEL_SR_BIG(false, true);
    COBEL_SR_BIG(false, false);
#undef COBEL_SR_BIG
  }
  const bool vec = (S % 4) == 0;
  if (world->succ_off) {   // the successor is drawn: the row-streaming kernel without its prefetch
    if (r.param_index) {
      if (vec) return occ ? launch_sr<true, true, false, true, true>(A, lds, st)
                          : launch_sr<true, false, false, true, true>(A, lds, st);
      return occ ? launch_sr<false, true, false, true, true>(A, lds, st)
                 : launch_sr<false, false, false, true, true>(A, lds, st);
    }
    if (vec) return occ ? launch_sr<true, true, false, false, true>(A, lds, st)
                        : launch_sr<true, false, false, false, true>(A, lds, st);
    return occ ? launch_sr<false, true, false, false, true>(A, lds, st)
               : launch_sr<false, false, false, false, true>(A, lds, st);
  }
  const bool pre = vec && S <= 1024 && !(r.flags & COBEL_F_NO_PREFETCH);
#define COBEL_SR(V, PF)                                                                   \
  do {                                                                                    \
    if (r.param_index)                                                                    \
      return occ ? launch_sr<V, true, PF, true>(A, lds, st)                               \
                 : launch_sr<V, false, PF, true>(A, lds, st);                             \
    return occ ? launch_sr<V, true, PF, false>(A, lds, st)                                \
               : launch_sr<V, false, PF, false>(A, lds, st);                              \
  } while (0)
  if (pre) COBEL_SR(true, true);
  if (vec) COBEL_SR(true, false);
  COBEL_SR(false, false);
#undef COBEL_SR
}

extern "C" int cobel_sr_retrieve_q(const float* sr, const uint16_t* trans, const float* rewards,
                                   const int32_t* states, float* q_out, int32_t n,
                                   int32_t n_states, void* stream) {
  COBEL_REQUIRE(sr && trans && rewards && states && q_out, COBEL_E_ARG,
                "cobel_sr_retrieve_q: NULL argument");
  COBEL_REQUIRE(n >= 0 && n_states > 0 && n_states <= 65535, COBEL_E_RANGE,
                "cobel_sr_retrieve_q: bad sizes");
  if (n == 0) return COBEL_OK;
  const int leaves = count_leaves(n_states);
  const size_t lds = sr_lds_bytes(n_states, leaves, false);
  hipStream_t st = (hipStream_t)stream;
  if (leaves > kMaxLeaves || lds > kLdsLimit) {   // (rows read where they lie, see cobel_sr_run)
    COBEL_REQUIRE(leaves <= kMaxLeavesBig, COBEL_E_RANGE, "cobel_sr_retrieve_q: %d states", n_states);
    hipLaunchKernelGGL((k_sr_q<false, true>), dim3(n), dim3(256), sr_lds_bytes_big(leaves), st, sr,
                       trans, rewards, states, q_out, n_states, leaves);
    COBEL_HIP_TRY(hipGetLastError());
    return COBEL_OK;
  }
  if ((n_states % 4) == 0 && ((uintptr_t)sr & 15u) == 0) {
    if (lds > 64 * 1024)
      COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sr_q<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_sr_q<true>), dim3(n), dim3(256), lds, st, sr, trans, rewards, states,
                       q_out, n_states, leaves);
  } else {
    if (lds > 64 * 1024)
      COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sr_q<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_sr_q<false>), dim3(n), dim3(256), lds, st, sr, trans, rewards, states,
                       q_out, n_states, leaves);
  }
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
