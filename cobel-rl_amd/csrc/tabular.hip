// Fused tabular agents (Q-learning / Dyna-Q) — wave-per-instance kernel for gfx950.
//
// One 64-lane wavefront owns one agent–env instance for the whole call:
//   * the instance's Q table (16 B per state) lives in LDS from the first step to the last, so
//     every Q access of the select / TD / planning chain is an LDS access and HBM sees each
//     table exactly twice per call (load, store);
//   * the B planning updates of a step are spread over lanes 0..B-1.  The reference applies
//     them one after another (agent/dyna_q.py:327-330), so lane j may only run once every
//     earlier lane it depends on has written: lane j depends on lane i < j iff i writes a cell
//     of the row j maximises over (s_i == ns_j) or the cell j itself reads (idx_i == idx_j).
//     Dependencies are rare (about S^-1 per pair); the wave finds them with a readlane sweep
//     and executes maximal conflict-free prefixes in parallel — results are identical to the
//     sequential order;
//   * the world model (Dyna-Q) / experience log (QAgent) stays in HBM as packed 8-byte records.
//     The memory stream is counter based, so the records a step will sample are known one step
//     ahead: they are gathered while the previous step's planning runs, then patched in
//     registers with the one record that step itself writes;
//   * lanes 0..3 prefetch the four candidate successor records of the current state, so the
//     env transition costs no dependent global load.
//
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   agent/dyna_q.py:164-215 (train loop), :217-273 (test), :290-299 (TD), :327-330 (replay)
//   memory/dyna_q.py:92-96 (store), :137-155 (retrieve_batch)
//   agent/q.py:183-228, :305-313, :353-354
//   monitor/behavior.py:82 (latency = logs['steps'] = index of the last executed step)
#include <stdlib.h>

#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct tab_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  cobel_tab_run_t r;
  cobel_eps_consts eps;
  float alpha_f, gamma_f, model_lr_f;
  int32_t lpw;         // lane-per-instance kernel: instances per wave (64, 32 or 16)
  // transition rows that are distributions (cobel_world_set_transitions): the generic
  // (!FAST) instantiations of k_tab_wpi draw the successor in the step, else NULL
  const uint32_t* succ_off;
  const uint16_t* succ_state;
  const double* succ_cdf;
};

__device__ __forceinline__ size_t mon_stripe_offset(const cobel_tab_run_t& r) {
  return cobel_mon_offset(r.mon_stripes, r.trial_cap);
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// max of a Q row in two instructions (fmaxf() costs two more: it first quiets each operand)
__device__ __forceinline__ float max4(const float4 v) {
  float m;
  asm("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4"
      : "=&v"(m)
      : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  return m;
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}

// LDS carve-up (bytes): Q 16*S | M16 8*S (Dyna-Q) | H 2048 (replay) | world 16*S (WLDS) | occ 4*S
// (26 KiB at S = 1024 for Dyna-Q: six instances per CU — LDS is allocated in 1 KiB units, so every
// byte above 26 624 would cost a whole resident wave)
struct tab_lds {
  float4* Qs;
  float* Qf;
  uint16_t* M16;
  uint4* Wl;
  uint32_t* occ;
};
constexpr int kThrBytes = 16 * 3 * 8;
constexpr int kPassLanes = COBEL_MAX_BATCH;   // planning updates one wavefront takes per pass

// LDS is handed out in blocks of 1 280 bytes, 128 per CU — not in KiB (measured on MI355X with
// scripts/experiments/exp_occupancy.py: the launch time of k_tab_wpi steps down at 15 360 and at 14 080 bytes
// per workgroup and is flat in between; 16 384 B, a 32 x 32 world, are 13 blocks: nine per CU —
// k_tab_pwg's ONE workgroup per CU takes all 128 blocks for ten).
inline int lds_workgroups_per_cu(size_t bytes) {
  const size_t blocks = (bytes + 1279) / 1280;
  return blocks ? (int)(128 / blocks) : 128;
}

__host__ __device__ __forceinline__ size_t tab_lds_bytes(int S, int agent, bool replay, bool wlds,
                                                         bool occ, bool midx = false) {
  size_t b = (size_t)S * 16;
  if (agent == COBEL_AGENT_DYNAQ && !midx) b += (size_t)S * 8;
  (void)replay;   // (round 6: a batch's dependencies are found in the Q table itself — no tables of lane masks)
  if (wlds) b += (size_t)S * 16;
  if (occ) b += (size_t)S * 4;
  return b;
}

#if defined(COBEL_STAMPS)
// Diagnostic build only: per-phase cycle sums, written to last_exp[i][0..5] (never to an output).
#define STAMP(k)                                                                          \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long now_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    stamp_sum[k] += now_ - stamp_last;                                                    \
    stamp_last = now_;                                                                    \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// FAST: the run is the plain training case — learning on, planning after every step, no action
// mask, no per-step host log — so those run-time switches become constants (fewer live scalar
// registers and branches in the step loop; the generic instantiation spills SGPRs).
// MIDX (with FAST): the 16-bit model entries live in a caller-provided HBM table instead of LDS
// (run.model_index, 8 B per state, L2 resident for the instances in flight).  The memory stream is
// counter based, so the entries a step will sample are gathered one step ahead and patched in
// registers with the one entry that step itself writes.  LDS shrinks to Q = 16 KiB and
// nine instances fit on a CU instead of six.
// PSETS: hyper-parameters come from per-instance parameter sets (run.param_index).  A template
// switch rather than a run-time one: the mere possibility of taking the loop constants from
// memory changes the register allocation of the whole step loop.
// STOCH (generic form only): the world's transition rows are distributions, the successor is
// drawn in the step.  A template switch for the reason PSETS is one: as a run-time branch it cost
// the one-hot runs of the generic kernel 3 % more vector instructions per step.
// MULTI (plain training with the digest in HBM only): more planning updates per step than one
// wavefront takes in a pass — further passes of up to COBEL_MAX_BATCH updates behind the first,
// their pairs drawn and gathered inside the step.
template <int AGENT, bool OCC, bool WLDS, bool FAST, bool MIDX, bool PSETS, bool STOCH = false,
          bool MULTI = false>
__global__ __launch_bounds__(64) void k_tab_wpi(const tab_args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int S = A.S;
  tab_lds L;
  {
    size_t off = 0;
    L.Qs = reinterpret_cast<float4*>(lds_raw);
    L.Qf = reinterpret_cast<float*>(lds_raw);
    off += (size_t)S * 16;
    L.M16 = reinterpret_cast<uint16_t*>(lds_raw + off);
    if (AGENT == COBEL_AGENT_DYNAQ && !MIDX) off += (size_t)S * 8;
    L.Wl = reinterpret_cast<uint4*>(lds_raw + off);
    if (WLDS) off += (size_t)S * 16;
    L.occ = reinterpret_cast<uint32_t*>(lds_raw + off);
  }
  float4* const Qs = L.Qs;
  float* const Qf = L.Qf;

  const int lane = (int)threadIdx.x;
  const int i = (int)blockIdx.x;
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  float4* const Qg = reinterpret_cast<float4*>(A.r.q) + (size_t)i * S;
  uint64_t* const model = (AGENT == COBEL_AGENT_DYNAQ) ? A.r.model + (size_t)i * S * 4 : nullptr;
  const uint32_t* const model32 = reinterpret_cast<const uint32_t*>(model);
  uint64_t* const rlog =
      (AGENT == COBEL_AGENT_Q && A.r.replay_log) ? A.r.replay_log + (size_t)i * A.r.log_cap
                                                 : nullptr;
  const uint32_t SA = (uint32_t)S * 4u;
  uint16_t* const Mg = MIDX ? A.r.model_index + (size_t)i * SA : nullptr;

  // ---- stage the instance: Q, compact model, world records --------------------------------
  // (nonzero: does anything that planning could propagate exist yet — a Q cell or a reward
  //  estimate of the model that is not +0.0f?  See COBEL_IF_NONZERO below.)
  uint32_t nonzero = 0u;
  for (int s = lane; s < S; s += 64) {
    const float4 qv = Qg[s];
    Qs[s] = qv;
    nonzero |= __builtin_bit_cast(uint32_t, qv.x) | __builtin_bit_cast(uint32_t, qv.y) |
               __builtin_bit_cast(uint32_t, qv.z) | __builtin_bit_cast(uint32_t, qv.w);
    if (WLDS) L.Wl[s] = W4[s];
    if (OCC) L.occ[s] = 0u;
  }
  if (AGENT == COBEL_AGENT_DYNAQ && !MIDX) {
    // 16-bit model entry: next state | nonterminal << 14 | (reward estimate != +0.0f) << 15.
    // The float32 reward estimates stay in HBM and are fetched only for flagged entries.
    for (uint32_t e = (uint32_t)lane; e < SA; e += 64u) {
      const uint64_t rec = model[e];
      const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
      L.M16[e] = (uint16_t)((hi & 0x3fffu) | (((hi >> 16) & 1u) << 14) | (lo ? 0x8000u : 0u));
      nonzero |= lo;
    }
  }
  __syncthreads();

  int32_t* const inst = A.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t cm = (uint32_t)inst[COBEL_I_CTR_MEMORY];
  uint32_t loglen = (uint32_t)inst[COBEL_I_LOG_LEN];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  // COBEL_IF_NONZERO (bit 1 of the instance flags, Dyna-Q): the instance's Q table or its model's
  // reward estimates hold something other than +0.0f.  Until then every planning update is
  // 0 + alpha (0 + gamma nt 0 - 0) = 0: it leaves Q as it is, whatever pairs are drawn, so only the
  // batch counter moves (in a maze with one rewarded goal that is every step before the agent
  // first reaches it).  Derived here from the tables as the caller handed them over (the digest
  // is only scanned while Q is all zero), set when a reward other than +0.0f arrives, not kept in
  // `inst`.
  if (AGENT == COBEL_AGENT_DYNAQ) {
    if (MIDX && !__ballot(nonzero != 0u))
      for (uint32_t e = (uint32_t)lane; e < SA; e += 64u) nonzero |= (uint32_t)Mg[e] & 0x8000u;
    if (__ballot(nonzero != 0u)) iflags |= 2u;
  }
  // trial reward: a per-lane (vector) register on purpose — it is only ever accumulated, and as a
  // wave-uniform value it would occupy scalar registers the step loop is short of
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO + (lane & 0));
  asm volatile("" : "+v"(trew));

  const uint32_t flags = A.r.flags;
  const bool learn = FAST || (flags & COBEL_F_LEARN);
  const bool episodic = !FAST && (AGENT == COBEL_AGENT_DYNAQ) && (flags & COBEL_F_EPISODIC);
  const int B = FAST ? A.r.batch
                     : ((learn && !(flags & COBEL_F_NO_REPLAY) &&
                         (AGENT == COBEL_AGENT_DYNAQ || rlog != nullptr))
                            ? A.r.batch
                            : 0);
  const bool replay_each_step = FAST || (B > 0 && !episodic);
  // Batches beyond one wavefront (the reference has no limit, dyna_q.py:319-330): the generic
  // kernel plans them in passes of kPassLanes updates, one after the other — the passes are
  // sequential, so later updates see earlier ones as the reference's loop does, and inside a pass
  // the conflict analysis below applies unchanged.  BP = updates of the pass being planned.
  int Bp = B > kPassLanes ? kPassLanes : B;
#define BP ((FAST && !MULTI) ? B : Bp)
  const uint32_t pol_stream = (!FAST && (flags & COBEL_F_TEST_STREAM)) ? COBEL_STREAM_POLICY_TEST
                                                                       : COBEL_STREAM_POLICY;
  const uint8_t* const amask =
      (!FAST && (flags & COBEL_F_MASK_ACTIONS)) ? A.r.action_mask : nullptr;
  uint64_t seed = A.r.seed;
  asm volatile("" : "+v"(seed));
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  // Loop constants that only ever feed vector instructions are pinned to vector registers: the
  // step loop is short of scalar registers (spills cost instructions), not of vector ones.
  // hyper-parameters: launch-wide, or this instance's parameter set (grid-search fan-out)
  const cobel_param_set_t* P = nullptr;
  if (PSETS) {
    const int k = (int)A.r.param_index[i];
    P = A.r.param_sets + (k < A.r.n_param_sets ? k : A.r.n_param_sets - 1);
  }
  double alpha = PSETS ? P->alpha : A.r.alpha, gamma = PSETS ? P->gamma : A.r.gamma;
  float alpha_f = PSETS ? P->alpha_f : A.alpha_f, gamma_f = PSETS ? P->gamma_f : A.gamma_f,
        mlr_f = PSETS ? P->model_lr_f : A.model_lr_f;
  cobel_eps_bb ebb;
  ebb.base[0] = ebb.bonus[0] = 0.0;
#pragma unroll
  for (int n = 1; n <= 4; ++n) {
    ebb.base[n] = PSETS ? P->eps_base[n] : A.eps.base[n];
    ebb.bonus[n] = PSETS ? P->eps_bonus[n] : A.eps.bonus[n];
  }
  asm volatile("" : "+v"(alpha), "+v"(gamma), "+v"(alpha_f), "+v"(gamma_f), "+v"(mlr_f));

  // epsilon-greedy thresholds (cobel_eps_consts::thr), entry e = t * 3 + k in lane e < 48
  const uint64_t thr_mine = PSETS ? P->eps_thr[(lane % 48) / 3][lane % 3]
                              : A.eps.thr[(lane % 48) / 3][lane % 3];
#if defined(COBEL_STAMPS)
  unsigned long long stamp_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0;
#endif
  // ---- values carried from one step to the next ------------------------------------------
  uint32_t cw0 = 0, cw1 = 0;   // next[0..3] of the current state (uniform)
  uint4 cand = {0, 0, 0, 0};   // !WLDS: lane k < 4 holds the world record of next[state][k]
  uint32_t mask_cur = 15u;
  // Cached Philox output: lanes < B hold memory block mb_idx (four consecutive batches), lanes 62
  // and 63 hold policy blocks 2 pb_idx and 2 pb_idx + 1 (action draws 4 pb_idx .. 4 pb_idx + 3).
  cobel_u4 blk = {0, 0, 0, 0};
  uint32_t mb_idx = ~0u, pb_idx = 0x7fffffffu;
  int refresh_in = 0;          // MIDX: steps left before the cached draws must be renewed
  uint2 m4 = {0u, 0u};         // !MIDX: the four 16-bit model entries of the current state
  uint32_t mrec = 0u;          // MIDX, lane k < 4: reward estimate (bits) of model record (state, k),
                               //       gathered one step ahead
  uint32_t fix_sa = ~0u;       // MIDX: the pair whose record the previous step rewrote (possibly
  float fix_r = 0.0f;          //       after that gather had been issued), and its new reward
  uint32_t idx_cur = 0;        // MIDX, lane j < B: pair sampled by this step's replay j ...
  uint32_t mg_cur = 0;         // ... and its model entry, gathered one step ahead
  uint32_t qx = 0;             // QAgent replay: memory draw of the upcoming batch
  uint64_t qrec = 0;           // ... and the logged experience it selects, gathered ahead

  auto draw_m = [&](uint32_t counter) -> uint32_t {   // uncached memory draw (QAgent log replay)
    const cobel_u4 b = cobel_philox(counter >> 2, (uint32_t)lane, g, COBEL_STREAM_MEMORY, seed);
    return cobel_word(b, counter & 3u);
  };
  const bool cached_mem = AGENT == COBEL_AGENT_DYNAQ && B > 0;
  // One Philox evaluation refills both caches.  The policy lanes hold the aligned block pair
  // {2 pq, 2 pq + 1} = action draws 4 pq .. 4 pq + 3.  With MIDX the memory block is needed one
  // batch ahead, so there the policy pair is also fetched one draw ahead (after this step's draw
  // has been taken from the old pair): both caches then run dry in the same step whenever the two
  // counters advance together, and one evaluation serves four steps.
  auto refresh_draws = [&](uint32_t pq, bool force = false) {
    const uint32_t mi = (MIDX ? cm + 1u : cm) >> 2;
    const bool hit = !force && pq == pb_idx && (!cached_mem || mi == mb_idx);
    if (__builtin_expect(hit, 1)) return;
    const bool p0 = lane == 62, p1 = lane == 63;
    blk = cobel_philox(p1 ? 2u * pq + 1u : (p0 ? 2u * pq : mi), (p0 || p1) ? 0u : (uint32_t)lane,
                       g, (p0 || p1) ? pol_stream : COBEL_STREAM_MEMORY, seed);
    pb_idx = pq;
    mb_idx = mi;
  };
  auto log_gather = [&](uint32_t x, uint32_t bound, uint32_t have) -> uint64_t {
    uint64_t rec = 0;
    if (lane < BP && bound > 0u) {
      const uint32_t idx = cobel_bounded(x, bound);
      if (idx < have) rec = rlog[idx];
    }
    return rec;
  };
  auto enter_state = [&](int s) {  // what depends only on the state being entered
    const uint4 c = WLDS ? L.Wl[s] : W4[s];
    cw0 = rfl(c.x);
    cw1 = rfl(c.y);
    if (!WLDS && lane < 4) cand = W4[next_of(cw0, cw1, lane)];
    if (MIDX && lane < 4) mrec = model32[2u * ((uint32_t)s * 4u + (uint32_t)lane)];
    mask_cur = amask ? (uint32_t)amask[s] & 15u : 15u;
  };

  // ---- B sequential TD updates, executed as speculative rounds / conflict-free prefixes ------
  // Lane j < B holds replay j = (idx -> (s, a), r, ns, nt).  The reference applies them in order
  // (agent/dyna_q.py:329-330); j may run once every earlier lane that writes a cell j reads —
  // s_i == ns_j (row of the max) or idx_i == idx_j (the cell itself) — has written.
  // (inside == true: the caller already runs under `lane < B`; one predicated region instead of
  //  one per table access)
  auto run_batch = [&](uint32_t idx, uint32_t ns, uint32_t nt, float r, bool inside = false) {
    const bool on = inside || lane < BP;
#if defined(COBEL_ABLATE) && COBEL_ABLATE == 3
    return;
#endif
    // The rounds are speculative: every remaining lane computes its update from the table as it
    // stands; a lane's result holds unless an earlier lane of this round that writes a cell it
    // reads has CHANGED that cell — an update that leaves its cell as it was (all-zero regions
    // of Q, converged entries) blocks nobody.  The lanes before the first one whose inputs moved
    // are committed (only changed cells are written, so two lanes of one round never write the
    // same cell), the rest goes again.  Same order of effects as the reference's loop; never
    // fewer lanes per round than the conflict-free prefix.
    // Who is held back is found IN the table (round 6, as in k_tab_pwg; until then two tables of
    // lane masks keyed by the pair index, exact up to 1 024 states, and buckets with a verifying
    // loop beyond): a lane that changes its cell raises it to the TAG ~lane with ds_max_u32 —
    // tags are the bit patterns 0xffffffc0 .. 0xffffffff, above every float that is not a NaN of
    // exactly that payload, so the cell then holds the tag of the EARLIEST lane that writes it —,
    // every lane reads its five inputs again and is held back iff one of them is a tag above its
    // own: an earlier writer of a cell it reads.  The earliest writer of a cell then stores the new
    // value (committed) or puts the old one back (held back).  Exact for any state count, no byte
    // of LDS beside the table.
    auto td_of = [&](float q, float m) -> float {
      if (AGENT == COBEL_AGENT_DYNAQ) {
        // planning TD in float64, one rounding on store (NumPy promotion of the reference's
        // expression with a float32 table; see include/cobel_hip.h)
        const double gnt = gamma * (double)nt;
        double td = (double)r + gnt * (double)m;
        td = td - (double)q;
        return (float)((double)q + alpha * td);
      }
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * m;
      td = td - q;
      return q + alpha_f * td;
    };
    uint32_t* const Qu = reinterpret_cast<uint32_t*>(Qf);
    const uint4* const Qs4u = reinterpret_cast<const uint4*>(Qs);
    const uint32_t tag_mine = ~(uint32_t)lane;
    // (`lo` = the first lane of the round: only a table that held a tag pattern to begin with — a NaN
    //  no arithmetic produces — can hold it back; it is committed regardless, so every round ends
    //  one lane further: garbage in, garbage out, never a batch that does not end)
    auto tag_round = [&](bool act, bool ch, float q, float qn, int lo) -> int {
      if (ch) atomicMax(&Qu[idx], tag_mine);
      __builtin_amdgcn_wave_barrier();
      uint32_t t = 0u, c2 = 0u;
      if (act) {
        const uint4 r2 = Qs4u[ns];
        c2 = Qu[idx];
        t = max(max(max(r2.x, r2.y), r2.z), max(r2.w, c2));
      }
#if defined(COBEL_ABLATE) && COBEL_ABLATE == 1
      t = 0u;
#endif
      const unsigned long long blocked = __builtin_amdgcn_ballot_w64(act && t > tag_mine);
      const int stop = max(blocked ? __ffsll((long long)blocked) - 1 : BP, lo + 1);
      if (ch && c2 == tag_mine) Qf[idx] = lane < stop ? qn : q;
      __builtin_amdgcn_wave_barrier();
      return stop;
    };
    // (the first round written out in front of the loop over the rounds — most batches end with
    //  it —, as in k_tab_pwg §4.1d: trained agents on 16 x 16 / 24 x 24 mazes +1.3 / +0.7 %,
    //  scripts/experiments/exp_occ_trained.py)
    int first;
    {
      float q = 0.0f, qn = 0.0f;
      if (on) {
        const float4 row = Qs[ns];
        q = Qf[idx];
        qn = td_of(q, max4(row));
      }
      const bool ch = on && __builtin_bit_cast(uint32_t, qn) != __builtin_bit_cast(uint32_t, q);
      if (!__builtin_amdgcn_ballot_w64(ch)) {
        STAMP(4);
        return;
      }
      first = tag_round(on, ch, q, qn, 0);
    }
    while (first < BP) {
      const bool act = on && lane >= first;
      float q = 0.0f, qn = 0.0f;
      if (act) {
        const float4 row = Qs[ns];
        q = Qf[idx];
        qn = td_of(q, max4(row));
      }
      const bool ch = act && __builtin_bit_cast(uint32_t, qn) != __builtin_bit_cast(uint32_t, q);
      if (!__ballot(ch)) break;
      first = tag_round(act, ch, q, qn, first);
    }
    STAMP(4);
  };
  // Dyna-Q batch drawn with x: model entries from LDS, reward estimates from HBM where flagged.
  // (fresh_idx, fresh_r): the entry this step wrote, whose HBM copy may still be in flight.
  auto plan_dynaq = [&](uint32_t x, uint32_t fresh_idx, float fresh_r) {
    uint32_t idx = 0, ns = 0, nt = 0;
    float r = 0.0f;
    if (lane < BP) {
      idx = cobel_bounded(x, SA);
      const uint32_t m = L.M16[idx];
      ns = m & 0x3fffu;
      nt = (m >> 14) & 1u;
      if (m & 0x8000u) {
        r = __builtin_bit_cast(float, model32[2u * idx]);
        asm volatile("" : "+v"(r));   // (consumed inside the branch, see the MIDX path)
      }
      if (idx == fresh_idx) r = fresh_r;
    }
    run_batch(idx, ns, nt, r);
  };
  // a whole batch: the first pass from the cached draw x, further passes (B > kPassLanes) from
  // their own Philox blocks — element j of the vector draw comes from sub-stream j
  auto plan_dynaq_batch = [&](uint32_t x, uint32_t fresh_idx, float fresh_r) {
    plan_dynaq(x, fresh_idx, fresh_r);
    if (!FAST && B > kPassLanes) {
      for (int j0 = kPassLanes; j0 < B; j0 += kPassLanes) {
        Bp = B - j0 < kPassLanes ? B - j0 : kPassLanes;
        const cobel_u4 b = cobel_philox(cm >> 2, (uint32_t)(j0 + lane), g, COBEL_STREAM_MEMORY, seed);
        plan_dynaq(cobel_word(b, cm & 3u), fresh_idx, fresh_r);
      }
      Bp = kPassLanes;
    }
  };
  auto replay_log = [&](uint64_t rec) {
    const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
    run_batch((hi & 0x3fffu) * 4u + ((hi >> 28) & 3u), (hi >> 14) & 0x3fffu, (hi >> 30) & 1u,
              __builtin_bit_cast(float, lo));
  };

  // interface/gridworld.py:142 — draw the start state of the next trial (false: all trials done)
  auto begin_trial = [&]() -> bool {
    if (trial >= A.r.trials_target) return false;
    state = (int)A.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                             start_cnt)];
    ce += 1u;
    step = 0;
    trew = 0.0;
    asm volatile("" : "+v"(trew));
    iflags |= 1u;
    enter_state(state);
    return true;
  };

  // ---- prologue -----------------------------------------------------------------------------
  // (a trial is begun where the previous one ends — also when the step budget is used up, as the
  //  start draw belongs to the trial count, not to the budget — so the loop itself never has to
  //  ask whether it is inside a trial)
  bool live = true;
  if (iflags & 1u) enter_state(state);
  else live = begin_trial();
  if (MIDX) {
    refresh_draws(cp >> 2);   // the pair this call's first action draw comes from
    {
      const int jp = 3 - (int)(cp & 3u), jm = 4 - (int)((cm + 1u) & 3u);
      refresh_in = jp < jm ? jp : jm;   // steps until one of the two caches runs dry
    }
    idx_cur = lane < B ? cobel_bounded(draw_m(cm), SA) : 0u;
    mg_cur = lane < B ? (uint32_t)Mg[idx_cur] : 0u;
  }
  if (AGENT == COBEL_AGENT_Q && replay_each_step) {
    qx = draw_m(cm);
    const uint32_t room = loglen < (uint32_t)A.r.log_cap ? 1u : 0u;
    qrec = log_gather(qx, loglen + room, loglen);
  }

  const int budget0 = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;
  int budget = budget0;   // steps executed by this call = budget0 - budget
  uint32_t batches = 0u;  // batches evaluated (a per-lane register on purpose: the loop is short
  asm volatile("" : "+v"(batches));   // of scalar ones)
#if defined(COBEL_STAMPS)
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

  while (live) {
    STAMP(5);
    if (budget == 0) break;
    budget -= 1;

    // ---- draws of this step (cached blocks; one Philox evaluation per ~4 steps) ---------------
    if (!MIDX) refresh_draws(cp >> 2);
#if defined(COBEL_STAMPS_FINE)
    STAMP(6);
#endif
    const int src_lane = 62 + (int)((cp >> 1) & 1u);
    const uint32_t w0 = rl((cp & 1u) ? blk.z : blk.x, src_lane);
    const uint32_t w1 = rl((cp & 1u) ? blk.w : blk.y, src_lane);
    if (MIDX) {
      // both counters advance by one per step here, so the step at which a cache runs dry is
      // known in advance: a countdown instead of two index comparisons per step
      if (__builtin_expect(refresh_in == 0, 0)) {
        refresh_draws((cp + 1u) >> 2, true);
        const int jp = 4 - (int)((cp + 1u) & 3u), jm = 4 - (int)((cm + 1u) & 3u);
        refresh_in = jp < jm ? jp : jm;
      }
      refresh_in -= 1;
    }
    const uint32_t mdraw = cobel_word(blk, cm & 3u);   // lanes < B: this step's batch
    cp += 1u;

    // ---- one batch of LDS reads serves the whole scalar part of the step ---------------------
    // Only four successors are possible, so lane k < 4 reads the Q row (and, for small worlds, the
    // world record) of next[state][k] now; once the action is known the chosen one is a readlane
    // away and the select -> step -> TD chain pays a single LDS round trip.
    // (lane l holds component l % 4 of Q[state]: ONE compare then gives the tie pattern — the low
    //  four bits of its ballot —, the row's maximum is two fused quad permutes away, and Q[s][a] a
    //  readlane: eight vector compares / selects and six scalar instructions less on the chain from
    //  the LDS answer to the action; k_tab_pwg 12.10 -> 11.52 ms per C3 launch with the same change)
    const float qc = reinterpret_cast<const float*>(Qs)[(uint32_t)state * 4u + (uint32_t)(lane & 3)];
    const uint32_t succ = next_of(cw0, cw1, lane & 3);
    const float4 srow = Qs[succ];
    const float smax = max4(srow);
    if (WLDS) cand = L.Wl[succ];
    if (AGENT == COBEL_AGENT_DYNAQ && !MIDX)   // the four 16-bit model entries of this state
      m4 = *reinterpret_cast<const uint2*>(&L.M16[state * 4]);

#if defined(COBEL_STAMPS_FINE)
    STAMP(7);
#endif
    // ---- select (policy/greedy.py:40-88) ------------------------------------------------------
    int a;
    if (mask_cur == 15u) {
      // integer thresholds of the tie pattern's CDF, held one per lane in thr_lo / thr_hi
      // (two wait states between a VALU write of a register and a DPP read of it: the compiler does
      //  not look for hazards inside an asm block.  Every lane of the wave is active here — a
      //  permute that reads a disabled lane leaves its destination unwritten)
      float m;
      asm("s_nop 1\n\t"
          "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
          : "=&v"(m)
          : "v"(qc));
      const int t = (int)((uint32_t)__ballot(qc == m) & 15u);
      // every lane e < 48 compares its own threshold (entry e = t * 3 + k) with the draw; the
      // three bits of this tie pattern in the ballot count the thresholds passed
      const uint64_t K = cobel_u53(w0, w1);
      const unsigned long long passed = __ballot(thr_mine <= K);
      a = __popcll((passed >> (t * 3)) & 7ull);
    } else {
      const float qx = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, qc), 0));
      const float qy = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, qc), 1));
      const float qz = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, qc), 2));
      const float qw = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, qc), 3));
      a = (int)rfl((uint32_t)cobel_eps_greedy_select_wave(qx, qy, qz, qw, mask_cur,
                                                          cobel_u01(w0, w1), ebb, lane));
    }
    STAMP(0);
    // ---- env.step (interface/gridworld.py:115-126) ----------------------------------------------
    int ns;
    uint32_t nw0, nw1, r_bits, end;
    float ns_max;
    if (STOCH) {
      // interface/gridworld.py:119-123: the successor is drawn from the row of sas — one double
      // of the env stream per step (sub-stream 1 of the counter the trial starts share, as
      // cobel_env_step_draw and k_tab_general draw it); any state may be entered, so its world
      // record and Q row are fetched behind the draw instead of taken from the prefetched four
      const double ue = cobel_draw_u01(ce, COBEL_SUB_DOUBLE, g, COBEL_STREAM_ENV, seed);
      ce += 1u;
      ns = (int)rfl((uint32_t)cobel_draw_successor(
          A.succ_off, A.succ_state, A.succ_cdf, ((size_t)world * S + (size_t)state) * 4 + a, ue));
      const uint4 c = WLDS ? L.Wl[ns] : W4[ns];
      nw0 = rfl(c.x);
      nw1 = rfl(c.y);
      r_bits = rfl(c.z);
      end = rfl(c.w);
      ns_max = __builtin_bit_cast(float, rfl(__builtin_bit_cast(uint32_t, max4(Qs[ns]))));
    } else {
      ns = (int)next_of(cw0, cw1, a);
      nw0 = rl(cand.x, a);
      nw1 = rl(cand.y, a);
      r_bits = rl(cand.z, a);
      end = rl(cand.w, a);
      ns_max = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, smax), a));
    }
    const float r = __builtin_bit_cast(float, r_bits);
    // COBEL_IF_NONZERO: while Q and the model's reward estimates are all +0.0f the only thing
    // that can change that is a reward other than +0.0f (TD = r + gamma 0 - 0)
    if (AGENT == COBEL_AGENT_DYNAQ && learn && r_bits != 0u) iflags |= 2u;
    const uint32_t nt = 1u - end;
    const uint32_t sa = (uint32_t)state * 4u + (uint32_t)a;
    const bool trial_over = end || (step + 1 >= A.r.steps_per_trial);
#if defined(COBEL_STAMPS_FINE)
    STAMP(8);
#endif
    // Successor records for the next step: issued before this step's stores (a wave's memory
    // operations retire in order, so a load issued behind a store would also wait for the
    // store's acknowledgement) and consumed one step later, behind the planning.
    // (MIDX: the model record gathered for THIS state is consumed before the next requests go out — a
    //  wait for a register loaded across the loop edge is a wait for every load in flight, and a
    //  64-bit load whose high half is dead lets the allocator reuse that half at once, which is a
    //  wait for the load as well)
    uint32_t mrec_now = 0u;
    if (MIDX) {
      mrec_now = rl(mrec, a);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t mrec_next = 0u;
    if (!trial_over) {
      if (lane < 4) {
        if (!WLDS) cand = W4[next_of(nw0, nw1, lane)];
        if (MIDX) mrec_next = model32[2u * ((uint32_t)ns * 4u + (uint32_t)lane)];
      }
      mask_cur = amask ? (uint32_t)amask[ns] & 15u : 15u;
    }

#if defined(COBEL_STAMPS_FINE)
    STAMP(9);
#endif
    uint32_t fresh_idx = ~0u;   // model / log entry written by this step
    uint32_t fresh_m = 0u;
    float fresh_r = 0.0f;
    uint64_t fresh_rec = 0;
    float td_online = 0.0f;
    if (learn) {
      if (AGENT == COBEL_AGENT_DYNAQ) {
        // memory/dyna_q.py:92-96 (float32): rewards[s,a] += lr * (r - rewards[s,a])
        float R = 0.0f;
        if (MIDX) {
          // the record of (state, a) came in with the successor records, one step ahead; if the
          // previous step rewrote it after that gather was issued, its value is still at hand
          R = __builtin_bit_cast(float, mrec_now);
          if (sa == fix_sa) R = fix_r;
        } else {
          const uint32_t pair = (a & 2) ? m4.y : m4.x;
          const uint32_t old = (a & 1) ? (pair >> 16) : (pair & 0xffffu);
          if (__builtin_expect((old & 0x8000u) != 0u, 0))
            R = __builtin_bit_cast(float, rfl(model32[2u * sa]));
        }
        const float d = r - R;
        const float Rn = R + mlr_f * d;
        const uint32_t rbits = __builtin_bit_cast(uint32_t, Rn);
        fresh_idx = sa;
        fresh_r = Rn;
        fresh_m = (uint32_t)ns | (nt << 14) | (rbits ? 0x8000u : 0u);
        // (if the next step is in this state again — a bumping move, or a trial that starts
        //  here — the records gathered for it may predate this store; the one entry concerned
        //  is replaced when it is consumed)
        fix_sa = sa;
        fix_r = Rn;
      } else if (rlog) {
        if (loglen < (uint32_t)A.r.log_cap) {
          fresh_rec = cobel_log_pack(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt);
          fresh_idx = loglen;
        }
      }
      // online TD (agent/dyna_q.py:290-299), float32; Q[ns] and Q[s][a] were read above
      const float q = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, qc), a));
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * ns_max;
      td = td - q;
      const float qn = q + alpha_f * td;
      if (lane == 0) {   // all of this step's stores in one predicated block
        Qf[sa] = qn;
        if (AGENT == COBEL_AGENT_DYNAQ) {
          model[sa] = cobel_model_pack(fresh_r, (uint32_t)ns, nt);
          if (MIDX) Mg[sa] = (uint16_t)fresh_m;
          else L.M16[sa] = (uint16_t)fresh_m;
          if (!MIDX && A.r.model_index) A.r.model_index[(size_t)i * SA + sa] = (uint16_t)fresh_m;
        } else if (rlog && fresh_idx != ~0u) {
          rlog[fresh_idx] = fresh_rec;
        }
      }
      if (AGENT == COBEL_AGENT_Q && rlog && fresh_idx != ~0u) loglen += 1u;
      td_online = td;
      __builtin_amdgcn_wave_barrier();
    }
    if (!FAST && A.r.last_exp && lane == 0) {
      int32_t* const e = A.r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = __builtin_bit_cast(int32_t, td_online);
    }

    STAMP(1);
    // ---- bookkeeping --------------------------------------------------------------------------
    trew += (double)r;
    if (OCC && lane == 0) L.occ[ns] += 1u;
    state = ns;
    cw0 = nw0;
    cw1 = nw1;

    // ---- planning / replay ----------------------------------------------------------------------
    if (replay_each_step) {
      if (AGENT == COBEL_AGENT_DYNAQ && MIDX) {
        // next step's batch: draw, start the gather; then run this step's batch
        uint32_t idx_next = 0u, mg_next = 0u;
        const uint32_t m = (idx_cur == fresh_idx) ? fresh_m : mg_cur;
        if (lane < (MULTI ? Bp : B)) {   // one predicated region for everything the replay lanes do
          float r = 0.0f;
          idx_next = cobel_bounded(cobel_word(blk, (cm + 1u) & 3u), SA);
          mg_next = (uint32_t)Mg[idx_next];
          if (iflags & 2u) {   // (COBEL_IF_NONZERO: otherwise no update of the batch can move Q)
            batches += 1u;
            if (__builtin_expect((m & 0x8000u) != 0u, 0)) {
              r = __builtin_bit_cast(float, model32[2u * idx_cur]);
              // (consumed here: a wait for this rare load after the branches have joined would be
              //  a wait for every gather this step has just sent out)
              asm volatile("" : "+v"(r));
            }
            if (idx_cur == fresh_idx) r = fresh_r;
            run_batch(idx_cur, m & 0x3fffu, (m >> 14) & 1u, r, true);
          }
          if (idx_next == fresh_idx) mg_next = fresh_m;   // the gather may have passed the store
        }
        if (MULTI && (iflags & 2u)) {
          // updates kPassLanes .. B - 1 of this step's batch: element j of the vector draw comes
          // from sub-stream j of the same counter (as the generic kernel's passes)
          for (int j0 = kPassLanes; j0 < B; j0 += kPassLanes) {
            Bp = B - j0 < kPassLanes ? B - j0 : kPassLanes;
            const cobel_u4 b =
                cobel_philox(cm >> 2, (uint32_t)(j0 + lane), g, COBEL_STREAM_MEMORY, seed);
            uint32_t idx2 = 0u, m2 = 0u;
            float r2 = 0.0f;
            if (lane < Bp) {
              idx2 = cobel_bounded(cobel_word(b, cm & 3u), SA);
              m2 = (uint32_t)Mg[idx2];
              if (idx2 == fresh_idx) m2 = fresh_m;   // (this step's own store may still be in flight)
              if (__builtin_expect((m2 & 0x8000u) != 0u, 0)) {
                r2 = __builtin_bit_cast(float, model32[2u * idx2]);
                asm volatile("" : "+v"(r2));
              }
              if (idx2 == fresh_idx) r2 = fresh_r;
            }
            run_batch(idx2, m2 & 0x3fffu, (m2 >> 14) & 1u, r2);
          }
          Bp = kPassLanes;
        }
        idx_cur = idx_next;
        mg_cur = mg_next;
        mrec = mrec_next;
        cm += 1u;
      } else if (AGENT == COBEL_AGENT_DYNAQ) {
        if (iflags & 2u) {
          plan_dynaq_batch(mdraw, fresh_idx, fresh_r);
          batches += 1u;
        }
        cm += 1u;
      } else {
        // experiences of the next batch are gathered now, behind this batch's updates
        const uint32_t bound_now = loglen;
        const uint32_t qx_next = draw_m(cm + 1u);
        const uint32_t room = loglen < (uint32_t)A.r.log_cap ? 1u : 0u;
        const uint32_t bound_next = loglen + room;
        uint64_t qrec_next = log_gather(qx_next, bound_next, loglen);
        if (bound_now > 0u) {
          if (lane < BP && cobel_bounded(qx, bound_now) == fresh_idx) qrec = fresh_rec;
          replay_log(qrec);
          // updates kPassLanes .. B - 1 of this step's batch (agent/q.py:344-354 draws ONE vector
          // of indices: element j comes from sub-stream j), gathered here — the log as it stands,
          // this step's own record from registers — and applied pass by pass, in order
          if (__builtin_expect(B > kPassLanes, 0)) {
            for (int j0 = kPassLanes; j0 < B; j0 += kPassLanes) {
              Bp = B - j0 < kPassLanes ? B - j0 : kPassLanes;
              const cobel_u4 b = cobel_philox(cm >> 2, (uint32_t)(j0 + lane), g, COBEL_STREAM_MEMORY, seed);
              uint64_t rec2 = 0;
              if (lane < Bp) {
                const uint32_t idx2 = cobel_bounded(cobel_word(b, cm & 3u), bound_now);
                rec2 = idx2 == fresh_idx ? fresh_rec : rlog[idx2];
              }
              replay_log(rec2);
            }
            Bp = kPassLanes;
          }
          batches += 1u;
        }
        cm += 1u;
        qx = qx_next;
        // the gather may have passed this step's own append
        if (lane < BP && bound_next > 0u && cobel_bounded(qx, bound_next) == fresh_idx)
          qrec_next = fresh_rec;
        qrec = qrec_next;
      }
    }

    if (__builtin_expect(trial_over, 0)) {
      // agent/dyna_q.py:207-212: current_trial += 1; logs['steps'] = step (0-based)
      if (lane == 0 && trial >= 0 && trial < A.r.trial_cap) {
        const size_t m = mon_stripe_offset(A.r) + (size_t)trial;
        if (A.r.lat_sum) atomicAdd(A.r.lat_sum + m, (unsigned long long)step);
        if (A.r.lat_cnt) atomicAdd(A.r.lat_cnt + m, 1ull);
        if (A.r.reward_sum) atomicAdd(A.r.reward_sum + m, trew);
        if (A.r.resp_cnt && trew > 0.0) atomicAdd(A.r.resp_cnt + m, 1ull);
        if (A.r.lat_trace) A.r.lat_trace[(size_t)i * A.r.trial_cap + trial] = step;
      }
      trial += 1;
      iflags &= ~1u;
      if (episodic && B > 0) {
        refresh_draws(cp >> 2);
        if (iflags & 2u) {
          plan_dynaq_batch(cobel_word(blk, cm & 3u), fresh_idx, fresh_r);
          batches += 1u;
        }
        cm += 1u;
      }
      if (!begin_trial()) break;
    } else {
      step += 1;
    }
  }

#if defined(COBEL_STAMPS)
  if (A.r.last_exp && lane == 0)
    for (int k = 0; k < 6; ++k) A.r.last_exp[(size_t)i * 6 + k] = (int32_t)(stamp_sum[k] >> 4);
#if defined(COBEL_STAMPS_FINE)
  if (A.r.last_exp && lane == 0) {
    A.r.last_exp[(size_t)i * 6 + 2] = (int32_t)(stamp_sum[6] >> 4);
    A.r.last_exp[(size_t)i * 6 + 3] = (int32_t)(stamp_sum[7] >> 4);
    A.r.last_exp[(size_t)i * 6 + 4] = (int32_t)(stamp_sum[8] >> 4);
    A.r.last_exp[(size_t)i * 6 + 5] = (int32_t)(stamp_sum[9] >> 4);
  }
#endif
#endif
  // ---- write back ---------------------------------------------------------------------------
  __syncthreads();
  for (int s = lane; s < S; s += 64) {
    Qg[s] = Qs[s];
    if (OCC) {
      const uint32_t c = L.occ[s];
      if (c && A.r.occupancy) atomicAdd(A.r.occupancy + (size_t)world * S + s, (unsigned long long)c);
    }
  }
  if (lane == 0) {
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_CTR_MEMORY] = (int32_t)cm;
    inst[COBEL_I_LOG_LEN] = (int32_t)loglen;
    inst[COBEL_I_FLAGS] = (int32_t)(iflags & ~2u);   // (COBEL_IF_NONZERO lives in the launch only)
    const unsigned long long executed = (unsigned long long)(budget0 - budget);
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) += executed;
    if (A.r.steps_done && executed) atomicAdd(A.r.steps_done, executed);
    if (A.r.batches_done && batches) atomicAdd(A.r.batches_done, (unsigned long long)batches);
  }
}

// ---------------------------------------------------------------------------------------------
// Lane-per-instance variant for runs without planning (QAgent with batch 0, or Agent.test()):
// with nothing to spread over the lanes of a wave, 64 instances share one wavefront and every
// instruction of the select -> step -> TD chain serves 64 env steps.  Each lane keeps its
// instance's Q table in its own LDS column (row s of lane l at ((s * LPW + l) * 16) bytes, so a
// 16-byte row read is bank-conflict free across the wave), plus its own stream counters and
// cached Philox blocks.  Usable while 1 KiB * S fits in LDS.
//
// Per-trial monitors: thousands of lanes finishing trials every step would serialise on global
// atomics, so each wave accumulates them in LDS and adds them to the global arrays once, at the
// end of the call.  The accumulators are a dense window of kMonSlots trial indices starting at
// the smallest trial index any lane of the wave has at launch (slot = trial - base): the lanes of
// a wave drift apart by a few hundred trials at most and advance a few hundred per launch.  A
// trial beyond the window goes to the global arrays directly.  (An earlier direct-mapped cache
// with tags and evictions cost more than the step itself once the lanes had drifted apart: 2.1 ms
// per launch on C2 against 0.8 ms without monitors.)
constexpr int kMonSlots = 512;
constexpr int kMonBytes = kMonSlots * 16;

// EXTRA: action masks, the last experience, per-trial latency traces or a run that does not learn
// — each a wave-uniform branch, TAKEN in every step of a plain training run.  One wave per SIMD has
// nobody to run while its instruction buffer refills behind a taken branch (the counters of
// exp_c2_parts.py / r05: 17 branches and 48 instruction fetches per wave-step, no LDS wait to speak
// of), so plain training is its own instantiation without them.
template <bool ONE_WORLD, bool MON, bool LOG, bool EXTRA>
__global__ __launch_bounds__(64) void k_tab_lpi(const tab_args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int S = A.S;
  float4* const Ql = reinterpret_cast<float4*>(lds_raw);
  float* const Qlf = reinterpret_cast<float*>(lds_raw);
  // LPW instances per wave (lanes >= LPW idle): 64, or fewer when the Q columns of 64 instances
  // would not fit in LDS.
  const int LPW = A.lpw;
  const size_t qbytes = (size_t)S * LPW * 16;
  uint64_t* const thr = reinterpret_cast<uint64_t*>(lds_raw + qbytes);
  uint4* const Wl = reinterpret_cast<uint4*>(lds_raw + qbytes + kThrBytes);
  unsigned char* const mon_raw = lds_raw + qbytes + kThrBytes + (size_t)S * 16;
  double* const wrew = reinterpret_cast<double*>(mon_raw);                      // [kMonSlots]
  uint32_t* const wsum = reinterpret_cast<uint32_t*>(mon_raw + kMonSlots * 8);  // [kMonSlots]
  uint32_t* const wcnt = wsum + kMonSlots;                                      // [kMonSlots]
  // ONE_WORLD: the start list next to the world records (a trial start is then LDS-only)
  uint16_t* const Sl = reinterpret_cast<uint16_t*>(mon_raw + (MON ? kMonBytes : 0));

  const int lane = (int)threadIdx.x;
  const int i = (int)blockIdx.x * LPW + lane;
  const bool active = lane < LPW && i < A.r.n;
  const int ii = active ? i : 0;
  const uint32_t g = A.r.instance_base + (uint32_t)ii;
  const int world = ONE_WORLD ? 0 : (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  float4* const Qg = reinterpret_cast<float4*>(A.r.q) + (size_t)ii * S;

  if (lane < LPW)
    for (int s = 0; s < S; ++s) Ql[s * LPW + lane] = Qg[s];
  if (ONE_WORLD) {
    for (int s = lane; s < S; s += 64) Wl[s] = W4[s];
    const int ns0 = A.start_off[1] - A.start_off[0];
    for (int k = lane; k < ns0; k += 64) Sl[k] = A.starts[A.start_off[0] + k];
  }
  if (lane < 48) thr[lane] = A.eps.thr[lane / 3][lane % 3];
  if (MON)
    for (int k = lane; k < kMonSlots; k += 64) {
      wrew[k] = 0.0;
      wsum[k] = 0u;
      wcnt[k] = 0u;
    }
  __syncthreads();
  auto wrec = [&](int s) -> uint4 { return ONE_WORLD ? Wl[s] : W4[s]; };
  // add (steps, finished, rewarded, reward) of trial t to the global monitors
  auto to_global = [&](int t, uint32_t steps, uint32_t c, uint32_t rc, double rew) {
    if (c && t >= 0 && t < A.r.trial_cap) {
      const size_t m = mon_stripe_offset(A.r) + (size_t)t;
      if (A.r.lat_sum) atomicAdd(A.r.lat_sum + m, (unsigned long long)steps);
      if (A.r.lat_cnt) atomicAdd(A.r.lat_cnt + m, (unsigned long long)c);
      if (A.r.reward_sum) atomicAdd(A.r.reward_sum + m, rew);
      if (A.r.resp_cnt && rc) atomicAdd(A.r.resp_cnt + m, (unsigned long long)rc);
    }
  };

  int32_t* const inst = A.r.inst + (size_t)ii * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  // the monitor window starts at the smallest trial index in the wave
  int mon_base = active ? trial : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mon_base = min(mon_base, __shfl_xor(mon_base, o));
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);
  unsigned long long nsteps = *reinterpret_cast<const unsigned long long*>(inst + COBEL_I_STEPS_LO);
  // LOG: QAgent appends every experience to its memory even when nothing is replayed
  // (agent/q.py:213) — one 8-byte record per step into the instance's row of the log
  uint64_t* const rlog = LOG ? A.r.replay_log + (size_t)ii * A.r.log_cap : nullptr;
  uint32_t loglen = LOG ? (uint32_t)inst[COBEL_I_LOG_LEN] : 0u;

  const uint32_t flags = A.r.flags;
  const bool learn = EXTRA ? (flags & COBEL_F_LEARN) != 0u : true;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (EXTRA && (flags & COBEL_F_MASK_ACTIONS)) ? A.r.action_mask : nullptr;
  int32_t* const last_exp = EXTRA ? A.r.last_exp : nullptr;
  const bool traces = EXTRA && A.r.lat_trace != nullptr;
  const uint64_t seed = A.r.seed;
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  const float alpha_f = A.alpha_f, gamma_f = A.gamma_f;

  cobel_u4 pblk = {0, 0, 0, 0}, eblk = {0, 0, 0, 0}, eblk2 = {0, 0, 0, 0};
  uint32_t pb_idx = ~0u, eb_idx = ~0u;
  bool eb2_ok = false;
  uint32_t cw0 = 0, cw1 = 0;
  if (iflags & 1u) {
    const uint4 c = wrec(state);
    cw0 = c.x;
    cw1 = c.y;
  }
  const int budget0 = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;
  int budget = budget0;   // steps executed by this lane = budget0 - budget (two 64-bit counters less per step)
  bool done = !active;

  for (;;) {
    // The block of the env stream a trial start draws from (four starts per block).  The lanes of a
    // wave start trials at different steps — with trained agents some lane does in almost every
    // step —, and a Philox block costs the WAVE its 18 multiplications whoever needs it.  So when
    // one lane has to have a block, every lane works one out: the one its next start needs, or,
    // having that, the one after it (eblk2).  A wave then pays for the rounds every ~20 steps
    // instead of every step; same blocks, same words (the stream is counter based).
    {
      const bool starts = !done && !(iflags & 1u) && trial < A.r.trials_target;
      const bool have = (ce >> 2) == eb_idx || (eb2_ok && (ce >> 2) == eb_idx + 1u);
      if (__builtin_expect(__any(starts && !have), 0)) {
        if (eb2_ok && (ce >> 2) == eb_idx + 1u) {   // the block after has become the current one
          eblk = eblk2;
          eb_idx += 1u;
          eb2_ok = false;
        }
        const bool cur = (ce >> 2) != eb_idx;
        if (!done && (cur || !eb2_ok)) {
          const cobel_u4 b = cobel_philox((ce >> 2) + (cur ? 0u : 1u), 0u, g, COBEL_STREAM_ENV, seed);
          if (cur) {
            eblk = b;
            eb_idx = ce >> 2;
            eb2_ok = false;
          } else {
            eblk2 = b;
            eb2_ok = true;
          }
        }
      }
    }
    if (!done && !(iflags & 1u)) {
      if (trial >= A.r.trials_target) {
        done = true;
      } else {
        const bool second = (ce >> 2) != eb_idx;   // (then: eb_idx + 1, in eblk2)
        const int pick = (int)cobel_bounded(
            second ? cobel_word(eblk2, ce & 3u) : cobel_word(eblk, ce & 3u), start_cnt);
        state = ONE_WORLD ? (int)Sl[pick] : (int)A.starts[start_lo + pick];
        ce += 1u;
        step = 0;
        trew = 0.0;
        iflags |= 1u;
        const uint4 c = wrec(state);
        cw0 = c.x;
        cw1 = c.y;
      }
    }
    if (!done && budget == 0) done = true;
    if (!__any(!done)) break;
    bool ended = false;
    if (!done) {
      budget -= 1;
      if ((cp >> 1) != pb_idx) {
        pb_idx = cp >> 1;
        pblk = cobel_philox(pb_idx, 0u, g, pol_stream, seed);
      }
      const uint32_t w0 = (cp & 1u) ? pblk.z : pblk.x, w1 = (cp & 1u) ? pblk.w : pblk.y;
      cp += 1u;
      const float4 q = Ql[state * LPW + lane];
      const uint32_t mask = amask ? (uint32_t)amask[state] & 15u : 15u;
      int a;
      if (mask == 15u) {
        const float m = max4(q);
        const int t = (int)(q.x == m) | ((int)(q.y == m) << 1) | ((int)(q.z == m) << 2) |
                      ((int)(q.w == m) << 3);
        const uint64_t K = cobel_u53(w0, w1);
        a = (int)(thr[t * 3] <= K) + (int)(thr[t * 3 + 1] <= K) + (int)(thr[t * 3 + 2] <= K);
      } else {
        a = cobel_eps_greedy_select(q.x, q.y, q.z, q.w, mask, cobel_u01(w0, w1), A.eps);
      }
      const int ns = (int)next_of(cw0, cw1, a);
      const uint4 nrec = wrec(ns);
      const float r = __builtin_bit_cast(float, nrec.z);
      const uint32_t end = nrec.w, nt = 1u - end;
      float td_online = 0.0f;
      if (learn) {   // online TD (agent/q.py:305-313), float32
        const float4 nrow = Ql[ns * LPW + lane];
        const int cell = (state * LPW + lane) * 4 + a;
        const float qsa = Qlf[cell];
        const float gnt = nt ? gamma_f : 0.0f;
        float td = r + gnt * max4(nrow);
        td = td - qsa;
        Qlf[cell] = qsa + alpha_f * td;
        td_online = td;
        if (LOG && loglen < (uint32_t)A.r.log_cap) {
          rlog[loglen] = cobel_log_pack(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt);
          loglen += 1u;
        }
      }
      if (last_exp) {
        int32_t* const e = last_exp + (size_t)ii * 6;
        e[0] = state;
        e[1] = a;
        e[2] = ns;
        e[3] = (int32_t)nt;
        e[4] = __builtin_bit_cast(int32_t, r);
        e[5] = __builtin_bit_cast(int32_t, td_online);
      }
      trew += (double)r;
      const bool trial_over = end || (step + 1 >= A.r.steps_per_trial);
      state = ns;
      cw0 = nrec.x;
      cw1 = nrec.y;
      ended = trial_over;
      if (!trial_over) step += 1;
    }
    // ---- trial ends (agent/q.py:222-224: current_trial += 1; logs['steps'] = step) ------------
    if (__any(ended)) {
      if (MON && ended) {
        const int slot = trial - mon_base;
        if (__builtin_expect(slot < kMonSlots, 1)) {
          // (low half of wcnt: instances that finished the trial; high half: those with a positive
          //  reward — one wave, so neither exceeds 64)
          atomicAdd(&wsum[slot], (uint32_t)step);
          atomicAdd(&wcnt[slot], trew > 0.0 ? 0x10001u : 1u);
          atomicAdd(&wrew[slot], trew);
        } else {
          to_global(trial, (uint32_t)step, 1u, trew > 0.0 ? 1u : 0u, trew);
        }
      }
      if (ended) {
        if (traces && trial >= 0 && trial < A.r.trial_cap)
          A.r.lat_trace[(size_t)ii * A.r.trial_cap + trial] = step;
        trial += 1;
        iflags &= ~1u;
      }
    }
  }
  if (MON) {
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < kMonSlots; k += 64)
      to_global(mon_base + k, wsum[k], wcnt[k] & 0xffffu, wcnt[k] >> 16, wrew[k]);
  }

  __syncthreads();
  if (active) {
    if (learn)
      for (int s = 0; s < S; ++s) Qg[s] = Ql[s * LPW + lane];
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_FLAGS] = (int32_t)iflags;
    if (LOG) inst[COBEL_I_LOG_LEN] = (int32_t)loglen;
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) =
        nsteps + (unsigned long long)(budget0 - budget);
  }
  if (A.r.steps_done) {   // one atomic per wave
    unsigned long long tot = active ? (unsigned long long)(budget0 - budget) : 0ull;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0 && tot) atomicAdd(A.r.steps_done, tot);
  }
}

__global__ __launch_bounds__(256) void k_model_init(uint64_t* __restrict__ model, size_t total,
                                                    int S4) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t s = (uint32_t)((t % (size_t)S4) >> 2);
  model[t] = cobel_model_pack(0.0f, s, 0u);
}

// 16-bit index entries from packed model records (see k_tab_wpi: next | nt << 14 | (R != 0) << 15)
__global__ __launch_bounds__(256) void k_model_index(const uint64_t* __restrict__ model,
                                                     uint16_t* __restrict__ index, size_t total) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint64_t rec = model[t];
  const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
  index[t] = (uint16_t)((hi & 0x3fffu) | (((hi >> 16) & 1u) << 14) | (lo ? 0x8000u : 0u));
}

template <int AGENT, bool OCC, bool WLDS, bool FAST, bool MIDX, bool PSETS, bool STOCH = false,
          bool MULTI = false>
int launch_wpi(const tab_args& A, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    COBEL_HIP_TRY(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_tab_wpi<AGENT, OCC, WLDS, FAST, MIDX, PSETS, STOCH, MULTI>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_tab_wpi<AGENT, OCC, WLDS, FAST, MIDX, PSETS, STOCH, MULTI>), dim3(A.r.n),
                     dim3(64), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

template <int AGENT, bool FAST, bool MIDX, bool PSETS>
int dispatch_wpi2(const tab_args& A, bool occ, bool wlds, size_t lds, hipStream_t st) {
  if (!FAST && A.succ_off) {   // (drawn successors: generic form only, see cobel_tab_run)
    if (occ) return wlds ? launch_wpi<AGENT, true, true, false, false, PSETS, true>(A, lds, st)
                         : launch_wpi<AGENT, true, false, false, false, PSETS, true>(A, lds, st);
    return wlds ? launch_wpi<AGENT, false, true, false, false, PSETS, true>(A, lds, st)
                : launch_wpi<AGENT, false, false, false, false, PSETS, true>(A, lds, st);
  }
  if (FAST && MIDX && A.r.batch > kPassLanes) {   // (several passes per step: see MULTI)
    if (occ) return wlds ? launch_wpi<AGENT, true, true, FAST, MIDX, PSETS, false, FAST && MIDX>(A, lds, st)
                         : launch_wpi<AGENT, true, false, FAST, MIDX, PSETS, false, FAST && MIDX>(A, lds, st);
    return wlds ? launch_wpi<AGENT, false, true, FAST, MIDX, PSETS, false, FAST && MIDX>(A, lds, st)
                : launch_wpi<AGENT, false, false, FAST, MIDX, PSETS, false, FAST && MIDX>(A, lds, st);
  }
  if (occ) return wlds ? launch_wpi<AGENT, true, true, FAST, MIDX, PSETS>(A, lds, st)
                       : launch_wpi<AGENT, true, false, FAST, MIDX, PSETS>(A, lds, st);
  return wlds ? launch_wpi<AGENT, false, true, FAST, MIDX, PSETS>(A, lds, st)
              : launch_wpi<AGENT, false, false, FAST, MIDX, PSETS>(A, lds, st);
}

// The plain-training kernels exist with and without parameter sets; the generic ones (masks,
// episodic replay, test runs, QAgent replay) always read them through the same code.
template <int AGENT>
int dispatch_wpi(const tab_args& A, bool occ, bool wlds, bool fast, bool midx, size_t lds,
                 hipStream_t st) {
  const bool psets = A.r.param_index != nullptr;
  if (AGENT == COBEL_AGENT_DYNAQ && fast && midx)
    return psets ? dispatch_wpi2<COBEL_AGENT_DYNAQ, true, true, true>(A, occ, wlds, lds, st)
                 : dispatch_wpi2<COBEL_AGENT_DYNAQ, true, true, false>(A, occ, wlds, lds, st);
  if (AGENT == COBEL_AGENT_DYNAQ && fast)
    return psets ? dispatch_wpi2<COBEL_AGENT_DYNAQ, true, false, true>(A, occ, wlds, lds, st)
                 : dispatch_wpi2<COBEL_AGENT_DYNAQ, true, false, false>(A, occ, wlds, lds, st);
  return psets ? dispatch_wpi2<AGENT, false, false, true>(A, occ, wlds, lds, st)
               : dispatch_wpi2<AGENT, false, false, false>(A, occ, wlds, lds, st);
}

}  // namespace

static const int kLdsLimit = 160 * 1024;
static const int kWorldLdsStates = 256;  // worlds up to this size keep their records in LDS

extern "C" int cobel_tab_query(int32_t n_states, int32_t agent, int32_t batch,
                               int32_t* lds_bytes, int32_t* instances_per_block) {
  COBEL_REQUIRE(n_states > 0, COBEL_E_RANGE, "cobel_tab_query: n_states = %d", n_states);
  COBEL_REQUIRE(agent == COBEL_AGENT_Q || agent == COBEL_AGENT_DYNAQ, COBEL_E_ARG,
                "cobel_tab_query: unknown agent %d", agent);
  COBEL_REQUIRE(batch >= 0 && batch <= COBEL_MAX_BATCH, COBEL_E_UNSUPPORTED,
                "cobel_tab_query: batch %d outside 0..%d", batch, COBEL_MAX_BATCH);
  // worst case: replay hash table and visit counters present
  const long long lds = (long long)tab_lds_bytes(n_states, agent, true, n_states <= kWorldLdsStates, true);
  COBEL_REQUIRE(lds <= kLdsLimit && n_states <= 16384, COBEL_E_UNSUPPORTED,
                "cobel_tab_query: %d states need %lld B of LDS per instance (limit %d)", n_states,
                lds, kLdsLimit);
  if (lds_bytes) *lds_bytes = (int32_t)lds;
  if (instances_per_block) *instances_per_block = 1;
  return COBEL_OK;
}
extern "C" int cobel_model_init(uint64_t* model, int32_t n, int32_t n_states, void* stream) {
  COBEL_REQUIRE(model, COBEL_E_ARG, "cobel_model_init: NULL model");
  COBEL_REQUIRE(n >= 0 && n_states > 0, COBEL_E_RANGE, "cobel_model_init: bad sizes");
  const size_t total = (size_t)n * n_states * 4;
  if (total == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_model_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, model, total, n_states * 4);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_model_index_build(const uint64_t* model, uint16_t* index, int32_t n,
                                       int32_t n_states, void* stream) {
  COBEL_REQUIRE(model && index, COBEL_E_ARG, "cobel_model_index_build: NULL argument");
  COBEL_REQUIRE(n >= 0 && n_states > 0 && n_states <= 16384, COBEL_E_RANGE,
                "cobel_model_index_build: bad sizes");
  const size_t total = (size_t)n * n_states * 4;
  if (total == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_model_index, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, model, index, total);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// describe != NULL: report which kernel the run would take instead of launching it
static int tab_run_impl(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream,
                        int32_t* describe) {
  COBEL_REQUIRE(world && run, COBEL_E_ARG, "cobel_tab_run: NULL world/run");
  if (int rc = cobel_world_check(world, "cobel_tab_run")) return rc;
  const cobel_tab_run_t& r = *run;
  COBEL_REQUIRE(r.q && r.inst, COBEL_E_ARG, "cobel_tab_run: q and inst are required");
  COBEL_REQUIRE(((uintptr_t)r.q & (world->n_actions == 4 ? 15u : 3u)) == 0 &&
                    ((uintptr_t)r.inst & 7u) == 0, COBEL_E_ARG,
                "cobel_tab_run: q must be 16-byte and inst 8-byte aligned");
  COBEL_REQUIRE(r.n >= 0, COBEL_E_RANGE, "cobel_tab_run: n = %d", r.n);
  COBEL_REQUIRE(r.agent == COBEL_AGENT_Q || r.agent == COBEL_AGENT_DYNAQ, COBEL_E_ARG,
                "cobel_tab_run: unknown agent %d", r.agent);
  COBEL_REQUIRE(r.agent != COBEL_AGENT_DYNAQ || r.model, COBEL_E_ARG,
                "cobel_tab_run: Dyna-Q needs the model table");
  COBEL_REQUIRE(r.agent != COBEL_AGENT_DYNAQ || world->n_actions == 4, COBEL_E_UNSUPPORTED,
                "cobel_tab_run: Dyna-Q model records are laid out for four-action worlds (this one "
                "has %d); Q-learning takes any action count", world->n_actions);
  COBEL_REQUIRE(r.batch >= 0, COBEL_E_RANGE, "cobel_tab_run: batch %d", r.batch);
  COBEL_REQUIRE(r.steps_per_trial > 0, COBEL_E_RANGE, "cobel_tab_run: steps_per_trial = %d",
                r.steps_per_trial);
  COBEL_REQUIRE(r.epsilon >= 0.0 && r.epsilon <= 1.0, COBEL_E_ARG,
                "cobel_tab_run: epsilon %g outside [0, 1]", r.epsilon);
  COBEL_REQUIRE(!(r.flags & COBEL_F_MASK_ACTIONS) || r.action_mask, COBEL_E_ARG,
                "cobel_tab_run: mask_actions set without an action mask");
  COBEL_REQUIRE(r.trial_cap >= 0 && r.log_cap >= 0, COBEL_E_RANGE, "cobel_tab_run: negative cap");
  COBEL_REQUIRE(!r.param_index || (r.param_sets && r.n_param_sets > 0), COBEL_E_ARG,
                "cobel_tab_run: param_index given without parameter sets");
  // Runs outside what the wavefront kernels are built for — an action count other than four
  // (hexagonal topologies), transition rows that are distributions (the successor is drawn),
  // more than COBEL_MAX_BATCH updates per step, tables beyond LDS — take
  // the general kernel (general.hip: one lane per instance, every update in sequence).
  int32_t lds_max = 0;
  // (batches above COBEL_MAX_BATCH run as several passes of the wavefront kernels: Dyna-Q's pairs
  //  and — round 6 — QAgent's log records beyond the first pass are drawn and gathered in the step)
  const int32_t pass = r.batch > COBEL_MAX_BATCH ? COBEL_MAX_BATCH : r.batch;
  // (worlds whose transition rows are distributions: the generic wavefront kernels draw the
  //  successor in the step — no fast / digest / lane-per-instance / persistent form for them)
  const bool draws = world->succ_off != nullptr;
  const bool replays = (r.flags & COBEL_F_LEARN) && !(r.flags & COBEL_F_NO_REPLAY) && r.batch > 0 &&
                       (r.agent == COBEL_AGENT_DYNAQ || r.replay_log != nullptr);
  // (without replayed updates a lane per instance is the better shape — what k_tab_lpi is for
  //  one-hot rows; measured on slippery 10 x 10 / 32 x 32 worlds, 65 536 instances x 512 steps:
  //  Q-learning 6.0 / 6.3 ms on k_tab_general against 11.4 / 20.8 on the wavefront kernel, Dyna-Q
  //  B 32 44 / 72 ms against 14.6 / 34.9: scripts/experiments/exp_tab_slippery.py)
  const bool general = world->n_actions != 4 || (draws && !replays) ||
                       (r.batch > COBEL_MAX_BATCH && r.agent != COBEL_AGENT_DYNAQ &&
                        (r.flags & COBEL_F_EPISODIC)) ||
                       (r.flags & COBEL_F_TAB_GENERAL) ||
                       cobel_tab_query(world->n_states, r.agent, pass, &lds_max, nullptr) != COBEL_OK;
  // Q-learning on worlds of other action counts whose tables fit the LDS: one wavefront per instance
  {
    size_t nact_lds = 0;
    int nact_ipw = 0;
    if (cobel_tab_nact_covers(world, r, &nact_lds, &nact_ipw)) {
      if (describe) {
        describe[0] = COBEL_TAB_KERNEL_WQN;
        describe[1] = (int32_t)nact_lds;
        describe[2] = lds_workgroups_per_cu(nact_lds);
        describe[3] = nact_ipw;
        return COBEL_OK;
      }
      return cobel_tab_nact_launch(world, r, (hipStream_t)stream);
    }
  }
  if (general) {
    COBEL_REQUIRE(world->n_actions == 4 || !r.model_index, COBEL_E_ARG,
                  "cobel_tab_run: the model digest exists for four-action worlds only");
    if (describe) {
      describe[0] = r.n ? COBEL_TAB_KERNEL_GENERAL : 0;
      describe[1] = 0;
      describe[2] = 0;
      describe[3] = r.n ? 64 : 0;
      return COBEL_OK;
    }
    if (r.n == 0) return COBEL_OK;
    return cobel_tab_general_launch(world, r, (hipStream_t)stream);
  }
  if (r.n == 0) return COBEL_OK;

  tab_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = world->n_states;
  A.n_worlds = world->n_worlds;
  A.r = r;
  A.eps = cobel_make_eps_consts(r.epsilon);
  A.alpha_f = (float)r.alpha;
  A.gamma_f = (float)r.gamma;
  A.model_lr_f = (float)r.model_lr;
  A.succ_off = world->succ_off;
  A.succ_state = world->succ_state;
  A.succ_cdf = world->succ_cdf;
  const bool occ = r.occupancy != nullptr;
  const bool wlds = world->n_states <= kWorldLdsStates;
  const bool replay = (r.flags & COBEL_F_LEARN) && !(r.flags & COBEL_F_NO_REPLAY) && r.batch > 0 &&
                      (r.agent == COBEL_AGENT_DYNAQ || r.replay_log != nullptr);
  // (more than COBEL_MAX_BATCH updates per step: plain training takes them in passes only with the
  //  digest in HBM — the MULTI instantiations)
  const bool digest = r.model_index != nullptr && !(r.flags & COBEL_F_FORCE_LDS_MODEL);
  const bool fast = !draws && r.agent == COBEL_AGENT_DYNAQ && replay && !(r.flags & COBEL_F_EPISODIC) &&
                    (r.batch <= COBEL_MAX_BATCH || digest) &&
                    !(r.flags & (COBEL_F_MASK_ACTIONS | COBEL_F_TEST_STREAM)) && !r.last_exp;
  const bool midx = fast && r.model_index != nullptr && !(r.flags & COBEL_F_FORCE_LDS_MODEL);
  // (per launch: scripts/experiments/exp_occupancy.py; validated against the limit below)
  static const char* const lpw_env = cobel_debug_env("COBEL_DEBUG_LPW");
  size_t lds = tab_lds_bytes(world->n_states, r.agent, replay, wlds, occ, midx);
  const size_t lds_pad = cobel_debug_lds_pad(lds, (size_t)kLdsLimit);   // occupancy experiments
  lds += lds_pad;
  hipStream_t st = (hipStream_t)stream;
  // No planning in this call and nothing but Q to keep per instance: 64 instances per wave.
  const bool learn = (r.flags & COBEL_F_LEARN) != 0;
  // instances per wave: 64 unless the Q columns of a full wave do not fit in LDS.  (Narrower waves
  // to give each SIMD more than one resident wave were measured on C2 — 65 536 instances are one
  // wave per SIMD — and lose: 1.25 ms per launch at 64, 1.45 ms at 32, 2.57 ms at 16 instances per
  // wave; the step is as much issue- as latency-bound.  COBEL_DEBUG_LPW overrides.)
  int lpw = 64;
  if (lpw_env) {
    const int v = atoi(lpw_env);
    if (v == 16 || v == 32 || v == 64) lpw = v;
  }
  while (lpw > 16 && (size_t)world->n_states * lpw * 16 + kThrBytes + (size_t)world->n_states * 18 +
                              16 + kMonBytes > (size_t)kLdsLimit)
    lpw >>= 1;   // larger worlds: fewer columns per wave so that the Q tables still fit
  A.lpw = lpw;
  const size_t lds_lpi = (size_t)world->n_states * lpw * 16 + kThrBytes +
                         (size_t)world->n_states * 16 + (((size_t)world->n_states * 2 + 15) & ~(size_t)15);
  // (per-instance parameter sets: the lane-per-instance kernel keeps ONE threshold table per wave)
  if (!draws && !replay && !occ && !r.param_index && (!learn || r.agent == COBEL_AGENT_Q) &&
      lds_lpi + kMonBytes <= (size_t)kLdsLimit &&
      r.n >= 64 && !(r.flags & COBEL_F_FORCE_WAVE)) {
    const bool one = world->n_worlds == 1;
    const bool mon = r.lat_sum || r.lat_cnt || r.reward_sum || r.resp_cnt;
    const size_t bytes = lds_lpi + (mon ? (size_t)kMonBytes : 0);
    const dim3 grid((unsigned)((r.n + lpw - 1) / lpw));
    if (describe) {
      describe[0] = COBEL_TAB_KERNEL_LPI;
      describe[1] = (int32_t)bytes;
      describe[2] = lds_workgroups_per_cu(bytes);
      describe[3] = lpw;
      return COBEL_OK;
    }
#define COBEL_LPI_X(ONE, MON, LOG, EXTRA)                                                      \
  do {                                                                                         \
    if (bytes > 64 * 1024)                                                                     \
      COBEL_HIP_TRY(hipFuncSetAttribute(                                                       \
          reinterpret_cast<const void*>(&k_tab_lpi<ONE, MON, LOG, EXTRA>),                     \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));                            \
    hipLaunchKernelGGL((k_tab_lpi<ONE, MON, LOG, EXTRA>), grid, dim3(64), bytes, st, A);       \
  } while (0)
#define COBEL_LPI(ONE, MON, LOG)                                                               \
  do {                                                                                         \
    if (extra) COBEL_LPI_X(ONE, MON, LOG, true);                                               \
    else COBEL_LPI_X(ONE, MON, LOG, false);                                                    \
  } while (0)
#define COBEL_LPI2(ONE, MON)                                                                   \
  do {                                                                                         \
    if (logs) COBEL_LPI(ONE, MON, true);                                                       \
    else COBEL_LPI(ONE, MON, false);                                                           \
  } while (0)
    // (QAgent with a log but no replay in this call still appends its experiences)
    const bool logs = r.agent == COBEL_AGENT_Q && learn && r.replay_log != nullptr && r.log_cap > 0;
    // (k_tab_lpi's EXTRA: anything but plain training without masks, traces and last experience)
    const bool extra = !learn || ((r.flags & COBEL_F_MASK_ACTIONS) && r.action_mask) || r.last_exp ||
                       r.lat_trace;
    if (one && mon) COBEL_LPI2(true, true);
    else if (one) COBEL_LPI2(true, false);
    else if (mon) COBEL_LPI2(false, true);
    else COBEL_LPI2(false, false);
#undef COBEL_LPI2
#undef COBEL_LPI
#undef COBEL_LPI_X
    COBEL_HIP_TRY(hipGetLastError());
    return COBEL_OK;
  }
  // plain training with the digest in HBM on worlds whose Q table lets LDS hold fewer instances
  // than the register file: one persistent workgroup per CU, part of its waves with Q in L2
  if (midx && !occ && !lds_pad && !(r.flags & (COBEL_F_NO_PWG | COBEL_F_FORCE_WAVE))) {
    int nl = 0, ng = 0;
    size_t wg_lds = 0;
    if (cobel_tab_pwg_plan(world, r, &nl, &ng, &wg_lds)) {
      if (describe) {
        describe[0] = COBEL_TAB_KERNEL_PWG;
        describe[1] = (int32_t)wg_lds;
        describe[2] = 1;
        describe[3] = nl + ng;
        return COBEL_OK;
      }
      return cobel_tab_pwg_launch(world, r, st);
    }
  }
  if (describe) {
    describe[0] = midx ? COBEL_TAB_KERNEL_WPI_INDEX
                       : (fast ? COBEL_TAB_KERNEL_WPI_FAST : COBEL_TAB_KERNEL_WPI);
    describe[1] = (int32_t)lds;
    describe[2] = lds_workgroups_per_cu(lds);
    describe[3] = 1;
    return COBEL_OK;
  }
  if (r.agent == COBEL_AGENT_DYNAQ)
    return dispatch_wpi<COBEL_AGENT_DYNAQ>(A, occ, wlds, fast, midx, lds, st);
  return dispatch_wpi<COBEL_AGENT_Q>(A, occ, wlds, false, false, lds, st);
}

extern "C" int cobel_tab_run(const cobel_world_t* world, const cobel_tab_run_t* run,
                             void* stream) {
  return tab_run_impl(world, run, stream, nullptr);
}

extern "C" int cobel_dynaq_run(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream) {
  COBEL_REQUIRE(run && run->agent == COBEL_AGENT_DYNAQ, COBEL_E_ARG,
                "cobel_dynaq_run: run->agent is not COBEL_AGENT_DYNAQ");
  return tab_run_impl(world, run, stream, nullptr);
}

extern "C" int cobel_q_run(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream) {
  COBEL_REQUIRE(run && run->agent == COBEL_AGENT_Q, COBEL_E_ARG,
                "cobel_q_run: run->agent is not COBEL_AGENT_Q");
  return tab_run_impl(world, run, stream, nullptr);
}

extern "C" int cobel_tab_scratch_check(const void* scratch, int64_t scratch_bytes, void* stream) {
  if (!scratch || scratch_bytes < (int64_t)COBEL_TAB_SCRATCH_BYTES(1)) return COBEL_OK;
  uint32_t word = 0;
  hipStream_t st = (hipStream_t)stream;
  COBEL_HIP_TRY(hipMemcpyAsync(&word, static_cast<const uint32_t*>(scratch) + COBEL_TAB_SCRATCH_ABORT_WORD,
                               sizeof(word), hipMemcpyDeviceToHost, st));
  COBEL_HIP_TRY(hipStreamSynchronize(st));
  if (word != 0u)
    return cobel_fail(COBEL_E_HIP, "cobel_tab_run: a sliced launch gave up waiting for a ring entry "
                                   "(a producer wavefront was lost); the tables are incomplete");
  return COBEL_OK;
}

extern "C" int cobel_tab_describe(const cobel_world_t* world, const cobel_tab_run_t* run,
                                  int32_t* out) {
  COBEL_REQUIRE(out, COBEL_E_ARG, "cobel_tab_describe: NULL out");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (world && run && run->n == 0) return COBEL_OK;
  return tab_run_impl(world, run, nullptr, out);
}
