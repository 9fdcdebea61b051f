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
};

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float max4(const float4 v) {
  return fmaxf(fmaxf(fmaxf(v.x, v.y), v.z), v.w);
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}

template <int AGENT, bool OCC>
__global__ __launch_bounds__(64) void k_tab_wpi(const tab_args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float4* const Qs = reinterpret_cast<float4*>(lds_raw);
  float* const Qf = reinterpret_cast<float*>(lds_raw);
  uint32_t* const occ = reinterpret_cast<uint32_t*>(lds_raw + (size_t)A.S * 16);

  const int lane = (int)threadIdx.x;
  const int i = (int)blockIdx.x;
  const int S = A.S;
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const cobel_wrec* const W = A.rec + (size_t)world * S;
  const uint4* const W4 = reinterpret_cast<const uint4*>(W);
  float4* const Qg = reinterpret_cast<float4*>(A.r.q) + (size_t)i * S;
  uint64_t* const model = (AGENT == COBEL_AGENT_DYNAQ) ? A.r.model + (size_t)i * S * 4 : nullptr;
  uint64_t* const rlog =
      (AGENT == COBEL_AGENT_Q && A.r.replay_log) ? A.r.replay_log + (size_t)i * A.r.log_cap
                                                 : nullptr;

  for (int s = lane; s < S; s += 64) {
    Qs[s] = Qg[s];
    if (OCC) occ[s] = 0u;
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
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);
  unsigned long long nsteps = *reinterpret_cast<const unsigned long long*>(inst + COBEL_I_STEPS_LO);

  const uint32_t flags = A.r.flags;
  const bool learn = flags & COBEL_F_LEARN;
  const bool episodic = (AGENT == COBEL_AGENT_DYNAQ) && (flags & COBEL_F_EPISODIC);
  const int B = (learn && !(flags & COBEL_F_NO_REPLAY) &&
                 (AGENT == COBEL_AGENT_DYNAQ || rlog != nullptr))
                    ? A.r.batch
                    : 0;
  const bool replay_each_step = B > 0 && !episodic;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (flags & COBEL_F_MASK_ACTIONS) ? A.r.action_mask : nullptr;
  const uint64_t seed = A.r.seed;
  const uint32_t SA = (uint32_t)S * 4u;
  const int start_lo = A.start_off[world];
  const uint32_t start_cnt = (uint32_t)(A.start_off[world + 1] - start_lo);
  const double alpha = A.r.alpha, gamma = A.r.gamma;
  const float alpha_f = A.alpha_f, gamma_f = A.gamma_f, mlr_f = A.model_lr_f;

  // ---- values carried from one step to the next ------------------------------------------
  uint32_t cw0 = 0, cw1 = 0;   // next[0..3] of the current state (uniform)
  uint4 cand = {0, 0, 0, 0};   // lane k < 4: world record of next[state][k]
  uint64_t mrow = 0;           // lane k < 4: model[state][k]            (Dyna-Q)
  uint32_t mask_cur = 15u;
  double u_cur = 0.0;          // policy draw of the upcoming step
  uint32_t mx = 0;             // lane j < B: memory draw of the upcoming step
  uint64_t gm = 0;             // lane j < B: record that draw selects, gathered ahead

  auto draw_u = [&](uint32_t index) -> double {
    const cobel_u4 x = cobel_philox(index, 0u, g, pol_stream, seed);
    return cobel_u01(x.x, x.y);
  };
  auto draw_m = [&](uint32_t index) -> uint32_t {
    return cobel_philox(index, (uint32_t)lane, g, COBEL_STREAM_MEMORY, seed).x;
  };
  // Start the gather of the records the batch drawn with `x` will use.  `bound` = number of
  // sampleable records at the time the batch runs; records with index >= `have` do not exist
  // yet and are patched in later.
  auto gather = [&](uint32_t x, uint32_t bound, uint32_t have) -> uint64_t {
    uint64_t rec = 0;
    if (lane < B && bound > 0u) {
      const uint32_t idx = cobel_bounded(x, bound);
      if (AGENT == COBEL_AGENT_DYNAQ) rec = model[idx];
      else if (idx < have) rec = rlog[idx];
    }
    return rec;
  };
  auto enter_state = [&](int s) {  // prefetches that depend only on the state being entered
    const uint4 c = W4[s];
    cw0 = rfl(c.x);
    cw1 = rfl(c.y);
    if (lane < 4) {
      cand = W4[next_of(cw0, cw1, lane)];
      if (AGENT == COBEL_AGENT_DYNAQ && learn) mrow = model[(uint32_t)s * 4u + (uint32_t)lane];
    }
    mask_cur = amask ? (uint32_t)amask[s] & 15u : 15u;
  };
  const bool log_room0 = loglen < (uint32_t)A.r.log_cap;

  // ---- the B sequential TD updates of one batch, run as conflict-free prefixes -------------
  auto run_batch = [&](uint32_t x, uint64_t rec, uint32_t bound) {
    if (bound == 0u) return;
    const bool on = lane < B;
    const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
    uint32_t idx, ns, nt;
    const float r = __builtin_bit_cast(float, lo);
    if (AGENT == COBEL_AGENT_DYNAQ) {
      idx = cobel_bounded(x, bound);
      ns = hi & 0xffffu;
      nt = (hi >> 16) & 1u;
    } else {
      idx = (hi & 0x3fffu) * 4u + ((hi >> 28) & 3u);
      ns = (hi >> 14) & 0x3fffu;
      nt = (hi >> 30) & 1u;
    }
    const uint32_t sj = idx >> 2;
    int dep = -1;  // latest earlier lane this lane must wait for
    for (int e = 0; e + 1 < B; ++e) {
      const uint32_t se = rl(sj, e), ie = rl(idx, e);
      const bool hit = (lane > e) && (ns == se || idx == ie);
      dep = hit ? e : dep;
    }
    int first = 0;
    while (first < B) {
      const unsigned long long blocked = __ballot(on && dep >= first);
      const int stop = blocked ? (__ffsll((long long)blocked) - 1) : B;
      if (lane >= first && lane < stop) {
        const float4 row = Qs[ns];
        const float q = Qf[idx];
        const float m = max4(row);
        float qn;
        if (AGENT == COBEL_AGENT_DYNAQ) {
          // planning TD in float64, one rounding on store (NumPy promotion of the reference's
          // expression with a float32 table; see header)
          const double gnt = gamma * (double)nt;
          double td = (double)r + gnt * (double)m;
          td = td - (double)q;
          qn = (float)((double)q + alpha * td);
        } else {
          const float gnt = nt ? gamma_f : 0.0f;
          float td = r + gnt * m;
          td = td - q;
          qn = q + alpha_f * td;
        }
        Qf[idx] = qn;
      }
      __builtin_amdgcn_wave_barrier();
      first = stop;
    }
  };

  // ---- prologue: draws and prefetches for the first step of this call -----------------------
  u_cur = draw_u(cp);
  if (iflags & 1u) enter_state(state);
  if (replay_each_step) {
    mx = draw_m(cm);
    const uint32_t bound = (AGENT == COBEL_AGENT_DYNAQ) ? SA : loglen + (log_room0 ? 1u : 0u);
    gm = gather(mx, bound, loglen);
  }

  int budget = A.r.step_budget > 0 ? A.r.step_budget : 0x7fffffff;
  unsigned long long executed = 0;

  while (true) {
    if (!(iflags & 1u)) {
      if (trial >= A.r.trials_target) break;
      const cobel_u4 x = cobel_philox(ce, 0u, g, COBEL_STREAM_ENV, seed);
      ce += 1u;
      state = (int)A.starts[start_lo + (int)cobel_bounded(x.x, start_cnt)];
      step = 0;
      trew = 0.0;
      iflags |= 1u;
      enter_state(state);
    }
    if (budget == 0) break;
    budget -= 1;

    // ---- select (policy/greedy.py) --------------------------------------------------------
    const float4 qrow = Qs[state];
    const int a = (int)rfl((uint32_t)cobel_eps_greedy_select_wave(qrow.x, qrow.y, qrow.z, qrow.w,
                                                                  mask_cur, u_cur, A.eps, lane));
    // ---- env.step (interface/gridworld.py:115-126) ------------------------------------------
    const int ns = (int)next_of(cw0, cw1, a);
    const uint32_t nw0 = rl(cand.x, a), nw1 = rl(cand.y, a);
    const float r = __builtin_bit_cast(float, rl(cand.z, a));
    const uint32_t end = rl(cand.w, a);
    const uint32_t nt = 1u - end;
    const uint32_t sa = (uint32_t)state * 4u + (uint32_t)a;

    uint64_t written = 0;       // the record this step adds to the model / log
    uint32_t written_at = ~0u;  // its index
    float td_online = 0.0f;
    if (learn) {
      if (AGENT == COBEL_AGENT_DYNAQ) {
        // memory/dyna_q.py:92-96 (float32 arithmetic)
        const float R = __builtin_bit_cast(float, rl((uint32_t)mrow, a));
        const float d = r - R;
        const float Rn = R + mlr_f * d;
        written = cobel_model_pack(Rn, (uint32_t)ns, nt);
        written_at = sa;
        if (lane == 0) model[sa] = written;
      } else if (rlog) {
        if (loglen < (uint32_t)A.r.log_cap) {
          written = cobel_log_pack(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt);
          written_at = loglen;
          if (lane == 0) rlog[loglen] = written;
          loglen += 1u;
        }
      }
      // online TD (agent/dyna_q.py:290-299), float32
      const float4 nrow = Qs[ns];
      const float q = Qf[sa];
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * max4(nrow);
      td = td - q;
      const float qn = q + alpha_f * td;
      if (lane == 0) Qf[sa] = qn;
      td_online = td;
      __builtin_amdgcn_wave_barrier();
    }
    if (A.r.last_exp && lane == 0) {
      int32_t* const e = A.r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = __builtin_bit_cast(int32_t, td_online);
    }

    // ---- bookkeeping ----------------------------------------------------------------------
    trew += (double)r;
    nsteps += 1ull;
    executed += 1ull;
    if (OCC && lane == 0) occ[ns] += 1u;
    const bool trial_over = end || (step + 1 >= A.r.steps_per_trial);
    const int prev_state = state;
    state = ns;
    cw0 = nw0;
    cw1 = nw1;
    cp += 1u;
    u_cur = draw_u(cp);

    // ---- prefetch for the next step, then this step's planning -------------------------------
    uint32_t mx_next = 0;
    uint64_t gm_next = 0;
    uint32_t bound_now = 0, bound_next = 0;
    if (replay_each_step) {
      bound_now = (AGENT == COBEL_AGENT_DYNAQ) ? SA : loglen;
      mx_next = draw_m(cm + 1u);
      const uint32_t room = (loglen < (uint32_t)A.r.log_cap) ? 1u : 0u;
      bound_next = (AGENT == COBEL_AGENT_DYNAQ) ? SA : loglen + room;
      gm_next = gather(mx_next, bound_next, loglen);
    }
    if (!trial_over) {
      if (lane < 4) {
        cand = W4[next_of(cw0, cw1, lane)];
        if (AGENT == COBEL_AGENT_DYNAQ && learn) {
          mrow = model[(uint32_t)state * 4u + (uint32_t)lane];
          // the load may pass the store above: forward the record just written
          if (state == prev_state && lane == a) mrow = written;
        }
      }
      mask_cur = amask ? (uint32_t)amask[state] & 15u : 15u;
    }
    if (replay_each_step) {
      // the record gathered ahead for this batch predates this step's own write
      if (lane < B && bound_now > 0u && cobel_bounded(mx, bound_now) == written_at) gm = written;
      run_batch(mx, gm, bound_now);
      cm += 1u;
      mx = mx_next;
      gm = gm_next;
      // ... and the gather for the next batch may have passed this step's store as well
      if (lane < B && bound_next > 0u && cobel_bounded(mx, bound_next) == written_at)
        gm = written;
    }

    if (trial_over) {
      // agent/dyna_q.py:207-212: current_trial += 1; logs['steps'] = step (0-based)
      if (lane == 0 && trial >= 0 && trial < A.r.trial_cap) {
        if (A.r.lat_sum) atomicAdd(A.r.lat_sum + trial, (unsigned long long)step);
        if (A.r.lat_cnt) atomicAdd(A.r.lat_cnt + trial, 1ull);
        if (A.r.reward_sum) atomicAdd(A.r.reward_sum + trial, trew);
        if (A.r.lat_trace) A.r.lat_trace[(size_t)i * A.r.trial_cap + trial] = step;
      }
      trial += 1;
      iflags &= ~1u;
      if (episodic && B > 0) {
        const uint32_t x = draw_m(cm);
        uint64_t rec = gather(x, SA, 0u);
        if (lane < B && cobel_bounded(x, SA) == written_at) rec = written;
        run_batch(x, rec, SA);
        cm += 1u;
      }
    } else {
      step += 1;
    }
  }

  // ---- write back ---------------------------------------------------------------------------
  __syncthreads();
  for (int s = lane; s < S; s += 64) {
    Qg[s] = Qs[s];
    if (OCC) {
      const uint32_t c = occ[s];
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
    inst[COBEL_I_FLAGS] = (int32_t)iflags;
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) = nsteps;
    if (A.r.steps_done && executed) atomicAdd(A.r.steps_done, executed);
  }
}

__global__ __launch_bounds__(256) void k_model_init(uint64_t* __restrict__ model, size_t total,
                                                    int S4) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t s = (uint32_t)((t % (size_t)S4) >> 2);
  model[t] = cobel_model_pack(0.0f, s, 0u);
}

template <int AGENT, bool OCC>
int launch_wpi(const tab_args& A, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tab_wpi<AGENT, OCC>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_tab_wpi<AGENT, OCC>), dim3(A.r.n), dim3(64), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

}  // namespace

static const int kLdsLimit = 160 * 1024;

extern "C" int cobel_tab_query(int32_t n_states, int32_t agent, int32_t batch,
                               int32_t* lds_bytes, int32_t* instances_per_block) {
  COBEL_REQUIRE(n_states > 0, COBEL_E_RANGE, "cobel_tab_query: n_states = %d", n_states);
  COBEL_REQUIRE(agent == COBEL_AGENT_Q || agent == COBEL_AGENT_DYNAQ, COBEL_E_ARG,
                "cobel_tab_query: unknown agent %d", agent);
  COBEL_REQUIRE(batch >= 0 && batch <= COBEL_MAX_BATCH, COBEL_E_UNSUPPORTED,
                "cobel_tab_query: batch %d outside 0..%d", batch, COBEL_MAX_BATCH);
  const long long lds = (long long)n_states * 20;  // Q row + visit counter per state
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

extern "C" int cobel_tab_run(const cobel_world_t* world, const cobel_tab_run_t* run,
                             void* stream) {
  COBEL_REQUIRE(world && run, COBEL_E_ARG, "cobel_tab_run: NULL world/run");
  const cobel_tab_run_t& r = *run;
  COBEL_REQUIRE(r.q && r.inst, COBEL_E_ARG, "cobel_tab_run: q and inst are required");
  COBEL_REQUIRE(((uintptr_t)r.q & 15u) == 0 && ((uintptr_t)r.inst & 7u) == 0, COBEL_E_ARG,
                "cobel_tab_run: q must be 16-byte and inst 8-byte aligned");
  COBEL_REQUIRE(r.n >= 0, COBEL_E_RANGE, "cobel_tab_run: n = %d", r.n);
  COBEL_REQUIRE(r.agent == COBEL_AGENT_Q || r.agent == COBEL_AGENT_DYNAQ, COBEL_E_ARG,
                "cobel_tab_run: unknown agent %d", r.agent);
  COBEL_REQUIRE(r.agent != COBEL_AGENT_DYNAQ || r.model, COBEL_E_ARG,
                "cobel_tab_run: Dyna-Q needs the model table");
  COBEL_REQUIRE(r.batch >= 0 && r.batch <= COBEL_MAX_BATCH, COBEL_E_UNSUPPORTED,
                "cobel_tab_run: batch %d outside 0..%d", r.batch, COBEL_MAX_BATCH);
  COBEL_REQUIRE(r.steps_per_trial > 0, COBEL_E_RANGE, "cobel_tab_run: steps_per_trial = %d",
                r.steps_per_trial);
  COBEL_REQUIRE(r.epsilon >= 0.0 && r.epsilon <= 1.0, COBEL_E_ARG,
                "cobel_tab_run: epsilon %g outside [0, 1]", r.epsilon);
  COBEL_REQUIRE(!(r.flags & COBEL_F_MASK_ACTIONS) || r.action_mask, COBEL_E_ARG,
                "cobel_tab_run: mask_actions set without an action mask");
  COBEL_REQUIRE(r.trial_cap >= 0 && r.log_cap >= 0, COBEL_E_RANGE, "cobel_tab_run: negative cap");
  COBEL_REQUIRE(r.agent != COBEL_AGENT_Q || world->n_states <= 16384, COBEL_E_UNSUPPORTED,
                "cobel_tab_run: replay records address at most 16384 states");
  int32_t lds = 0;
  if (int rc = cobel_tab_query(world->n_states, r.agent, r.batch, &lds, nullptr)) return rc;
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
  const bool occ = r.occupancy != nullptr;
  const size_t lds_q = (size_t)world->n_states * 16;
  hipStream_t st = (hipStream_t)stream;
  if (r.agent == COBEL_AGENT_DYNAQ)
    return occ ? launch_wpi<COBEL_AGENT_DYNAQ, true>(A, (size_t)lds, st)
               : launch_wpi<COBEL_AGENT_DYNAQ, false>(A, lds_q, st);
  return occ ? launch_wpi<COBEL_AGENT_Q, true>(A, (size_t)lds, st)
             : launch_wpi<COBEL_AGENT_Q, false>(A, lds_q, st);
}
