// Everything of one lockstep DQN training step that is not the network, one lane per instance
// (and one lane per sample for the batch draw at the end):
// epsilon-greedy on the Q-values the replay kernel left behind -> env.step -> append to the replay
// ring -> trial bookkeeping, monitors, auto-reset -> draw the replay batch.  With cobel_dqn_replay
// (mlp.hip) a training step is two launches; through PyTorch the same bookkeeping is ~90 small
// launches (index_put / gather / where / index_add ...), 0.5 ms per step at 8 192 instances.
//
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   agent/dqn.py:170-212       train loop: retrieve_q -> select_action -> step -> store -> replay,
//                              trial ends on `end_trial` or after `steps` steps
//   policy/greedy.py:40-88     epsilon-greedy (cobel_policy.h)
//   interface/topology.py:146-157, :170   step on the neighbour table, reset = draw a start node
//   memory/dqn.py:103-119, :137           FIFO store, uniform sampling with replacement
//   monitor/behavior.py:82               latency = index of the last executed step
// Per instance and step the draws are: one double on the policy stream, one bounded integer on
// the env stream when a trial restarts, `batch` bounded integers (sub = 0 .. batch-1 of ONE
// counter) on the memory stream — the same streams, counters and order as the stand-alone entry
// points (cobel_rng_uniform, cobel_env_reset, cobel_rng_bounded_each) the PyTorch loop uses.
#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct act_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  cobel_dqn_act_t r;
  cobel_eps_consts eps;
  // action counts other than four (GEN instantiations; worlds of cobel_world_create_n)
  int32_t n_actions;
  const uint16_t* next_n;      // [n_worlds][S][n_actions]
  const float* reward_s;       // [n_worlds][S]
  const uint8_t* terminal_s;   // [n_worlds][S]
};

// DYNA: the memory is DynaDQN's tabular world model instead of a replay ring.
// GEN: 1 .. 8 actions but four (interface/topology.py:110-112, six on hexagonal graphs): the world
// as neighbour / reward / terminal tables, the selection over n_actions values.
template <typename T, bool DYNA, bool GEN = false>
__global__ __launch_bounds__(64) void k_dqn_act(const act_args A) {
  const cobel_dqn_act_t& R = A.r;
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= R.n) return;
  if (!R.active[i]) {   // finished all its trials: frozen, consumes nothing
    R.stepped[i] = 0;
    return;
  }
  const uint32_t g = R.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const cobel_wrec* const W = A.rec + (size_t)world * A.S;
  const int D = R.n_obs;

  // ---- select (policy/greedy.py:40-88) -----------------------------------------------------------
  const int NA = GEN ? A.n_actions : 4;
  const T* const q = (const T*)R.q + (size_t)i * NA;
  const uint32_t pc = R.policy_ctr[i];
  const double u = cobel_draw_u01(pc, 0u, g, R.policy_stream, R.seed);
  R.policy_ctr[i] = pc + 1u;
  int a;
  if (GEN) {
    T qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = k < NA ? q[k] : (T)0;
    a = cobel_eps_greedy_select_n<T, 8>(qv, NA, 0xffu, u, R.epsilon, nullptr);
  } else {
    a = cobel_eps_greedy_select<T>(q[0], q[1], q[2], q[3], 15u, u, A.eps);
  }

  // ---- env.step ------------------------------------------------------------------------------------
  const int s = R.state[i];
  int ns;
  float reward;
  bool done;
  if (GEN) {
    const size_t wb = (size_t)world * A.S;
    ns = (int)A.next_n[(wb + s) * NA + a];
    reward = A.reward_s[wb + ns];
    done = A.terminal_s[wb + ns] != 0u;
  } else {
    ns = W[s].next[a & 3];
    const cobel_wrec entered = W[ns];
    reward = entered.reward;
    done = entered.terminal != 0u;
  }

  // ---- store ----------------------------------------------------------------------------------------
  const int slots = R.slots;
  int size = 0;
  long long head = 0;
  if (DYNA) {
    // memory/dyna_q.py:92-96 in float64: rewards[s, a] += lr * (r - rewards[s, a]); states[s, a] =
    // ns; terminals[s, a] = 1 - end_trial
    const size_t e = (size_t)i * R.n_states * 4 + (size_t)s * 4 + (a & 3);
    const double old = R.model_rewards[e];
    R.model_rewards[e] = old + R.model_lr * ((double)reward - old);
    R.model_states[e] = (int64_t)ns;
    R.model_nonterminal[e] = done ? 0.0 : 1.0;
  } else {
    // memory/dqn.py:103-119: FIFO ring, the oldest entry goes once it is full
    size = R.ring_size[i];
    head = R.ring_head[i];
    const bool full = size >= slots;
    const int slot = (int)((head + (long long)size) % slots);
    const size_t row = (size_t)i * slots + slot;
    T* const ds = (T*)R.ring_states + row * D;
    T* const dn = (T*)R.ring_next_states + row * D;
    const double* const os = R.obs_table + (size_t)s * D;
    const double* const on = R.obs_table + (size_t)ns * D;
    for (int d = 0; d < D; ++d) {
      ds[d] = (T)os[d];
      dn[d] = (T)on[d];
    }
    R.ring_actions[row] = (int64_t)a;
    ((T*)R.ring_rewards)[row] = (T)reward;
    ((T*)R.ring_nonterminal)[row] = done ? (T)0 : (T)1;
    if (full) head = (head + 1) % slots;
    else size += 1;
    R.ring_size[i] = size;
    R.ring_head[i] = head;
  }

  // ---- trial bookkeeping (agent/dqn.py:186-212) -------------------------------------------------------
  int trial = R.trial[i];
  int step = R.step[i];
  double trew = R.trial_reward[i] + (double)reward;
  const bool over = done || (step + 1 >= R.steps_per_trial);
  int state = ns;
  bool active = true;
  if (over) {
    if (trial < R.trial_cap) {
      const size_t m = (size_t)(i % (R.mon_stripes > 1 ? R.mon_stripes : 1)) * R.trial_cap + trial;
      if (R.lat_sum) atomicAdd((unsigned long long*)R.lat_sum + m, (unsigned long long)step);
      if (R.lat_cnt) atomicAdd((unsigned long long*)R.lat_cnt + m, 1ull);
      if (R.reward_sum) atomicAdd(R.reward_sum + m, trew);
    }
    trial += 1;
    trew = 0.0;
    step = 0;
    active = trial < R.trials_target;
    if (active) {   // interface/topology.py:170: a new trial starts from a drawn start node
      const int lo = A.start_off[world], cnt = A.start_off[world + 1] - lo;
      const uint32_t ec = R.env_ctr[i];
      state = A.starts[lo + (int)cobel_draw_bounded(ec, 0u, g, COBEL_STREAM_ENV, R.seed,
                                                    (uint32_t)cnt)];
      R.env_ctr[i] = ec + 1u;
    }
  } else {
    step += 1;
  }
  R.state[i] = state;
  R.trial[i] = trial;
  R.step[i] = step;
  R.trial_reward[i] = trew;
  R.active[i] = active ? 1 : 0;

  // ---- replay batch (memory/dqn.py:137): `batch` indices below the number of stored entries --------
  R.stepped[i] = 1;
  if (R.adam_steps) R.adam_steps[i] += 1.0;
  // (the batch itself is drawn by k_dqn_batch, one lane per SAMPLE: 32 Philox blocks and dependent
  //  gathers in a row per lane were two thirds of this kernel's time)
  if (DYNA || R.batch_slots) R.memory_ctr[i] += 1u;
}

// The replay batch of every instance that took part in this step, one lane per (instance, sample):
// sub-stream j of ONE counter of the memory stream (the value k_dqn_act has just counted past).
//   DYNA: memory/dyna_q.py:137-155 — `batch` pairs drawn uniformly from all n_states x 4 pairs, the
//         model's entries for them gathered;  ring: memory/dqn.py:137 — indices below the number of
//         stored entries, as ring slots.
template <typename T, bool DYNA>
__global__ __launch_bounds__(256) void k_dqn_batch(const act_args A) {
  const cobel_dqn_act_t& R = A.r;
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(tid / R.batch), j = (int)(tid - (long long)i * R.batch);
  if (i >= R.n || !R.stepped[i]) return;
  const uint32_t g = R.instance_base + (uint32_t)i;
  const uint32_t mc = R.memory_ctr[i] - 1u;
  if (DYNA) {
    const uint32_t pairs = (uint32_t)R.n_states * 4u;
    const size_t base = (size_t)i * pairs, out = (size_t)i * R.batch + j;
    const uint32_t idx = cobel_draw_bounded(mc, (uint32_t)j, g, COBEL_STREAM_MEMORY, R.seed, pairs);
    R.batch_state_index[out] = (int32_t)(idx >> 2);
    R.batch_next_index[out] = (int32_t)R.model_states[base + idx];
    R.batch_actions[out] = (int64_t)(idx & 3u);
    ((T*)R.batch_rewards)[out] = (T)R.model_rewards[base + idx];
    ((T*)R.batch_nonterminal)[out] = (T)R.model_nonterminal[base + idx];
  } else {
    const int size = R.ring_size[i], slots = R.slots;
    const long long head = R.ring_head[i];
    const uint32_t idx = cobel_draw_bounded(mc, (uint32_t)j, g, COBEL_STREAM_MEMORY, R.seed,
                                            (uint32_t)size);
    R.batch_slots[(size_t)i * R.batch + j] = (int32_t)((head + (long long)idx) % slots);
  }
}

}  // namespace

extern "C" int cobel_dqn_act(const cobel_world_t* world, const cobel_dqn_act_t* run,
                             void* stream) {
  COBEL_REQUIRE(world && run, COBEL_E_ARG, "cobel_dqn_act: NULL world/run");
  const bool gen = world->n_actions != 4;
  if (gen) {
    if (int rc = cobel_world_check(world, "cobel_dqn_act")) return rc;
    COBEL_REQUIRE(world->n_actions >= 1 && world->n_actions <= 8 && world->next_n &&
                      !world->succ_off && !run->model_rewards,
                  COBEL_E_UNSUPPORTED,
                  "cobel_dqn_act: worlds of 1 .. 8 actions with one-hot rows and a replay ring are "
                  "served (this one has %d actions)", world->n_actions);
  } else if (int rc = cobel_world_check4(world, "cobel_dqn_act")) {
    return rc;
  }
  const cobel_dqn_act_t& r = *run;
  COBEL_REQUIRE(r.state && r.env_ctr && r.obs_table && r.q && r.policy_ctr, COBEL_E_ARG,
                "cobel_dqn_act: NULL env / policy argument");
  const bool dyna = r.model_rewards != nullptr;
  if (dyna)
    COBEL_REQUIRE(r.model_states && r.model_nonterminal && r.memory_ctr && r.batch_state_index &&
                      r.batch_next_index && r.batch_actions && r.batch_rewards &&
                      r.batch_nonterminal && r.n_states == world->n_states,
                  COBEL_E_ARG, "cobel_dqn_act: incomplete world-model arguments");
  else
    COBEL_REQUIRE(r.ring_states && r.ring_next_states && r.ring_actions && r.ring_rewards &&
                      r.ring_nonterminal && r.ring_size && r.ring_head,
                  COBEL_E_ARG, "cobel_dqn_act: NULL replay ring argument");
  COBEL_REQUIRE(r.trial && r.step && r.trial_reward && r.active && r.stepped, COBEL_E_ARG,
                "cobel_dqn_act: NULL bookkeeping argument");
  COBEL_REQUIRE(!r.batch_slots || r.memory_ctr, COBEL_E_ARG,
                "cobel_dqn_act: batch_slots given without memory_ctr");
  COBEL_REQUIRE(r.n >= 0 && r.n_obs > 0 && (dyna || r.slots > 0) && r.batch >= 0 &&
                    r.steps_per_trial > 0 && r.trial_cap >= 0,
                COBEL_E_RANGE, "cobel_dqn_act: bad sizes");
  COBEL_REQUIRE(r.epsilon >= 0.0 && r.epsilon <= 1.0, COBEL_E_ARG,
                "cobel_dqn_act: epsilon %g outside [0, 1]", r.epsilon);
  if (r.n == 0) return COBEL_OK;
  act_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = world->n_states;
  A.n_worlds = world->n_worlds;
  A.r = r;
  A.eps = cobel_make_eps_consts(r.epsilon);
  A.n_actions = world->n_actions;
  A.next_n = world->next_n;
  A.reward_s = world->reward_s;
  A.terminal_s = world->terminal_s;
  const dim3 grid((unsigned)((r.n + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  if (gen) {
    if (r.is_float64) hipLaunchKernelGGL((k_dqn_act<double, false, true>), grid, dim3(64), 0, st, A);
    else hipLaunchKernelGGL((k_dqn_act<float, false, true>), grid, dim3(64), 0, st, A);
  } else if (r.is_float64 && dyna) hipLaunchKernelGGL((k_dqn_act<double, true>), grid, dim3(64), 0, st, A);
  else if (r.is_float64) hipLaunchKernelGGL((k_dqn_act<double, false>), grid, dim3(64), 0, st, A);
  else if (dyna) hipLaunchKernelGGL((k_dqn_act<float, true>), grid, dim3(64), 0, st, A);
  else hipLaunchKernelGGL((k_dqn_act<float, false>), grid, dim3(64), 0, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  if (r.batch > 0 && (dyna || r.batch_slots)) {
    const dim3 bgrid((unsigned)(((long long)r.n * r.batch + 255) / 256));
    if (r.is_float64 && dyna) hipLaunchKernelGGL((k_dqn_batch<double, true>), bgrid, dim3(256), 0, st, A);
    else if (dyna) hipLaunchKernelGGL((k_dqn_batch<float, true>), bgrid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_dqn_batch<float, false>), bgrid, dim3(256), 0, st, A);   // (slots: no dtype)
    COBEL_HIP_TRY(hipGetLastError());
  }
  return COBEL_OK;
}
