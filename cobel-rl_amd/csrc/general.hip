// General tabular path: any action count up to 8, any batch size, any state count.
//
// The fast kernels (tabular.hip) are built around four actions — 16-byte Q rows, 48 integer CDF
// thresholds, one wavefront planning one batch of at most 62 updates, tables in LDS.  The
// reference has none of these limits: a Topology's action space is the neighbour count of its
// start node (interface/topology.py:110-112; six on the hexagonal graphs of
// misc/topology_tools.py:175-272), DynaQ.replay / QAgent.replay take any batch_size
// (agent/dyna_q.py:319-330, agent/q.py:344-354).  This file serves those runs: one LANE per
// instance, tables in HBM / L2 (Q f32 [N][S][A]), every update executed in the reference's own
// sequential order — no conflict analysis needed.  Same streams, counters, arithmetic and
// monitors as the fast kernels; for four actions and a batch <= 62 both produce identical
// tables (tested), so a run may mix them freely.
//
// Also here: worlds with an action count other than four (cobel_world_create_n) and the
// epsilon-greedy selection over n values (cobel_eps_greedy_n), policy/greedy.py:40-88.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct gen_args {
  const cobel_wrec* rec;        // four-action worlds
  const uint16_t* next_n;       // other action counts: [W][S][A]
  const float* reward_s;        // [W][S]
  const uint8_t* terminal_s;    // [W][S]
  const uint16_t* starts;
  const int32_t* start_off;
  // transition rows that are distributions (cobel_world_set_transitions), or NULL
  const uint32_t* succ_off;
  const uint16_t* succ_state;
  const double* succ_cdf;
  int32_t S, n_worlds, A;
  cobel_tab_run_t r;
  float alpha_f, gamma_f, model_lr_f;
};

// QAgent replay record: lo = f32 reward, hi = s | ns << 14 | action << 28 | nonterminal << 30 for
// up to four actions (the layout of the fast kernels), nonterminal << 31 up to eight, and beyond
// eight actions (up to 32; worlds of at most 8 192 states) s | ns << 13 | action << 26 |
// nonterminal << 31.
// (worlds whose states or actions do not fit the packed word — cobel_hip.h COBEL_LOG_WORDS — keep
//  TWO words per entry: {f32 reward, action | nonterminal << 8}, {state, next state})
__device__ __forceinline__ uint64_t log_pack_n(float r, uint32_t s, uint32_t a, uint32_t ns,
                                               uint32_t nt, int A) {
  const uint32_t hi = A <= 8 ? (s | (ns << 14) | (a << 28) | (nt << (A <= 4 ? 30 : 31)))
                             : (s | (ns << 13) | (a << 26) | (nt << 31));
  return (uint64_t)__builtin_bit_cast(uint32_t, r) | ((uint64_t)hi << 32);
}

}  // namespace

namespace {

// The action mask of row i (bit a = action a allowed; policy/greedy.py:79-81): one byte per row in
// worlds of up to eight actions, one 32-bit word per row beyond (the AMAX = 16 / 32 instantiations).
template <int AMAX>
__device__ __forceinline__ uint32_t mask_word(const uint8_t* mask, int i) {
  if (!mask) return 0xffffffffu;
  if (AMAX > 8) return reinterpret_cast<const uint32_t*>(mask)[i];
  return (uint32_t)mask[i];
}

template <typename V, int AMAX>
__global__ __launch_bounds__(256) void k_eps_greedy_n(const V* __restrict__ values,
                                                      const uint8_t* __restrict__ mask,
                                                      const double* __restrict__ u, double eps,
                                                      uint8_t* __restrict__ action_out,
                                                      double* __restrict__ probs_out, int n,
                                                      int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V v[AMAX];
#pragma unroll
  for (int a = 0; a < AMAX; ++a) v[a] = a < A ? values[(size_t)i * A + a] : (V)0;
  double p[AMAX];
  const int act = cobel_eps_greedy_select_n<V, AMAX>(v, A, mask_word<AMAX>(mask, i), u[i], eps, p);
  action_out[i] = (uint8_t)act;
  if (probs_out)
    for (int a = 0; a < A; ++a) probs_out[(size_t)i * A + a] = p[a];
}

template <typename V>
int eps_greedy_n(const V* values, const uint8_t* mask, const double* u, double epsilon,
                 uint8_t* action_out, double* probs_out, int32_t n, int32_t n_actions,
                 void* stream, const char* who) {
  COBEL_REQUIRE(values && u && action_out, COBEL_E_ARG, "%s: NULL argument", who);
  COBEL_REQUIRE(epsilon >= 0.0 && epsilon <= 1.0, COBEL_E_ARG, "%s: epsilon %g outside [0, 1]", who,
                epsilon);
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "%s: n = %d", who, n);
  COBEL_REQUIRE(n_actions >= 1 && n_actions <= COBEL_MAX_ACTIONS, COBEL_E_UNSUPPORTED,
                "%s: %d actions (1..%d are served)", who, n_actions, COBEL_MAX_ACTIONS);
  COBEL_REQUIRE(!mask || n_actions <= 8 || ((uintptr_t)mask & 3u) == 0, COBEL_E_ARG,
                "%s: the masks of a %d-action row are 32-bit words, 4-byte aligned", who, n_actions);
  if (n == 0) return COBEL_OK;
  const dim3 grid((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (n_actions <= 8)
    hipLaunchKernelGGL((k_eps_greedy_n<V, 8>), grid, dim3(256), 0, st, values, mask, u, epsilon,
                       action_out, probs_out, n, n_actions);
  else if (n_actions <= 16)
    hipLaunchKernelGGL((k_eps_greedy_n<V, 16>), grid, dim3(256), 0, st, values, mask, u, epsilon,
                       action_out, probs_out, n, n_actions);
  else
    hipLaunchKernelGGL((k_eps_greedy_n<V, 32>), grid, dim3(256), 0, st, values, mask, u, epsilon,
                       action_out, probs_out, n, n_actions);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

}  // namespace

extern "C" int cobel_eps_greedy_n(const float* values, const uint8_t* mask, const double* u,
                                  double epsilon, uint8_t* action_out, double* probs_out,
                                  int32_t n, int32_t n_actions, void* stream) {
  return eps_greedy_n<float>(values, mask, u, epsilon, action_out, probs_out, n, n_actions,
                             stream, "cobel_eps_greedy_n");
}
extern "C" int cobel_eps_greedy_n_f64(const double* values, const uint8_t* mask, const double* u,
                                      double epsilon, uint8_t* action_out, double* probs_out,
                                      int32_t n, int32_t n_actions, void* stream) {
  return eps_greedy_n<double>(values, mask, u, epsilon, action_out, probs_out, n, n_actions,
                              stream, "cobel_eps_greedy_n_f64");
}

// ---------------------------------------------------------------------------------------------
// Worlds with any action count.
extern "C" int cobel_world_create_n(const uint16_t* next, const float* reward,
                                    const uint8_t* terminal, const uint16_t* starts,
                                    const int32_t* start_offsets, int32_t n_states,
                                    int32_t n_worlds, int32_t n_actions, int32_t device,
                                    cobel_world_t** out) {
  if (n_actions == 4)
    return cobel_world_create(next, reward, terminal, starts, start_offsets, n_states, n_worlds,
                              device, out);
  COBEL_REQUIRE(next && reward && terminal && starts && start_offsets && out, COBEL_E_ARG,
                "cobel_world_create_n: NULL argument");
  COBEL_REQUIRE(n_actions >= 1 && n_actions <= COBEL_MAX_ACTIONS, COBEL_E_UNSUPPORTED,
                "cobel_world_create_n: %d actions (1..%d are served)", n_actions,
                COBEL_MAX_ACTIONS);
  COBEL_REQUIRE(n_states > 0 && n_states <= 16384, COBEL_E_RANGE,
                "cobel_world_create_n: n_states %d outside 1..16384", n_states);
  COBEL_REQUIRE(n_worlds > 0, COBEL_E_RANGE, "cobel_world_create_n: n_worlds %d", n_worlds);
  COBEL_REQUIRE(start_offsets[0] == 0, COBEL_E_ARG, "cobel_world_create_n: start_offsets[0] != 0");
  for (int w = 0; w < n_worlds; ++w)
    COBEL_REQUIRE(start_offsets[w + 1] > start_offsets[w], COBEL_E_ARG,
                  "cobel_world_create_n: world %d has no starting state", w);
  const size_t total = (size_t)n_worlds * (size_t)n_states;
  for (size_t k = 0; k < total * (size_t)n_actions; ++k)
    COBEL_REQUIRE(next[k] < n_states, COBEL_E_RANGE, "cobel_world_create_n: next[%zu] = %u >= n_states",
                  k, (unsigned)next[k]);
  const int32_t n_starts = start_offsets[n_worlds];
  for (int32_t k = 0; k < n_starts; ++k)
    COBEL_REQUIRE(starts[k] < n_states, COBEL_E_RANGE, "cobel_world_create_n: start %u >= n_states",
                  (unsigned)starts[k]);
  std::vector<uint8_t> term(total);
  for (size_t k = 0; k < total; ++k) term[k] = terminal[k] ? 1 : 0;

  COBEL_HIP_TRY(hipSetDevice(device));
  cobel_world* w = (cobel_world*)calloc(1, sizeof(cobel_world));
  COBEL_REQUIRE(w, COBEL_E_ARG, "cobel_world_create_n: out of host memory");
  w->n_states = n_states;
  w->n_worlds = n_worlds;
  w->device = device;
  w->n_actions = n_actions;
  for (int k = 0; k < n_worlds; ++k) {
    int32_t rewarded = 0;
    for (int32_t s = 0; s < n_states; ++s) rewarded += reward[(size_t)k * n_states + s] != 0.0f;
    if (rewarded > w->max_rewarded_states) w->max_rewarded_states = rewarded;
  }
  w->h_start_off = (int32_t*)malloc(sizeof(int32_t) * (n_worlds + 1));
  memcpy(w->h_start_off, start_offsets, sizeof(int32_t) * (n_worlds + 1));
  hipError_t e = hipMalloc((void**)&w->next_n, total * n_actions * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&w->reward_s, total * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&w->terminal_s, total);
  if (e == hipSuccess) e = hipMalloc((void**)&w->starts, sizeof(uint16_t) * n_starts);
  if (e == hipSuccess) e = hipMalloc((void**)&w->start_off, sizeof(int32_t) * (n_worlds + 1));
  if (e == hipSuccess)
    e = hipMemcpy(w->next_n, next, total * n_actions * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->reward_s, reward, total * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(w->terminal_s, term.data(), total, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->starts, starts, sizeof(uint16_t) * n_starts, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->start_off, start_offsets, sizeof(int32_t) * (n_worlds + 1),
                  hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    cobel_world_destroy(w);
    return cobel_fail(COBEL_E_HIP, "cobel_world_create_n: %s", hipGetErrorString(e));
  }
  *out = w;
  return COBEL_OK;
}

extern "C" int cobel_world_actions(const cobel_world_t* w, int32_t* n_actions) {
  COBEL_REQUIRE(w && n_actions, COBEL_E_ARG, "cobel_world_actions: NULL argument");
  *n_actions = w->n_actions;
  return COBEL_OK;
}

namespace {

__global__ __launch_bounds__(256) void k_env_step_n(const cobel_wrec* __restrict__ rec,
                                                    const uint16_t* __restrict__ next_n,
                                                    const float* __restrict__ reward_s,
                                                    const uint8_t* __restrict__ terminal_s,
                                                    const uint32_t* __restrict__ succ_off,
                                                    const uint16_t* __restrict__ succ_state,
                                                    const double* __restrict__ succ_cdf,
                                                    uint32_t* __restrict__ env_ctr, uint64_t seed,
                                                    int S, int A, int n_worlds,
                                                    int32_t* __restrict__ state,
                                                    const uint8_t* __restrict__ action,
                                                    float* __restrict__ reward_out,
                                                    uint8_t* __restrict__ done_out, int n,
                                                    uint32_t base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = base + (uint32_t)i;
  const size_t w = (size_t)(g % (uint32_t)n_worlds) * S;
  const int s = min(max(state[i], 0), S - 1);
  const int a = min((int)action[i], A - 1);
  int ns;
  if (succ_off) {   // interface/gridworld.py:119-123: the successor is drawn from the row
    const uint32_t c = env_ctr[i];
    const double u = cobel_draw_u01(c, COBEL_SUB_DOUBLE, g, COBEL_STREAM_ENV, seed);
    env_ctr[i] = c + 1u;
    ns = cobel_draw_successor(succ_off, succ_state, succ_cdf, (w + s) * A + a, u);
  } else {
    ns = next_n[(w + s) * A + a];
  }
  state[i] = ns;
  if (reward_out) reward_out[i] = rec ? rec[w + ns].reward : reward_s[w + ns];
  if (done_out) done_out[i] = rec ? (uint8_t)rec[w + ns].terminal : terminal_s[w + ns];
}

}  // namespace

// cobel_env_step for worlds created with an action count other than four, and cobel_env_step_draw
// for worlds whose transition rows are distributions (env_ctr / seed: their draws).
int cobel_env_step_general(const cobel_world* world, int32_t* state, const uint8_t* action,
                           float* reward_out, uint8_t* done_out, uint32_t* env_ctr, uint64_t seed,
                           int32_t n, uint32_t instance_base, hipStream_t st) {
  hipLaunchKernelGGL(k_env_step_n, dim3((n + 255) / 256), dim3(256), 0, st, world->rec,
                     world->next_n, world->reward_s, world->terminal_s, world->succ_off,
                     world->succ_state, world->succ_cdf, env_ctr, seed, world->n_states,
                     world->n_actions, world->n_worlds, state, action, reward_out, done_out, n,
                     instance_base);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
namespace {

// One lane per instance; nothing is shared between lanes, so lanes return as they finish.
// AMAX: the action counts an instantiation serves (8 / 16 / 32): rows of Q travel through AMAX
// registers per lane.
template <int AGENT, int AMAX>
__global__ __launch_bounds__(64) void k_tab_general(const gen_args G) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (blocks of 8 .. 64 lanes, see the launch)
  if (i >= G.r.n) return;
  const int S = G.S, A = G.A;
  const uint32_t g = G.r.instance_base + (uint32_t)i;
  const size_t wbase = (size_t)(g % (uint32_t)G.n_worlds) * S;
  const int world = (int)(g % (uint32_t)G.n_worlds);
  float* const Q = G.r.q + (size_t)i * S * A;
  uint64_t* const model = AGENT == COBEL_AGENT_DYNAQ ? G.r.model + (size_t)i * S * A : nullptr;
  uint16_t* const mindex =
      (AGENT == COBEL_AGENT_DYNAQ && G.r.model_index) ? G.r.model_index + (size_t)i * S * A : nullptr;
  const bool wide = COBEL_LOG_WORDS(A, S) == 2;
  uint64_t* const rlog = (AGENT == COBEL_AGENT_Q && G.r.replay_log)
                             ? G.r.replay_log + (size_t)i * G.r.log_cap * (wide ? 2 : 1)
                             : nullptr;

  auto next_of = [&](int s, int a) -> int {
    return G.rec ? (int)G.rec[wbase + s].next[a] : (int)G.next_n[(wbase + s) * A + a];
  };
  auto reward_of = [&](int s) -> float { return G.rec ? G.rec[wbase + s].reward : G.reward_s[wbase + s]; };
  auto terminal_of = [&](int s) -> uint32_t {
    return G.rec ? G.rec[wbase + s].terminal : (uint32_t)G.terminal_s[wbase + s];
  };
  auto row_max = [&](int s) -> float {
    float m = Q[(size_t)s * A];
    for (int a = 1; a < A; ++a) m = fmaxf(m, Q[(size_t)s * A + a]);
    return m;
  };

  int32_t* const inst = G.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t cm = (uint32_t)inst[COBEL_I_CTR_MEMORY];
  uint32_t loglen = (uint32_t)inst[COBEL_I_LOG_LEN];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO);
  unsigned long long executed = 0, batches = 0;

  const uint32_t flags = G.r.flags;
  const bool learn = flags & COBEL_F_LEARN;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  const uint8_t* const amask = (flags & COBEL_F_MASK_ACTIONS) ? G.r.action_mask : nullptr;
  const uint64_t seed = G.r.seed;
  const int start_lo = G.start_off[world];
  const uint32_t start_cnt = (uint32_t)(G.start_off[world + 1] - start_lo);
  const int B = G.r.batch;
  const bool wants_replay = learn && !(flags & COBEL_F_NO_REPLAY) && B > 0;
  const bool episodic = AGENT == COBEL_AGENT_DYNAQ && wants_replay && (flags & COBEL_F_EPISODIC);
  const bool replay_each_step =
      wants_replay && (AGENT == COBEL_AGENT_DYNAQ ? !(flags & COBEL_F_EPISODIC) : rlog != nullptr);
  double alpha = G.r.alpha, gamma = G.r.gamma, eps = G.r.epsilon;
  float alpha_f = G.alpha_f, gamma_f = G.gamma_f, mlr_f = G.model_lr_f;
  if (G.r.param_index) {
    const int k = (int)G.r.param_index[i];
    const cobel_param_set_t* const P =
        G.r.param_sets + (k < G.r.n_param_sets ? k : G.r.n_param_sets - 1);
    alpha = P->alpha;
    gamma = P->gamma;
    eps = P->epsilon;
    alpha_f = P->alpha_f;
    gamma_f = P->gamma_f;
    mlr_f = P->model_lr_f;
  }
  const uint32_t SA = (uint32_t)(S * A);

  // A planned / replayed update is in the arithmetic of the reference run with float32 tables:
  // Dyna-Q planning in float64 rounded once on store (the sampled `terminal` is np.int64), QAgent
  // replay in float32 (agent/dyna_q.py:290-299, agent/q.py:305-313).
  // One batch of B sequential updates (j = 0 .. B - 1), software-pipelined: this lane is
  // the only wave of its SIMD (65 536 instances are one wave per SIMD), so nothing else covers the
  // two dependent trips to memory of an update — record, then Q cells.  The records do not depend
  // on Q: record j + 2 and the Q cells of update j + 1 (the row of its successor, its own cell) are
  // requested before update j is finished; what those loads cannot have seen — the cell update j
  // writes (and, for good measure, update j - 1's) — is patched into the prefetched values.  Same
  // values, same order of effects as the plain loop: identical tables (tests, fuzz).
  struct upd_t {
    int s, a, ns;
    float r;
    uint32_t nt;
  };
  struct rec_t {       // a drawn index and the 64-bit record found there (wide logs: two)
    uint32_t idx;
    uint64_t bits, bits2;
  };
  // (slot k of a prefetched row holds action k; beyond A the last action — or, when rows are loaded
  //  two values at a time, the last PAIR — once more)
  auto slot_action = [&](int k) -> int {
    if (A & 1) return k < A ? k : A - 1;
    return k < A ? k : A - 2 + (k & 1);
  };
  auto fetch_cells = [&](const upd_t& u, float (&row)[AMAX], float& q) {
    // (eight unconditional loads — the last valid action again beyond A — so that the count of
    //  loads in flight is a constant: with a load under a condition the compiler waits for ALL
    //  outstanding loads, the prefetch just issued included, before it uses the previous one)
    if (A & 1) {
#pragma unroll
      for (int k = 0; k < AMAX; ++k) row[k] = Q[(size_t)u.ns * A + (k < A ? k : A - 1)];
    } else {   // an even action count: rows are 8-byte aligned, four two-value loads (the lanes of a
               // wave are different instances: every load instruction is 64 separate lines)
      const float2* const r2 = reinterpret_cast<const float2*>(Q + (size_t)u.ns * A);
#pragma unroll
      for (int k = 0; k < AMAX / 2; ++k) {
        const float2 v = r2[2 * k < A ? k : A / 2 - 1];
        row[2 * k] = v.x;
        row[2 * k + 1] = v.y;
      }
    }
    q = Q[(size_t)u.s * A + u.a];
  };
  auto patch_cells = [&](const upd_t& u, float (&row)[AMAX], float& q, int ws, int wa, float wv) {
    if (ws == u.ns) {
#pragma unroll
      for (int k = 0; k < AMAX; ++k) row[k] = slot_action(k) == wa ? wv : row[k];
    }
    if (ws == u.s && wa == u.a) q = wv;
  };
  auto run_batch = [&](auto load_rec, auto decode, bool f64) {
    if (B <= 0) return;
    upd_t cur = decode(load_rec(0));
    float row[AMAX], q;
    fetch_cells(cur, row, q);
    rec_t rec_next = load_rec(B > 1 ? 1 : 0);
    int w1s = -1, w1a = 0, w2s = -1, w2a = 0;   // the cells of the last two updates and their values
    float w1v = 0.0f, w2v = 0.0f;
    for (int j = 0; j < B; ++j) {
      // (unconditional: the last iterations request the last update's record and cells once more)
      const upd_t nxt = decode(rec_next);
      float nrow[AMAX], nq;
      fetch_cells(nxt, nrow, nq);
      rec_next = load_rec(j + 2 < B ? j + 2 : B - 1);
      patch_cells(cur, row, q, w2s, w2a, w2v);
      patch_cells(cur, row, q, w1s, w1a, w1v);
      float m = row[0];
#pragma unroll
      for (int k = 1; k < AMAX; ++k) m = fmaxf(m, row[k]);   // (entries beyond A repeat the last action)
      float qn;
      if (f64) {
        const double gnt = gamma * (double)cur.nt;
        double td = (double)cur.r + gnt * (double)m;
        td = td - (double)q;
        qn = (float)((double)q + alpha * td);
      } else {
        const float gnt = cur.nt ? gamma_f : 0.0f;
        float td = cur.r + gnt * m;
        td = td - q;
        qn = q + alpha_f * td;
      }
      Q[(size_t)cur.s * A + cur.a] = qn;
      w2s = w1s; w2a = w1a; w2v = w1v;
      w1s = cur.s; w1a = cur.a; w1v = qn;
      cur = nxt;
#pragma unroll
      for (int k = 0; k < AMAX; ++k) row[k] = nrow[k];
      q = nq;
    }
  };
  auto plan_dynaq = [&]() {   // memory/dyna_q.py:137-155 + dyna_q.py:329-330
    run_batch(
        [&](int j) -> rec_t {
          const uint32_t idx =
              cobel_draw_bounded(cm, (uint32_t)j, g, COBEL_STREAM_MEMORY, seed, SA);
          return rec_t{idx, model[idx], 0ull};
        },
        [&](const rec_t& rc) -> upd_t {
          const uint32_t hi = (uint32_t)(rc.bits >> 32);
          return upd_t{(int)(rc.idx / (uint32_t)A), (int)(rc.idx % (uint32_t)A), (int)(hi & 0xffffu),
                       __builtin_bit_cast(float, (uint32_t)rc.bits), (hi >> 16) & 1u};
        },
        true);
    cm += 1u;
  };

  int budget = G.r.step_budget > 0 ? G.r.step_budget : 0x7fffffff;
  while (true) {
    if (!(iflags & 1u)) {
      if (trial >= G.r.trials_target) break;
      state = (int)G.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                               start_cnt)];
      ce += 1u;
      step = 0;
      trew = 0.0;
      iflags |= 1u;
    }
    if (budget == 0) break;
    budget -= 1;

    // ---- select + env.step --------------------------------------------------------------------
    float qv[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) qv[a] = a < A ? Q[(size_t)state * A + a] : 0.0f;
    const double u = cobel_draw_u01(cp, 0u, g, pol_stream, seed);
    cp += 1u;
    const int a = cobel_eps_greedy_select_n<float, AMAX>(qv, A, mask_word<AMAX>(amask, state), u,
                                                         eps, nullptr);
    int ns;
    if (G.succ_off) {   // the successor is drawn from the row of sas (gridworld.py:119-123): one
                        // double of the env stream, at the counter the trial starts share
      const double ue = cobel_draw_u01(ce, COBEL_SUB_DOUBLE, g, COBEL_STREAM_ENV, seed);
      ce += 1u;
      ns = cobel_draw_successor(G.succ_off, G.succ_state, G.succ_cdf,
                                (wbase + (size_t)state) * A + a, ue);
    } else {
      ns = next_of(state, a);
    }
    const float r = reward_of(ns);
    const uint32_t end = terminal_of(ns), nt = 1u - end;
    float td_online = 0.0f;
    if (learn) {
      if (AGENT == COBEL_AGENT_DYNAQ) {   // memory/dyna_q.py:92-96, float32
        const size_t sa = (size_t)state * A + a;
        const float R = __builtin_bit_cast(float, (uint32_t)model[sa]);
        const float d = r - R;
        const float Rn = R + mlr_f * d;
        model[sa] = cobel_model_pack(Rn, (uint32_t)ns, nt);
        if (mindex)
          mindex[sa] = (uint16_t)((uint32_t)ns | (nt << 14) |
                                  (__builtin_bit_cast(uint32_t, Rn) ? 0x8000u : 0u));
      }
      {   // online TD, float32
        const float m = row_max(ns);
        const float qsa = Q[(size_t)state * A + a];
        const float gnt = nt ? gamma_f : 0.0f;
        float td = r + gnt * m;
        td = td - qsa;
        Q[(size_t)state * A + a] = qsa + alpha_f * td;
        td_online = td;
      }
      if (rlog && loglen < (uint32_t)G.r.log_cap) {   // q.py:213
        if (wide) {
          rlog[2u * loglen] = (uint64_t)__builtin_bit_cast(uint32_t, r) |
                              ((uint64_t)((uint32_t)a | (nt << 8)) << 32);
          rlog[2u * loglen + 1u] = (uint64_t)(uint32_t)state | ((uint64_t)(uint32_t)ns << 32);
        } else {
          rlog[loglen] = log_pack_n(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt, A);
        }
        loglen += 1u;
      }
    }
    if (G.r.last_exp) {
      int32_t* const e = G.r.last_exp + (size_t)i * 6;
      e[0] = state;
      e[1] = a;
      e[2] = ns;
      e[3] = (int32_t)nt;
      e[4] = __builtin_bit_cast(int32_t, r);
      e[5] = __builtin_bit_cast(int32_t, td_online);
    }
    trew += (double)r;
    executed += 1ull;
    if (G.r.occupancy) atomicAdd(G.r.occupancy + wbase + ns, 1ull);
    const bool trial_over = end || (step + 1 >= G.r.steps_per_trial);
    state = ns;

    // ---- planning / replay, every update in the reference's order -----------------------------
    if (replay_each_step) {
      if (AGENT == COBEL_AGENT_DYNAQ) {
        plan_dynaq();
        batches += 1ull;
      } else {
        if (loglen > 0u) {   // q.py:353-354: idx = rng.choice(len(M), batch_size), one vector draw
          batches += 1ull;
          run_batch(
              [&](int j) -> rec_t {
                const uint32_t idx =
                    cobel_draw_bounded(cm, (uint32_t)j, g, COBEL_STREAM_MEMORY, seed, loglen);
                if (wide) return rec_t{idx, rlog[2u * idx], rlog[2u * idx + 1u]};
                return rec_t{idx, rlog[idx], 0ull};
              },
              [&](const rec_t& rc) -> upd_t {
                const uint32_t hi = (uint32_t)(rc.bits >> 32);
                if (wide)
                  return upd_t{(int)(uint32_t)rc.bits2, (int)(hi & 0xffu), (int)(rc.bits2 >> 32),
                               __builtin_bit_cast(float, (uint32_t)rc.bits), (hi >> 8) & 1u};
                if (AMAX > 8 && A > 8)   // (s | ns << 13 | action << 26 | nonterminal << 31)
                  return upd_t{(int)(hi & 0x1fffu), (int)((hi >> 26) & 31u), (int)((hi >> 13) & 0x1fffu),
                               __builtin_bit_cast(float, (uint32_t)rc.bits), (hi >> 31) & 1u};
                const uint32_t ra = A <= 4 ? (hi >> 28) & 3u : (hi >> 28) & 7u;
                const uint32_t rnt = A <= 4 ? (hi >> 30) & 1u : (hi >> 31) & 1u;
                return upd_t{(int)(hi & 0x3fffu), (int)ra, (int)((hi >> 14) & 0x3fffu),
                             __builtin_bit_cast(float, (uint32_t)rc.bits), rnt};
              },
              false);
        }
        cm += 1u;
      }
    }

    if (trial_over) {
      if (trial >= 0 && trial < G.r.trial_cap) {
        const size_t m = cobel_mon_offset(G.r.mon_stripes, G.r.trial_cap) + (size_t)trial;
        if (G.r.lat_sum) atomicAdd(G.r.lat_sum + m, (unsigned long long)step);
        if (G.r.lat_cnt) atomicAdd(G.r.lat_cnt + m, 1ull);
        if (G.r.reward_sum) atomicAdd(G.r.reward_sum + m, trew);
        if (G.r.resp_cnt && trew > 0.0) atomicAdd(G.r.resp_cnt + m, 1ull);
        if (G.r.lat_trace) G.r.lat_trace[(size_t)i * G.r.trial_cap + trial] = step;
      }
      trial += 1;
      iflags &= ~1u;
      if (episodic) {   // dyna_q.py:210-211
        plan_dynaq();
        batches += 1ull;
      }
    } else {
      step += 1;
    }
  }

  inst[COBEL_I_STATE] = state;
  inst[COBEL_I_STEP] = step;
  inst[COBEL_I_TRIAL] = trial;
  inst[COBEL_I_CTR_ENV] = (int32_t)ce;
  inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
  inst[COBEL_I_CTR_MEMORY] = (int32_t)cm;
  inst[COBEL_I_LOG_LEN] = (int32_t)loglen;
  inst[COBEL_I_FLAGS] = (int32_t)iflags;
  *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
  *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) += executed;
  if (G.r.steps_done && executed) atomicAdd(G.r.steps_done, executed);
  if (G.r.batches_done && batches) atomicAdd(G.r.batches_done, batches);
}

}  // namespace

// Arguments already checked by cobel_tab_run.
int cobel_tab_general_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st) {
  gen_args G;
  G.rec = world->rec;
  G.next_n = world->next_n;
  G.reward_s = world->reward_s;
  G.terminal_s = world->terminal_s;
  G.starts = world->starts;
  G.start_off = world->start_off;
  G.succ_off = world->succ_off;
  G.succ_state = world->succ_state;
  G.succ_cdf = world->succ_cdf;
  G.S = world->n_states;
  G.n_worlds = world->n_worlds;
  G.A = world->n_actions;
  G.r = r;
  G.alpha_f = (float)r.alpha;
  G.gamma_f = (float)r.gamma;
  G.model_lr_f = (float)r.model_lr;
  // Lanes per workgroup.  The lanes share nothing, and a lane spends its time waiting for its own
  // dependent table reads: 65 536 instances as 1 024 full waves are ONE wave per SIMD, with nobody to
  // run while it waits.  Narrower workgroups (a wave each, partly filled) put up to eight waves on
  // a SIMD: six-action hexagonal QAgent, 65 536 instances: 3.73 (64 lanes) / 4.08 (32) / 4.11 (16) /
  // 4.23e8 env-steps/s (8).  Full waves again once there are enough instances to fill the chip.
  int lpb = 64;
  while (lpb > 8 && (long long)r.n < 8192ll * lpb) lpb >>= 1;
  const dim3 grid((unsigned)((r.n + lpb - 1) / lpb));
  COBEL_REQUIRE(G.A <= 8 || !(r.flags & COBEL_F_MASK_ACTIONS) || ((uintptr_t)r.action_mask & 3u) == 0,
                COBEL_E_ARG, "cobel_tab_run: the action masks of a %d-action world are 32-bit words, "
                "4-byte aligned", G.A);
#define COBEL_GENERAL(AGENT)                                                                     \
  do {                                                                                           \
    if (G.A <= 8) hipLaunchKernelGGL((k_tab_general<AGENT, 8>), grid, dim3(lpb), 0, st, G);      \
    else if (G.A <= 16) hipLaunchKernelGGL((k_tab_general<AGENT, 16>), grid, dim3(lpb), 0, st, G); \
    else hipLaunchKernelGGL((k_tab_general<AGENT, 32>), grid, dim3(lpb), 0, st, G);              \
  } while (0)
  if (r.agent == COBEL_AGENT_DYNAQ) COBEL_GENERAL(COBEL_AGENT_DYNAQ);
  else COBEL_GENERAL(COBEL_AGENT_Q);
#undef COBEL_GENERAL
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
