/*
 * cobel_hip.h — C ABI of libcobel_hip.so, the MI355X (gfx950) hot path of the
 * vectorised navigation-RL loop.
 *
 * The reference (sencheng/CoBeL-RL) is pure Python and has no FFI layer; its
 * boundary for this path is the duck-typed Python API.  Each entry point below
 * names the reference interface it replaces (paths relative to
 * /root/reference/src/cobel).  The binding a maintainer would add is a ctypes
 * stub; see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every pointer marked [dev] is DEVICE memory owned by the
 *     caller (e.g. a torch tensor's data_ptr()), [host] is host memory;
 *   - every call returns 0 on success or a negative COBEL_E_* code and never
 *     throws; cobel_last_error() returns a thread-local message;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and calls return without synchronising;
 *   - the library keeps no global mutable state besides the last-error string; a
 *     world handle is immutable after creation (cobel_world_create*, then at most one
 *     cobel_world_set_transitions before its first use) and may be shared by any number
 *     of runs on its device.
 */
#ifndef COBEL_HIP_H
#define COBEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COBEL_API __attribute__((visibility("default")))

#define COBEL_OK 0
#define COBEL_E_ARG (-1)         /* NULL / malformed argument            (reference: AssertionError) */
#define COBEL_E_RANGE (-2)       /* index or size out of range            (reference: IndexError)     */
#define COBEL_E_HIP (-3)         /* HIP runtime failure (message has hipGetErrorString)               */
#define COBEL_E_UNSUPPORTED (-4) /* valid request this build cannot serve (e.g. table exceeds LDS)   */

#define COBEL_ACTIONS 4 /* gridworld / 4-neighbour topology action count (gridworld.py:86) */
#define COBEL_MAX_ACTIONS 32 /* largest action count of cobel_world_create_n (hexagonal topology: 6).
                               Action masks (bit a = action a allowed, policy/greedy.py:79-81) are
                               one byte per row in worlds of up to eight actions and one 32-bit
                               word per row (4-byte aligned) in worlds of nine to 32 */
/* 64-bit words per entry of a QAgent's replay log (cobel_tab_run_t.replay_log) in a world of A
 * actions and S states: ONE packed word while states and action fit beside the reward, TWO where
 * they do not — the reference's memory (agent/q.py:213) is a list of tuples of any size. */
#define COBEL_LOG_WORDS(A, S) ((((A) > 8 && (S) > 8192) || (S) > 16384) ? 2 : 1)

/* Random streams: Philox-4x32-10, key = (seed lo, seed hi), ctr = (block, sub, instance, stream).
 * Each stream is consumed through a per-instance draw counter c:
 *   bounded integer c = mulhi32(word (c & 3) of block (c >> 2), n)
 *   uniform double  c = 53-bit double from words 2(c & 1), 2(c & 1) + 1 of block (c >> 1)      */
#define COBEL_STREAM_ENV 0u    /* c = resets so far                (gridworld.py:142)     */
#define COBEL_STREAM_POLICY 1u /* c = select_action calls          (greedy.py:58)         */
#define COBEL_STREAM_MEMORY 2u /* c = replay batches, sub = j      (memory/dyna_q.py:137) */
#define COBEL_STREAM_POLICY_TEST 3u
#define COBEL_STREAM_AGENT 4u  /* c = draws of the agent's own generator (agent/sfma.py:312) */
/* A generator that mixes integer and double draws (SFMAMemory.rng) takes its doubles from
 * sub = 1 + j (j = position in a vector draw): integer and double draws never share a block. */
#define COBEL_SUB_DOUBLE 1u

COBEL_API const char* cobel_last_error(void);
/* ABI version of this header: major * 1000 + minor. */
COBEL_API int cobel_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Raw stream access (used by the host facade for single calls of Policy.select_action /
 * Interface.reset, and by the parity tests of the generator itself).
 *   uniform: out[i]    = double draw number index[i] (sub 0) of instance instance_base + i
 *   bounded: out[i][j] = integer draw number index[i], sub j, j < per_instance
 * If advance != 0, index[i] += 1 afterwards.
 * ------------------------------------------------------------------------------------------ */
COBEL_API int cobel_rng_uniform(uint32_t* index /* [dev] [N] */, uint64_t seed, uint32_t stream,
                                uint32_t instance_base, double* out /* [dev] [N] */, int32_t n,
                                int32_t advance, void* stream_handle);
COBEL_API int cobel_rng_bounded(uint32_t* index /* [dev] [N] */, uint64_t seed, uint32_t stream,
                                uint32_t instance_base, uint32_t bound,
                                int32_t* out /* [dev] [N][per_instance] */, int32_t n,
                                int32_t per_instance, int32_t advance, void* stream_handle);
/* Same with one bound per instance (replay memories of different fill); a bound of 0 yields 0
 * and leaves that instance's counter untouched. */
COBEL_API int cobel_rng_bounded_each(uint32_t* index /* [dev] [N] */, uint64_t seed,
                                     uint32_t stream, uint32_t instance_base,
                                     const uint32_t* bounds /* [dev] [N] */,
                                     int32_t* out /* [dev] [N][per_instance] */, int32_t n,
                                     int32_t per_instance, int32_t advance, void* stream_handle);

/* ------------------------------------------------------------------------------------------
 * World tables.  Replaces WorldDict (interface/gridworld.py:17-30) as produced by
 * make_gridworld (misc/gridworld_tools.py:10-136) and Topology's node dict
 * (interface/topology.py:21-26): the dense one-hot sas[S,4,S] becomes next[S,4].
 * n_worlds distinct worlds of the same state count can live in one handle; instance g uses
 * world (g % n_worlds).
 * ------------------------------------------------------------------------------------------ */
typedef struct cobel_world cobel_world_t;

COBEL_API int cobel_world_create(const uint16_t* next /* [host] [n_worlds][S][4] */,
                       const float* reward /* [host] [n_worlds][S] reward on ENTERING the state */,
                       const uint8_t* terminal /* [host] [n_worlds][S] */,
                       const uint16_t* starts /* [host] concatenated start lists */,
                       const int32_t* start_offsets /* [host] [n_worlds + 1] into starts */,
                       int32_t n_states, int32_t n_worlds, int32_t device, cobel_world_t** out);
/* Worlds whose action count is not four: a Topology's action space is the neighbour count of its
 * start node (interface/topology.py:110-112), six on the hexagonal graphs of
 * misc/topology_tools.py:175-272.  next is [n_worlds][S][n_actions]; n_actions = 4 is
 * cobel_world_create.  Such a world is served by cobel_env_step / cobel_env_reset, and by
 * cobel_tab_run (Q rows of n_actions entries: QAgent on the wavefront kernel of
 * csrc/tabular_nact.hip while the tables fit, else the general kernel); the entry points built
 * around four actions (cobel_sr_run, cobel_sfma_run; cobel_dqn_act beyond eight) refuse it with
 * COBEL_E_UNSUPPORTED. */
COBEL_API int cobel_world_create_n(const uint16_t* next /* [host] [n_worlds][S][n_actions] */,
                       const float* reward /* [host] [n_worlds][S] */,
                       const uint8_t* terminal /* [host] [n_worlds][S] */,
                       const uint16_t* starts, const int32_t* start_offsets, int32_t n_states,
                       int32_t n_worlds, int32_t n_actions /* 1..COBEL_MAX_ACTIONS */,
                       int32_t device, cobel_world_t** out);
COBEL_API int cobel_world_actions(const cobel_world_t* world, int32_t* n_actions);
COBEL_API int cobel_world_destroy(cobel_world_t* world);
COBEL_API int cobel_world_info(const cobel_world_t* world, int32_t* n_states, int32_t* n_worlds,
                     int32_t* device);
/* Transition rows that are DISTRIBUTIONS.  The reference's Gridworld.step reads
 * p = world['sas'][s][a] and, when world['deterministic'] is off, draws the successor with
 * rng.choice(arange(S), p=p) (interface/gridworld.py:115-123).  Every builder writes one-hot rows;
 * a world whose dense sas was edited (slippery floors, ...) is handed over here in list form: for
 * pair p = (world * S + s) * n_actions + a the possible successors succ_state[succ_off[p] ..
 * succ_off[p + 1]) in ascending state order and the normalised cumulative sum of their
 * probabilities (cumsum(p) / cumsum(p)[-1], as Generator.choice forms it; the last entry of a row
 * is 1).  Afterwards cobel_env_step_draw steps the world and cobel_tab_run, cobel_sr_run and
 * cobel_sfma_run draw the successor inside their kernels (cobel_tab_run: the generic wavefront
 * kernel for runs with replayed updates, the general kernel otherwise; cobel_sr_run: the wavefront
 * kernel for plain runs, the row-streaming one with occupancy counters; per-instance parameter
 * sets together with drawn successors are refused there): every step draws one double of
 * COBEL_STREAM_ENV (sub-stream 1) at the instance's env counter, which trial starts share (they
 * draw integers, sub-stream 0).  cobel_env_step and cobel_dqn_act refuse such a world
 * (COBEL_E_UNSUPPORTED); the table given to cobel_world_create (most likely successors) stays in
 * place for them to see. */
COBEL_API int cobel_world_set_transitions(cobel_world_t* world,
                       const uint32_t* succ_off /* [host] [n_worlds * S * n_actions + 1] */,
                       const uint16_t* succ_state /* [host] [nnz] */,
                       const double* succ_cdf /* [host] [nnz] */, int64_t nnz);

/* ------------------------------------------------------------------------------------------
 * Stand-alone vectorised environment.  Replaces Gridworld.step / Gridworld.reset
 * (interface/gridworld.py:92-129, :131-145) and Topology.step / reset
 * (interface/topology.py:126-172) for N instances at once.
 *   step : ns = next[s][a]; reward_out = reward[ns]; done_out = terminal[ns]; state <- ns
 *   reset: where reset_mask[i] != 0 (or reset_mask == NULL):
 *          state[i] = starts[bounded draw number env_ctr[i] of COBEL_STREAM_ENV, n = n_starts];
 *          env_ctr[i] += 1
 * ------------------------------------------------------------------------------------------ */
COBEL_API int cobel_env_step(const cobel_world_t* world, int32_t* state /* [dev] [N] in/out */,
                   const uint8_t* action /* [dev] [N] */, float* reward_out /* [dev] [N] */,
                   uint8_t* done_out /* [dev] [N] */, int32_t n, uint32_t instance_base,
                   void* stream);
/* step of a world with distribution rows (cobel_world_set_transitions): u = double draw number
 * env_ctr[i] of COBEL_STREAM_ENV, sub-stream 1; ns = first successor of (s, a) whose cumulative
 * probability exceeds u; env_ctr[i] += 1.  Worlds without such rows step as cobel_env_step does
 * and leave the counters alone. */
COBEL_API int cobel_env_step_draw(const cobel_world_t* world, int32_t* state /* [dev] [N] in/out */,
                   const uint8_t* action /* [dev] [N] */, float* reward_out /* [dev] [N] */,
                   uint8_t* done_out /* [dev] [N] */, uint32_t* env_ctr /* [dev] [N] in/out */,
                   uint64_t seed, int32_t n, uint32_t instance_base, void* stream);
COBEL_API int cobel_env_reset(const cobel_world_t* world, int32_t* state /* [dev] [N] */,
                    const uint8_t* reset_mask /* [dev] [N] or NULL */,
                    uint32_t* env_ctr /* [dev] [N] in/out */, uint64_t seed, int32_t n,
                    uint32_t instance_base, void* stream);

/* Observation lookup: out[i][0..width) = table[index[i]][0..width) (float64 rows).  Replaces
 * Topology.get_observation for pose observations (interface/topology.py:174-193) and the
 * coordinate lookup of Gridworld.get_position (interface/gridworld.py:147-156). */
COBEL_API int cobel_gather_rows(const double* table /* [dev] [rows][width] */,
                                const int32_t* index /* [dev] [N] */,
                                double* out /* [dev] [N][width] */, int32_t n, int32_t width,
                                int32_t rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Epsilon-greedy.  Replaces EpsilonGreedy.get_action_probs / select_action
 * (policy/greedy.py:40-88) incl. the Generator.choice draw
 * searchsorted(cumsum(p) / cumsum(p)[-1], u, 'right').
 *
 * Probabilities are float64 as in the reference (probs = np.zeros(v.shape)); ties are
 * detected with exact equality on the float32 values.
 * cobel_eps_greedy selects actions for N rows of 4 values with injected uniforms (KAT entry
 * point; the fused run kernels draw u from COBEL_STREAM_POLICY instead) and optionally
 * returns get_action_probs().
 * ------------------------------------------------------------------------------------------ */
COBEL_API int cobel_eps_greedy(const float* values /* [dev] [N][4], 16-byte aligned */,
                               const uint8_t* mask /* [dev] [N] 4-bit masks, or NULL = all */,
                               const double* u /* [dev] [N] in [0,1) */, double epsilon,
                               uint8_t* action_out /* [dev] [N] */,
                               double* probs_out /* [dev] [N][4] or NULL */, int32_t n,
                               void* stream);

/* Same for float64 value rows (network outputs of the DQN path, which the reference keeps in
 * float64): ties are exact equality on the float64 values. */
COBEL_API int cobel_eps_greedy_f64(const double* values /* [dev] [N][4] */,
                                   const uint8_t* mask /* [dev] [N] or NULL */,
                                   const double* u /* [dev] [N] */, double epsilon,
                                   uint8_t* action_out /* [dev] [N] */,
                                   double* probs_out /* [dev] [N][4] or NULL */, int32_t n,
                                   void* stream);

/* The same selection over rows of n_actions values (1..COBEL_MAX_ACTIONS; mask bit a = action a
 * allowed: bytes up to eight actions, 32-bit words beyond): action spaces other than four, e.g. the
 * six neighbours of a hexagonal Topology. */
COBEL_API int cobel_eps_greedy_n(const float* values /* [dev] [N][n_actions] */,
                                 const uint8_t* mask /* [dev] [N] bytes (n_actions <= 8) or
                                                        32-bit words, or NULL */,
                                 const double* u /* [dev] [N] */, double epsilon,
                                 uint8_t* action_out /* [dev] [N] */,
                                 double* probs_out /* [dev] [N][n_actions] or NULL */, int32_t n,
                                 int32_t n_actions, void* stream);
COBEL_API int cobel_eps_greedy_n_f64(const double* values /* [dev] [N][n_actions] */,
                                     const uint8_t* mask, const double* u, double epsilon,
                                     uint8_t* action_out, double* probs_out, int32_t n,
                                     int32_t n_actions, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused tabular agents.  One call advances every instance through
 *   select -> env.step -> [model store] -> TD update -> [B planning / replay TD updates]
 * with per-instance auto-reset at trial ends, entirely on device.
 *
 * Replaces the loops DynaQ.train / DynaQ.test (agent/dyna_q.py:140-273) with update_q
 * (:275-301), replay (:319-330), DynaQMemory.store / retrieve_batch (memory/dyna_q.py:77-96,
 * :122-157); QAgent.train / update_q / replay (agent/q.py:160-228, :289-315, :344-354); and
 * the on_trial_end monitors EscapeLatencyMonitor / RewardMonitor (monitor/behavior.py:73-97,
 * :111-) plus the visit counts behind get_occupancy_map (analysis/behavior_spatial.py:9-73).
 *
 * Numerics ("f32 tables"): tables are float32; the arithmetic follows what the reference
 * itself computes when its tables are cast to float32 under NumPy >= 2 promotion — online TD
 * and QAgent replay in float32, Dyna-Q planning TD in float64 rounded once on store, model
 * reward update in float32 — so results are bit-exact against that run of the reference.
 * ------------------------------------------------------------------------------------------ */

/* Per-instance scalar state: 12 x 32-bit words, caller-owned [dev] [N][12]. */
enum {
  COBEL_I_STATE = 0,      /* int32  current env state                                        */
  COBEL_I_STEP = 1,       /* int32  steps taken in the running trial                         */
  COBEL_I_TRIAL = 2,      /* int32  trials finished (agent.current_trial, agent.py:58)       */
  COBEL_I_CTR_ENV = 3,    /* uint32 next index on COBEL_STREAM_ENV                           */
  COBEL_I_CTR_POLICY = 4, /* uint32 next index on the policy stream                          */
  COBEL_I_CTR_MEMORY = 5, /* uint32 next index on COBEL_STREAM_MEMORY                        */
  COBEL_I_LOG_LEN = 6,    /* uint32 QAgent: experiences logged so far                        */
  COBEL_I_FLAGS = 7,      /* bit 0: a trial is in progress                                   */
  COBEL_I_REWARD_LO = 8,  /* float64 reward collected in the running trial (2 words)         */
  COBEL_I_REWARD_HI = 9,
  COBEL_I_STEPS_LO = 10,  /* uint64 env steps executed over the instance's lifetime (2 words) */
  COBEL_I_STEPS_HI = 11,
  COBEL_I_WORDS = 12
};

enum { COBEL_AGENT_Q = 0, COBEL_AGENT_DYNAQ = 1 };

#define COBEL_F_LEARN 1u          /* 0 = Agent.test(): act only                              */
#define COBEL_F_NO_REPLAY 2u      /* DynaQ.train(no_replay=True)                             */
#define COBEL_F_EPISODIC 4u       /* DynaQ.episodic_replay: one batch per trial              */
#define COBEL_F_MASK_ACTIONS 8u   /* agent.mask_actions                                      */
#define COBEL_F_TEST_STREAM 16u   /* draw u from COBEL_STREAM_POLICY_TEST (separate policy)  */
#define COBEL_F_FORCE_WAVE 32u    /* always use the wave-per-instance kernel (testing); SFMA: always
                                     use the general kernel instead of a specialised one      */
#define COBEL_F_FORCE_LDS_MODEL 64u /* ignore model_index, keep the model digest in LDS (testing) */
#define COBEL_F_NO_PREFETCH 128u   /* SR: load value rows at the top of each step (testing)        */
#define COBEL_F_TAB_GENERAL 512u   /* cobel_tab_run: always take the general kernel (testing)          */
#define COBEL_F_NO_PWG 1024u       /* Dyna-Q: never take the persistent-workgroup kernel (testing,
                                      A/B measurements): k_tab_wpi, one workgroup per instance      */
#define COBEL_F_PWG_GLOBAL 2048u   /* ... its wavefronts ALL work on Q in global memory (testing)          */
#define COBEL_F_SR_STREAM_ROWS 256u /* SR: always take the row-streaming kernel, also where the
                                      sparse-reward kernel applies (testing, A/B measurements)    */

/* Per-instance hyper-parameters.  The reference explores hyper-parameters by running one
 * simulation per combination (optimizer/grid_search.py:173-262: `simulation(task, parameters)`
 * for every entry of `parameter_combinations`, nb_runs times); here the combinations ride on the
 * instance axis of ONE launch: instance i uses param_sets[param_index[i]].  A set carries the
 * agent's learning rate and discount (agent/dyna_q.py:112-113, q.py:121-122, sr.py:115-116), the
 * policy's epsilon (policy/greedy.py:35) and the world model's learning rate
 * (memory/dyna_q.py:66), plus the constants derived from them on the host in float64 — fill it
 * with cobel_param_set_fill, never by hand.  512 bytes. */
typedef struct {
  double alpha, gamma, epsilon, model_lr;
  float alpha_f, gamma_f, model_lr_f, reserved_;
  double eps_base[5];      /* eps / n,        n = 1..4 (index 0 unused)                       */
  double eps_bonus[5];     /* (1 - eps) / t,  t = 1..4                                        */
  uint64_t eps_thr[16][3]; /* per tie pattern: ceil(cdf_k * 2^53) of the unmasked CDF         */
} cobel_param_set_t;
COBEL_API int cobel_param_set_fill(double alpha, double gamma, double epsilon, double model_lr,
                                   cobel_param_set_t* out /* [host] */);

typedef struct {
  /* tables, all caller-owned device memory */
  float* q;              /* [N][S][4] float32 Q.  The replayed updates of a step use the bit
                            patterns 0xffffffc0 .. 0xffffffff inside the table as lane tags while
                            a batch runs (NaNs of a payload no arithmetic produces: the reference's
                            tables never hold them).  A table that holds such a pattern — e.g.
                            uninitialised memory — is garbage in, garbage out (the cell may be
                            taken for a tag); every batch still ends.                           */
  uint64_t* model;       /* DYNAQ: [N][S][4] packed {f32 R; u16 NS; u8 nonterminal; u8 0}    */
  uint16_t* model_index; /* DYNAQ, optional: [N][S][4] 16-bit digest of `model`
                            (next | nonterminal << 14 | (R != +0) << 15, cobel_model_index_build),
                            kept in sync by the kernel.  When present the planning kernel reads
                            it from HBM/L2 instead of holding a copy in LDS: nine instances per
                            CU instead of six.                                                 */
  uint64_t* replay_log;  /* Q: [N][log_cap][COBEL_LOG_WORDS(actions, states)] experiences
                            (q.py:213), or NULL (needed for batch > 0).  Every learning step
                            appends one while there is room, also at batch 0 (the reference's
                            memory grows whether it replays or not).  One word per entry:
                            lo = f32 reward, hi = s | ns << 14 | action << 28 | nonterminal << 30
                            (worlds of five to eight actions: nonterminal << 31; of nine to 32
                            actions, at most 8 192 states: s | ns << 13 | action << 26 |
                            nonterminal << 31).  Two words per entry (more than 16 384 states, or
                            more than eight actions and more than 8 192 states): word 0 lo = f32
                            reward, hi = action | nonterminal << 8; word 1 lo = s, hi = ns      */
  int32_t* inst;         /* [N][COBEL_I_WORDS]                                               */
  const uint8_t* action_mask; /* [S] masks (bit a = action a) shared by all instances, or NULL:
                                 bytes in worlds of up to eight actions, 32-bit words beyond */
  /* monitors (any may be NULL) */
  unsigned long long* lat_sum;  /* [trial_cap] sum over instances of logs['steps']           */
  unsigned long long* lat_cnt;  /* [trial_cap] instances that finished that trial            */
  double* reward_sum;           /* [trial_cap] sum of trial rewards                          */
  unsigned long long* resp_cnt; /* [trial_cap] instances whose trial reward was > 0: the default
                                   response of ResponseMonitor (monitor/behavior.py:286-289)    */
  int32_t* lat_trace;           /* [N][trial_cap] per-instance logs['steps'], or NULL        */
  unsigned long long* occupancy;/* [n_worlds][S] visits of next_state                        */
  unsigned long long* steps_done; /* [1] env steps executed by this call (added)             */
  int32_t* last_exp;     /* [N][6] last experience {s, a, ns, nonterminal, f32 r, f32 td} for
                            per-step host callbacks (use with step_budget = 1), or NULL       */
  /* sizes */
  int32_t n;             /* instances on this device                                         */
  int32_t log_cap;       /* entries per instance in replay_log                               */
  int32_t trial_cap;     /* length of the monitor arrays                                     */
  uint32_t instance_base;/* global id of local instance 0 (sharding; keeps draws G-independent) */
  /* run parameters */
  int32_t agent;         /* COBEL_AGENT_*                                                    */
  uint32_t flags;        /* COBEL_F_*                                                        */
  int32_t trials_target; /* stop an instance when inst[TRIAL] reaches this                   */
  int32_t steps_per_trial;
  int32_t step_budget;   /* max env steps per instance in this call; <= 0 = unlimited        */
  int32_t batch;         /* B: planning / replay updates per step (0..COBEL_MAX_BATCH)       */
  double alpha, gamma, epsilon, model_lr;
  uint64_t seed;
  /* optional per-instance hyper-parameters: with param_index != NULL instance i uses
     param_sets[param_index[i]] (index < n_param_sets) and the four scalars above are ignored */
  const cobel_param_set_t* param_sets; /* [dev] [n_param_sets]                               */
  const uint16_t* param_index;         /* [dev] [N]                                          */
  int32_t n_param_sets;
  int32_t mon_stripes;   /* > 1: lat_sum / lat_cnt / reward_sum / resp_cnt are [mon_stripes][trial_cap]
                            and workgroup b adds into copy b % mon_stripes (spreads the atomics of
                            many instances finishing the same trials over L2 channels); the caller
                            sums the copies.  0 or 1: one copy.                                 */
  unsigned long long* batches_done; /* [1] or NULL: planning / replay batches EVALUATED by this call
                            (added).  A Dyna-Q batch is drawn in every learning step but evaluated
                            only if it can change a table: not while the instance's Q and the
                            reward estimates of its model are all +0.0f (every update of such a
                            batch is 0 + alpha (0 + gamma nt 0 - 0) = 0).  steps_done minus this is
                            the number of batches skipped that way (episodic replay: per trial).  */
  void* scratch;         /* optional [dev] work area of the call, caller-owned (the library allocates
                            nothing in a run): COBEL_TAB_SCRATCH_BYTES(n) bytes, zeroed once by the
                            caller after allocating it, contents otherwise undefined between
                            calls, never shared by two calls in flight.  The
                            persistent-workgroup Dyna-Q kernel keeps its ticket counters and the
                            rings of its ready slices there; with it a launch of few instances
                            per GPU (a shard of a split batch) is handed out in slices of an
                            instance's steps instead of whole instances, so that the last round of
                            work fills the chip.  NULL: whole instances,
                            counters in a buffer of the world handle (one call in flight per
                            handle).                                                           */
  int64_t scratch_bytes;
} cobel_tab_run_t;
#define COBEL_TAB_SCRATCH_BYTES(n) ((256 + 7 * ((int64_t)(n) + 8)) * 4)
/* Word (uint32) of the scratch area a sliced launch raises when one of its wavefronts gave up
 * waiting for a ring entry that never came (a producer wavefront that faulted or was killed: the
 * wait is bounded — seconds — so that the grid drains instead of hanging the GPU).  Valid from the
 * end of a call to the start of the next one on the same area; read by cobel_tab_scratch_check.
 * What slicing relies on, for a maintainer: (1) the hand-off of an instance from the wavefront that
 * ran slice k to the one that runs slice k + 1 goes through ONE XCD's L2 — a queue of tickets is
 * served by the workgroups of a single XCD (HW_REG_XCC_ID, claimed through the `owner` words), the
 * producer waits for its stores (vmcnt(0), a workgroup-scope release) before it publishes the
 * entry, the consumer drops its CU's L1 (agent-scope acquire) before it reads; the per-XCD L2s are
 * not written back per hand-off; (2) every ticket that exists is held by a resident wavefront that
 * finishes (one persistent workgroup per CU, grid <= CU count), so a waiting wavefront waits for
 * work in progress, never for work not yet scheduled. */
#define COBEL_TAB_SCRATCH_ABORT_WORD 255

/* Largest batch the wavefront kernels plan in one pass (one lane per update).  Larger batches —
 * the reference has no limit (agent/dyna_q.py:319-330, memory/dyna_q.py:137) — are planned by
 * Dyna-Q and QAgent in ceil(batch / 62) passes per step (the passes run one after the other, so
 * the sequential order of the reference's loop is kept).  The general kernel — one lane per
 * instance, every update in the reference's sequential order, tables in HBM / L2 — is taken where
 * no wavefront kernel covers a world with an action
 * count other than four (Q is then [N][S][n_actions], replay records carry the action in bits 28-30
 * and the nonterminal flag in bit 31 of the high word; QAgent runs one wavefront per instance on
 * 1 .. 32 actions) and for state counts whose tables
 * exceed the LDS. */
#define COBEL_MAX_BATCH 62

/* 0 = this (S, agent, batch) combination is supported; fills *lds_bytes with the LDS per instance. */
COBEL_API int cobel_tab_query(int32_t n_states, int32_t agent, int32_t batch, int32_t* lds_bytes,
                    int32_t* instances_per_block);
COBEL_API int cobel_tab_run(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream);
/* SURVEY.md section 8b names the two agents' entry points separately; they are cobel_tab_run with
 * run->agent checked: COBEL_E_ARG unless it is COBEL_AGENT_DYNAQ (agent/dyna_q.py:140-330) /
 * COBEL_AGENT_Q (agent/q.py:115-354).  Same arguments, same kernels, same results. */
COBEL_API int cobel_dynaq_run(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream);
COBEL_API int cobel_q_run(const cobel_world_t* world, const cobel_tab_run_t* run, void* stream);
/* Which kernel cobel_tab_run would take for this run, without launching anything (same argument
 * checks): out[0] = COBEL_TAB_KERNEL_*, out[1] = LDS bytes per workgroup, out[2] = workgroups a CU
 * can hold by LDS (1 280-byte blocks, 128 per CU), out[3] = instances per workgroup.  For tests
 * and benchmark reports; all zero for n = 0. */
enum {
  COBEL_TAB_KERNEL_LPI = 0,       /* one lane per instance (runs without planning)              */
  COBEL_TAB_KERNEL_WPI = 1,       /* one wavefront per instance, every run-time switch          */
  COBEL_TAB_KERNEL_WPI_FAST = 2,  /* ... plain Dyna-Q training, model digest in LDS             */
  COBEL_TAB_KERNEL_WPI_INDEX = 3, /* ... plain Dyna-Q training, model digest in HBM (model_index) */
  COBEL_TAB_KERNEL_PWG = 5,       /* plain Dyna-Q training, one persistent workgroup of 16 wavefronts
                                     per CU: some keep Q in LDS, the others work on it in L2; out[1]
                                     = LDS of the workgroup, out[2] = 1, out[3] = its wavefronts    */
  COBEL_TAB_KERNEL_WQN = 6,       /* Q-learning on worlds of 1..8 (not four) actions whose tables fit
                                     the LDS: one wavefront per instance, Q rows padded to eight
                                     values; out[3] = instances per workgroup                      */
  COBEL_TAB_KERNEL_GENERAL = 4    /* one lane per instance, tables in HBM: any action count, batch
                                     size and state count                                        */
};
COBEL_API int cobel_tab_describe(const cobel_world_t* world, const cobel_tab_run_t* run,
                                 int32_t* out /* [host] [4] */);

/* After a cobel_tab_run call that was given a scratch area: waits for `stream`, reads the area's
 * abort word and returns COBEL_OK, or COBEL_E_HIP when a sliced launch gave up (the tables of the
 * call are then incomplete and must be discarded).  The word is STICKY: a launch zeroes its
 * counters and rings but never this word (only the caller does, by zeroing the area it owns), so
 * the check may be made once after any number of launches and reports an abort in ANY of them.  scratch NULL or smaller than
 * COBEL_TAB_SCRATCH_BYTES(1): COBEL_OK (nothing was sliced).  The reference has no counterpart:
 * its loop cannot lose a producer (agent/dyna_q.py:140-215 is one thread). */
COBEL_API int cobel_tab_scratch_check(const void* scratch /* [dev] */, int64_t scratch_bytes,
                                      void* stream);

/* The additions NumPy's pairwise summation (np.sum over a contiguous float32 vector, agent/sr.py:
 * 302-306 `np.sum(SR[j] * rewards)`) performs on a vector of n elements of which only the k <= 32 at
 * `pos` (ascending) are not zero: step t is value[dst[t]] += value[src[t]] on slots 0..k-1, k - 1
 * steps (fewer never: every non-zero element is added once), *root = the slot holding the sum (-1
 * for k = 0).  Pure host function: the SR kernel's sparse-reward form takes its order from it. */
COBEL_API int cobel_pairwise_order(int32_t n, const int32_t* pos, int32_t k, uint8_t* dst /* [31] */,
                                   uint8_t* src /* [31] */, int32_t* root);

/* Host helpers for the packed 8-byte records (so bindings never re-derive the layout). */
COBEL_API uint64_t cobel_pack_model(float reward, uint16_t next_state, uint8_t nonterminal);
COBEL_API void cobel_unpack_model(uint64_t rec, float* reward, uint16_t* next_state, uint8_t* nonterminal);
/* Initialise a model table the way DynaQMemory.__init__ does (memory/dyna_q.py:72-75):
 * R = 0, NS[s][a] = s, nonterminal = 0. */
COBEL_API int cobel_model_init(uint64_t* model /* [dev] [N][S][4] */, int32_t n, int32_t n_states,
                     void* stream);
/* (Re)build the 16-bit digest of a model table — after cobel_model_init or whenever the caller
 * has edited `model` by hand. */
COBEL_API int cobel_model_index_build(const uint64_t* model /* [dev] [N][S][4] */,
                                      uint16_t* index /* [dev] [N][S][4] */, int32_t n,
                                      int32_t n_states, void* stream);

/* ------------------------------------------------------------------------------------------
 * Successor-representation agent.  Replaces SR.train / SR.update / SR.retrieve_q
 * (agent/sr.py:142-197, :255-286, :288-308).  Tables: SR f32 [N][S][S] (= eye at init),
 * agent transition table T u16 [N][S][4] (= s at init, sr.py:131-135), reward estimate
 * f32 [N][S].  The row TD error is evaluated in float64 and rounded once on store, as the
 * reference does with a float32 SR (np.eye is float64, sr.py:276-284).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  float* sr;          /* [N][S][S] */
  uint16_t* trans;    /* [N][S][4] */
  float* rewards;     /* [N][S]    */
  int32_t* inst;      /* [N][COBEL_I_WORDS] */
  const uint8_t* action_mask;
  unsigned long long* lat_sum;
  unsigned long long* lat_cnt;
  double* reward_sum;
  unsigned long long* resp_cnt;
  int32_t* lat_trace;
  unsigned long long* occupancy;
  unsigned long long* steps_done;
  int32_t* last_exp;  /* [N][6] {s, a, ns, nonterminal, f32 r, 0} or NULL */
  int32_t n, trial_cap;
  uint32_t instance_base;
  uint32_t flags;
  int32_t trials_target, steps_per_trial, step_budget;
  double alpha, gamma, epsilon;
  uint64_t seed;
  const cobel_param_set_t* param_sets; /* as in cobel_tab_run_t (model_lr unused)            */
  const uint16_t* param_index;
  int32_t n_param_sets;
  int32_t mon_stripes;   /* as in cobel_tab_run_t                                             */
  /* optional [dev] [4] counters the sparse-reward kernel adds to: SR rows read, SR rows
     written, 4-byte value gathers, instances that fell back to full row sums (NULL: not counted;
     untouched by the row-streaming kernel) */
  unsigned long long* traffic;
} cobel_sr_run_t;

COBEL_API int cobel_sr_init(float* sr, uint16_t* trans, float* rewards, int32_t n, int32_t n_states,
                  void* stream);
COBEL_API int cobel_sr_run(const cobel_world_t* world, const cobel_sr_run_t* run, void* stream);
/* q[i][a] = V[T[s_i][a]], V[j] = sum_k SR[j][k] * rewards[k] (sr.py:302-306) for given states. */
COBEL_API int cobel_sr_retrieve_q(const float* sr, const uint16_t* trans, const float* rewards,
                        const int32_t* states /* [dev] [N] */, float* q_out /* [dev] [N][4] */,
                        int32_t n, int32_t n_states, void* stream);

/* ------------------------------------------------------------------------------------------
 * SFMA agent: Dyna-Q whose replay is driven by Spatial structure and Frequency-weighted Memory
 * Access.  Replaces SFMA.train / test / replay / update_q (agent/sfma.py:233-458) and
 * SFMAMemory.store / replay / softmax / retrieve_random_batch (memory/sfma.py:195-416) for N
 * instances; the similarity metric (memory/utils/metrics.py) comes in as a matrix.
 *
 * Per trial: online steps (select -> env.step -> store -> TD), then — unless NO_REPLAY —
 * nb_replays replays of `batch` reactivations.  One reactivation ranks all 4S experiences
 * j = a * S + s by R = C * D * (1 - I) [* T] (strength x similarity x (1 - inhibition) [x
 * recency]), thresholds, normalises, and draws one from softmax(R) = exp(beta R) - 1
 * (memory/sfma.py:280-333); the drawn experience is applied with a TD update at once (the batch
 * the reference builds first does not depend on Q, so the order of effects is the same).
 *
 * Numerics: Q and the model's reward estimates are float32 and follow the reference run with
 * float32 tables (online TD float32, replayed TD float64 rounded once on store, |TD| sum float32
 * until the first replayed TD after a reset); strengths, inhibition, recency, similarities and
 * priorities are float64 as in the reference.  The drawn index is
 * #{k : cumsum(w)_k / cumsum(w)_last <= u}: the reference normalises w three times before the same
 * count and sums sequentially, here the sum is a wave scan and exp() is the device's — equal up
 * to a few ulp of the CDF, so a draw differs from the reference's only if u falls within ~1e-14
 * of a CDF edge.
 * ------------------------------------------------------------------------------------------ */
enum {
  COBEL_SFMA_DEFAULT = 0, COBEL_SFMA_REVERSE = 1, COBEL_SFMA_FORWARD = 2,
  COBEL_SFMA_BLEND_FORWARD = 3, COBEL_SFMA_BLEND_REVERSE = 4, COBEL_SFMA_INTERPOLATE = 5,
  COBEL_SFMA_SWEEPING = 6, COBEL_SFMA_MODES = 7
};

#define COBEL_SF_RANDOM 1u            /* agent.random: uniform batches (memory/sfma.py:374-416) */
#define COBEL_SF_DYNAMIC 2u           /* agent.dynamic: reverse/default drawn from the |TD| sum  */
#define COBEL_SF_START_REPLAY 4u      /* agent.start_replay: a replay (without TD) at trial start */
#define COBEL_SF_DETERMINISTIC 8u     /* M.deterministic: argmax instead of the softmax draw      */
#define COBEL_SF_RECENCY 16u          /* M.recency                                                */
#define COBEL_SF_C_NORMALIZE 32u      /* M.C_normalize                                            */
#define COBEL_SF_D_NORMALIZE 64u      /* M.D_normalize                                            */
#define COBEL_SF_R_NORMALIZE 128u     /* M.R_normalize (default on)                               */
#define COBEL_SF_REWARD_MOD_LOCAL 256u/* M.reward_mod_local                                       */
#define COBEL_SF_REWARD_MOD 512u      /* M.reward_mod                                             */
#define COBEL_SF_STATE_MOD 1024u      /* M.state_mod                                              */

/* Per-instance SFMA state: 8 x 32-bit words, caller-owned [dev] [N][8]. */
enum {
  COBEL_SI_CLOCK = 0,     /* uint32 experiences stored so far (the recency clock)              */
  COBEL_SI_EPOCH = 1,     /* uint32 clock at the last T.fill(0) (agent/sfma.py:325)            */
  COBEL_SI_MODE = 2,      /* int32  M.mode (COBEL_SFMA_*); rewritten in dynamic mode           */
  COBEL_SI_FLAGS = 3,     /* bit 0: the |TD| sum currently has float32 type                    */
  COBEL_SI_TD_LO = 4,     /* float64 agent.td, the |TD| sum (2 words)                          */
  COBEL_SI_TD_HI = 5,
  COBEL_SI_CTR_AGENT = 6, /* uint32 next index on COBEL_STREAM_AGENT                           */
  COBEL_SI_RESERVED = 7,
  COBEL_SI_WORDS = 8
};

/* One replayed experience (logs['replay'], agent/sfma.py:273,322), 24 bytes. */
typedef struct {
  uint32_t sa;     /* state | action << 16 | nonterminal << 24 | kind << 25 (1 = trial start)  */
  uint32_t next;   /* next state                                                               */
  float reward;    /* M.rewards[s][a] at replay time                                           */
  int32_t trial;   /* agent.current_trial of the trial the replay belongs to                   */
  double td;       /* TD error of the update (NaN for start-of-trial replays)                  */
} cobel_sfma_event_t;

typedef struct {
  /* tables, caller-owned device memory */
  float* q;               /* [N][S][4] float32 Q                                              */
  uint64_t* model;        /* [N][S][4] packed records as in cobel_tab_run_t (cobel_model_init) */
  double* strength;       /* [N][4S]  M.C, index a * S + s                                    */
  uint32_t* stamp;        /* [N][4S]  clock value at which (s, a) was last stored; M.T[j] =
                             recency_tab[clock - stamp[j]] if stamp[j] > epoch else 0         */
  int32_t* inst;          /* [N][COBEL_I_WORDS]                                               */
  int32_t* sfma_inst;     /* [N][COBEL_SI_WORDS]                                              */
  const double* metric;   /* [n_worlds][S][S] similarity matrix metric.D of each world        */
  const double* recency_tab; /* [recency_len] 1, d, fl(d*d), ... by repeated multiplication
                             (M.T *= decay_recency per store); ages beyond the end use the last */
  const double* random_cdf;  /* [4S] RANDOM: cumsum(p) / cumsum(p)[-1] of the masked uniform p */
  const uint8_t* action_mask; /* [S] 4-bit masks or NULL                                       */
  /* monitors (any may be NULL), as in cobel_tab_run_t */
  unsigned long long* lat_sum;
  unsigned long long* lat_cnt;
  double* reward_sum;
  unsigned long long* resp_cnt;
  int32_t* lat_trace;
  unsigned long long* occupancy;
  unsigned long long* steps_done;
  unsigned long long* replays_done; /* [1] experiences reactivated by this call (added)       */
  int32_t* last_exp;      /* [N][6] as in cobel_tab_run_t                                     */
  cobel_sfma_event_t* replay_trace; /* [N][trace_cap] or NULL                                 */
  int32_t* trace_len;     /* [N] in/out: events appended so far (counts past trace_cap too)   */
  /* sizes */
  int32_t n, trial_cap, trace_cap, recency_len;
  uint32_t instance_base;
  /* run parameters */
  uint32_t flags;         /* COBEL_F_LEARN | NO_REPLAY | MASK_ACTIONS | TEST_STREAM           */
  uint32_t sfma_flags;    /* COBEL_SF_*                                                       */
  int32_t trials_target, steps_per_trial, step_budget;
  int32_t batch;          /* replay length                                                    */
  int32_t nb_replays;     /* agent.nb_replays                                                 */
  int32_t mon_stripes;    /* as in cobel_tab_run_t                                            */
  double alpha, gamma, epsilon, model_lr;        /* agent lr, discount, policy eps, M.learning_rate */
  double decay_inhibition, decay_strength;       /* M.decay_inhibition, M.decay_strength      */
  double c_step, i_step, r_threshold, beta;      /* M.C_step, M.I_step, M.R_threshold, M.beta */
  double reward_modulation, blend, interp_fwd, interp_rev;
  uint64_t seed;
} cobel_sfma_run_t;

/* 0 = supported; fills *lds_bytes with the LDS one instance needs. */
COBEL_API int cobel_sfma_query(int32_t n_states, int32_t* lds_bytes);
COBEL_API int cobel_sfma_run(const cobel_world_t* world, const cobel_sfma_run_t* run, void* stream);
/* For tests: exp(x[i]) by the routine the plain-training SFMA kernel uses for its softmax weights
 * (arguments in [0, 700], memory/sfma.py:349-372) and by the device library's exp, which every
 * other instantiation calls; the two must agree bit for bit on that range.  With `divisor`: x[i] /
 * divisor[i] through the reciprocal the kernels share over a reactivation's experiences, and by
 * the division it replaces (the normalisation R / max R, memory/sfma.py:319-321).  Device pointers. */
COBEL_API int cobel_sfma_exp_check(const double* x, double* in_range, double* library,
                                   const double* divisor, double* quotient_by_reciprocal,
                                   double* quotient, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused Adam step for network parameters stacked over instances (the DQN path keeps one network
 * per agent-env instance, parameters [N][...] per tensor).  Replaces optimizer.step() behind
 * TorchNetwork.train_on_batch (network/network_torch.py:160-167) for torch.optim.Adam without
 * amsgrad: one pass over param / grad / exp_avg / exp_avg_sq instead of ~10 elementwise kernels.
 *   steps[i]  : Adam step count of instance i INCLUDING this step (bias corrections 1 - beta^step)
 *   active[i] : 0 = instance i keeps parameters and optimizer state untouched; NULL = all step
 * Elements [i * per_instance, (i + 1) * per_instance) of every tensor belong to instance i.
 * ------------------------------------------------------------------------------------------ */
COBEL_API int cobel_adam_step(void* param /* [dev] [N][per_instance] */,
                              const void* grad /* [dev] */, void* exp_avg /* [dev] */,
                              void* exp_avg_sq /* [dev] */, const double* steps /* [dev] [N] */,
                              const uint8_t* active /* [dev] [N] or NULL */, int64_t n_instances,
                              int64_t per_instance, int32_t is_float64, double lr, double beta1,
                              double beta2, double eps, double weight_decay,
                              void* target /* [dev] same shape, or NULL: target[e] += tau *
                                              (param_new[e] - target[e]) for stepping instances
                                              (the DQN target blend, agent/dqn.py:366-371) */,
                              double tau, void* stream);

/* ------------------------------------------------------------------------------------------
 * One DQN replay step per instance in ONE kernel, for networks of the shape the reference's DQN
 * demos and tests use — Linear(D, 64) - ReLU - Linear(64, 64) - ReLU - Linear(64, A), D <= 32,
 * A = 4 (1 .. 8 on the streaming form: the six neighbours of a hexagonal Topology),
 * batches of 32, float64 or float32, MSE loss, torch.optim.Adam without amsgrad:
 *   targets = Q_online(s); targets[a] = r + gamma * nt * max_a' Q_target(s')   (agent/dqn.py:346-364;
 *             ddqn != 0: a' = argmax Q_online(s'), :352-355)
 *   loss = mean((Q_online(s) - targets)^2); backward; Adam step   (network/network_torch.py:160-167)
 *   w_target += tau * (w_online - w_target)                       (agent/dqn.py:366-371; tau 0 = none)
 * Parameters are torch.nn.Linear tensors stacked over instances: w[l] [N][out][in], b[l] [N][out]
 * for the online network, *_target for the target network, m_* / v_* Adam's exp_avg / exp_avg_sq
 * of the online network.  steps / active as for cobel_adam_step.  The batch is given gathered:
 * states / next_states [N][32][D], actions int64 [N][32], rewards and the non-terminal flags
 * (1 - end_trial, agent/dqn.py:191) [N][32] in the network's dtype.  All device memory, caller-owned.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  void* w[3];
  void* b[3];
  void* w_target[3];
  void* b_target[3];
  void* m_w[3];
  void* m_b[3];
  void* v_w[3];
  void* v_b[3];
  const double* steps;      /* [N] Adam step count INCLUDING this step                          */
  const uint8_t* active;    /* [N] or NULL                                                      */
  const void* states;       /* [N][batch][n_inputs]                                             */
  const void* next_states;
  const int64_t* actions;   /* [N][batch]                                                       */
  const void* rewards;      /* [N][batch]                                                       */
  const void* nonterminal;  /* [N][batch]                                                       */
  int32_t n, n_inputs, n_hidden1, n_hidden2, n_actions, batch;
  int32_t is_float64, ddqn;
  double gamma, lr, beta1, beta2, eps, weight_decay, tau;
  /* optional: the kernel gathers the batch from the replay rings itself.  With batch_slots != NULL
     states / next_states / actions / rewards / nonterminal are the rings [N][ring_slots][..] and
     sample s of instance i is row batch_slots[i][s] (as written by cobel_dqn_act). */
  const int32_t* batch_slots; /* [N][batch] or NULL                                             */
  int32_t ring_slots;
  int32_t reserved_;
  /* optional: Q-values of each instance's NEXT observation, computed with the updated online
     network at the end of the step (what the next action selection needs, agent/dqn.py:174):
     observation of instance i = row obs_index[i] of obs_table.  Instances outside `active` are
     left alone. */
  const int32_t* obs_index;   /* [N]                                                            */
  const double* obs_table;    /* [rows][n_inputs] float64                                       */
  void* q_out;                /* [N][n_actions] network dtype, or NULL                          */
  /* optional: the batch's observations as rows of obs_table (DynaDQN: the model is indexed by
     integer states, agent/dyna_q.py:333-708).  With state_index != NULL states / next_states are
     not read; actions / rewards / nonterminal are the gathered [N][batch] arrays. */
  const int32_t* state_index; /* [N][batch] or NULL                                             */
  const int32_t* next_index;  /* [N][batch]                                                     */
} cobel_dqn_replay_t;
/* 0 = the fused step covers this network / batch shape; fills *lds_bytes (per instance). */
COBEL_API int cobel_dqn_replay_query(int32_t n_inputs, int32_t n_hidden1, int32_t n_hidden2,
                                     int32_t n_actions, int32_t batch, int32_t is_float64,
                                     int32_t* lds_bytes);
COBEL_API int cobel_dqn_replay(const cobel_dqn_replay_t* run, void* stream);

/* ------------------------------------------------------------------------------------------
 * TorchNetwork.train_on_batch / predict_on_batch for STACKS of small networks — Linear(D <= 32,
 * 64)-ReLU-Linear(64, 64)-ReLU-Linear(64, O <= 32), batches of 32, float64 or float32 — one
 * kernel each (network/network_torch.py:110-167).  They carry DynaDSR.replay
 * (agent/dyna_q.py:1042-1150): per agent four online + four target successor networks
 * (D -> 64 -> 64 -> D) and one reward network (D -> 64 -> 64 -> 1).
 *
 * cobel_mlp_forward: out[j] = net[j / net_div](inputs of instance j), 32 rows per instance; the
 *   inputs are rows in_table[in_index[j / in_div][s]] of a float64 table or a dense block
 *   in_dense[j][32][D] in the network's dtype.
 * cobel_mlp_fit: network j takes ONE optimisation step towards targets[j / tgt_div][32][O] on its
 *   32 input rows: loss = mean over the samples marked in sample_mask[j] (NULL: all) and the O
 *   outputs of (out - target)^2, torch.optim.Adam with the network's own step count steps[j]
 *   (incremented here), then w_target += tau * (w - w_target).  train[j] == 0: no optimiser step
 *   (parameters, moments and step count untouched) but the blend still happens, as the reference
 *   blends every network pair every step (agent/dyna_q.py:1134-1143).  active[j / act_div] == 0:
 *   nothing at all.  Afterwards ep_out[j][ep_rows][O] receives the (updated) network's outputs
 *   for ep_rows <= 4 extra rows — row ep_table[ep_index[j / ep_div]] or ep_dense[j][ep_rows][D] —
 *   which is what the next action selection needs.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* w[3];        /* [M][out][in], network dtype */
  const void* b[3];        /* [M][out]                    */
  const uint8_t* active;   /* [n / act_div] or NULL       */
  const double* in_table;  /* [rows][D] float64, or NULL  */
  const int32_t* in_index; /* [n / in_div][32]            */
  const void* in_dense;    /* [n][32][D], network dtype   */
  void* out;               /* [n][32][O]                  */
  int32_t n, n_inputs, n_outputs, is_float64;
  int32_t net_div, in_div, act_div, reserved_;
} cobel_mlp_forward_t;

typedef struct {
  void* w[3];              /* [n][out][in] */
  void* b[3];
  void* w_target[3];       /* blend targets, or NULL with tau == 0 */
  void* b_target[3];
  void* m_w[3];            /* Adam exp_avg / exp_avg_sq */
  void* m_b[3];
  void* v_w[3];
  void* v_b[3];
  double* steps;           /* [n] Adam step counts, += 1 for every network that trains            */
  const uint8_t* train;    /* [n] or NULL (= all)                                                 */
  const uint8_t* active;   /* [n / act_div] or NULL                                               */
  const double* in_table;  /* inputs as for cobel_mlp_forward                                     */
  const int32_t* in_index;
  const void* in_dense;
  const void* targets;     /* [n / tgt_div][32][O], network dtype                                 */
  const uint8_t* sample_mask; /* [n][32] or NULL                                                  */
  const double* ep_table;  /* extra rows: one row of a float64 table ...                          */
  const int32_t* ep_index; /* ... [n / ep_div]                                                    */
  const void* ep_dense;    /* ... or [n][ep_rows][D], network dtype                               */
  void* ep_out;            /* [n][ep_rows][O] or NULL                                             */
  int32_t n, n_inputs, n_outputs, is_float64;
  int32_t in_div, tgt_div, act_div, ep_div, ep_rows;
  int32_t debug_stage;     /* 0; k = 1..4: return after the forward pass / the output layer / the
                              second layer's gradient / delta1 (phase timings, scripts/exp_mlp_fit.py) */
  double lr, beta1, beta2, eps, weight_decay, tau;
} cobel_mlp_fit_t;
/* 0 = this shape is covered; fills *lds_bytes (per workgroup). */
COBEL_API int cobel_mlp_query(int32_t n_inputs, int32_t n_hidden1, int32_t n_hidden2,
                              int32_t n_outputs, int32_t batch, int32_t is_float64,
                              int32_t* lds_bytes);
COBEL_API int cobel_mlp_forward(const cobel_mlp_forward_t* run, void* stream);
COBEL_API int cobel_mlp_fit(const cobel_mlp_fit_t* run, void* stream);

/* The regression targets of DynaDSR.replay (agent/dyna_q.py:1079-1131) for all samples of all
 * agents in ONE launch — what sits between the two forward passes and the two fits of a step:
 *   best[s]   = argmax_a value[a][s]   (first maximum)      | use_dr: the mean over the actions
 *   boot_sr   = successor[best[s]][s][:]                     |         of successor[a][s][:]
 *   nt        = nonterminal[s] != 0
 *   boot      = next_obs * ((1 - follow_up)(1 - ignore)) * (1 - nt) + boot_sr * min(nt + ignore, 1)
 *   targets[s][:] = (follow_up ? next_obs : obs) + gamma * boot
 *   took[a][s] = actions[s] == a;  train[a] = any_s took[a][s]
 * with obs / next_obs = rows state_index[s] / next_index[s] of the float64 observation table
 * converted to the network dtype.  Operation order as in the reference's expressions (and in the
 * PyTorch path of cobel_amd.agent.DynaDSR), element for element. */
typedef struct {
  const void* successor;      /* [n][A][32][O] network dtype: target networks on the next states  */
  const void* value;          /* [n][A][32]    network dtype: reward network on those             */
  const double* table;        /* [rows][O] float64 observation table                              */
  const int32_t* state_index; /* [n][32]                                                          */
  const int32_t* next_index;  /* [n][32]                                                          */
  const int64_t* actions;     /* [n][32]                                                          */
  const void* nonterminal;    /* [n][32] network dtype                                            */
  void* targets;              /* out [n][32][O] network dtype                                     */
  uint8_t* took;              /* out [n * A][32]                                                  */
  uint8_t* train;             /* out [n * A]                                                      */
  int32_t n, n_actions, n_outputs, is_float64;
  int32_t use_dr, follow_up, ignore_terminality, reserved_;
  double gamma;
} cobel_dsr_targets_t;
COBEL_API int cobel_dsr_targets(const cobel_dsr_targets_t* run, void* stream);

/* ------------------------------------------------------------------------------------------
 * Everything of one lockstep DQN training step that is not the network: per instance
 *   epsilon-greedy on the given Q-values (policy/greedy.py:40-88) -> env.step
 *   (interface/topology.py:146-157, gridworld.py:115-126) -> append the experience to the replay
 *   ring (memory/dqn.py:103-119; FIFO at capacity) -> trial bookkeeping and monitors
 *   (agent/dqn.py:186-212, monitor/behavior.py:73-97) with auto-reset (topology.py:170) ->
 *   draw the replay batch (memory/dqn.py:137: uniform over the stored entries, with replacement).
 * One lane per instance; instances whose `active` flag is 0 are left untouched (they consume no
 * draws).  Observations are rows of obs_table indexed by the env state (Topology poses, one-hot
 * rows for a Gridworld).  Outputs for cobel_dqn_replay: `stepped` (1 = the instance took part
 * in this step; its Adam step count adam_steps[i] has been incremented) and `batch_slots`.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  /* environment */
  int32_t* state;           /* [N] in/out                                                       */
  uint32_t* env_ctr;        /* [N] next index on COBEL_STREAM_ENV                               */
  const double* obs_table;  /* [S][n_obs] float64                                               */
  /* policy */
  const void* q;            /* [N][A] Q-values of the current observations, network dtype; A =
                               the world's action count: four, or 1 .. 8 on a world of
                               cobel_world_create_n (replay ring only, one-hot rows)            */
  uint32_t* policy_ctr;     /* [N] next index on the policy stream                              */
  uint32_t policy_stream;   /* COBEL_STREAM_POLICY or COBEL_STREAM_POLICY_TEST                  */
  int32_t is_float64;       /* dtype of q and of the ring's floating-point arrays               */
  double epsilon;
  /* replay ring, [N][slots][..] */
  void* ring_states;
  void* ring_next_states;
  int64_t* ring_actions;
  void* ring_rewards;
  void* ring_nonterminal;
  int32_t* ring_size;       /* [N] stored entries                                               */
  int64_t* ring_head;       /* [N] slot of the oldest entry                                     */
  uint32_t* memory_ctr;     /* [N] next index on COBEL_STREAM_MEMORY                            */
  /* trial bookkeeping */
  int32_t* trial;           /* [N] trials finished                                              */
  int32_t* step;            /* [N] steps taken in the running trial                             */
  double* trial_reward;     /* [N]                                                              */
  uint8_t* active;          /* [N] in/out: 0 once trial == trials_target                        */
  double* adam_steps;       /* [N] += 1 for every instance that steps                           */
  long long* lat_sum;       /* [mon_stripes][trial_cap] (instance i adds into copy i % mon_stripes) */
  long long* lat_cnt;
  double* reward_sum;
  /* outputs */
  uint8_t* stepped;         /* [N]                                                              */
  int32_t* batch_slots;     /* [N][batch]                                                       */
  /* sizes and parameters */
  int32_t n, n_obs, slots, batch;
  int32_t steps_per_trial, trials_target, trial_cap, mon_stripes;
  uint32_t instance_base;
  uint32_t reserved_;
  uint64_t seed;
  /* optional, instead of the replay ring (ring_* may then be NULL): the tabular world model of
     DynaDQN (memory/dyna_q.py:62-157 in float64, as the reference keeps it): store = running
     reward estimate + successor + non-terminal flag of (state, action); the batch = `batch` pairs
     drawn uniformly from ALL n_states x 4 pairs, written out gathered. */
  double* model_rewards;        /* [N][n_states * 4]                                            */
  int64_t* model_states;        /* [N][n_states * 4]                                            */
  double* model_nonterminal;    /* [N][n_states * 4]                                            */
  double model_lr;
  int32_t n_states;
  int32_t reserved2_;
  int32_t* batch_state_index;   /* [N][batch] out: state of each drawn pair                     */
  int32_t* batch_next_index;    /* [N][batch] out: its modelled successor                       */
  int64_t* batch_actions;       /* [N][batch] out                                               */
  void* batch_rewards;          /* [N][batch] out, network dtype                                */
  void* batch_nonterminal;      /* [N][batch] out, network dtype                                */
} cobel_dqn_act_t;
COBEL_API int cobel_dqn_act(const cobel_world_t* world, const cobel_dqn_act_t* run, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COBEL_HIP_H */
