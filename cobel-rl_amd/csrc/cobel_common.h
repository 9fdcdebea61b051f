// Internal declarations shared by the translation units of libcobel_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cobel_hip.h"
#include "cobel_rng.h"

// World record, one per state, 16 B so that a single dwordx4 load returns everything the
// loop needs about a state: where each action leads, and what entering the state pays.
struct __attribute__((aligned(16))) cobel_wrec {
  uint16_t next[4];
  float reward;
  uint32_t terminal;  // 0 / 1
};
static_assert(sizeof(cobel_wrec) == 16, "world record must be 16 bytes");

// The rewarded states of one world (reward != 0), at most 32 of them, and the order in which
// NumPy's pairwise summation (numpy/core/src/umath/loops_utils.h) adds the products of a row with a
// vector that is zero everywhere else: a sum over S elements of which all but k are zero is the sum
// of those k in the grouping the summation tree gives them (an addition of zero changes nothing),
// so k - 1 additions in this order return np.sum(row * R) bit for bit (up to the sign of a zero).
struct cobel_rw_info {
  uint16_t pos[32];     // ascending states
  uint8_t k;            // how many (0 .. 32); 255: more
  uint8_t root;         // slot that holds the sum after the last step
  uint8_t dst[31], src[31];   // step t: value[dst[t]] += value[src[t]]
};
static_assert(sizeof(cobel_rw_info) == 128, "cobel_rw_info layout");

struct cobel_world {
  int32_t n_states, n_worlds, device;
  cobel_wrec* rec;       // [dev] [n_worlds][S]
  uint16_t* starts;      // [dev] concatenated
  int32_t* start_off;    // [dev] [n_worlds + 1]
  int32_t* h_start_off;  // [host] copy for argument checks
  int32_t max_rewarded_states;  // max over worlds of #{s : reward[s] != 0}
  uint32_t* queue;       // [dev] 256 B: ticket counters of the persistent-workgroup kernel for calls
                         // that bring no scratch area (cobel_tab_run_t.scratch)
  cobel_rw_info* rw;     // [dev] [n_worlds] rewarded states + pairwise combine order (four-action worlds)
  // action counts other than four (cobel_world_create_n): `rec` is NULL and these hold the world
  int32_t n_actions;
  uint16_t* next_n;     // [dev] [n_worlds][S][n_actions]
  float* reward_s;      // [dev] [n_worlds][S]
  uint8_t* terminal_s;  // [dev] [n_worlds][S]
  // transition rows that are distributions (cobel_world_set_transitions), else NULL: successors
  // of pair p = (world * S + s) * n_actions + a are succ_state[succ_off[p] .. succ_off[p + 1]) with
  // the normalised cumulative probabilities succ_cdf (last entry of a row = 1)
  uint32_t* succ_off;   // [dev] [n_worlds * S * n_actions + 1]
  uint16_t* succ_state; // [dev] [nnz]
  double* succ_cdf;     // [dev] [nnz]
};

// the successor Generator.choice(arange(S), p=row) returns for the uniform u (interface/
// gridworld.py:119-123): first entry whose cumulative probability exceeds u
__device__ __forceinline__ int cobel_draw_successor(const uint32_t* __restrict__ off,
                                                    const uint16_t* __restrict__ succ,
                                                    const double* __restrict__ cdf, size_t pair,
                                                    double u) {
  const uint32_t lo = off[pair], hi = off[pair + 1];
  uint32_t k = lo;
  while (k + 1u < hi && !(cdf[k] > u)) ++k;
  return (int)succ[k];
}

// general.hip: any action count / batch size / state count (one lane per instance)
int cobel_env_step_general(const cobel_world* world, int32_t* state, const uint8_t* action,
                           float* reward_out, uint8_t* done_out, uint32_t* env_ctr, uint64_t seed,
                           int32_t n, uint32_t instance_base, hipStream_t st);
int cobel_tab_general_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st);

// tabular_pwg.hip: plain Dyna-Q training as one persistent workgroup per CU (Q in LDS + Q in L2)
bool cobel_tab_pwg_plan(const cobel_world* world, const cobel_tab_run_t& r, int* nl, int* ng,
                        size_t* lds_bytes);
int cobel_tab_pwg_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st);

// tabular_nact.hip: Q-learning on worlds of 1..32 (not four) actions, one wavefront per instance
bool cobel_tab_nact_covers(const cobel_world* world, const cobel_tab_run_t& r, size_t* lds_bytes,
                           int* instances_per_workgroup);
int cobel_tab_nact_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st);

// world.hip: the additions NumPy's pairwise sum performs on k non-zero elements at `pos` (ascending)
// of a vector of n elements; returns the slot holding the result (-1: k == 0)
int cobel_pairwise_schedule(int n, const int* pos, int k, uint8_t* dst, uint8_t* src);

// sr_wave.hip: the sparse-reward form of the SR agent (one wavefront per instance)
bool cobel_sr_wave_covers(const cobel_world* world, const cobel_sr_run_t& r);
int cobel_sr_wave_launch(const cobel_world* world, const cobel_sr_run_t& r, hipStream_t st);

int cobel_fail(int code, const char* fmt, ...);
int cobel_device_limits(int device, int* n_cu, size_t* lds_per_cu);   // world.hip (cached per device)
// COBEL_DEBUG_LDS_PAD=<bytes> (occupancy experiments, scripts/experiments/exp_occ*.py): extra dynamic LDS per
// workgroup.  Honoured only if it is a plain number that keeps `base + pad` within `limit`; anything
// else (a stray or malformed variable) is ignored, so it can change occupancy, never break a launch.
size_t cobel_debug_lds_pad(size_t base, size_t limit);
// getenv(name) under the master switch COBEL_DEBUG=1, else NULL (world.hip)
const char* cobel_debug_env(const char* name);
// mlp.hip: the parameter-staging DQN replay kernel (cobel_dqn_replay, mlp_fit.hip, dispatches)
size_t cobel_dqn_replay_lds_bytes(int32_t n_inputs, int32_t is_float64);
int cobel_dqn_replay_lds_launch(const cobel_dqn_replay_t& r, hipStream_t st,
                                unsigned long long* trace /* experiments, or NULL */);
int cobel_world_check(const cobel_world* w, const char* who);   // non-NULL, on the current device
int cobel_world_check4(const cobel_world* w, const char* who);  // ... and a four-action world

#define COBEL_HIP_TRY(expr)                                                                  \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return cobel_fail(COBEL_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
  } while (0)

#define COBEL_REQUIRE(cond, code, ...) \
  do {                                 \
    if (!(cond)) return cobel_fail(code, __VA_ARGS__); \
  } while (0)

// Packed 8-byte records.
//   model entry  : lo = f32 reward estimate, hi = next_state | nonterminal << 16
//   replay entry : lo = f32 reward,          hi = state | next_state << 14 | action << 28 | nonterminal << 30
__host__ __device__ __forceinline__ uint64_t cobel_model_pack(float r, uint32_t ns, uint32_t nt) {
  return (uint64_t)__builtin_bit_cast(uint32_t, r) | ((uint64_t)(ns | (nt << 16)) << 32);
}
__host__ __device__ __forceinline__ uint64_t cobel_log_pack(float r, uint32_t s, uint32_t a,
                                                            uint32_t ns, uint32_t nt) {
  return (uint64_t)__builtin_bit_cast(uint32_t, r) |
         ((uint64_t)(s | (ns << 14) | (a << 28) | (nt << 30)) << 32);
}

// Device-side view of the epsilon-greedy CDF table (cobel_policy_table), passed by value.
struct cobel_cdf_table {
  double cdf[16][16][4];
};

// Per-trial monitors are striped: workgroup b adds into copy b % mon_stripes of each array, so
// that the atomics of thousands of workgroups finishing the same trial indices do not all queue on
// the same few cache lines of one L2 channel (measured on C2: 600 000 atomics per launch onto ~150
// hot addresses cost 1.1 ms of a 2.8 ms launch; Dyna-Q on 65 536 5x5 worlds ran 2x slower).  The
// caller sums the copies.
#if defined(__HIPCC__)
__device__ __forceinline__ size_t cobel_mon_offset(int32_t mon_stripes, int32_t trial_cap) {
  const unsigned stripes = mon_stripes > 1 ? (unsigned)mon_stripes : 1u;
  return (size_t)((unsigned)blockIdx.x % stripes) * (size_t)trial_cap;
}
#endif
