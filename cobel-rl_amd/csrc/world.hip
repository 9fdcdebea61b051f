// World handle, stand-alone vectorised env step/reset and the epsilon-greedy KAT entry point.
// Reference behaviour restated (paths relative to /root/reference/src/cobel):
//   interface/gridworld.py:115-126  step  = next-state lookup, reward/terminal of the state entered
//   interface/gridworld.py:142      reset = uniform draw from starting_states
//   interface/topology.py:146-157   same shape of step on a neighbour table
//   policy/greedy.py:40-88          epsilon-greedy
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "cobel_common.h"
#include "cobel_policy.h"

static thread_local char g_err[512] = "";

int cobel_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// The experiment switches (COBEL_DEBUG_*: occupancy padding, kernel choice, phase cuts, slice plans)
// are honoured only under the master switch COBEL_DEBUG=1: a stray variable in a production
// environment changes nothing.
const char* cobel_debug_env(const char* name) {
  const char* const on = getenv("COBEL_DEBUG");
  if (!on || on[0] != '1' || on[1] != '\0') return nullptr;
  return getenv(name);
}

size_t cobel_debug_lds_pad(size_t base, size_t limit) {
  const char* const v = cobel_debug_env("COBEL_DEBUG_LDS_PAD");
  if (!v || !*v) return 0;
  char* end = nullptr;
  const long pad = strtol(v, &end, 10);
  if (*end != '\0' || pad <= 0 || base + (size_t)pad > limit) return 0;
  return (size_t)pad;
}

// CUs and LDS bytes per CU of a device, asked once per device (the persistent-workgroup kernel
// sizes its grid and its workgroup by them).
int cobel_device_limits(int device, int* n_cu, size_t* lds_per_cu) {
  struct limits {
    std::once_flag once;
    int cu = 0;
    size_t lds = 0;
    hipError_t err = hipSuccess;
  };
  static limits table[64];
  COBEL_REQUIRE(device >= 0 && device < 64, COBEL_E_ARG, "cobel_device_limits: device %d", device);
  limits& l = table[device];
  // (agents of several host threads may ask at once: filled exactly once per device, and a reader
  //  sees both values or waits)
  std::call_once(l.once, [&l, device] {
    hipDeviceProp_t prop;
    l.err = hipGetDeviceProperties(&prop, device);
    if (l.err != hipSuccess) return;
    size_t b = prop.maxSharedMemoryPerMultiProcessor;
    if (b == 0 || b > 160 * 1024) b = 160 * 1024;   // (gfx950: 160 KiB; the kernels are laid out for it)
    l.lds = b;
    l.cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  });
  COBEL_HIP_TRY(l.err);
  *n_cu = l.cu;
  *lds_per_cu = l.lds;
  return COBEL_OK;
}

extern "C" const char* cobel_last_error(void) { return g_err; }
extern "C" int cobel_abi_version(void) { return 1017; }

extern "C" int cobel_param_set_fill(double alpha, double gamma, double epsilon, double model_lr,
                                    cobel_param_set_t* out) {
  COBEL_REQUIRE(out, COBEL_E_ARG, "cobel_param_set_fill: NULL out");
  COBEL_REQUIRE(epsilon >= 0.0 && epsilon <= 1.0, COBEL_E_ARG,
                "cobel_param_set_fill: epsilon %g outside [0, 1]", epsilon);
  static_assert(sizeof(cobel_param_set_t) == 512, "cobel_param_set_t layout");
  const cobel_eps_consts c = cobel_make_eps_consts(epsilon);
  out->alpha = alpha;
  out->gamma = gamma;
  out->epsilon = epsilon;
  out->model_lr = model_lr;
  out->alpha_f = (float)alpha;
  out->gamma_f = (float)gamma;
  out->model_lr_f = (float)model_lr;
  out->reserved_ = 0.0f;
  for (int n = 0; n < 5; ++n) {
    out->eps_base[n] = c.base[n];
    out->eps_bonus[n] = c.bonus[n];
  }
  for (int t = 0; t < 16; ++t)
    for (int k = 0; k < 3; ++k) out->eps_thr[t][k] = c.thr[t][k];
  return COBEL_OK;
}

extern "C" uint64_t cobel_pack_model(float reward, uint16_t next_state, uint8_t nonterminal) {
  return cobel_model_pack(reward, next_state, nonterminal ? 1u : 0u);
}
extern "C" void cobel_unpack_model(uint64_t rec, float* reward, uint16_t* next_state,
                                   uint8_t* nonterminal) {
  const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
  if (reward) *reward = __builtin_bit_cast(float, lo);
  if (next_state) *next_state = (uint16_t)(hi & 0xffffu);
  if (nonterminal) *nonterminal = (uint8_t)((hi >> 16) & 1u);
}


// ---------------------------------------------------------------------------------------------
// NumPy's pairwise summation restricted to the elements that are not zero (see cobel_rw_info).
// Values are slots 0 .. k-1 (element pos[slot]), -1 stands for an exact zero.
namespace {
struct pw_builder {
  const int* pos;
  int k, steps;
  uint8_t* dst;
  uint8_t* src;
  int add(int x, int y) {
    if (x < 0) return y;
    if (y < 0) return x;
    dst[steps] = (uint8_t)x;
    src[steps] = (uint8_t)y;
    steps += 1;
    return x;
  }
  int slot_at(int e) const {
    for (int s = 0; s < k; ++s)
      if (pos[s] == e) return s;
    return -1;
  }
  bool any_in(int lo, int len) const {
    for (int s = 0; s < k; ++s)
      if (pos[s] >= lo && pos[s] < lo + len) return true;
    return false;
  }
  int sum(int lo, int len) {
    if (!any_in(lo, len)) return -1;
    if (len < 8) {
      int res = -1;
      for (int i = 0; i < len; ++i) res = add(res, slot_at(lo + i));
      return res;
    }
    if (len <= 128) {
      int r[8];
      for (int j = 0; j < 8; ++j) r[j] = slot_at(lo + j);
      int i = 8;
      for (; i < len - (len % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] = add(r[j], slot_at(lo + i + j));
      int res = add(add(add(r[0], r[1]), add(r[2], r[3])), add(add(r[4], r[5]), add(r[6], r[7])));
      for (; i < len; ++i) res = add(res, slot_at(lo + i));
      return res;
    }
    int n2 = len / 2;
    n2 -= n2 % 8;
    const int left = sum(lo, n2);
    const int right = sum(lo + n2, len - n2);
    return add(left, right);
  }
};
}  // namespace

int cobel_pairwise_schedule(int n, const int* pos, int k, uint8_t* dst, uint8_t* src) {
  pw_builder b{pos, k, 0, dst, src};
  return b.sum(0, n);
}

// exported for tests: evaluates nothing on the device
extern "C" int cobel_pairwise_order(int32_t n, const int32_t* pos, int32_t k, uint8_t* dst,
                                    uint8_t* src, int32_t* root) {
  COBEL_REQUIRE(n > 0 && pos && dst && src && root && k >= 0 && k <= 32, COBEL_E_ARG,
                "cobel_pairwise_order: bad arguments");
  for (int j = 0; j < k; ++j)
    COBEL_REQUIRE(pos[j] >= 0 && pos[j] < n && (j == 0 || pos[j] > pos[j - 1]), COBEL_E_RANGE,
                  "cobel_pairwise_order: positions must ascend inside [0, n)");
  int p[32];
  for (int j = 0; j < k; ++j) p[j] = pos[j];
  *root = cobel_pairwise_schedule(n, p, k, dst, src);
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int cobel_world_create(const uint16_t* next, const float* reward,
                                  const uint8_t* terminal, const uint16_t* starts,
                                  const int32_t* start_offsets, int32_t n_states,
                                  int32_t n_worlds, int32_t device, cobel_world_t** out) {
  COBEL_REQUIRE(next && reward && terminal && starts && start_offsets && out, COBEL_E_ARG,
                "cobel_world_create: NULL argument");
  COBEL_REQUIRE(n_states > 0 && n_states <= 16384, COBEL_E_RANGE,
                "cobel_world_create: n_states %d outside 1..16384", n_states);
  COBEL_REQUIRE(n_worlds > 0, COBEL_E_RANGE, "cobel_world_create: n_worlds %d", n_worlds);
  COBEL_REQUIRE(start_offsets[0] == 0, COBEL_E_ARG, "cobel_world_create: start_offsets[0] != 0");
  for (int w = 0; w < n_worlds; ++w)
    COBEL_REQUIRE(start_offsets[w + 1] > start_offsets[w], COBEL_E_ARG,
                  "cobel_world_create: world %d has no starting state", w);
  const size_t total = (size_t)n_worlds * (size_t)n_states;
  std::vector<cobel_wrec> rec(total);
  for (size_t i = 0; i < total; ++i) {
    for (int a = 0; a < 4; ++a) {
      COBEL_REQUIRE(next[i * 4 + a] < n_states, COBEL_E_RANGE,
                    "cobel_world_create: next[%zu][%d] = %u >= n_states", i, a,
                    (unsigned)next[i * 4 + a]);
      rec[i].next[a] = next[i * 4 + a];
    }
    rec[i].reward = reward[i];
    rec[i].terminal = terminal[i] ? 1u : 0u;
  }
  const int32_t n_starts = start_offsets[n_worlds];
  for (int32_t i = 0; i < n_starts; ++i)
    COBEL_REQUIRE(starts[i] < n_states, COBEL_E_RANGE, "cobel_world_create: start %u >= n_states",
                  (unsigned)starts[i]);

  COBEL_HIP_TRY(hipSetDevice(device));
  cobel_world* w = (cobel_world*)calloc(1, sizeof(cobel_world));
  COBEL_REQUIRE(w, COBEL_E_ARG, "cobel_world_create: out of host memory");
  w->n_states = n_states;
  w->n_worlds = n_worlds;
  w->device = device;
  w->n_actions = 4;
  std::vector<cobel_rw_info> rw((size_t)n_worlds);
  for (int k = 0; k < n_worlds; ++k) {
    int32_t rewarded = 0;
    int pos[32];
    for (int32_t s = 0; s < n_states; ++s)
      if (reward[(size_t)k * n_states + s] != 0.0f) {
        if (rewarded < 32) pos[rewarded] = s;
        rewarded += 1;
      }
    if (rewarded > w->max_rewarded_states) w->max_rewarded_states = rewarded;
    cobel_rw_info& info = rw[(size_t)k];
    memset(&info, 0, sizeof(info));
    info.k = rewarded <= 32 ? (uint8_t)rewarded : (uint8_t)255;
    if (rewarded <= 32) {
      for (int j = 0; j < rewarded; ++j) info.pos[j] = (uint16_t)pos[j];
      const int root = cobel_pairwise_schedule(n_states, pos, rewarded, info.dst, info.src);
      info.root = (uint8_t)(root < 0 ? 0 : root);
    }
  }
  w->h_start_off = (int32_t*)malloc(sizeof(int32_t) * (n_worlds + 1));
  memcpy(w->h_start_off, start_offsets, sizeof(int32_t) * (n_worlds + 1));
  hipError_t e = hipMalloc((void**)&w->rec, total * sizeof(cobel_wrec));
  if (e == hipSuccess) e = hipMalloc((void**)&w->starts, sizeof(uint16_t) * n_starts);
  if (e == hipSuccess) e = hipMalloc((void**)&w->start_off, sizeof(int32_t) * (n_worlds + 1));
  if (e == hipSuccess) e = hipMalloc((void**)&w->queue, 256);
  if (e == hipSuccess) e = hipMalloc((void**)&w->rw, sizeof(cobel_rw_info) * (size_t)n_worlds);
  if (e == hipSuccess)
    e = hipMemcpy(w->rw, rw.data(), sizeof(cobel_rw_info) * (size_t)n_worlds, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->rec, rec.data(), total * sizeof(cobel_wrec), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->starts, starts, sizeof(uint16_t) * n_starts, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->start_off, start_offsets, sizeof(int32_t) * (n_worlds + 1),
                  hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    cobel_world_destroy(w);
    return cobel_fail(COBEL_E_HIP, "cobel_world_create: %s", hipGetErrorString(e));
  }
  *out = w;
  return COBEL_OK;
}

extern "C" int cobel_world_destroy(cobel_world_t* w) {
  if (!w) return COBEL_OK;
  if (w->rec) (void)hipFree(w->rec);
  if (w->starts) (void)hipFree(w->starts);
  if (w->start_off) (void)hipFree(w->start_off);
  if (w->queue) (void)hipFree(w->queue);
  if (w->rw) (void)hipFree(w->rw);
  if (w->next_n) (void)hipFree(w->next_n);
  if (w->reward_s) (void)hipFree(w->reward_s);
  if (w->terminal_s) (void)hipFree(w->terminal_s);
  if (w->succ_off) (void)hipFree(w->succ_off);
  if (w->succ_state) (void)hipFree(w->succ_state);
  if (w->succ_cdf) (void)hipFree(w->succ_cdf);
  free(w->h_start_off);
  free(w);
  return COBEL_OK;
}

extern "C" int cobel_world_set_transitions(cobel_world_t* w, const uint32_t* succ_off,
                                           const uint16_t* succ_state, const double* succ_cdf,
                                           int64_t nnz) {
  if (int rc = cobel_world_check(w, "cobel_world_set_transitions")) return rc;
  COBEL_REQUIRE(succ_off && succ_state && succ_cdf, COBEL_E_ARG,
                "cobel_world_set_transitions: NULL table");
  COBEL_REQUIRE(!w->succ_off, COBEL_E_ARG, "cobel_world_set_transitions: already set");
  const size_t pairs = (size_t)w->n_worlds * w->n_states * w->n_actions;
  COBEL_REQUIRE(nnz >= (int64_t)pairs && nnz < ((int64_t)1 << 32), COBEL_E_RANGE,
                "cobel_world_set_transitions: %lld successors for %zu pairs", (long long)nnz, pairs);
  COBEL_REQUIRE(succ_off[0] == 0 && succ_off[pairs] == (uint32_t)nnz, COBEL_E_ARG,
                "cobel_world_set_transitions: offsets do not span the lists");
  for (size_t p = 0; p < pairs; ++p) {
    const uint32_t lo = succ_off[p], hi = succ_off[p + 1];
    COBEL_REQUIRE(hi > lo && hi <= (uint32_t)nnz, COBEL_E_ARG,
                  "cobel_world_set_transitions: pair %zu has no successor", p);
    double prev = 0.0;
    for (uint32_t k = lo; k < hi; ++k) {
      COBEL_REQUIRE((int)succ_state[k] < w->n_states, COBEL_E_RANGE,
                    "cobel_world_set_transitions: successor %u outside the world", succ_state[k]);
      COBEL_REQUIRE(succ_cdf[k] > prev && succ_cdf[k] <= 1.0, COBEL_E_ARG,
                    "cobel_world_set_transitions: pair %zu: cumulative probabilities must increase "
                    "to 1", p);
      prev = succ_cdf[k];
    }
    COBEL_REQUIRE(prev == 1.0, COBEL_E_ARG,
                  "cobel_world_set_transitions: pair %zu: the last cumulative probability is %g, not 1",
                  p, prev);
  }
  hipError_t e = hipMalloc((void**)&w->succ_off, (pairs + 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&w->succ_state, (size_t)nnz * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&w->succ_cdf, (size_t)nnz * sizeof(double));
  if (e == hipSuccess)
    e = hipMemcpy(w->succ_off, succ_off, (pairs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->succ_state, succ_state, (size_t)nnz * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(w->succ_cdf, succ_cdf, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (w->succ_off) (void)hipFree(w->succ_off);
    if (w->succ_state) (void)hipFree(w->succ_state);
    if (w->succ_cdf) (void)hipFree(w->succ_cdf);
    w->succ_off = nullptr;
    w->succ_state = nullptr;
    w->succ_cdf = nullptr;
    return cobel_fail(COBEL_E_HIP, "cobel_world_set_transitions: %s", hipGetErrorString(e));
  }
  return COBEL_OK;
}

extern "C" int cobel_world_info(const cobel_world_t* w, int32_t* n_states, int32_t* n_worlds,
                                int32_t* device) {
  COBEL_REQUIRE(w, COBEL_E_ARG, "cobel_world_info: NULL world");
  if (n_states) *n_states = w->n_states;
  if (n_worlds) *n_worlds = w->n_worlds;
  if (device) *device = w->device;
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
// One lane per instance; state/action/reward/done are SoA so every access is coalesced, the world
// records are shared and stay in L2.
__global__ __launch_bounds__(256) void k_env_step(const cobel_wrec* __restrict__ rec, int S,
                                                  int n_worlds, int32_t* __restrict__ state,
                                                  const uint8_t* __restrict__ action,
                                                  float* __restrict__ reward_out,
                                                  uint8_t* __restrict__ done_out, int n,
                                                  uint32_t base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cobel_wrec* w = rec + (size_t)((base + (uint32_t)i) % (uint32_t)n_worlds) * S;
  // (the host setters refuse states outside the world; the clamp only keeps a buffer the caller
  //  filled by other means from reading outside the tables)
  const int s = min(max(state[i], 0), S - 1);
  const int a = action[i] & 3;
  const int ns = w[s].next[a];
  const cobel_wrec r = w[ns];
  state[i] = ns;
  if (reward_out) reward_out[i] = r.reward;
  if (done_out) done_out[i] = (uint8_t)r.terminal;
}

__global__ __launch_bounds__(256) void k_env_reset(const uint16_t* __restrict__ starts,
                                                   const int32_t* __restrict__ start_off,
                                                   int n_worlds, int32_t* __restrict__ state,
                                                   const uint8_t* __restrict__ reset_mask,
                                                   uint32_t* __restrict__ env_ctr, uint64_t seed,
                                                   int n, uint32_t base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (reset_mask && !reset_mask[i]) return;
  const uint32_t g = base + (uint32_t)i;
  const int w = (int)(g % (uint32_t)n_worlds);
  const int lo = start_off[w], cnt = start_off[w + 1] - lo;
  const uint32_t idx = env_ctr[i];
  state[i] = starts[lo + (int)cobel_draw_bounded(idx, 0u, g, COBEL_STREAM_ENV, seed, (uint32_t)cnt)];
  env_ctr[i] = idx + 1u;
}

// A world handle may only be used on the device it was created on (its tables live there).
int cobel_world_check(const cobel_world_t* w, const char* who) {
  COBEL_REQUIRE(w, COBEL_E_ARG, "%s: NULL world", who);
  int dev = -1;
  COBEL_HIP_TRY(hipGetDevice(&dev));
  COBEL_REQUIRE(dev == w->device, COBEL_E_ARG,
                "%s: the world lives on device %d, the current device is %d", who, w->device, dev);
  return COBEL_OK;
}
int cobel_world_check4(const cobel_world_t* w, const char* who) {
  if (int rc = cobel_world_check(w, who)) return rc;
  COBEL_REQUIRE(w->n_actions == 4, COBEL_E_UNSUPPORTED,
                "%s: the world has %d actions, this entry point serves four-action worlds", who,
                w->n_actions);
  COBEL_REQUIRE(!w->succ_off, COBEL_E_UNSUPPORTED,
                "%s: the world's transition rows are distributions (cobel_world_set_transitions); "
                "this entry point steps transition tables", who);
  return COBEL_OK;
}
static int check_states_dev(const cobel_world_t* w, const char* who) {
  return cobel_world_check(w, who);
}

extern "C" int cobel_env_step(const cobel_world_t* world, int32_t* state, const uint8_t* action,
                              float* reward_out, uint8_t* done_out, int32_t n,
                              uint32_t instance_base, void* stream) {
  if (int rc = check_states_dev(world, "cobel_env_step")) return rc;
  COBEL_REQUIRE(state && action, COBEL_E_ARG, "cobel_env_step: NULL state/action");
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_env_step: n = %d", n);
  COBEL_REQUIRE(!world->succ_off, COBEL_E_UNSUPPORTED,
                "cobel_env_step: the world's transition rows are distributions: "
                "cobel_env_step_draw steps it (it needs the env counters and the seed)");
  if (n == 0) return COBEL_OK;
  if (world->n_actions != 4)
    return cobel_env_step_general(world, state, action, reward_out, done_out, nullptr, 0, n,
                                  instance_base, (hipStream_t)stream);
  hipLaunchKernelGGL(k_env_step, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     world->rec, world->n_states, world->n_worlds, state, action, reward_out,
                     done_out, n, instance_base);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_env_step_draw(const cobel_world_t* world, int32_t* state, const uint8_t* action,
                                   float* reward_out, uint8_t* done_out, uint32_t* env_ctr,
                                   uint64_t seed, int32_t n, uint32_t instance_base, void* stream) {
  if (int rc = check_states_dev(world, "cobel_env_step_draw")) return rc;
  if (!world->succ_off)
    return cobel_env_step(world, state, action, reward_out, done_out, n, instance_base, stream);
  COBEL_REQUIRE(state && action && env_ctr, COBEL_E_ARG,
                "cobel_env_step_draw: NULL state / action / env_ctr");
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_env_step_draw: n = %d", n);
  if (n == 0) return COBEL_OK;
  return cobel_env_step_general(world, state, action, reward_out, done_out, env_ctr, seed, n,
                                instance_base, (hipStream_t)stream);
}

extern "C" int cobel_env_reset(const cobel_world_t* world, int32_t* state,
                               const uint8_t* reset_mask, uint32_t* env_ctr, uint64_t seed,
                               int32_t n, uint32_t instance_base, void* stream) {
  if (int rc = check_states_dev(world, "cobel_env_reset")) return rc;
  COBEL_REQUIRE(state && env_ctr, COBEL_E_ARG, "cobel_env_reset: NULL state/env_ctr");
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_env_reset: n = %d", n);
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_env_reset, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     world->starts, world->start_off, world->n_worlds, state, reset_mask, env_ctr,
                     seed, n, instance_base);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void k_eps_greedy(const V* __restrict__ values,
                                                    const uint8_t* __restrict__ mask,
                                                    const double* __restrict__ u,
                                                    cobel_eps_consts k,
                                                    uint8_t* __restrict__ action_out,
                                                    double* __restrict__ probs_out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V v0 = values[4 * i], v1 = values[4 * i + 1], v2 = values[4 * i + 2],
          v3 = values[4 * i + 3];
  const uint32_t m = mask ? (mask[i] & 15u) : 15u;
  double p[4];
  const int a = cobel_eps_greedy_select<V>(v0, v1, v2, v3, m, u[i], k, p);
  action_out[i] = (uint8_t)a;
  if (probs_out) {
    probs_out[4 * i + 0] = p[0];
    probs_out[4 * i + 1] = p[1];
    probs_out[4 * i + 2] = p[2];
    probs_out[4 * i + 3] = p[3];
  }
}

extern "C" int cobel_eps_greedy(const float* values, const uint8_t* mask, const double* u,
                                double epsilon, uint8_t* action_out, double* probs_out,
                                int32_t n, void* stream) {
  COBEL_REQUIRE(values && u && action_out, COBEL_E_ARG, "cobel_eps_greedy: NULL argument");
  COBEL_REQUIRE(epsilon >= 0.0 && epsilon <= 1.0, COBEL_E_ARG,
                "cobel_eps_greedy: epsilon %g outside [0, 1]", epsilon);
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_eps_greedy: n = %d", n);
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL((k_eps_greedy<float>), dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, values, mask, u, cobel_make_eps_consts(epsilon),
                     action_out, probs_out, n);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_eps_greedy_f64(const double* values, const uint8_t* mask, const double* u,
                                    double epsilon, uint8_t* action_out, double* probs_out,
                                    int32_t n, void* stream) {
  COBEL_REQUIRE(values && u && action_out, COBEL_E_ARG, "cobel_eps_greedy_f64: NULL argument");
  COBEL_REQUIRE(epsilon >= 0.0 && epsilon <= 1.0, COBEL_E_ARG,
                "cobel_eps_greedy_f64: epsilon %g outside [0, 1]", epsilon);
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_eps_greedy_f64: n = %d", n);
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL((k_eps_greedy<double>), dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, values, mask, u, cobel_make_eps_consts(epsilon),
                     action_out, probs_out, n);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rng_uniform(uint32_t* __restrict__ index, uint64_t seed,
                                                     uint32_t stream, uint32_t base,
                                                     double* __restrict__ out, int n,
                                                     int advance) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t idx = index[i];
  out[i] = cobel_draw_u01(idx, 0u, base + (uint32_t)i, stream, seed);
  if (advance) index[i] = idx + 1u;
}

__global__ __launch_bounds__(256) void k_rng_bounded(uint32_t* __restrict__ index, uint64_t seed,
                                                     uint32_t stream, uint32_t base,
                                                     uint32_t bound, int32_t* __restrict__ out,
                                                     int n, int per, int advance) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * per) return;
  const int i = t / per, j = t % per;
  const uint32_t idx = index[i];
  out[t] = (int32_t)cobel_draw_bounded(idx, (uint32_t)j, base + (uint32_t)i, stream, seed, bound);
}

__global__ __launch_bounds__(256) void k_rng_advance(uint32_t* __restrict__ index, int n,
                                                     const uint32_t* __restrict__ bounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (!bounds || bounds[i])) index[i] += 1u;
}

__global__ __launch_bounds__(256) void k_rng_bounded_each(uint32_t* __restrict__ index,
                                                          uint64_t seed, uint32_t stream,
                                                          uint32_t base,
                                                          const uint32_t* __restrict__ bounds,
                                                          int32_t* __restrict__ out, int n,
                                                          int per) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * per) return;
  const int i = t / per, j = t % per;
  const uint32_t b = bounds[i];
  out[t] = b ? (int32_t)cobel_draw_bounded(index[i], (uint32_t)j, base + (uint32_t)i, stream, seed, b)
             : 0;
}

extern "C" int cobel_rng_uniform(uint32_t* index, uint64_t seed, uint32_t stream,
                                 uint32_t instance_base, double* out, int32_t n, int32_t advance,
                                 void* sh) {
  COBEL_REQUIRE(index && out, COBEL_E_ARG, "cobel_rng_uniform: NULL argument");
  COBEL_REQUIRE(n >= 0, COBEL_E_RANGE, "cobel_rng_uniform: n = %d", n);
  if (n == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_rng_uniform, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)sh, index,
                     seed, stream, instance_base, out, n, advance);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_rng_bounded(uint32_t* index, uint64_t seed, uint32_t stream,
                                 uint32_t instance_base, uint32_t bound, int32_t* out, int32_t n,
                                 int32_t per_instance, int32_t advance, void* sh) {
  COBEL_REQUIRE(index && out, COBEL_E_ARG, "cobel_rng_bounded: NULL argument");
  COBEL_REQUIRE(n >= 0 && per_instance >= 0, COBEL_E_RANGE, "cobel_rng_bounded: bad sizes");
  COBEL_REQUIRE(bound > 0, COBEL_E_RANGE, "cobel_rng_bounded: bound must be positive");
  if (n == 0) return COBEL_OK;
  const long long total = (long long)n * per_instance;
  COBEL_REQUIRE(total < (1ll << 31), COBEL_E_RANGE, "cobel_rng_bounded: too many draws");
  if (total > 0)
    hipLaunchKernelGGL(k_rng_bounded, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)sh, index, seed, stream, instance_base, bound, out, n,
                       per_instance, advance);
  if (advance)
    hipLaunchKernelGGL(k_rng_advance, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)sh, index,
                       n, (const uint32_t*)nullptr);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_rng_bounded_each(uint32_t* index, uint64_t seed, uint32_t stream,
                                      uint32_t instance_base, const uint32_t* bounds,
                                      int32_t* out, int32_t n, int32_t per_instance,
                                      int32_t advance, void* sh) {
  COBEL_REQUIRE(index && out && bounds, COBEL_E_ARG, "cobel_rng_bounded_each: NULL argument");
  COBEL_REQUIRE(n >= 0 && per_instance >= 0, COBEL_E_RANGE, "cobel_rng_bounded_each: bad sizes");
  if (n == 0) return COBEL_OK;
  const long long total = (long long)n * per_instance;
  COBEL_REQUIRE(total < (1ll << 31), COBEL_E_RANGE, "cobel_rng_bounded_each: too many draws");
  if (total > 0)
    hipLaunchKernelGGL(k_rng_bounded_each, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)sh, index, seed, stream, instance_base, bounds, out, n,
                       per_instance);
  if (advance)
    hipLaunchKernelGGL(k_rng_advance, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)sh, index,
                       n, bounds);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ table,
                                                     const int32_t* __restrict__ index,
                                                     double* __restrict__ out, int n, int width,
                                                     int rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  const int i = t / width, c = t % width;
  const int r = index[i];
  out[t] = (r >= 0 && r < rows) ? table[(size_t)r * width + c] : __builtin_nan("");
}

extern "C" int cobel_gather_rows(const double* table, const int32_t* index, double* out,
                                 int32_t n, int32_t width, int32_t rows, void* stream) {
  COBEL_REQUIRE(table && index && out, COBEL_E_ARG, "cobel_gather_rows: NULL argument");
  COBEL_REQUIRE(n >= 0 && width > 0 && rows > 0, COBEL_E_RANGE, "cobel_gather_rows: bad sizes");
  const long long total = (long long)n * width;
  COBEL_REQUIRE(total < (1ll << 31), COBEL_E_RANGE, "cobel_gather_rows: too many elements");
  if (total == 0) return COBEL_OK;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, table, index, out, n, width, rows);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
