// DynaDSR.replay's regression targets (agent/dyna_q.py:1079-1131) for all agents in one launch:
// the dozen elementwise passes between the forward launches and the fits of a lockstep step
// (argmax over the actions' values, gather of the chosen successor features, bootstrap, targets,
// per-action sample masks) — 0.29 ms of torch kernels per step of 8 192 agents, one 30 us launch here.
// One workgroup of 256 threads per agent; arithmetic element for element in the order of the
// reference's expressions (see include/cobel_hip.h).
#include "cobel_common.h"

namespace {

constexpr int kB = 32;
constexpr int kMaxA = 8;

struct dsr_args {
  cobel_dsr_targets_t r;
};

template <typename T>
__global__ __launch_bounds__(256) void k_dsr_targets(const dsr_args G) {
  const cobel_dsr_targets_t& R = G.r;
  const int i = (int)blockIdx.x, t = (int)threadIdx.x;
  const int A = R.n_actions, O = R.n_outputs;
  __shared__ int best[kB];
  __shared__ int act[kB];
  const T* const val = (const T*)R.value + (size_t)i * A * kB;
  if (t < kB) {
    int b = 0;
    T bv = val[t];
    for (int a = 1; a < A; ++a) {
      const T v = val[a * kB + t];
      if (v > bv) {   // first maximum, as torch.argmax
        bv = v;
        b = a;
      }
    }
    best[t] = b;
    act[t] = (int)R.actions[(size_t)i * kB + t];
  }
  __syncthreads();
  // took / train: thread (a, s) for a < A
  if (t < A * kB) {
    const int a = t >> 5, s = t & 31;
    const bool mine = act[s] == a;
    R.took[((size_t)i * A + a) * kB + s] = mine ? 1 : 0;
    const unsigned long long any = __ballot(mine);
    // (a wave holds two actions' 32 samples: lanes 0..31 and 32..63)
    if (s == 0) R.train[(size_t)i * A + a] = ((any >> ((t & 32) ? 32 : 0)) & 0xffffffffull) ? 1 : 0;
  }
  const T* const fsr = (const T*)R.successor + (size_t)i * A * kB * O;
  const T* const ntp = (const T*)R.nonterminal + (size_t)i * kB;
  const int32_t* const si = R.state_index + (size_t)i * kB;
  const int32_t* const ni = R.next_index + (size_t)i * kB;
  T* const y = (T*)R.targets + (size_t)i * kB * O;
  const T follow = (T)(R.follow_up ? 1.0 : 0.0), ignore = (T)(R.ignore_terminality ? 1.0 : 0.0);
  const T c1 = (T)((1.0 - (double)follow) * (1.0 - (double)ignore));
  const T gamma = (T)R.gamma;
  for (int e = t; e < kB * O; e += 256) {
    const int s = e / O, o = e - s * O;
    const T nxt = (T)R.table[(size_t)ni[s] * O + o];
    const T base = R.follow_up ? nxt : (T)R.table[(size_t)si[s] * O + o];
    const T nt = ntp[s] != (T)0 ? (T)1 : (T)0;
    T boot_sr;
    if (R.use_dr) {
      T sum = fsr[(size_t)s * O + o];
      for (int a = 1; a < A; ++a) sum = sum + fsr[((size_t)a * kB + s) * O + o];
      boot_sr = sum / (T)A;
    } else {
      boot_sr = fsr[((size_t)best[s] * kB + s) * O + o];
    }
    T boot = (nxt * c1) * ((T)1 - nt);
    const T w = nt + ignore;
    boot = boot + boot_sr * (w > (T)1 ? (T)1 : w);
    y[e] = base + gamma * boot;
  }
}

}  // namespace

extern "C" int cobel_dsr_targets(const cobel_dsr_targets_t* run, void* stream) {
  COBEL_REQUIRE(run, COBEL_E_ARG, "cobel_dsr_targets: NULL run");
  const cobel_dsr_targets_t& r = *run;
  COBEL_REQUIRE(r.successor && r.value && r.table && r.state_index && r.next_index && r.actions &&
                    r.nonterminal && r.targets && r.took && r.train,
                COBEL_E_ARG, "cobel_dsr_targets: NULL tensor");
  COBEL_REQUIRE(r.n >= 0 && r.n_actions >= 1 && r.n_actions <= kMaxA && r.n_outputs >= 1, COBEL_E_RANGE,
                "cobel_dsr_targets: n %d, %d actions (at most %d), %d outputs", r.n, r.n_actions, kMaxA,
                r.n_outputs);
  if (r.n == 0) return COBEL_OK;
  dsr_args G;
  G.r = r;
  hipStream_t st = (hipStream_t)stream;
  if (r.is_float64) hipLaunchKernelGGL(k_dsr_targets<double>, dim3(r.n), dim3(256), 0, st, G);
  else hipLaunchKernelGGL(k_dsr_targets<float>, dim3(r.n), dim3(256), 0, st, G);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
