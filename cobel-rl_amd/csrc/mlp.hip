// One DQN replay step per instance as ONE kernel: forward of the target network on the sampled
// next states, forward of the online network on the sampled states, the Q-learning targets, the
// backward pass of the mean-squared error, torch.optim.Adam's update of the online network and the
// blend of the new weights into the target network — for networks of the shape every DQN demo and
// test of the reference uses: Linear(D, 64) - ReLU - Linear(64, 64) - ReLU - Linear(64, 4).
// This is the PARAMETER-STAGING form of the step (parameters of one network at a time in LDS): 39 KB
// of LDS per workgroup in float32, 79 KB in float64 at 6 inputs.  cobel_dqn_replay (mlp_fit.hip) uses
// it wherever at least two of its workgroups fit a CU and the form that streams its weight operands
// from memory (k_dqn_replay there, 52 KB whatever the inputs) otherwise; see DESIGN.md section 4.4.
//
// Replaces, per step and instance (paths relative to /root/reference/src/cobel):
//   agent/dqn.py:346-364      targets = Q_online(s); targets[a] = r + gamma * nt * max_a' Q_target(s')
//                             (DDQN, :352-355: the action is chosen by the online network)
//   network/network_torch.py:160-167  train_on_batch: MSELoss(reduction='none')(model(s), targets)
//                             .mean().backward(); optimizer.step()
//   agent/dqn.py:366-371      w_target += tau * (w_online - w_target)
// which through PyTorch is three batched forward passes, one backward pass and the optimizer: ~60
// GEMM / elementwise launches and ~25 passes over the stacked parameters (8 192 instances x 4 932
// float64 parameters = 323 MB per pass).  Here a workgroup of 256 threads owns one instance:
//   * the parameters of one network at a time are staged in LDS (first the target network's, then
//     the online network's into the same buffer), the 64 x 64 matrix transposed so that every
//     product below reads it along its contiguous axis;
//   * activations stay in LDS with a row stride of 66 elements (rows of different samples fall
//     into different banks); the backward pass overwrites them in place with the deltas;
//   * every thread applies Adam to the gradient elements it has just accumulated in registers —
//     gradients never exist in memory; p, m, v are read once and written once, the target network's
//     copy is read once (its forward pass; each thread keeps its tile of the 64 x 64 matrix for the
//     blend; the small tensors, a sixth of the parameters, are read a second time) and written once.
//   * requests go out in the order their answers are needed and nothing is waited for before
//     everything that does not depend on it has been asked: slots, both networks' parameters, then
//     both batches' rows / rewards / flags / actions; the optimizer state in the order the backward
//     pass consumes it, a phase or two ahead; tensor pointers are formed where they are used
//     (round 4: three dependent trips to memory per step where the first version made eleven).
// HBM traffic per instance and step: 8 streams over the parameters (online read + write, two
// moments read + write, target read + write) = 8 x 39 KB (float64).
//
// Arithmetic: the three 64 x 64 products are 16 x 16 x 4 MFMAs in the network's dtype (see
// mfma_acc below), the thin layers fused multiply-adds with one accumulator per output in index
// order (float32: the output layer's sums in two halves, the next observation's in four / sixteen
// parts) — not torch's GEMM order, so results agree with the PyTorch path to rounding (1e-10
// relative in float64 after ten steps; tests bound it), not bit for bit.  The optimizer update is
// k_adam's (adam.hip), operation for operation.
#include <stdlib.h>

#include "cobel_common.h"

namespace {

constexpr int kH = 64;        // hidden width (both layers)
constexpr int kA = 4;         // actions
constexpr int kB = 32;        // replay batch
constexpr int kRow = 66;      // LDS row stride of activations and of the transposed 64 x 64 matrix
constexpr int kMaxD = 32;     // input width limit
// First-layer elements (64 D) and batch-row elements (32 D) per thread of the 256: the kernels are
// instantiated for DI = 2, 4, 8 first-layer elements per thread (inputs up to 8, 16, 32) — every
// one of them is an unconditional load in flight at once, so the count follows the input width.
constexpr int kMaxW1Iters = kH * kMaxD / 256;
static_assert(kMaxW1Iters == 8, "three instantiations: 2, 4, 8");

struct mlp_args {
  cobel_dqn_replay_t r;
  unsigned long long* trace;   // experiments: [n][16] wall-clock stamps of thread 0 (or NULL)
};

template <typename T>
struct mlp_lds {
  T* wt2;   // [64][66]  wt2[k * 66 + j] = W2[j][k]
  T* wt1;   // [D][64]   wt1[d * 64 + j] = W1[j][d]
  T* w3;    // [4][64]
  T* b1;    // [64]
  T* b2;    // [64]
  T* b3;    // [4] (+ 4 pad)
  T* x;     // [32][D]
  T* h1;    // [32][66]
  T* h2;    // [32][66]
  T* q;     // [32][4]   online Q(s) -> delta3
  T* qt;    // [32][4]   target Q(s')
  T* boot;  // [32]
  int* pick;  // [32] DDQN: argmax_a Q_online(s')
  int* slot;  // [32] row of each sample inside the instance's batch / replay ring
  int* idx_s; // [32] world-model mode: observation-table rows of the sampled states ...
  int* idx_n; // [32] ... and of their successors
};

__host__ __device__ inline size_t mlp_lds_elems(int D) {
  return (size_t)kH * kRow + (size_t)D * kH + kA * kH + kH + kH + 8 + (size_t)kB * D +
         2 * (size_t)kB * kRow + 2 * kB * kA + kB + 4 * kB /* pick, slot, two index rows: as T-sized cells */;
}

// Threads of a workgroup talk through LDS only, so its barriers wait for the LDS counter, not for
// memory (__syncthreads() also drains vmcnt: the optimizer-state and parameter loads issued ahead
// of their use, and the parameter stores of the previous phase, are meant to stay in flight).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The three 64 x 64 products of a step (second-layer forward, its weight gradient, the delta of
// the first layer: 85 % of the arithmetic) run on the matrix cores as 16 x 16 x 4 MFMAs in the
// network's dtype.  Operand layout of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (probed on
// gfx950): lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; of the 16 x 16 result it holds
// column l % 16 and, in accumulator element v, row 4 v + l / 16 (float64) or 4 (l / 16) + v
// (float32).
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename T>
struct mfma_acc;
template <>
struct mfma_acc<double> {
  typedef v4d type;
  static __device__ __forceinline__ int row(int lane, int v) { return 4 * v + (lane >> 4); }
};
template <>
struct mfma_acc<float> {
  typedef v4f type;
  static __device__ __forceinline__ int row(int lane, int v) { return 4 * (lane >> 4) + v; }
};
__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ v4f mfma(float a, float b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ T fma_t(T a, T b, T c);
template <>
__device__ __forceinline__ double fma_t<double>(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}
template <>
__device__ __forceinline__ float fma_t<float>(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}

// torch.optim.Adam, one element (same operation order as k_adam in adam.hip)
template <typename T>
struct adam_consts {
  T bc2_sqrt, step_size, one_m_b1, b2, one_m_b2, eps, wd, tau;
  bool has_wd, blend;
};

// The optimizer state of an element (and the target network's copy of the parameter) is loaded
// at the start of the kernel, long before the gradient exists: by the time the update runs, the
// loads of all of a thread's elements have returned together instead of one after another.
template <typename T>
struct adam_slot {
  T m, v, target;
};

template <typename T>
__device__ __forceinline__ void adam_update(T p_old, T g, const adam_slot<T>& s,
                                            const adam_consts<T>& c, T& pn, T& mn, T& vn, T& tn) {
  if (c.has_wd) g = g + c.wd * p_old;
  mn = s.m + c.one_m_b1 * (g - s.m);
  vn = s.v * c.b2 + (c.one_m_b2 * g) * g;
  const T denom = sqrt(vn) / c.bc2_sqrt + c.eps;
  pn = p_old - c.step_size * (mn / denom);
  tn = s.target + c.tau * (pn - s.target);
}

template <typename T>
__device__ __forceinline__ T adam_apply(T* __restrict__ p, T* __restrict__ m,
                                        T* __restrict__ v, T* __restrict__ tgt, size_t e,
                                        T p_old, T g, const adam_slot<T>& s,
                                        const adam_consts<T>& c) {
  T pn, mn, vn, tn;
  adam_update<T>(p_old, g, s, c, pn, mn, vn, tn);
  m[e] = mn;
  v[e] = vn;
  p[e] = pn;
  if (c.blend) tgt[e] = tn;
  return pn;
}

// One network's parameters (torch.nn.Linear layout [out][in]) on their way into LDS: the 64 x 64
// matrix, the output layer and the biases are loaded into registers (the online network's while
// the target network's forward pass runs) and written to LDS when the buffer is free; the small
// first layer goes straight from memory to LDS at that point.
// Float32 at more than eight inputs keeps the LATE form of the small requests (first layer of a
// network, second batch of rows, first layer's optimizer state: each where it is used): the early
// form holds them in registers across the forward passes, and the 128 registers of the
// four-workgroups-per-CU build do not have that room (measured at 12 / 25 inputs: 0.41 / 0.52 ms per
// Dyna-DQN step late, 0.45 / 0.64 early with spills; at 6 inputs early wins, 0.323 -> 0.310).
template <typename T, int DI>
struct early_requests {
  static constexpr bool value = sizeof(T) == 8 || DI == 2;
};

template <typename T, int DI>
struct param_regs {
  T w2[16], w1[early_requests<T, DI>::value ? DI : 1], w3, b1, b2, b3;
};

template <typename T, int DI>
__device__ __forceinline__ void params_load(param_regs<T, DI>& P, const T* __restrict__ w1,
                                            const T* __restrict__ b1, const T* __restrict__ w2,
                                            const T* __restrict__ b2, const T* __restrict__ w3,
                                            const T* __restrict__ b3, int D, int t) {
#pragma unroll
  for (int u = 0; u < 16; ++u) P.w2[u] = __builtin_nontemporal_load(w2 + t + 256 * u);   // coalesced along k
  if (early_requests<T, DI>::value) {
#pragma unroll
    for (int u = 0; u < DI; ++u) {
      const int e = t + 256 * u;
      P.w1[u] = w1[e < kH * D ? e : kH * D - 1];   // (unconditional: a load under a condition is a
                                                   //  branch and a wait for every load before it)
    }
  }
  P.w3 = w3[t];   // 4 * 64 = 256 elements
  P.b1 = t < kH ? b1[t] : (T)0;
  P.b2 = t < kH ? b2[t] : (T)0;
  P.b3 = t < kA ? b3[t] : (T)0;
}

template <typename T, int DI>
__device__ __forceinline__ void params_store(const mlp_lds<T>& L, const param_regs<T, DI>& P,
                                             const T* __restrict__ w1, int D, int t) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {   // transposed write
    const int e = t + 256 * u;
    L.wt2[(e & 63) * kRow + (e >> 6)] = P.w2[u];
  }
  if (early_requests<T, DI>::value) {
#pragma unroll
    for (int u = 0; u < DI; ++u) {
      const int e = t + 256 * u;
      if (e < kH * D) {
        const int j = e / D, d = e - j * D;
        L.wt1[d * kH + j] = P.w1[u];
      }
    }
  } else {
    for (int e = t; e < kH * D; e += 256) {
      const int j = e / D, d = e - j * D;
      L.wt1[d * kH + j] = w1[e];
    }
  }
  L.w3[t] = P.w3;
  if (t < kH) {
    L.b1[t] = P.b1;
    L.b2[t] = P.b2;
  }
  if (t < kA) L.b3[t] = P.b3;
}

// h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2), out[s][a] = W3 h2 + b3 for the 32 rows of L.x.
// Thread tile of the two hidden layers: 2 samples x 4 neurons.
template <typename T>
__device__ void forward(const mlp_lds<T>& L, T* out, int D, int t) {
  const int jg = t & 15, sg = t >> 4;
  const int j0 = jg * 4, s0 = sg * 2;
  {
    T acc[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[0][c] = acc[1][c] = L.b1[j0 + c];
    for (int d = 0; d < D; ++d) {
      const T x0 = L.x[s0 * D + d], x1 = L.x[(s0 + 1) * D + d];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T w = L.wt1[d * kH + j0 + c];
        acc[0][c] = fma_t<T>(w, x0, acc[0][c]);
        acc[1][c] = fma_t<T>(w, x1, acc[1][c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      L.h1[s0 * kRow + j0 + c] = acc[0][c] > (T)0 ? acc[0][c] : (T)0;
      L.h1[(s0 + 1) * kRow + j0 + c] = acc[1][c] > (T)0 ? acc[1][c] : (T)0;
    }
  }
  lds_barrier();
  {
    // h2[s][j] = relu(b2[j] + sum_k h1[s][k] W2[j][k]): M = s (two tiles), N = j (wave w takes
    // columns 16 w ..), K = k.  A = h1 rows, B = wt2 (k-major: the transposed copy of W2).
    typedef typename mfma_acc<T>::type acc_t;
    const int lane = t & 63, jt = (t >> 6) * 16;
    const int li = lane & 15, lq = lane >> 4;
    const T bias = L.b2[jt + li];
    acc_t acc0 = {bias, bias, bias, bias}, acc1 = acc0;
#pragma unroll 4
    for (int k0 = 0; k0 < kH; k0 += 4) {
      const T b = L.wt2[(k0 + lq) * kRow + jt + li];
      const T a0 = L.h1[li * kRow + k0 + lq];
      const T a1 = L.h1[(16 + li) * kRow + k0 + lq];
      acc0 = mfma(a0, b, acc0);
      acc1 = mfma(a1, b, acc1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = mfma_acc<T>::row(lane, v);
      L.h2[r * kRow + jt + li] = acc0[v] > (T)0 ? acc0[v] : (T)0;
      L.h2[(16 + r) * kRow + jt + li] = acc1[v] > (T)0 ? acc1[v] : (T)0;
    }
  }
  lds_barrier();
  if (sizeof(T) == 4) {   // (float32: all 256 threads, two per output, half of the sum each — C5 -1 %;
                          //  float64 measured 3 % SLOWER with the split forms here and below and keeps
                          //  one thread per output)
    const int s = t >> 3, a = (t >> 1) & 3, half = t & 1;
    T acc = half ? (T)0 : L.b3[a];
#pragma unroll 8
    for (int k = 32 * half; k < 32 * half + 32; ++k)
      acc = fma_t<T>(L.w3[a * kH + k], L.h2[s * kRow + k], acc);
    acc = acc + __shfl_xor(acc, 1);
    if (!half) out[s * kA + a] = acc;
  } else if (t < kB * kA) {
    const int s = t >> 2, a = t & 3;
    T acc = L.b3[a];
#pragma unroll 8
    for (int k = 0; k < kH; ++k) acc = fma_t<T>(L.w3[a * kH + k], L.h2[s * kRow + k], acc);
    out[s * kA + a] = acc;
  }
  lds_barrier();
}

// The 32 rows of a batch (rows `slot[s]` of the instance's ring / gathered batch, or rows
// `index[s]` of the observation table in world-model mode): requested into registers, written to
// LDS when L.x is free — both batches of a step are requested together, right behind the
// parameters.
template <typename T, int DI>
struct row_regs {
  T r[DI / 2];
};

template <typename T, int DI>
__device__ __forceinline__ void rows_request(row_regs<T, DI>& X, const T* src, const int* slot,
                                             const double* table, const int32_t* index, int D,
                                             int t) {
  if (table) {
#pragma unroll
    for (int u = 0; u < (DI / 2); ++u) {
      const int e = t + 256 * u;
      const int ec = e < kB * D ? e : kB * D - 1;   // (unconditional loads, see params_load)
      const int s = ec / D, d = ec - s * D;
      X.r[u] = (T)table[(size_t)index[s] * D + d];
    }
  } else {
#pragma unroll
    for (int u = 0; u < (DI / 2); ++u) {
      const int e = t + 256 * u;
      const int ec = e < kB * D ? e : kB * D - 1;
      const int s = ec / D, d = ec - s * D;
      X.r[u] = src[(size_t)slot[s] * D + d];
    }
  }
}

template <typename T, int DI>
__device__ __forceinline__ void rows_store(T* dst, const row_regs<T, DI>& X, int D, int t) {
#pragma unroll
  for (int u = 0; u < (DI / 2); ++u) {
    const int e = t + 256 * u;
    if (e < kB * D) dst[e] = X.r[u];
  }
}

// The kernel arguments as they lie in the kernarg segment, through a pointer the optimizer cannot
// see through (constant address space: scalar loads).  The thirty tensor pointers of a step are
// formed where they are used — two scalar loads and a multiply-add each — instead of at the top
// of the kernel, from where they lived (and were spilled: ~240 v_readlane / v_writelane of ~2 600
// vector instructions per wave) across the whole step.
typedef const __attribute__((address_space(4))) mlp_args* mlp_kargs;
__device__ __forceinline__ mlp_kargs kr() {
#if defined(__HIP_DEVICE_COMPILE__)
  mlp_kargs p = (mlp_kargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
#else
  return nullptr;
#endif
}

template <typename T, int DI>
__device__ __forceinline__ void dqn_replay_body(const mlp_args& A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const cobel_dqn_replay_t& R = A.r;
  const int i = (int)blockIdx.x;
  if (R.active && !R.active[i]) return;
  const int t = (int)threadIdx.x;
  const int D = R.n_inputs;
  auto stamp = [&](int k) {   // (scripts/experiments/exp_mlp_trace.py)
    if (A.trace && t == 0) A.trace[(size_t)i * 16 + k] = wall_clock64();
  };
  stamp(0);
  mlp_lds<T> L;
  {
    T* p = reinterpret_cast<T*>(lds_raw);
    L.wt2 = p; p += kH * kRow;
    L.wt1 = p; p += D * kH;
    L.w3 = p;  p += kA * kH;
    L.b1 = p;  p += kH;
    L.b2 = p;  p += kH;
    L.b3 = p;  p += 8;
    L.x = p;   p += kB * D;
    L.h1 = p;  p += kB * kRow;
    L.h2 = p;  p += kB * kRow;
    L.q = p;   p += kB * kA;
    L.qt = p;  p += kB * kA;
    L.boot = p; p += kB;
    L.pick = reinterpret_cast<int*>(p); p += kB;
    L.slot = reinterpret_cast<int*>(p); p += kB;
    L.idx_s = reinterpret_cast<int*>(p); p += kB;
    L.idx_n = reinterpret_cast<int*>(p);
  }
  const size_t n1 = (size_t)kH * D, n2 = (size_t)kH * kH, n3 = (size_t)kA * kH;
  auto w1 = [&]() -> T* { return (T*)kr()->r.w[0] + (size_t)i * n1; };
  auto b1 = [&]() -> T* { return (T*)kr()->r.b[0] + (size_t)i * kH; };
  auto w2 = [&]() -> T* { return (T*)kr()->r.w[1] + (size_t)i * n2; };
  auto b2 = [&]() -> T* { return (T*)kr()->r.b[1] + (size_t)i * kH; };
  auto w3 = [&]() -> T* { return (T*)kr()->r.w[2] + (size_t)i * n3; };
  auto b3 = [&]() -> T* { return (T*)kr()->r.b[2] + (size_t)i * kA; };
  auto tw1 = [&]() -> T* { return (T*)kr()->r.w_target[0] + (size_t)i * n1; };
  auto tb1 = [&]() -> T* { return (T*)kr()->r.b_target[0] + (size_t)i * kH; };
  auto tw2 = [&]() -> T* { return (T*)kr()->r.w_target[1] + (size_t)i * n2; };
  auto tb2 = [&]() -> T* { return (T*)kr()->r.b_target[1] + (size_t)i * kH; };
  auto tw3 = [&]() -> T* { return (T*)kr()->r.w_target[2] + (size_t)i * n3; };
  auto tb3 = [&]() -> T* { return (T*)kr()->r.b_target[2] + (size_t)i * kA; };
  // the batch: gathered tensors [N][32][..], or rows batch_slots[i][s] of the replay rings
  const size_t rows = R.batch_slots ? (size_t)R.ring_slots : (size_t)kB;
  const T* const xs = (const T*)R.states + (size_t)i * rows * D;
  const T* const xn = (const T*)R.next_states + (size_t)i * rows * D;
  // Requests go out in the order their answers are needed, and nothing waits for an answer before
  // everything that does not depend on it has been requested: the batch's slots first, then both
  // networks' parameters, then (once the slots are there) both batches' rows, rewards, flags and
  // actions — one trip to memory where the first version of this kernel made five, each behind a
  // barrier.
  constexpr bool kEarly = early_requests<T, DI>::value;
  int my_slot = t, my_is = 0, my_in = 0;
  if (t < kB) {
    if (R.batch_slots) my_slot = R.batch_slots[(size_t)i * kB + t];
    if (R.state_index) {   // world-model mode: the batch's observations are rows of a table
      my_is = R.state_index[(size_t)i * kB + t];
      my_in = R.next_index[(size_t)i * kB + t];
    }
  }
  const int my_obs = R.q_out ? R.obs_index[i] : 0;
  auto m_w1 = [&]() -> T* { return (T*)kr()->r.m_w[0] + (size_t)i * n1; }; auto v_w1 = [&]() -> T* { return (T*)kr()->r.v_w[0] + (size_t)i * n1; };
  auto m_w2 = [&]() -> T* { return (T*)kr()->r.m_w[1] + (size_t)i * n2; }; auto v_w2 = [&]() -> T* { return (T*)kr()->r.v_w[1] + (size_t)i * n2; };
  auto m_w3 = [&]() -> T* { return (T*)kr()->r.m_w[2] + (size_t)i * n3; }; auto v_w3 = [&]() -> T* { return (T*)kr()->r.v_w[2] + (size_t)i * n3; };
  auto m_b1 = [&]() -> T* { return (T*)kr()->r.m_b[0] + (size_t)i * kH; }; auto v_b1 = [&]() -> T* { return (T*)kr()->r.v_b[0] + (size_t)i * kH; };
  auto m_b2 = [&]() -> T* { return (T*)kr()->r.m_b[1] + (size_t)i * kH; }; auto v_b2 = [&]() -> T* { return (T*)kr()->r.v_b[1] + (size_t)i * kH; };
  auto m_b3 = [&]() -> T* { return (T*)kr()->r.m_b[2] + (size_t)i * kA; }; auto v_b3 = [&]() -> T* { return (T*)kr()->r.v_b[2] + (size_t)i * kA; };

  // ---- the elements this thread will update (see adam_slot) -----------------------------------
  // second layer: the 4 x 4 tile (j0 .., k0 ..) of the backward pass; first layer: elements
  // t, t + 256, ...; output layer: element t; biases: threads < 64 / < 4.  The moments are loaded
  // at the start of the backward pass (requested before the forward passes they cost registers
  // there: measured slower, 0.80 -> 1.17 ms per C5 step in float64).
  // second layer: element (kt, v) of this thread is W2[j][k] with j = 16 (t / 64) + row(lane, v),
  // k = 16 kt + lane % 16 — the accumulator layout of the MFMA tiles of the backward pass
  const int lane2 = t & 63, jt2 = (t >> 6) * 16, li2 = lane2 & 15;
  adam_slot<T> s2[4][4], s3, sb1, sb2, sb3;
  // ---- Q_target(s') ---------------------------------------------------------------------------
  // Both networks' parameters are requested at once (the online network's stay in registers until
  // the target network's forward pass has released the LDS buffer): twice the bytes in flight while
  // the workgroup has nothing to compute.
  param_regs<T, DI> P, PT;
  params_load<T, DI>(PT, tw1(), tb1(), tw2(), tb2(), tw3(), tb3(), D, t);
  params_load<T, DI>(P, w1(), b1(), w2(), b2(), w3(), b3(), D, t);
  if (t < kB) {
    L.slot[t] = my_slot;
    L.idx_s[t] = my_is;
    L.idx_n[t] = my_in;
  }
  lds_barrier();
  stamp(1);
  row_regs<T, DI> XN, XS;
  rows_request<T, DI>(XN, xn, L.slot, R.state_index ? R.obs_table : nullptr, L.idx_n, D, t);
  if (kEarly) rows_request<T, DI>(XS, xs, L.slot, R.state_index ? R.obs_table : nullptr, L.idx_s, D, t);
  // (thread t < 32: reward and flag of sample t; thread t < 128: the action of sample t / 4)
  const size_t my_row = (size_t)i * rows + L.slot[t < kB ? t : 0];
  const T my_r = ((const T*)R.rewards)[my_row];
  const T my_nt = ((const T*)R.nonterminal)[my_row];
  const int my_act = (int)R.actions[(size_t)i * rows + L.slot[t < kB * kA ? t >> 2 : 0]];
  // Adam's bias corrections (two float64 pow() and a sqrt: ~300 float64 instructions) by ONE wave
  // while the parameter loads are in flight, through two spare LDS words behind b3() — every wave
  // used to evaluate them in front of the backward pass: 9 % of a C5 step in float32.
  if (t < 64) {
    const double st = R.steps[i];
    const T bc1 = (T)(1.0 - pow(R.beta1, st));
    const T bc2_sqrt = (T)sqrt(1.0 - pow(R.beta2, st));
    if (t == 0) {
      L.b3[4] = (T)R.lr / bc1;
      L.b3[5] = bc2_sqrt;
    }
  }
  params_store<T, DI>(L, PT, tw1(), D, t);
  rows_store<T, DI>(L.x, XN, D, t);
  lds_barrier();
  stamp(2);
  // the target network's copies of this thread's tile of the 64 x 64 matrix, for the blend at the
  // end (the other, small tensors are read again with their moments in the backward pass)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      s2[a][b].target = L.wt2[(16 * a + li2) * kRow + jt2 + mfma_acc<T>::row(lane2, b)];
  forward<T>(L, L.qt, D, t);
  stamp(3);
  // Optimizer state is requested in the order the backward pass consumes it, each group early
  // enough to have arrived: the output layer's and the small tensors' here (a dozen registers over
  // the online pass), the 64 x 64 tensor's after that pass (32 registers: requested any earlier
  // they crowd the forward passes out of the register file), the first layer's behind it.
  s3.m = m_w3()[t];
  s3.v = v_w3()[t];
  s3.target = tw3()[t];
  sb2.m = sb2.v = sb2.target = (T)0;
  sb3.m = sb3.v = sb3.target = (T)0;
  if (t < kH) {
    sb2.m = m_b2()[t]; sb2.v = v_b2()[t]; sb2.target = tb2()[t];
  }
  if (t < kA) {
    sb3.m = m_b3()[t]; sb3.v = v_b3()[t]; sb3.target = tb3()[t];
  }
  // ---- online network -------------------------------------------------------------------------
  params_store<T, DI>(L, P, w1(), D, t);
  if (!kEarly) rows_request<T, DI>(XS, xs, L.slot, R.state_index ? R.obs_table : nullptr, L.idx_s, D, t);
  if (!R.ddqn) rows_store<T, DI>(L.x, XS, D, t);   // (DDQN: the online pass on s' comes first)
  lds_barrier();
  if (R.ddqn) {   // agent/dqn.py:352-355: the online network picks the action, the target rates it
    forward<T>(L, L.q, D, t);
    if (t < kB) {
      int best = 0;
      T bv = L.q[t * kA];
#pragma unroll
      for (int a = 1; a < kA; ++a)
        if (L.q[t * kA + a] > bv) {   // first maximum, as torch.argmax
          bv = L.q[t * kA + a];
          best = a;
        }
      L.pick[t] = best;
    }
    rows_store<T, DI>(L.x, XS, D, t);
    lds_barrier();
  }
  forward<T>(L, L.q, D, t);
  stamp(4);
  T* const pm2 = m_w2();
  T* const pv2 = v_w2();
#pragma unroll
  for (int a = 0; a < 4; ++a)   // (16 lanes read 16 consecutive elements of a row)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const size_t e = (size_t)(jt2 + mfma_acc<T>::row(lane2, b)) * kH + 16 * a + li2;
      s2[a][b].m = __builtin_nontemporal_load(pm2 + e);
      s2[a][b].v = __builtin_nontemporal_load(pv2 + e);
    }

  // ---- targets and the loss gradient at the output ----------------------------------------------
  // new = r + (boot * nt) * gamma (the reference's operation order); loss = mean over the 32 x 4
  // outputs of (Q - targets)^2 with targets == Q except at the action taken, so the gradient is
  // 2 (Q[s][a] - new[s]) / 128 there and zero elsewhere.
  if (t < kB) {
    T boot;
    if (R.ddqn) {
      boot = L.qt[t * kA + L.pick[t]];
    } else {
      boot = L.qt[t * kA];
#pragma unroll
      for (int a = 1; a < kA; ++a) boot = L.qt[t * kA + a] > boot ? L.qt[t * kA + a] : boot;
    }
    L.boot[t] = my_r + (boot * my_nt) * (T)R.gamma;
  }
  lds_barrier();
  if (t < kB * kA) {
    const int s = t >> 2, a = t & 3;
    const int act = my_act;
    const T d = L.q[t] - L.boot[s];
    const T g = ((T)2 * d) * ((T)1 / (T)(kB * kA));
    L.q[t] = (a == act) ? g : (T)0;   // delta3
  }
  lds_barrier();
  stamp(5);

  // ---- Adam constants of this instance ----------------------------------------------------------
  adam_consts<T> c;
  {
    c.step_size = L.b3[4];   // (written at the start of the kernel, several barriers ago)
    c.bc2_sqrt = L.b3[5];
    c.one_m_b1 = (T)(1.0 - R.beta1);
    c.b2 = (T)R.beta2;
    c.one_m_b2 = (T)(1.0 - R.beta2);
    c.eps = (T)R.eps;
    c.wd = (T)R.weight_decay;
    c.has_wd = R.weight_decay != 0.0;
    c.tau = (T)R.tau;
    c.blend = R.tau != 0.0;
  }
  // ... and what the END of the step needs — the first layer's optimizer state and the instance's
  // next observation (they used to be requested where they are used: three more trips to memory,
  // one after the other, with the workgroup waiting)
  adam_slot<T> s1[DI];
  auto request_s1 = [&]() {
    const T* const pm = m_w1();
    const T* const pv = v_w1();
    const T* const pt = tw1();
#pragma unroll
    for (int u = 0; u < DI; ++u) {
      const int e = t + 256 * u;
      const int ec = e < kH * D ? e : kH * D - 1;   // (unconditional loads, see params_load)
      s1[u].m = pm[ec];
      s1[u].v = pv[ec];
      s1[u].target = pt[ec];
    }
  };
  if (kEarly) request_s1();
  sb1.m = sb1.v = sb1.target = (T)0;
  if (t < kH) {
    sb1.m = m_b1()[t]; sb1.v = v_b1()[t]; sb1.target = tb1()[t];
  }
  T my_x = (T)0;   // (no next observation asked for: obs_table may be absent)
  if (R.q_out) my_x = (T)R.obs_table[(size_t)my_obs * D + (t < D ? t : 0)];

  // The updated parameters also replace the old ones in LDS as soon as the backward pass no longer
  // needs those (the output layer's after delta2, the second layer's after delta1): the Q-values of
  // the next observation are computed from there at the end.
  T new_w3 = (T)0;
  // ---- output layer: dW3[a][k] = sum_s delta3[s][a] h2[s][k], db3[a] = sum_s delta3[s][a] --------
  {
    const int a = t >> 6, k = t & 63;
    T g = (T)0;
#pragma unroll 8
    for (int s = 0; s < kB; ++s) g = fma_t<T>(L.q[s * kA + a], L.h2[s * kRow + k], g);
    new_w3 = adam_apply<T>(w3(), m_w3(), v_w3(), tw3(), (size_t)t, L.w3[t], g, s3, c);
    if (t < kA) {
      T gb = (T)0;
      for (int s = 0; s < kB; ++s) gb = gb + L.q[s * kA + t];
      L.b3[t] = adam_apply<T>(b3(), m_b3(), v_b3(), tb3(), (size_t)t, L.b3[t], gb, sb3, c);
    }
  }
  lds_barrier();
  stamp(6);
  // delta2[s][k] = (sum_a W3[a][k] delta3[s][a]) * (h2[s][k] > 0), in place over h2
  for (int e = t; e < kB * kH; e += 256) {
    const int s = e >> 6, k = e & 63;
    T d = (T)0;
#pragma unroll
    for (int a = 0; a < kA; ++a) d = fma_t<T>(L.w3[a * kH + k], L.q[s * kA + a], d);
    const T h = L.h2[s * kRow + k];
    L.h2[s * kRow + k] = h > (T)0 ? d : (T)0;
  }
  lds_barrier();
  stamp(7);
  L.w3[t] = new_w3;

  // ---- second layer: dW2[j][k] = sum_s delta2[s][j] h1[s][k] ------------------------------------
  // M = j (wave w takes rows 16 w ..), N = k (four tiles), K = s.  A = delta2 (in h2), B = h1.
  T new_w2[4][4];   // the updated elements: into LDS once delta1 no longer needs the old weights
  {
    typedef typename mfma_acc<T>::type acc_t;
    const int lq = lane2 >> 4;
    acc_t g2[4];
    T* const qm2 = m_w2();
    T* const qv2 = v_w2();
    T* const qw2 = w2();
    T* const qt2 = tw2();
#pragma unroll
    for (int a = 0; a < 4; ++a) g2[a] = acc_t{(T)0, (T)0, (T)0, (T)0};
#pragma unroll 2
    for (int s0 = 0; s0 < kB; s0 += 4) {
      const T a = L.h2[(s0 + lq) * kRow + jt2 + li2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        g2[kt] = mfma(a, L.h1[(s0 + lq) * kRow + 16 * kt + li2], g2[kt]);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int j = jt2 + mfma_acc<T>::row(lane2, v), k = 16 * kt + li2;
        T pn, mn, vn, tn;
        adam_update<T>(L.wt2[k * kRow + j], g2[kt][v], s2[kt][v], c, pn, mn, vn, tn);
        new_w2[kt][v] = pn;
        const size_t e = (size_t)j * kH + k;
        __builtin_nontemporal_store(mn, qm2 + e);
        __builtin_nontemporal_store(vn, qv2 + e);
        __builtin_nontemporal_store(pn, qw2 + e);
        if (c.blend) __builtin_nontemporal_store(tn, qt2 + e);
      }
    if (t < kH) {
      T gb = (T)0;
      for (int s = 0; s < kB; ++s) gb = gb + L.h2[s * kRow + t];
      L.b2[t] = adam_apply<T>(b2(), m_b2(), v_b2(), tb2(), (size_t)t, L.b2[t], gb, sb2, c);
    }
  }
  lds_barrier();
  stamp(8);
  // delta1[s][k] = (sum_j delta2[s][j] W2[j][k]) * (h1[s][k] > 0), in place over h1, from the
  // weights this step started from (LDS still holds them: the update above went to memory only).
  // M = s (two tiles), N = k (wave w takes columns 16 w ..), K = j.  A = delta2 (in h2), B = W2
  // read from its transposed copy wt2[k][j] (rows 66 apart: the 16 lanes of a group hit 16 banks).
  {
    typedef typename mfma_acc<T>::type acc_t;
    const int lq = lane2 >> 4;
    acc_t d0 = {(T)0, (T)0, (T)0, (T)0}, d1 = d0;
#pragma unroll 4
    for (int j0 = 0; j0 < kH; j0 += 4) {
      const T b = L.wt2[(jt2 + li2) * kRow + j0 + lq];
      d0 = mfma(L.h2[li2 * kRow + j0 + lq], b, d0);
      d1 = mfma(L.h2[(16 + li2) * kRow + j0 + lq], b, d1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {   // (each cell of h1 is read and written by this lane only)
      const int r = mfma_acc<T>::row(lane2, v), k = jt2 + li2;
      const T h0 = L.h1[r * kRow + k], h1v = L.h1[(16 + r) * kRow + k];
      L.h1[r * kRow + k] = h0 > (T)0 ? d0[v] : (T)0;
      L.h1[(16 + r) * kRow + k] = h1v > (T)0 ? d1[v] : (T)0;
    }
  }
  lds_barrier();   // every read of the old second-layer weights is done
  stamp(9);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      L.wt2[(16 * kt + li2) * kRow + jt2 + mfma_acc<T>::row(lane2, v)] = new_w2[kt][v];

  // ---- first layer: dW1[j][d] = sum_s delta1[s][j] x[s][d] -----------------------------------------
  if (!kEarly) request_s1();
  T* const qw1 = w1();
  T* const qm1 = m_w1();
  T* const qv1 = v_w1();
  T* const qt1 = tw1();
#pragma unroll
  for (int u = 0; u < DI; ++u) {
    const int e = t + 256 * u;
    if (e < kH * D) {
      const int j = e / D, d = e - j * D;
      T g = (T)0;
#pragma unroll 8
      for (int s = 0; s < kB; ++s) g = fma_t<T>(L.h1[s * kRow + j], L.x[s * D + d], g);
      L.wt1[d * kH + j] =
          adam_apply<T>(qw1, qm1, qv1, qt1, (size_t)e, L.wt1[d * kH + j], g, s1[u], c);
    }
  }
  if (t < kH) {
    T gb = (T)0;
    for (int s = 0; s < kB; ++s) gb = gb + L.h1[s * kRow + t];
    const T nb = adam_apply<T>(b1(), m_b1(), v_b1(), tb1(), (size_t)t, L.b1[t], gb, sb1, c);
    L.b1[t] = nb;
  }

  stamp(10);
  // ---- Q-values of the next observation with the updated online network ------------------------
  // (what the next step's action selection needs: agent/dqn.py:174 -> retrieve_q)
  if (R.q_out) {
    lds_barrier();   // LDS holds the updated parameters; h1 / h2 / x are free
    if (t < D) L.x[t] = my_x;
    lds_barrier();
    if (t < kH) {
      T acc = L.b1[t];
      for (int d = 0; d < D; ++d) acc = fma_t<T>(L.wt1[d * kH + t], L.x[d], acc);
      L.h1[t] = acc > (T)0 ? acc : (T)0;
    }
    lds_barrier();
    if (sizeof(T) == 4) {
      // (one row through the 64 x 64 layer and the output layer: as chains of 64 dependent
      //  multiply-adds on one wave these two were 1.3 of the workgroup's 35 us; every wave takes a
      //  quarter of the 64 x 64 sum, sixteen lanes a sixteenth of an output's.  float32 only.)
      {
        const int j = t & 63, part = t >> 6;
        T acc = (T)0;
#pragma unroll
        for (int k = 16 * part; k < 16 * part + 16; ++k) acc = fma_t<T>(L.wt2[k * kRow + j], L.h1[k], acc);
        L.h2[(1 + part) * kRow + j] = acc;   // (rows 1 .. 4 of h2: free by now)
      }
      lds_barrier();
      if (t < kH) {
        T acc = L.b2[t];
#pragma unroll
        for (int part = 0; part < 4; ++part) acc = acc + L.h2[(1 + part) * kRow + t];
        L.h2[t] = acc > (T)0 ? acc : (T)0;
      }
      lds_barrier();
      if (t < 16 * kA) {
        const int a = t >> 4, part = t & 15;
        T acc = (T)0;
#pragma unroll
        for (int k = 4 * part; k < 4 * part + 4; ++k) acc = fma_t<T>(L.w3[a * kH + k], L.h2[k], acc);
        acc = acc + __shfl_xor(acc, 1);
        acc = acc + __shfl_xor(acc, 2);
        acc = acc + __shfl_xor(acc, 4);
        acc = acc + __shfl_xor(acc, 8);
        if (part == 0) ((T*)R.q_out)[(size_t)i * kA + a] = L.b3[a] + acc;
      }
    } else {
      if (t < kH) {
        T acc = L.b2[t];
#pragma unroll 8
        for (int k = 0; k < kH; ++k) acc = fma_t<T>(L.wt2[k * kRow + t], L.h1[k], acc);
        L.h2[t] = acc > (T)0 ? acc : (T)0;
      }
      lds_barrier();
      if (t < kA) {
        T acc = L.b3[t];
#pragma unroll 8
        for (int k = 0; k < kH; ++k) acc = fma_t<T>(L.w3[t * kH + k], L.h2[k], acc);
        ((T*)R.q_out)[(size_t)i * kA + t] = acc;
      }
    }
  }
  stamp(11);
}

// float64: 78 KB of LDS allow two workgroups per CU, so the kernel may use 256 registers.
// float32: 39 KB allow four, and four resident workgroups per CU with a few spilled registers
// (128-register cap) beat three without: C5 float32 0.69 -> 0.51 ms per step.
#ifndef COBEL_MLP_WAVES_F32
#define COBEL_MLP_WAVES_F32 4
#endif
template <int DI>
__global__ __launch_bounds__(256) void k_dqn_replay_lds_f64(const mlp_args A) {
  dqn_replay_body<double, DI>(A);
}
template <int DI>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(COBEL_MLP_WAVES_F32, COBEL_MLP_WAVES_F32)))
void k_dqn_replay_lds_f32(const mlp_args A) {
  dqn_replay_body<float, DI>(A);
}

template <int DI>
int launch_lds(const mlp_args& A, int32_t lds, hipStream_t st) {
  const cobel_dqn_replay_t& r = A.r;
  if (r.is_float64) {
    // (raised once per device and instantiation: the call is not free and this entry point runs
    //  every step; a race between two host threads at worst raises the limit twice)
    static int raised_to[64] = {0};
    int dev = 0;
    COBEL_HIP_TRY(hipGetDevice(&dev));
    if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || raised_to[dev] < lds)) {
      COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dqn_replay_lds_f64<DI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      if (dev >= 0 && dev < 64) raised_to[dev] = lds;
    }
    hipLaunchKernelGGL(k_dqn_replay_lds_f64<DI>, dim3(r.n), dim3(256), lds, st, A);
  } else {
    hipLaunchKernelGGL(k_dqn_replay_lds_f32<DI>, dim3(r.n), dim3(256), lds, st, A);
  }
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

}  // namespace

// LDS bytes of a workgroup of this kernel (inputs D, float64 or float32)
size_t cobel_dqn_replay_lds_bytes(int32_t n_inputs, int32_t is_float64) {
  return mlp_lds_elems(n_inputs) * (is_float64 ? 8 : 4);
}

// The launch (arguments checked by cobel_dqn_replay, mlp_fit.hip, which picks between this kernel
// and the one that streams its weight operands from memory).
int cobel_dqn_replay_lds_launch(const cobel_dqn_replay_t& r, hipStream_t st,
                                unsigned long long* trace) {
  const int32_t lds = (int32_t)cobel_dqn_replay_lds_bytes(r.n_inputs, r.is_float64);
  mlp_args A;
  A.r = r;
  A.trace = trace;
  if (r.n_inputs <= 8) return launch_lds<2>(A, lds, st);
  if (r.n_inputs <= 16) return launch_lds<4>(A, lds, st);
  return launch_lds<8>(A, lds, st);
}
