// TorchNetwork.train_on_batch / predict_on_batch for stacks of small networks, one kernel each:
// Linear(D <= 32, 64) - ReLU - Linear(64, 64) - ReLU - Linear(64, O <= 32) on batches of 32.
//
//   cobel_mlp_fit      one optimisation step per network towards GIVEN targets: forward, the
//                      gradient of MSELoss(reduction='none')(out, targets) averaged over the marked
//                      samples and the outputs (network/network_torch.py:160-167, with the
//                      sub-batch selection of agent/dyna_q.py:1079-1131 as a sample mask),
//                      torch.optim.Adam, the blend of the new weights into a target network
//                      (agent/dyna_q.py:1134-1143) — also for networks that do not train in this
//                      step — and the updated network's outputs for a few extra rows (what the
//                      next action selection needs);
//   cobel_mlp_forward  forward only, for 32 rows per instance; several instances may share one
//                      network (the reward network of Dyna-DSR rates the successor features of all
//                      four actions).
// Together they carry DynaDSR.replay (agent/dyna_q.py:1042-1150): nine networks per agent.
//
//   cobel_dqn_replay   the DQN replay step (agent/dqn.py:346-371) = the same optimisation step with
//                      the Q-learning targets r + gamma nt max_a' Q_target(s') [DDQN: the online
//                      network picks a'] at the action taken, i.e. two (three) more forward passes
//                      in front of it.
//
// The kernels take the weight operand of every product straight from memory in the MFMA's operand
// layout and keep only the activations in LDS (see "Layout" below): all products are 16 x 16 x 4
// MFMAs in the network's dtype, gradients only ever live in accumulator registers, Adam is applied
// by the lane that holds the element — not torch's GEMM order, so results agree with the PyTorch
// path to rounding (1e-10 relative in float64 after ten steps; tests bound it), not bit for bit.
#include <stdlib.h>
#include <string.h>

#include <cstdlib>

#include "cobel_common.h"

namespace {

constexpr int kH = 64;
constexpr int kB = 32;
constexpr int kRow = 66;
constexpr int kMaxD = 32;
constexpr int kMaxO = 32;
constexpr int kMaxEp = 4;
constexpr int kA = 4;              // actions of the DQN step's parameter-staging form (mlp.hip)
constexpr int kAMax = 8;           // action counts the streaming form of the DQN step serves
constexpr int kFitThreads = 512;   // threads of a training workgroup

struct fit_args {
  cobel_mlp_fit_t r;
  // cobel_dqn_replay: the regression targets are the Q-learning targets
  int32_t dqn, ddqn;
  int32_t rows;                    // rows per instance of the dense inputs (32, or the ring's slots)
  int32_t steps_given;             // the step counts already include this step (and stay as they are)
  const void* next_dense;          // [n][rows][D] next states ...
  const int32_t* next_index;       // ... or [n][32] rows of in_table
  const int32_t* slots;            // [n][32] row of each sample, or NULL (sample s = row s)
  const int64_t* actions;          // [n][rows]
  const void* rewards;             // [n][rows]
  const void* nonterminal;         // [n][rows]
  double gamma;
  unsigned long long* trace;       // experiments: [n][16] wall-clock stamps of thread 0 (or NULL)
  int32_t n_actions;               // DQN: outputs = actions of the step (1 .. kAMax)
};
struct fwd_args {
  cobel_mlp_forward_t r;
  int32_t group;   // consecutive instances per workgroup (they share a network)
};

// Threads of a workgroup talk through LDS only, so its barriers wait for the LDS counter, not for
// memory (__syncthreads() also drains vmcnt: the operand and optimizer-state loads issued ahead of
// their use, and the parameter stores of the previous phase, are meant to stay in flight).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Operand layout of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (probed on gfx950): lane l
// supplies A[l % 16][l / 16] and B[l / 16][l % 16]; of the 16 x 16 result it holds column l % 16
// and, in accumulator element v, row 4 v + l / 16 (float64) or 4 (l / 16) + v (float32).
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
__device__ __forceinline__ float uniform(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double uniform(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T>
struct adam_consts {
  T bc2_sqrt, step_size, one_m_b1, b2, one_m_b2, eps, wd, tau;
  bool has_wd, blend;
};

// torch.optim.Adam, one element (the operation order of k_adam in adam.hip).  The element's
// parameter, moments and — when blending — the target network's copy are requested together
// (slot_load), a matrix product later the update is applied and written back (adam_apply).
template <typename T>
struct adam_slot {
  T p, m, v, target;
};
template <typename T>
__device__ __forceinline__ adam_slot<T> slot_load(const T* __restrict__ p, const T* __restrict__ m,
                                                  const T* __restrict__ v,
                                                  const T* __restrict__ tgt, uint32_t e, bool blend,
                                                  bool valid, bool with_target = true) {
  adam_slot<T> s;
  s.p = s.m = s.v = s.target = (T)0;
  if (valid) {
    s.p = p[e];
    s.m = __builtin_nontemporal_load(m + e);
    s.v = __builtin_nontemporal_load(v + e);
    // (without a blend: tgt == p, unused)
    if (with_target) s.target = __builtin_nontemporal_load(tgt + e);
  }
  return s;
}

template <typename T>
__device__ __forceinline__ T adam_apply(T* __restrict__ p, T* __restrict__ m, T* __restrict__ v,
                                        T* __restrict__ tgt, uint32_t e, T g, const adam_slot<T>& s,
                                        const adam_consts<T>& c) {
  if (c.has_wd) g = g + c.wd * s.p;
  const T m0 = s.m, v0 = s.v;
  const T mn = m0 + c.one_m_b1 * (g - m0);
  const T vn = v0 * c.b2 + (c.one_m_b2 * g) * g;
  const T denom = sqrt(vn) / c.bc2_sqrt + c.eps;
  const T pn = s.p - c.step_size * (mn / denom);
  __builtin_nontemporal_store(mn, m + e);
  __builtin_nontemporal_store(vn, v + e);
  p[e] = pn;
  if (c.blend) __builtin_nontemporal_store(s.target + c.tau * (pn - s.target), tgt + e);
  return pn;
}

// ---------------------------------------------------------------------------------------------
// Layout.  A network's parameters are used by exactly 32 rows, i.e. by two 16-row MFMA tiles:
// nothing is gained by parking them in LDS first.  Every product takes its weight operand straight
// from memory in the MFMA's B layout (lane (li, lq) supplies B[k][n0 + li] for ITS quarter of the
// summation index, k = lq * KS + step — a product may sum in any order, so each lane reads KS
// consecutive elements of one row: whole cache lines for the 64-wide layers), and LDS holds only
// the activations: x [32][33], h1 / h2 [32][66], q [32][33] — 52 KB in float64, two training
// workgroups per CU (128 registers) where the parameter-staging layout (107 KB) had one.
//   forward   h1 = relu(x W1' + b1)     M = rows, N = 64,  K = D   B from W1 (memory)
//             h2 = relu(h1 W2' + b2)                        K = 64  B from W2
//             q  = h2 W3' + b3          N = O               K = 64  B from W3
//   backward  delta2 = (delta3 W3) . (h2 > 0)   K = O   B from W3 (the weights the step started from)
//             dW3 = delta3' h2                  K = 32  both operands from LDS
//             delta1 = (delta2 W2) . (h1 > 0)   K = 64  B from W2
//             dW2 = delta2' h1,  dW1 = delta1' x
// A gradient tile stays in the accumulator registers of the wave that summed it; that wave applies
// Adam to its 4 elements per lane and tile (consecutive lanes = consecutive addresses).  Operands and
// optimizer state are requested a phase ahead of their use (registers permitting).
constexpr int kXRow = 33;

template <typename T>
struct act_lds {
  T* x;     // [32][33]  inputs, columns >= D zero
  T* h1;    // [32][66]  -> delta1
  T* h2;    // [32][66]  -> delta2
  T* q;     // [32][33]  outputs -> delta3, columns >= O zero
  T* qt;    // [32][NA]  DQN: Q_target(s'), NA <= 8 actions
  int* aux; // [64]      DQN: row of each sample; the action taken in it
  T* cst;   // [2]       Adam's bias corrections of this step (computed by one wave)
  T* tg;    // [32]      DQN: the Q-learning target of each sample
};

__host__ __device__ inline size_t fit_lds_elems() {
  return 2 * (size_t)kB * kXRow + 2 * (size_t)kB * kRow + kB * kAMax + 64 + 2 + kB;
}
// forward only: the inputs sit where h2 will be written (read for the last time before that)
__host__ __device__ inline size_t fwd_lds_elems() { return 2 * (size_t)kB * kRow; }

template <typename T>
__device__ __forceinline__ act_lds<T> carve_fit(unsigned char* raw) {
  act_lds<T> L;
  T* p = reinterpret_cast<T*>(raw);
  L.h1 = p; p += kB * kRow;
  L.h2 = p; p += kB * kRow;
  L.x = p;  p += kB * kXRow;
  L.q = p;  p += kB * kXRow;
  L.qt = p; p += kB * kAMax;
  L.aux = reinterpret_cast<int*>(p);
  L.cst = p + 64;
  L.tg = p + 64 + 2;
  return L;
}

// the 32 input rows of an instance — rows of a float64 table by index, or rows slot[s] (NULL: s) of
// a dense block — on their way into x [32][33], columns D .. 31 zero: `request` asks for all of a
// thread's elements, `store` writes them to LDS (as late as the caller likes)
// BATCH: every load unconditional — column D - 1 again beyond D, zeroed afterwards.  A load under a
// condition is a branch, and the compiler then sends a thread's loads out one at a time, each behind
// a wait for the one before (four dependent trips for the four rows of a forward thread).  The
// forward kernel has the registers for it; the training kernels, at their 128-register cap, spill
// prefetched values when everything is in flight at once (measured slower) and keep the plain form.
template <typename T, int NT, bool BATCH = false>
struct input_rows {
  static constexpr int U = kB * 32 / NT;
  T r[U];
  int row_ahead[U];
  bool ahead = false;
  // (the row numbers of a table-fed batch, requested before anything else of the step: what they
  //  gate — the gather of the rows — is then one trip to memory behind the kernel's first one)
  __device__ __forceinline__ void request_index(const int32_t* index, int t) {
#pragma unroll
    for (int u = 0; u < U; ++u) row_ahead[u] = index[(t + NT * u) >> 5];
    ahead = true;
  }
  __device__ __forceinline__ void request(const double* table, const int32_t* index, const T* dense,
                                          const int* slot, int D, int t) {
    if (BATCH) {
      if (table) {
        int row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) row[u] = index[(t + NT * u) >> 5];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int d = (t + NT * u) & 31;
          r[u] = (T)table[(size_t)row[u] * D + (d < D ? d : D - 1)];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = t + NT * u, d = e & 31;
          const int s = slot ? slot[e >> 5] : e >> 5;
          r[u] = dense[(size_t)s * D + (d < D ? d : D - 1)];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = ((t + NT * u) & 31) < D ? r[u] : (T)0;
      return;
    }
    if (table) {
      int row[U];
#pragma unroll
      for (int u = 0; u < U; ++u) row[u] = ahead ? row_ahead[u] : index[(t + NT * u) >> 5];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int d = (t + NT * u) & 31;
        r[u] = d < D ? (T)table[(size_t)row[u] * D + d] : (T)0;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = t + NT * u, d = e & 31;
        const int s = slot ? slot[e >> 5] : e >> 5;
        r[u] = d < D ? dense[(size_t)s * D + d] : (T)0;
      }
    }
  }
  __device__ __forceinline__ void store(T* x, int t) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = t + NT * u;
      x[(e >> 5) * kXRow + (e & 31)] = r[u];
    }
  }
};
template <typename T, int NT, bool BATCH = false>
__device__ __forceinline__ void load_inputs(T* x, const double* table, const int32_t* index,
                                            const T* dense, const int* slot, int D, int t) {
  input_rows<T, NT, BATCH> in;
  in.request(table, index, dense, slot, D, t);
  in.store(x, t);
}

template <typename T>
__device__ __forceinline__ typename mfma_acc<T>::type splat(T v) {
  typename mfma_acc<T>::type a = {v, v, v, v};
  return a;
}

// The weight operand of a forward product for the 16 outputs n0 .. of a layer, requested from
// memory: lane (li, lq) takes W[n0 + li][lq KS .. lq KS + KS) (k >= kmax, n >= nmax: zero), and
// the output's bias.  Requested at the START of a pass for all three layers — a layer's operand is
// then in registers when the layer before it has finished, instead of one more trip to memory per
// layer.  ALIGNED: rows of 64 elements, 16-byte aligned (whole vector loads).
template <typename T, int KS>
struct weight_op {
  T b[KS];
  T bias;
};
template <typename T, int KS, bool ALIGNED, bool BATCH = false>
__device__ __forceinline__ weight_op<T, KS> weight_rows(const T* __restrict__ W,
                                                        const T* __restrict__ bias, int ld, int kmax,
                                                        int nmax, int n0, int lane) {
  weight_op<T, KS> w;
  const int li = lane & 15, lq = lane >> 4;
  const bool valid = n0 + li < nmax;
  if (BATCH) {   // unconditional loads of clamped addresses, zeroed afterwards (see input_rows)
    const int row = valid ? n0 + li : 0;
    if (ALIGNED) {
      const T* const wr = reinterpret_cast<const T*>(
          __builtin_assume_aligned(W + (uint32_t)(row * ld + lq * KS), 16));
#pragma unroll
      for (int s = 0; s < KS; ++s) w.b[s] = wr[s];
    } else {
      const T* const wr = W + (uint32_t)(row * ld);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = lq * KS + s;
        w.b[s] = wr[k < kmax ? k : kmax - 1];
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) w.b[s] = valid && (ALIGNED || lq * KS + s < kmax) ? w.b[s] : (T)0;
    w.bias = bias[row];
    w.bias = valid ? w.bias : (T)0;
    return w;
  }
  const T* wr = W + (uint32_t)((valid ? n0 + li : 0) * ld + lq * KS);
  if (ALIGNED) wr = reinterpret_cast<const T*>(__builtin_assume_aligned(wr, 16));
#pragma unroll
  for (int s = 0; s < KS; ++s) w.b[s] = valid && (ALIGNED || lq * KS + s < kmax) ? wr[s] : (T)0;
  w.bias = valid ? bias[n0 + li] : (T)0;
  return w;
}

// One hidden layer for MT row tiles: out[m][n0 + li] = relu(bias + sum_k in[m][k] W[n0 + li][k]),
// K = 4 KS, `in` with row stride IS.
template <typename T, int KS, int MT>
__device__ __forceinline__ void dense_relu(const weight_op<T, KS>& w, const T* in, int IS, T* out,
                                           int m0, int n0, int lane) {
  typedef typename mfma_acc<T>::type acc_t;
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    acc_t acc = splat<T>(w.bias);
    const T* const ar = in + (m0 + 16 * i + li) * IS + lq * KS;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma(ar[s], w.b[s], acc);
#pragma unroll
    for (int v = 0; v < 4; ++v)
      out[(m0 + 16 * i + mfma_acc<T>::row(lane, v)) * kRow + n0 + li] =
          acc[v] > (T)0 ? acc[v] : (T)0;
  }
}

// The output layer's tile (rows m0 .., outputs n0 ..): q[m][n0 + li] = b3 + sum_k h2[m][k] W3[n][k]
template <typename T>
__device__ __forceinline__ typename mfma_acc<T>::type output_tile(const weight_op<T, 16>& w,
                                                                  const T* h2, int m0, int lane) {
  typedef typename mfma_acc<T>::type acc_t;
  const int li = lane & 15, lq = lane >> 4;
  acc_t acc = splat<T>(w.bias);
  const T* const ar = h2 + (m0 + li) * kRow + lq * 16;
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = mfma(ar[s], w.b[s], acc);
  return acc;
}

// ---------------------------------------------------------------------------------------------
// Forward only: four waves per instance; wave w owns the 16 neurons n0 = 16 w of a hidden layer
// for both row tiles (its weights are loaded once), and one of the (up to) four output tiles.
template <typename T, bool GROUPED>
__device__ __forceinline__ void mlp_forward_body(const fwd_args& A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const cobel_mlp_forward_t& R = A.r;
  const int t = (int)threadIdx.x;
  const int D = R.n_inputs, O = R.n_outputs;
  // A workgroup serves `group` consecutive instances that share ONE network (group divides net_div:
  // the reward network of Dyna-DSR rates the successor features of an agent's four actions): the
  // weight operands are requested once and stay in registers for all of them.
  // (GROUPED is a separate instantiation: holding all three operands across the loop costs the
  //  one-instance form its fourth workgroup per CU)
  const int G = GROUPED ? A.group : 1;
  const int j0 = (int)blockIdx.x * G;
  T* const h1 = reinterpret_cast<T*>(lds_raw);   // [32][66]
  T* const h2 = h1 + kB * kRow;                  // [32][66]; before that the inputs x [32][33]
  T* const x = h2;
  const int lane = t & 63, wave = t >> 6;
  const size_t net = (size_t)(j0 / R.net_div);
  const T* const w1 = (const T*)R.w[0] + net * (size_t)kH * D;
  const T* const w2 = (const T*)R.w[1] + net * (size_t)kH * kH;
  const T* const w3 = (const T*)R.w[2] + net * (size_t)O * kH;
  const int m3 = 16 * (wave >> 1), n3 = 16 * (wave & 1);
  const weight_op<T, 8> o1 =
      weight_rows<T, 8, false, true>(w1, (const T*)R.b[0] + net * kH, D, D, kH, 16 * wave, lane);
  const weight_op<T, 16> o2 =
      weight_rows<T, 16, true, true>(w2, (const T*)R.b[1] + net * kH, kH, kH, kH, 16 * wave, lane);
  weight_op<T, 16> o3;
  bool have_o3 = false;
  input_rows<T, 256, true> xin;
  auto request_inputs = [&](int j) {
    xin.request(R.in_table, R.in_table ? R.in_index + (size_t)(j / R.in_div) * kB : nullptr,
                R.in_table ? nullptr : (const T*)R.in_dense + (size_t)j * kB * D, nullptr, D, t);
  };
  request_inputs(j0);
  for (int g = 0; g < G; ++g) {
    const int j = j0 + g;
    // (uniform: the whole workgroup skips an instance that sits the step out; its inputs were
    //  requested all the same — addresses of rows that exist)
    const bool skip = R.active && !R.active[j / R.act_div];
    if (!skip) xin.store(x, t);
    if (g + 1 < G) request_inputs(j + 1);   // (the next instance's rows, behind this one's layers)
    if (skip) continue;
    lds_barrier();
    dense_relu<T, 8, 2>(o1, x, kXRow, h1, 0, 16 * wave, lane);
    // (the output layer's operand once the first layer has been through: registers)
    if (!have_o3) {
      __builtin_amdgcn_sched_barrier(0);
      o3 = weight_rows<T, 16, true, true>(w3, (const T*)R.b[2] + net * O, kH, kH, O, n3, lane);
      have_o3 = true;
    }
    lds_barrier();   // (every read of x is done: h2 takes its place)
    dense_relu<T, 16, 2>(o2, h1, kRow, h2, 0, 16 * wave, lane);
    lds_barrier();
    if (n3 < O) {
      const typename mfma_acc<T>::type acc = output_tile<T>(o3, h2, m3, lane);
      T* const out = (T*)R.out + (size_t)j * kB * O;
      const int a = n3 + (lane & 15);
      if (a < O) {
#pragma unroll
        for (int v = 0; v < 4; ++v) out[(m3 + mfma_acc<T>::row(lane, v)) * O + a] = acc[v];
      }
    }
    if (g + 1 < G) lds_barrier();   // (the next instance's inputs go where h2 is being read)
  }
}

template <typename T, bool GROUPED>
__global__ __launch_bounds__(256) void k_mlp_forward(const fwd_args A) {
  mlp_forward_body<T, GROUPED>(A);
}

// ---------------------------------------------------------------------------------------------
// A forward pass of the eight-wave workgroup over the rows in L.x (all 32, or — ONE_TILE — the
// first 16 by waves 0 .. 3), in two parts: `request` asks memory for the weight operands (ahead of
// whatever fills L.x), `run` returns the output tile of the waves that own one (waves 0 .. 3: rows
// 16 (w / 2) .., outputs 16 (w % 2) ..) and starts with a barrier (the rows are in LDS).
template <typename T, bool ONE_TILE>
struct forward_pass {
  weight_op<T, 8> o1;
  weight_op<T, 16> o2, o3;
  bool hidden, has_out;
  int m0, n0, r3;
  __device__ __forceinline__ void request(const T* w1, const T* b1, const T* w2, const T* b2,
                                          const T* w3, const T* b3, int D, int O, int wave,
                                          int lane) {
    m0 = 16 * (wave >> 2);
    n0 = 16 * (wave & 3);
    r3 = 16 * ((wave >> 1) & 1);
    const int a3 = 16 * (wave & 1);
    hidden = !ONE_TILE || wave < 4;
    has_out = wave < 4 && a3 < O && (!ONE_TILE || r3 == 0);
    if (hidden) {
      o1 = weight_rows<T, 8, false>(w1, b1, D, D, kH, n0, lane);
      o2 = weight_rows<T, 16, true>(w2, b2, kH, kH, kH, n0, lane);
    }
    o3 = weight_rows<T, 16, true>(w3, b3, kH, kH, has_out ? O : 0, a3, lane);
  }
  __device__ __forceinline__ typename mfma_acc<T>::type run(const act_lds<T>& L, int lane) {
    lds_barrier();
    if (hidden) dense_relu<T, 8, 1>(o1, L.x, kXRow, L.h1, m0, n0, lane);
    lds_barrier();
    if (hidden) dense_relu<T, 16, 1>(o2, L.h1, kRow, L.h2, m0, n0, lane);
    lds_barrier();
    typename mfma_acc<T>::type acc = splat<T>((T)0);
    if (has_out) acc = output_tile<T>(o3, L.h2, r3, lane);
    return acc;
  }
};

// One optimisation step: eight waves per network.  Wave w owns the tile (rows 16 (w / 4) ..,
// columns 16 (w % 4) ..) of every rows x 64 product and one or two tiles of every gradient.
// The kernel arguments as they lie in the kernarg segment, through a pointer the optimizer cannot
// see through (constant address space: scalar loads): the step's tensor pointers are formed again
// at the start of the phase that uses them instead of living — spilled — in scalar registers from
// the top of the kernel (as in mlp.hip).
typedef const __attribute__((address_space(4))) fit_args* fit_kargs;
__device__ __forceinline__ fit_kargs fit_kr() {
#if defined(__HIP_DEVICE_COMPILE__)
  fit_kargs p = (fit_kargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
#else
  return nullptr;
#endif
}

// NAC: the DQN step's action count when it is known at compile time (4: every gridworld / linear
// track; as a run-time value the loops over the actions cost the four-action step 13 %), 0 = A.n_actions.
template <typename T, bool DQN, int NAC = 0>
__device__ __forceinline__ void mlp_fit_body(const fit_args& A) {
  constexpr int NT = kFitThreads;
  typedef typename mfma_acc<T>::type acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const cobel_mlp_fit_t& R = A.r;
  const int j = (int)blockIdx.x, t = (int)threadIdx.x;
  // The first trip to memory carries everything that depends on nothing: the two flags that decide
  // whether this network does anything in this step, and the row numbers of its batch.  (They used
  // to be three trips in a row — flag, flag, row numbers behind the weight requests — before the
  // gather of the rows could start: 15.6 of a workgroup's 53.5 us, exp_mlp_trace.py dsr.)
  input_rows<T, kFitThreads, DQN> xin;   // (DQN: batched requests — its prologue has four row sets to fetch)
  if (!DQN && R.in_table) xin.request_index(R.in_index + (size_t)(j / R.in_div) * kB, t);
  const bool is_active = !R.active || R.active[j / R.act_div] != 0;
  const bool is_training = DQN || !R.train || R.train[j] != 0;
  if (!is_active) return;
  const int NA = DQN ? (NAC ? NAC : A.n_actions) : 0;   // (DQN: 1 .. kAMax actions)
  const int D = R.n_inputs, O = DQN ? NA : R.n_outputs;
  const act_lds<T> L = carve_fit<T>(lds_raw);
  const int lane = t & 63, wave = t >> 6, li = lane & 15, lq = lane >> 4;
  const int m0 = 16 * (wave >> 2), n0 = 16 * (wave & 3);
  auto stamp = [&](int k) {
    if (A.trace && t == 0) A.trace[(size_t)j * 16 + k] = wall_clock64();
  };
  stamp(0);
  const bool train = is_training;
  const size_t n1 = (size_t)kH * D, n2 = (size_t)kH * kH, n3 = (size_t)O * kH;
  const bool has_target = R.w_target[0] != nullptr;
  const bool blend = R.tau != 0.0 && has_target;
  // (pointer makers: each call reads the kernarg segment again; without a target network the
  //  target pointers alias the parameters, so that loads stay unconditional)
  auto mk_w = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.w[l] + (size_t)j * n; };
  auto mk_b = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.b[l] + (size_t)j * n; };
  auto mk_tw = [&](int l, size_t n) -> T* {
    return has_target ? (T*)fit_kr()->r.w_target[l] + (size_t)j * n : mk_w(l, n);
  };
  auto mk_tb = [&](int l, size_t n) -> T* {
    return has_target ? (T*)fit_kr()->r.b_target[l] + (size_t)j * n : mk_b(l, n);
  };
  auto mk_mw = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.m_w[l] + (size_t)j * n; };
  auto mk_vw = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.v_w[l] + (size_t)j * n; };
  auto mk_mb = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.m_b[l] + (size_t)j * n; };
  auto mk_vb = [&](int l, size_t n) -> T* { return (T*)fit_kr()->r.v_b[l] + (size_t)j * n; };
  // (not const: re-formed at the start of every phase that uses them — REFRESH below — so that
  //  none is live across the phases in between)
  T* w1 = mk_w(0, n1);
  T* b1 = mk_b(0, kH);
  T* w2 = mk_w(1, n2);
  T* b2 = mk_b(1, kH);
  T* w3 = mk_w(2, n3);
  T* b3 = mk_b(2, (size_t)O);
  T* tw1 = mk_tw(0, n1);
  T* tb1 = mk_tb(0, kH);
  T* tw2 = mk_tw(1, n2);
  T* tb2 = mk_tb(1, kH);
  T* tw3 = mk_tw(2, n3);
  T* tb3 = mk_tb(2, (size_t)O);

  // (the extra row's number: a scalar, fetched long before the row is)
  const int ep_row = R.ep_out && R.ep_rows > 0 && R.ep_table ? R.ep_index[j / R.ep_div] : 0;

  if (!train) {
    // No samples for this network in this step: parameters and optimizer state stay as they are;
    // the target network still moves towards it (agent/dyna_q.py:1134-1143 blends every action's
    // pair every step).
    if (blend) {
      const T tau = (T)R.tau;
      for (size_t e = t; e < n1; e += NT) tw1[e] = tw1[e] + tau * (w1[e] - tw1[e]);
      for (size_t e = t; e < n2; e += NT) tw2[e] = tw2[e] + tau * (w2[e] - tw2[e]);
      for (size_t e = t; e < n3; e += NT) tw3[e] = tw3[e] + tau * (w3[e] - tw3[e]);
      if (t < kH) {
        tb1[t] = tb1[t] + tau * (b1[t] - tb1[t]);
        tb2[t] = tb2[t] + tau * (b2[t] - tb2[t]);
      }
      if (t < O) tb3[t] = tb3[t] + tau * (b3[t] - tb3[t]);
    }
  } else {
    T *m_w1, *v_w1, *m_w2, *v_w2, *m_w3, *v_w3, *m_b1, *v_b1, *m_b2, *v_b2, *m_b3, *v_b3;
    // waves 0 .. 3 also own an output tile: rows r3 .., outputs a3 ..
    const int r3 = 16 * ((wave >> 1) & 1), a3 = 16 * (wave & 1);
    const bool has_out = wave < 4 && a3 < O;
    const int rows = DQN ? A.rows : kB;
    const int* const slot = DQN ? L.aux : nullptr;
    const int32_t* const in_index =
        R.in_table ? R.in_index + (size_t)(j / R.in_div) * kB : nullptr;
    const T* const in_dense =
        R.in_table ? nullptr : (const T*)R.in_dense + (size_t)j * rows * D;
    for (int e = t; e < kB * kXRow; e += NT) L.q[e] = (T)0;
    int my_action = 0, my_best = 0;
    T my_reward = (T)0, my_nt = (T)0;

    if (DQN) {
      // ---- Q_target(s') [and the online network's choice among the next actions] ---------------
      forward_pass<T, false> tp;
      tp.request(tw1, tb1, tw2, tb2, tw3, tb3, D, O, wave, lane);
      if (t < kB) {
        const int sl = A.slots ? A.slots[(size_t)j * kB + t] : t;
        L.aux[t] = sl;
        // (the sample's action, reward and terminal flag, requested by the thread that will form
        //  its target: three trips to memory that now start with the kernel's first ones)
        const size_t row = (size_t)j * rows + sl;
        my_action = (int)A.actions[row];
        my_reward = ((const T*)A.rewards)[row];
        my_nt = ((const T*)A.nonterminal)[row];
      }
      lds_barrier();
      load_inputs<T, NT, true>(L.x, R.in_table, R.in_table ? A.next_index + (size_t)j * kB : nullptr,
                         R.in_table ? nullptr : (const T*)A.next_dense + (size_t)j * rows * D, slot,
                         D, t);
      xin.request(R.in_table, in_index, in_dense, slot, D, t);   // (for the pass after this one)
      {
        const acc_t acc = tp.run(L, lane);
        if (has_out && li < NA) {
#pragma unroll
          for (int v = 0; v < 4; ++v) L.qt[(r3 + mfma_acc<T>::row(lane, v)) * NA + li] = acc[v];
        }
      }
      stamp(1);
      if (R.debug_stage == 5) return;
      if (A.ddqn) {   // agent/dqn.py:352-355: the online network picks the action, the target rates it
        forward_pass<T, false> op;
        op.request(w1, b1, w2, b2, w3, b3, D, O, wave, lane);
        const acc_t acc = op.run(L, lane);
        if (has_out && li < NA) {
#pragma unroll
          for (int v = 0; v < 4; ++v) L.q[(r3 + mfma_acc<T>::row(lane, v)) * kXRow + li] = acc[v];
        }
        lds_barrier();
        if (t < kB) {
          int best = 0;
          T bv = L.q[t * kXRow];
          for (int a = 1; a < NA; ++a)
            if (L.q[t * kXRow + a] > bv) {   // first maximum, as torch.argmax
              bv = L.q[t * kXRow + a];
              best = a;
            }
          my_best = best;
        }
      }
    }

    weight_op<T, 8> o1 = weight_rows<T, 8, false>(w1, b1, D, D, kH, n0, lane);
    weight_op<T, 16> o2 = weight_rows<T, 16, true>(w2, b2, kH, kH, kH, n0, lane);
    weight_op<T, 16> o3 = weight_rows<T, 16, true>(w3, b3, kH, kH, has_out ? O : 0, a3, lane);
    if (!DQN) xin.request(R.in_table, in_index, in_dense, slot, D, t);
    xin.store(L.x, t);
    lds_barrier();
    stamp(2);
    if (DQN && t < kB) {
      // new = r + (boot * nt) * gamma (the reference's operation order, agent/dqn.py:356-360); read by
      // the output tiles two barriers further down
      T boot;
      if (A.ddqn) {
        boot = L.qt[t * NA + my_best];
      } else {
        boot = L.qt[t * NA];
        for (int a = 1; a < NA; ++a) boot = L.qt[t * NA + a] > boot ? L.qt[t * NA + a] : boot;
      }
      L.tg[t] = my_reward + (boot * my_nt) * (T)A.gamma;
      L.aux[kB + t] = my_action;
    }

    // ---- forward --------------------------------------------------------------------------------
    dense_relu<T, 8, 1>(o1, L.x, kXRow, L.h1, m0, n0, lane);
    lds_barrier();
    dense_relu<T, 16, 1>(o2, L.h1, kRow, L.h2, m0, n0, lane);
    // Requested here, a barrier and a product ahead of their use: the targets of the output tile
    // (the mask and the targets, or the action, reward and terminal flag of the sample), the
    // weights behind delta2 (the column block W3[.][n0 ..], lane (li, lq) takes the outputs
    // a = 8 lq ..) and the optimizer state of this wave's output-layer gradient tile dW3[a][k]
    // (outputs g0 = 16 (w / 4) .., neurons n0 ..).
    __builtin_amdgcn_sched_barrier(0);
    // REFRESH: what the output layer's phase reads and writes
    w3 = mk_w(2, n3); tw3 = mk_tw(2, n3); m_w3 = mk_mw(2, n3); v_w3 = mk_vw(2, n3);
    T yv[4];
    bool on[4];
    int count = kB;
    if (has_out) {
      if (!DQN) {
        const uint8_t* const mask = R.sample_mask ? R.sample_mask + (size_t)j * kB : nullptr;
        const T* const y = (const T*)R.targets + (size_t)(j / R.tgt_div) * kB * O;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int s = r3 + mfma_acc<T>::row(lane, v);
          yv[v] = a3 + li < O ? y[s * O + a3 + li] : (T)0;
          on[v] = !mask || mask[s];
        }
        if (mask) {
          const bool mine = mask[lane & 31] != 0;
          count = __popcll(__ballot(mine && lane < kB));
          count = count > 0 ? count : 1;
        }
      }
    }
    const int g0 = 16 * (wave >> 2);
    const bool has_g = g0 < O;
    T cw3[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int a = lq * 8 + s;
      cw3[s] = a < O ? w3[(uint32_t)(a * kH + n0 + li)] : (T)0;
    }
    adam_slot<T> s3[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = g0 + mfma_acc<T>::row(lane, v);
      s3[v] = slot_load<T>(w3, m_w3, v_w3, tw3, (uint32_t)(a * kH + n0 + li), blend, a < O);
    }
    lds_barrier();
    // output tiles and the loss gradient delta3
    if (has_out) {
      const acc_t acc = output_tile<T>(o3, L.h2, r3, lane);
      if (a3 + li < O) {
        if (DQN) {
          // loss = mean over the 32 x NA outputs of (Q - targets)^2 with targets == Q except at the
          // action taken (there: the sample's Q-learning target, L.tg), so the gradient is
          // 2 (Q[s][a] - new[s]) / (32 NA) there and zero elsewhere
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int s = r3 + mfma_acc<T>::row(lane, v);
            const T d = acc[v] - L.tg[s];
            const T g = ((T)2 * d) * ((T)1 / (T)(kB * NA));
            L.q[s * kXRow + li] = li == L.aux[kB + s] ? g : (T)0;
          }
        } else {
          // mean over the marked samples and the O outputs of (out - target)^2
          const T scale = (T)1 / (T)(count * O);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int s = r3 + mfma_acc<T>::row(lane, v);
            const T d = acc[v] - yv[v];
            L.q[s * kXRow + a3 + li] = on[v] ? ((T)2 * d) * scale : (T)0;
          }
        }
      }
    }
    stamp(3);
    if (R.debug_stage == 1) return;

    // ---- Adam constants of this network (its own step count) ------------------------------------
    // (the two powers are ~500 instructions of float64 arithmetic: one wave without an output tile
    //  works them out while the output tiles are being summed, the barrier below publishes them)
    if (wave == 7) {
      const double st = A.steps_given ? R.steps[j] : R.steps[j] + 1.0;
      if (lane == 0) {
        L.cst[0] = (T)(1.0 - pow(R.beta1, st));
        L.cst[1] = (T)sqrt(1.0 - pow(R.beta2, st));
      }
    }
    lds_barrier();
    adam_consts<T> c;
    {
      // (the same two numbers in every lane: kept in scalar registers through the three Adam phases)
      const T bc1 = L.cst[0];
      c.bc2_sqrt = uniform(L.cst[1]);
      c.step_size = uniform((T)R.lr / bc1);
      c.one_m_b1 = (T)(1.0 - R.beta1);
      c.b2 = (T)R.beta2;
      c.one_m_b2 = (T)(1.0 - R.beta2);
      c.eps = (T)R.eps;
      c.wd = (T)R.weight_decay;
      c.has_wd = R.weight_decay != 0.0;
      c.tau = (T)R.tau;
      c.blend = blend;
    }

    // ---- output layer ---------------------------------------------------------------------------
    // delta2 tile (rows m0 .., neurons n0 ..) from the weights the step started from, and the
    // gradient tile; the new weights are written once every wave has taken its share of the old
    // ones (barrier).  cw2: the weights behind delta1 (column block W2[.][n0 ..]), for the next phase.
    T cw2[16];
    {
      acc_t d2 = splat<T>((T)0);
      {
        const T* const ar = L.q + (m0 + li) * kXRow + lq * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) d2 = mfma(ar[s], cw3[s], d2);
      }
      acc_t g3 = splat<T>((T)0);
      if (has_g) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int r = lq * 8 + s;
          g3 = mfma(L.q[r * kXRow + g0 + li], L.h2[r * kRow + n0 + li], g3);
        }
      }
      T gb = (T)0;
      if (t < O)
        for (int s = 0; s < kB; ++s) gb = gb + L.q[s * kXRow + t];
      __builtin_amdgcn_sched_barrier(0);
      w2 = mk_w(1, n2);   // REFRESH
#pragma unroll
      for (int s = 0; s < 16; ++s) cw2[s] = w2[(uint32_t)((lq * 16 + s) * kH + n0 + li)];
      lds_barrier();
      // (h2 at this thread's own four places, read only now — everybody's reads of the tile for the
      //  gradient are behind the barrier — and replaced by delta2: four values fewer across the products)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        T* const h = L.h2 + (m0 + mfma_acc<T>::row(lane, v)) * kRow + n0 + li;
        *h = *h > (T)0 ? d2[v] : (T)0;
      }
      if (has_g) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a = g0 + mfma_acc<T>::row(lane, v);
          if (a < O)
            adam_apply<T>(w3, m_w3, v_w3, tw3, (uint32_t)(a * kH + n0 + li), g3[v], s3[v], c);
        }
      }
      if (t < O) {
        b3 = mk_b(2, (size_t)O); tb3 = mk_tb(2, (size_t)O); m_b3 = mk_mb(2, (size_t)O); v_b3 = mk_vb(2, (size_t)O);   // REFRESH
        const adam_slot<T> sb = slot_load<T>(b3, m_b3, v_b3, tb3, (uint32_t)t, blend, true);
        adam_apply<T>(b3, m_b3, v_b3, tb3, (uint32_t)t, gb, sb, c);
      }
    }
    stamp(4);
    if (R.debug_stage == 2) return;

    // ---- second layer: delta1 tile and two gradient tiles dW2[j][k] (rows n0 .., columns
    //      k0 = 32 (w / 4) .., + 16).  Optimizer state: the first tile's is requested once delta1's
    //      weight operand is used up, the second tile's once the first has been applied (128
    //      registers = two workgroups per CU). ----------------------------------------------------
    {
      const int k0 = 32 * (wave >> 2);
      adam_slot<T> s2a[4], s2b[4], s1[4];
      // REFRESH: the second and the first layer's tensors
      w2 = mk_w(1, n2); tw2 = mk_tw(1, n2); m_w2 = mk_mw(1, n2); v_w2 = mk_vw(1, n2);
      w1 = mk_w(0, n1); tw1 = mk_tw(0, n1); m_w1 = mk_mw(0, n1); v_w1 = mk_vw(0, n1);
      lds_barrier();
      acc_t d1 = splat<T>((T)0);
      {
        const T* const ar = L.h2 + (m0 + li) * kRow + lq * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) d1 = mfma(ar[s], cw2[s], d1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)   // (the registers of cw2 are free now)
        s2a[v] = slot_load<T>(w2, m_w2, v_w2, tw2,
                              (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * kH + k0 + li), blend, true);
      acc_t g2[2] = {splat<T>((T)0), splat<T>((T)0)};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int r = lq * 8 + s;
        const T a = L.h2[r * kRow + n0 + li];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) g2[kt] = mfma(a, L.h1[r * kRow + k0 + 16 * kt + li], g2[kt]);
      }
      T gb = (T)0;
      if (t < kH)
        for (int s = 0; s < kB; ++s) gb = gb + L.h2[s * kRow + t];
      lds_barrier();
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        T* const h = L.h1 + (m0 + mfma_acc<T>::row(lane, v)) * kRow + n0 + li;
        *h = *h > (T)0 ? d1[v] : (T)0;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        s2b[v] = slot_load<T>(w2, m_w2, v_w2, tw2,
                              (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * kH + k0 + 16 + li), blend,
                              true, false);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        adam_apply<T>(w2, m_w2, v_w2, tw2,
                      (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * kH + k0 + li), g2[0][v], s2a[v], c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)   // (the blend is the last thing the update needs)
        s2b[v].target = __builtin_nontemporal_load(
            tw2 + (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * kH + k0 + 16 + li));
      // first layer's gradient tile dW1[j][d] (rows n0 .., inputs d0 = 16 (w / 4) ..): its state
      const int d0 = 16 * (wave >> 2), d = d0 + li;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        s1[v] = slot_load<T>(w1, m_w1, v_w1, tw1,
                             (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * D + d), blend,
                             d0 < D && d < D, false);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        adam_apply<T>(w2, m_w2, v_w2, tw2,
                      (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * kH + k0 + 16 + li), g2[1][v],
                      s2b[v], c);
      __builtin_amdgcn_sched_barrier(0);
      if (d0 < D && d < D) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          s1[v].target = __builtin_nontemporal_load(
              tw1 + (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * D + d));
      }
      if (t < kH) {
        b2 = mk_b(1, kH); tb2 = mk_tb(1, kH); m_b2 = mk_mb(1, kH); v_b2 = mk_vb(1, kH);   // REFRESH
        const adam_slot<T> sb = slot_load<T>(b2, m_b2, v_b2, tb2, (uint32_t)t, blend, true);
        adam_apply<T>(b2, m_b2, v_b2, tb2, (uint32_t)t, gb, sb, c);
      }
      stamp(5);
      if (R.debug_stage == 3 || R.debug_stage == 4) return;
      lds_barrier();

      // ---- first layer ------------------------------------------------------------------------
      if (d0 < D) {
        acc_t g1 = splat<T>((T)0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int r = lq * 8 + s;
          g1 = mfma(L.h1[r * kRow + n0 + li], L.x[r * kXRow + d], g1);
        }
        if (d < D) {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            adam_apply<T>(w1, m_w1, v_w1, tw1, (uint32_t)((n0 + mfma_acc<T>::row(lane, v)) * D + d),
                          g1[v], s1[v], c);
        }
      }
      if (t < kH) {
        T gb1 = (T)0;
        for (int s = 0; s < kB; ++s) gb1 = gb1 + L.h1[s * kRow + t];
        b1 = mk_b(0, kH); tb1 = mk_tb(0, kH); m_b1 = mk_mb(0, kH); v_b1 = mk_vb(0, kH);   // REFRESH
        const adam_slot<T> sb = slot_load<T>(b1, m_b1, v_b1, tb1, (uint32_t)t, blend, true);
        adam_apply<T>(b1, m_b1, v_b1, tb1, (uint32_t)t, gb1, sb, c);
      }
    }
    if (t == 0 && !A.steps_given) R.steps[j] = R.steps[j] + 1.0;
  }
  stamp(6);

  // ---- outputs of the (updated) network for the extra rows -------------------------------------
  // (cobel_dqn_replay: the Q-values of the next observation, what the next action selection needs:
  //  agent/dqn.py:174 -> retrieve_q)
  if (R.ep_out && R.ep_rows > 0) {
    __syncthreads();   // the new parameters are in memory; x / h1 / h2 are free
    forward_pass<T, true> ep;
    w1 = mk_w(0, n1); b1 = mk_b(0, kH); w2 = mk_w(1, n2); b2 = mk_b(1, kH);   // REFRESH
    w3 = mk_w(2, n3); b3 = mk_b(2, (size_t)O);
    ep.request(w1, b1, w2, b2, w3, b3, D, O, wave, lane);
    const int E = R.ep_rows;
    if (t < kMaxEp * 32) {
      const int r = t >> 5, d = t & 31;
      T v = (T)0;
      if (r < E && d < D)
        v = R.ep_table ? (T)R.ep_table[(size_t)ep_row * D + d]
                       : ((const T*)R.ep_dense)[((size_t)j * E + r) * D + d];
      L.x[r * kXRow + d] = v;
    }
    // one row tile (rows >= E of it are not looked at)
    stamp(7);
    const acc_t acc = ep.run(L, lane);
    stamp(8);
    if (wave < 2 && 16 * wave < O) {
      const int a = 16 * wave + li;
      if (a < O) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = mfma_acc<T>::row(lane, v);
          if (r < E) ((T*)R.ep_out)[((size_t)j * E + r) * O + a] = acc[v];
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kFitThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_mlp_fit(const fit_args A) {
  mlp_fit_body<T, false>(A);
}
template <typename T, int NAC = kA>
__global__ __launch_bounds__(kFitThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_dqn_replay(const fit_args A) {
  mlp_fit_body<T, true, NAC>(A);
}

int shape_ok(int32_t D, int32_t O, const char* who) {
  COBEL_REQUIRE(D >= 1 && D <= kMaxD && O >= 1 && O <= kMaxO, COBEL_E_UNSUPPORTED,
                "%s: Linear(D <= %d, 64)-ReLU-Linear(64, 64)-ReLU-Linear(64, O <= %d) on batches "
                "of 32 is covered (got D %d, O %d)", who, kMaxD, kMaxO, D, O);
  return COBEL_OK;
}

// Experiments only (scripts/experiments/exp_mlp_trace.py): COBEL_DEBUG_MLP_TRACE = address of a device buffer of
// n x 16 uint64 for the per-phase stamps.  Taken only if it parses completely and names DEVICE memory
// of the current device; anything else is ignored, so a stray variable cannot send stores anywhere.
unsigned long long* debug_trace_buffer() {
  const char* const v = cobel_debug_env("COBEL_DEBUG_MLP_TRACE");
  if (!v || !*v) return nullptr;
  char* end = nullptr;
  const unsigned long long addr = strtoull(v, &end, 0);
  if (*end != '\0' || addr == 0 || (addr & 7u)) return nullptr;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, reinterpret_cast<const void*>(addr)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != dev)
    return nullptr;
  return reinterpret_cast<unsigned long long*>(addr);
}
// COBEL_DEBUG_MLP_STAGE = 1 .. 5: leave the step after that phase (phase timings; any other value: 0)
int debug_stage_env() {
  const char* const v = cobel_debug_env("COBEL_DEBUG_MLP_STAGE");
  if (!v || !*v) return 0;
  char* end = nullptr;
  const long k = strtol(v, &end, 10);
  return (*end == '\0' && k >= 1 && k <= 5) ? (int)k : 0;
}

template <typename K>
int raise_lds(K kernel, int32_t lds) {
  if (lds > 64 * 1024)
    COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return COBEL_OK;
}

}  // namespace

extern "C" int cobel_mlp_query(int32_t n_inputs, int32_t n_hidden1, int32_t n_hidden2,
                               int32_t n_outputs, int32_t batch, int32_t is_float64,
                               int32_t* lds_bytes) {
  COBEL_REQUIRE(n_hidden1 == kH && n_hidden2 == kH && batch == kB, COBEL_E_UNSUPPORTED,
                "cobel_mlp: hidden layers of 64 units and batches of 32 are covered (got %d-%d, "
                "batch %d)", n_hidden1, n_hidden2, batch);
  if (int rc = shape_ok(n_inputs, n_outputs, "cobel_mlp")) return rc;
  if (lds_bytes)
    *lds_bytes = (int32_t)(fit_lds_elems() * (is_float64 ? 8 : 4));
  return COBEL_OK;
}

extern "C" int cobel_mlp_forward(const cobel_mlp_forward_t* run, void* stream) {
  COBEL_REQUIRE(run, COBEL_E_ARG, "cobel_mlp_forward: NULL run");
  const cobel_mlp_forward_t& r = *run;
  int32_t lds = 0;
  if (int rc = cobel_mlp_query(r.n_inputs, kH, kH, r.n_outputs, kB, r.is_float64, &lds)) return rc;
  lds = (int32_t)(fwd_lds_elems() * (r.is_float64 ? 8 : 4));
  for (int l = 0; l < 3; ++l)
    COBEL_REQUIRE(r.w[l] && r.b[l], COBEL_E_ARG, "cobel_mlp_forward: NULL parameters (layer %d)", l);
  COBEL_REQUIRE(r.out && (r.in_table ? r.in_index != nullptr : r.in_dense != nullptr), COBEL_E_ARG,
                "cobel_mlp_forward: out and inputs (in_table + in_index, or in_dense) are required");
  COBEL_REQUIRE(r.n >= 0 && r.net_div >= 1 && r.in_div >= 1 && r.act_div >= 1, COBEL_E_RANGE,
                "cobel_mlp_forward: bad sizes");
  COBEL_REQUIRE(((uintptr_t)r.w[1] & 15u) == 0 && ((uintptr_t)r.w[2] & 15u) == 0, COBEL_E_ARG,
                "cobel_mlp_forward: the 64-wide weight matrices must be 16-byte aligned");
  if (r.n == 0) return COBEL_OK;
  fwd_args A;
  A.r = r;
  // instances that share a network go to one workgroup, four at most (its weights are fetched once)
  A.group = 1;
  for (int g = 4; g > 1; --g)
    if (r.net_div % g == 0 && r.n % g == 0) {
      A.group = g;
      break;
    }
  const int grid = r.n / A.group;
  hipStream_t st = (hipStream_t)stream;
  if (A.group > 1) {
    if (r.is_float64) hipLaunchKernelGGL((k_mlp_forward<double, true>), dim3(grid), dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_mlp_forward<float, true>), dim3(grid), dim3(256), lds, st, A);
  } else {
    if (r.is_float64) hipLaunchKernelGGL((k_mlp_forward<double, false>), dim3(grid), dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_mlp_forward<float, false>), dim3(grid), dim3(256), lds, st, A);
  }
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

extern "C" int cobel_mlp_fit(const cobel_mlp_fit_t* run, void* stream) {
  COBEL_REQUIRE(run, COBEL_E_ARG, "cobel_mlp_fit: NULL run");
  const cobel_mlp_fit_t& r = *run;
  int32_t lds = 0;
  if (int rc = cobel_mlp_query(r.n_inputs, kH, kH, r.n_outputs, kB, r.is_float64, &lds)) return rc;
  for (int l = 0; l < 3; ++l)
    COBEL_REQUIRE(r.w[l] && r.b[l] && r.m_w[l] && r.m_b[l] && r.v_w[l] && r.v_b[l], COBEL_E_ARG,
                  "cobel_mlp_fit: NULL parameter / moment tensor (layer %d)", l);
  COBEL_REQUIRE(r.tau == 0.0 || (r.w_target[0] && r.w_target[1] && r.w_target[2] &&
                                 r.b_target[0] && r.b_target[1] && r.b_target[2]),
                COBEL_E_ARG, "cobel_mlp_fit: tau != 0 needs the target network's tensors");
  COBEL_REQUIRE(r.steps && r.targets && (r.in_table ? r.in_index != nullptr : r.in_dense != nullptr),
                COBEL_E_ARG, "cobel_mlp_fit: steps, targets and inputs are required");
  COBEL_REQUIRE(r.n >= 0 && r.in_div >= 1 && r.tgt_div >= 1 && r.act_div >= 1 && r.ep_div >= 1 &&
                    r.ep_rows >= 0 && r.ep_rows <= kMaxEp,
                COBEL_E_RANGE, "cobel_mlp_fit: bad sizes (at most %d extra rows)", kMaxEp);
  COBEL_REQUIRE(!r.ep_out || r.ep_rows == 0 ||
                    (r.ep_table ? (r.ep_index != nullptr && r.ep_rows == 1) : r.ep_dense != nullptr),
                COBEL_E_ARG, "cobel_mlp_fit: extra rows need ep_table + ep_index (one row) or ep_dense");
  COBEL_REQUIRE(((uintptr_t)r.w[1] & 15u) == 0 && ((uintptr_t)r.w[2] & 15u) == 0, COBEL_E_ARG,
                "cobel_mlp_fit: the 64-wide weight matrices must be 16-byte aligned");
  if (r.n == 0) return COBEL_OK;
  fit_args A;
  memset(&A, 0, sizeof A);
  A.r = r;
  A.rows = kB;
  A.trace = debug_trace_buffer();
  hipStream_t st = (hipStream_t)stream;
  lds += cobel_debug_lds_pad(lds, 160 * 1024);   // (occupancy experiments only)
  if (r.is_float64) {
    if (int rc = raise_lds(&k_mlp_fit<double>, lds)) return rc;
    hipLaunchKernelGGL(k_mlp_fit<double>, dim3(r.n), dim3(kFitThreads), lds, st, A);
  } else {
    if (int rc = raise_lds(&k_mlp_fit<float>, lds)) return rc;   // (only a padded launch exceeds 64 KiB)
    hipLaunchKernelGGL(k_mlp_fit<float>, dim3(r.n), dim3(kFitThreads), lds, st, A);
  }
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}

// Two forms of the DQN step.  The parameter-staging kernel (mlp.hip: four waves, parameters in LDS)
// has the shorter dependent chain per instance when several of its workgroups share a CU — at
// least two fit: always in float32, up to 7 inputs in float64 — and is used there (C5: 6 inputs);
// beyond that it is alone on its CU and the streaming form (k_dqn_replay here: 52 KB whatever the
// inputs, two workgroups of eight waves per CU) wins: Dyna-DQN's 25 one-hot inputs in float64,
// 1.34 -> 1.0 ms per step of 8 192 instances.  COBEL_DEBUG_DQN_KERNEL = lds | stream pins one
// (experiments, tests).
static bool dqn_staged(int32_t n_inputs, int32_t is_float64, int32_t n_actions) {
  if (n_actions != kA) return false;   // (the parameter-staging form is laid out for four actions)
  if (const char* v = cobel_debug_env("COBEL_DEBUG_DQN_KERNEL")) {   // exactly "lds" or "stream", else ignored
    if (!strcmp(v, "lds")) return true;
    if (!strcmp(v, "stream")) return false;
  }
  // (two workgroups of the staging kernel per CU: the LDS of the device the call runs on)
  int dev = 0, n_cu = 0;
  size_t lds_per_cu = 160 * 1024;
  if (hipGetDevice(&dev) == hipSuccess) (void)cobel_device_limits(dev, &n_cu, &lds_per_cu);
  return 2 * cobel_dqn_replay_lds_bytes(n_inputs, is_float64) <= lds_per_cu;
}

extern "C" int cobel_dqn_replay_query(int32_t n_inputs, int32_t n_hidden1, int32_t n_hidden2,
                                      int32_t n_actions, int32_t batch, int32_t is_float64,
                                      int32_t* lds_bytes) {
  COBEL_REQUIRE(n_inputs >= 1 && n_inputs <= kMaxD && n_hidden1 == kH && n_hidden2 == kH &&
                    n_actions >= 1 && n_actions <= kAMax && batch == kB,
                COBEL_E_UNSUPPORTED,
                "cobel_dqn_replay: the fused step covers Linear(D <= %d, 64)-ReLU-Linear(64, 64)-"
                "ReLU-Linear(64, A <= %d) on batches of 32 (got D %d, %d-%d, %d actions, batch %d)",
                kMaxD, kAMax, n_inputs, n_hidden1, n_hidden2, n_actions, batch);
  if (lds_bytes)
    *lds_bytes = dqn_staged(n_inputs, is_float64, n_actions)
                     ? (int32_t)cobel_dqn_replay_lds_bytes(n_inputs, is_float64)
                     : (int32_t)(fit_lds_elems() * (is_float64 ? 8 : 4));
  return COBEL_OK;
}

extern "C" int cobel_dqn_replay(const cobel_dqn_replay_t* run, void* stream) {
  COBEL_REQUIRE(run, COBEL_E_ARG, "cobel_dqn_replay: NULL run");
  const cobel_dqn_replay_t& r = *run;
  int32_t lds = 0;
  if (int rc = cobel_dqn_replay_query(r.n_inputs, r.n_hidden1, r.n_hidden2, r.n_actions, r.batch,
                                      r.is_float64, &lds))
    return rc;
  for (int l = 0; l < 3; ++l)
    COBEL_REQUIRE(r.w[l] && r.b[l] && r.w_target[l] && r.b_target[l] && r.m_w[l] && r.m_b[l] &&
                      r.v_w[l] && r.v_b[l],
                  COBEL_E_ARG, "cobel_dqn_replay: NULL parameter / moment tensor (layer %d)", l);
  COBEL_REQUIRE(r.actions && r.rewards && r.nonterminal && r.steps, COBEL_E_ARG,
                "cobel_dqn_replay: NULL batch tensor or step counts");
  COBEL_REQUIRE(r.state_index ? (r.next_index && r.obs_table && !r.batch_slots)
                              : (r.states && r.next_states),
                COBEL_E_ARG, "cobel_dqn_replay: the batch's observations are missing (states + "
                "next_states, or state_index + next_index + obs_table without batch_slots)");
  COBEL_REQUIRE(r.n >= 0, COBEL_E_RANGE, "cobel_dqn_replay: n = %d", r.n);
  COBEL_REQUIRE(!r.batch_slots || r.ring_slots > 0, COBEL_E_RANGE,
                "cobel_dqn_replay: batch_slots given with ring_slots = %d", r.ring_slots);
  COBEL_REQUIRE(!r.q_out || (r.obs_index && r.obs_table), COBEL_E_ARG,
                "cobel_dqn_replay: q_out needs obs_index and obs_table");
  COBEL_REQUIRE((((uintptr_t)r.w[1] | (uintptr_t)r.w[2] | (uintptr_t)r.w_target[1] |
                  (uintptr_t)r.w_target[2]) & 15u) == 0,
                COBEL_E_ARG, "cobel_dqn_replay: the 64-wide weight matrices must be 16-byte aligned");
  if (r.n == 0) return COBEL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dqn_staged(r.n_inputs, r.is_float64, r.n_actions))
    return cobel_dqn_replay_lds_launch(r, st, debug_trace_buffer());
  // the optimisation step of cobel_mlp_fit towards the Q-learning targets
  fit_args A;
  memset(&A, 0, sizeof A);
  cobel_mlp_fit_t& f = A.r;
  for (int l = 0; l < 3; ++l) {
    f.w[l] = r.w[l];
    f.b[l] = r.b[l];
    f.w_target[l] = r.w_target[l];
    f.b_target[l] = r.b_target[l];
    f.m_w[l] = r.m_w[l];
    f.m_b[l] = r.m_b[l];
    f.v_w[l] = r.v_w[l];
    f.v_b[l] = r.v_b[l];
  }
  f.steps = const_cast<double*>(r.steps);
  f.active = r.active;
  if (r.state_index) {
    f.in_table = r.obs_table;
    f.in_index = r.state_index;
    A.next_index = r.next_index;
  } else {
    f.in_dense = r.states;
    A.next_dense = r.next_states;
  }
  if (r.q_out) {
    f.ep_table = r.obs_table;
    f.ep_index = r.obs_index;
    f.ep_rows = 1;
    f.ep_out = r.q_out;
  }
  f.n = r.n;
  f.n_inputs = r.n_inputs;
  f.n_outputs = r.n_actions;
  f.is_float64 = r.is_float64;
  f.in_div = f.tgt_div = f.act_div = f.ep_div = 1;
  f.lr = r.lr;
  f.beta1 = r.beta1;
  f.beta2 = r.beta2;
  f.eps = r.eps;
  f.weight_decay = r.weight_decay;
  f.tau = r.tau;
  A.dqn = 1;
  A.n_actions = r.n_actions;
  A.ddqn = r.ddqn;
  A.rows = r.batch_slots ? r.ring_slots : kB;
  A.steps_given = 1;
  A.slots = r.batch_slots;
  A.actions = r.actions;
  A.rewards = r.rewards;
  A.nonterminal = r.nonterminal;
  A.gamma = r.gamma;
  f.debug_stage = debug_stage_env();
  A.trace = debug_trace_buffer();
  lds += cobel_debug_lds_pad(lds, 160 * 1024);                                    // (occupancy)
  const bool four = r.n_actions == kA;
  const void* const kernel =
      r.is_float64 ? (four ? reinterpret_cast<const void*>(&k_dqn_replay<double, kA>)
                           : reinterpret_cast<const void*>(&k_dqn_replay<double, 0>))
                   : (four ? reinterpret_cast<const void*>(&k_dqn_replay<float, kA>)
                           : reinterpret_cast<const void*>(&k_dqn_replay<float, 0>));
  if (lds > 64 * 1024) {
    if (int rc = raise_lds(kernel, lds)) return rc;
  }
  if (r.is_float64 && four) hipLaunchKernelGGL((k_dqn_replay<double, kA>), dim3(r.n), dim3(kFitThreads), lds, st, A);
  else if (r.is_float64) hipLaunchKernelGGL((k_dqn_replay<double, 0>), dim3(r.n), dim3(kFitThreads), lds, st, A);
  else if (four) hipLaunchKernelGGL((k_dqn_replay<float, kA>), dim3(r.n), dim3(kFitThreads), lds, st, A);
  else hipLaunchKernelGGL((k_dqn_replay<float, 0>), dim3(r.n), dim3(kFitThreads), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
