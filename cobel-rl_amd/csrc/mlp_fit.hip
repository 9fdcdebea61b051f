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
// Same structure as k_dqn_replay (mlp.hip), which stays the specialised kernel of the DQN step: a
// workgroup of 256 threads per network, parameters staged in LDS (the 64 x 64 matrix transposed),
// activations in LDS and overwritten by the deltas, the three 64 x 64 products as 16 x 16 x 4
// MFMAs in the network's dtype, gradients only ever in registers, Adam applied by the thread that
// accumulated the element.  The output layer is O wide here, so its loops run over O x 64 elements.
#include <stdlib.h>

#include <cstdlib>

#include "cobel_common.h"

namespace {

constexpr int kH = 64;
constexpr int kB = 32;
constexpr int kRow = 66;
constexpr int kMaxD = 32;
constexpr int kMaxO = 32;
constexpr int kMaxEp = 4;
constexpr int kFitThreads = 512;   // threads of a training workgroup

struct fit_args {
  cobel_mlp_fit_t r;
};
struct fwd_args {
  cobel_mlp_forward_t r;
};

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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


// dst(e, value) for the elements e = t, t + NT, ... < n of src: ALL of the thread's loads first, then
// the writes.  Written as one plain loop each element waits for its own trip to memory
// (`s_waitcnt vmcnt(0)` per iteration) — with one workgroup per CU nothing else covers it.
template <int MAXI, int NT, typename T, typename Put>
__device__ __forceinline__ void staged_copy(const T* __restrict__ src, int n, int t, Put put) {
  T r[MAXI];
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int e = t + NT * u;
    r[u] = e < n ? src[e] : (T)0;
  }
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int e = t + NT * u;
    if (e < n) put(e, r[u]);
  }
}

template <typename T>
struct adam_consts {
  T bc2_sqrt, step_size, one_m_b1, b2, one_m_b2, eps, wd, tau;
  bool has_wd, blend;
};

// torch.optim.Adam, one element (the operation order of k_adam in adam.hip); returns the new
// parameter and writes parameter, moments and — when blending — the target network's copy.
// The optimizer state of an element (and the target network's copy of the parameter), requested at
// the START of the kernel — the workgroup owns a whole CU's register file (one wave per SIMD), so
// every element a thread will update fits — and consumed in the backward pass, by which time the
// loads have long returned.  Loaded where used, each update waited for HBM on its own.
template <typename T>
struct adam_slot {
  T m, v, target;
};
template <typename T>
__device__ __forceinline__ adam_slot<T> slot_load(const T* __restrict__ m, const T* __restrict__ v,
                                                  const T* __restrict__ tgt, size_t e, bool blend) {
  adam_slot<T> s;
  s.m = __builtin_nontemporal_load(m + e);
  s.v = __builtin_nontemporal_load(v + e);
  s.target = blend ? __builtin_nontemporal_load(tgt + e) : (T)0;
  return s;
}

template <typename T>
__device__ __forceinline__ T adam_apply(T* __restrict__ p, T* __restrict__ m, T* __restrict__ v,
                                        T* __restrict__ tgt, size_t e, T p_old, T g,
                                        const adam_slot<T>& s, const adam_consts<T>& c) {
  if (c.has_wd) g = g + c.wd * p_old;
  const T m0 = s.m, v0 = s.v;
  const T mn = m0 + c.one_m_b1 * (g - m0);
  const T vn = v0 * c.b2 + (c.one_m_b2 * g) * g;
  const T denom = sqrt(vn) / c.bc2_sqrt + c.eps;
  const T pn = p_old - c.step_size * (mn / denom);
  m[e] = mn;
  v[e] = vn;
  p[e] = pn;
  if (c.blend) tgt[e] = s.target + c.tau * (pn - s.target);
  return pn;
}

template <typename T>
struct net_lds {
  T* wt2;   // [64][66]  wt2[k * 66 + j] = W2[j][k]
  T* wt1;   // [D][64]   wt1[d * 64 + j] = W1[j][d]
  T* w3;    // [O][64]
  T* b1;    // [64]
  T* b2;    // [64]
  T* b3;    // [32]
  T* x;     // [32][D]
  T* h1;    // [32][66]
  T* h2;    // [32][66]
  T* q;     // [32][O]  outputs -> delta3
};

__host__ __device__ inline size_t net_lds_elems(int D, int O) {
  return (size_t)kH * kRow + (size_t)D * kH + (size_t)O * kH + kH + kH + kMaxO + (size_t)kB * D +
         2 * (size_t)kB * kRow + (size_t)kB * O;
}

template <typename T>
__device__ __forceinline__ net_lds<T> carve(unsigned char* raw, int D, int O) {
  net_lds<T> L;
  T* p = reinterpret_cast<T*>(raw);
  L.wt2 = p; p += kH * kRow;
  L.wt1 = p; p += D * kH;
  L.w3 = p;  p += O * kH;
  L.b1 = p;  p += kH;
  L.b2 = p;  p += kH;
  L.b3 = p;  p += kMaxO;
  L.x = p;   p += kB * D;
  L.h1 = p;  p += kB * kRow;
  L.h2 = p;  p += kB * kRow;
  L.q = p;
  return L;
}

// One network's parameters (torch.nn.Linear layout [out][in]) into LDS.
template <typename T, int NT>
__device__ void stage_params(const net_lds<T>& L, const T* __restrict__ w1,
                             const T* __restrict__ b1, const T* __restrict__ w2,
                             const T* __restrict__ b2, const T* __restrict__ w3,
                             const T* __restrict__ b3, int D, int O, int t) {
  constexpr int U = kH * kH / NT;
  T w2r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) w2r[u] = __builtin_nontemporal_load(w2 + t + NT * u);
#pragma unroll
  for (int u = 0; u < U; ++u) {   // transposed write
    const int e = t + NT * u;
    L.wt2[(e & 63) * kRow + (e >> 6)] = w2r[u];
  }
  staged_copy<(kH * kMaxD + NT - 1) / NT, NT, T>(w1, kH * D, t, [&](int e, T v) {
    const int j = e / D, d = e - j * D;
    L.wt1[d * kH + j] = v;
  });
  staged_copy<(kMaxO * kH + NT - 1) / NT, NT, T>(w3, O * kH, t, [&](int e, T v) { L.w3[e] = v; });
  if (t < kH) {
    L.b1[t] = b1[t];
    L.b2[t] = b2[t];
  }
  if (t < O) L.b3[t] = b3[t];
}

// h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2), out[s][a] = W3 h2 + b3 for the 32 rows of L.x, by a
// workgroup of NT = 512 threads: first layer one sample x four neurons per thread, second layer one
// 16 x 16 MFMA tile per wave (2 sample tiles x 4 neuron tiles = 8 waves).
template <typename T, int NT>
__device__ void forward32(const net_lds<T>& L, T* out, int D, int O, int t) {
  static_assert(NT == 512, "eight waves per workgroup");
  {
    const int j0 = (t & 15) * 4, s0 = t >> 4;
    T acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = L.b1[j0 + c];
    for (int d = 0; d < D; ++d) {
      const T x0 = L.x[s0 * D + d];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fma_t<T>(L.wt1[d * kH + j0 + c], x0, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) L.h1[s0 * kRow + j0 + c] = acc[c] > (T)0 ? acc[c] : (T)0;
  }
  lds_barrier();
  {
    typedef typename mfma_acc<T>::type acc_t;
    const int lane = t & 63, wave = t >> 6;
    const int jt = (wave & 3) * 16, st = (wave >> 2) * 16;
    const int li = lane & 15, lq = lane >> 4;
    const T bias = L.b2[jt + li];
    acc_t acc0 = {bias, bias, bias, bias};
#pragma unroll 4
    for (int k0 = 0; k0 < kH; k0 += 4)
      acc0 = mfma(L.h1[(st + li) * kRow + k0 + lq], L.wt2[(k0 + lq) * kRow + jt + li], acc0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = mfma_acc<T>::row(lane, v);
      L.h2[(st + r) * kRow + jt + li] = acc0[v] > (T)0 ? acc0[v] : (T)0;
    }
  }
  lds_barrier();
  for (int e = t; e < kB * O; e += NT) {
    const int s = e / O, a = e - s * O;
    T acc = L.b3[a];
#pragma unroll 8
    for (int k = 0; k < kH; ++k) acc = fma_t<T>(L.w3[a * kH + k], L.h2[s * kRow + k], acc);
    out[e] = acc;
  }
  lds_barrier();
}

// the 32 input rows of an instance: rows of a float64 table by index, or a dense [32][D] block
template <typename T, int NT>
__device__ void load_inputs(T* dst, const double* table, const int32_t* index, const T* dense,
                            int D, int t) {
  constexpr int U = (kB * kMaxD + NT - 1) / NT;
  if (table) {   // the row numbers first (all at once), then the rows (all at once)
    int row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = t + NT * u;
      row[u] = e < kB * D ? index[e / D] : 0;
    }
    double r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = t + NT * u;
      const int d = e - (e / D) * D;
      r[u] = e < kB * D ? table[(size_t)row[u] * D + d] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = t + NT * u;
      if (e < kB * D) dst[e] = (T)r[u];
    }
  } else {
    staged_copy<U, NT, T>(dense, kB * D, t, [&](int e, T v) { dst[e] = v; });
  }
}

// ---------------------------------------------------------------------------------------------
// Forward only.  No backward pass means no buffer has to outlive the layer that reads it: the first
// layer's weights sit where h2 will be written, the output layer's where h1 was, the inputs are
// read straight from memory and the outputs written straight to it — 69 KB in float64 instead of
// the 94 KB of the training layout: two workgroups per CU.
__host__ __device__ inline size_t fwd_lds_elems() {
  return (size_t)kH * kRow + 2 * (size_t)kB * kRow + 2 * kH + kMaxO;
}

template <typename T>
__device__ __forceinline__ void mlp_forward_body(const fwd_args& A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const cobel_mlp_forward_t& R = A.r;
  const int j = (int)blockIdx.x, t = (int)threadIdx.x;
  const int D = R.n_inputs, O = R.n_outputs;
  if (R.active && !R.active[j / R.act_div]) return;
  T* const wt2 = reinterpret_cast<T*>(lds_raw);          // [64][66]
  T* const h1 = wt2 + kH * kRow;                         // [32][66]; later w3 [O][64]
  T* const h2 = h1 + kB * kRow;                          // [32][66]; before that wt1 [D][64]
  T* const bias1 = h2 + kB * kRow;
  T* const bias2 = bias1 + kH;
  T* const bias3 = bias2 + kH;
  T* const wt1 = h2;
  T* const w3l = h1;
  const size_t net = (size_t)(j / R.net_div);
  const size_t n1 = (size_t)kH * D, n2 = (size_t)kH * kH, n3 = (size_t)O * kH;
  const T* const w1 = (const T*)R.w[0] + net * n1;
  const T* const w2 = (const T*)R.w[1] + net * n2;
  const T* const w3 = (const T*)R.w[2] + net * n3;
  {
    T w2r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w2r[u] = w2[t + 256 * u];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = t + 256 * u;
      wt2[(e & 63) * kRow + (e >> 6)] = w2r[u];
    }
    staged_copy<kH * kMaxD / 256, 256, T>(w1, kH * D, t, [&](int e, T v) {
      const int jj = e / D, d = e - jj * D;
      wt1[d * kH + jj] = v;
    });
    if (t < kH) {
      bias1[t] = ((const T*)R.b[0] + net * kH)[t];
      bias2[t] = ((const T*)R.b[1] + net * kH)[t];
    }
    if (t < O) bias3[t] = ((const T*)R.b[2] + net * O)[t];
  }
  lds_barrier();
  // layer 1: thread tile 2 samples x 4 neurons, inputs from memory (16 threads share a sample)
  {
    const int j0 = (t & 15) * 4, s0 = (t >> 4) * 2;
    const int32_t* const idx = R.in_table ? R.in_index + (size_t)(j / R.in_div) * kB : nullptr;
    const double* const r0 = R.in_table ? R.in_table + (size_t)idx[s0] * D : nullptr;
    const double* const r1 = R.in_table ? R.in_table + (size_t)idx[s0 + 1] * D : nullptr;
    const T* const dn = R.in_table ? nullptr : (const T*)R.in_dense + ((size_t)j * kB + s0) * D;
    T acc[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[0][c] = acc[1][c] = bias1[j0 + c];
    for (int d = 0; d < D; ++d) {
      const T x0 = R.in_table ? (T)r0[d] : dn[d], x1 = R.in_table ? (T)r1[d] : dn[D + d];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T w = wt1[d * kH + j0 + c];
        acc[0][c] = fma_t<T>(w, x0, acc[0][c]);
        acc[1][c] = fma_t<T>(w, x1, acc[1][c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h1[s0 * kRow + j0 + c] = acc[0][c] > (T)0 ? acc[0][c] : (T)0;
      h1[(s0 + 1) * kRow + j0 + c] = acc[1][c] > (T)0 ? acc[1][c] : (T)0;
    }
  }
  lds_barrier();   // (every read of wt1 is done: h2 may be written)
  {
    typedef typename mfma_acc<T>::type acc_t;
    const int lane = t & 63, jt = (t >> 6) * 16;
    const int li = lane & 15, lq = lane >> 4;
    const T bias = bias2[jt + li];
    acc_t acc0 = {bias, bias, bias, bias}, acc1 = acc0;
#pragma unroll 4
    for (int k0 = 0; k0 < kH; k0 += 4) {
      const T b = wt2[(k0 + lq) * kRow + jt + li];
      acc0 = mfma(h1[li * kRow + k0 + lq], b, acc0);
      acc1 = mfma(h1[(16 + li) * kRow + k0 + lq], b, acc1);
    }
    lds_barrier();   // (every read of h1 is done: the output layer's weights take its place)
    staged_copy<kMaxO * kH / 256, 256, T>(w3, O * kH, t, [&](int e, T v) { w3l[e] = v; });
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = mfma_acc<T>::row(lane, v);
      h2[r * kRow + jt + li] = acc0[v] > (T)0 ? acc0[v] : (T)0;
      h2[(16 + r) * kRow + jt + li] = acc1[v] > (T)0 ? acc1[v] : (T)0;
    }
  }
  lds_barrier();
  T* const out = (T*)R.out + (size_t)j * kB * O;
  for (int e = t; e < kB * O; e += 256) {
    const int s = e / O, a = e - s * O;
    T acc = bias3[a];
#pragma unroll 8
    for (int k = 0; k < kH; ++k) acc = fma_t<T>(w3l[a * kH + k], h2[s * kRow + k], acc);
    out[e] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_mlp_forward(const fwd_args A) {
  mlp_forward_body<T>(A);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mlp_fit_body(const fit_args& A) {
  constexpr int NT = kFitThreads;   // eight waves: the workgroup is alone on its CU (LDS), so the
                                    // parallelism inside it is all there is to hide latency
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const cobel_mlp_fit_t& R = A.r;
  const int j = (int)blockIdx.x, t = (int)threadIdx.x;
  if (R.active && !R.active[j / R.act_div]) return;
  const int D = R.n_inputs, O = R.n_outputs;
  const net_lds<T> L = carve<T>(lds_raw, D, O);
  const bool train = !R.train || R.train[j];
  const size_t n1 = (size_t)kH * D, n2 = (size_t)kH * kH, n3 = (size_t)O * kH;
  T* const w1 = (T*)R.w[0] + (size_t)j * n1;
  T* const b1 = (T*)R.b[0] + (size_t)j * kH;
  T* const w2 = (T*)R.w[1] + (size_t)j * n2;
  T* const b2 = (T*)R.b[1] + (size_t)j * kH;
  T* const w3 = (T*)R.w[2] + (size_t)j * n3;
  T* const b3 = (T*)R.b[2] + (size_t)j * O;
  const bool blend = R.tau != 0.0 && R.w_target[0] != nullptr;
  T* const tw1 = blend ? (T*)R.w_target[0] + (size_t)j * n1 : nullptr;
  T* const tb1 = blend ? (T*)R.b_target[0] + (size_t)j * kH : nullptr;
  T* const tw2 = blend ? (T*)R.w_target[1] + (size_t)j * n2 : nullptr;
  T* const tb2 = blend ? (T*)R.b_target[1] + (size_t)j * kH : nullptr;
  T* const tw3 = blend ? (T*)R.w_target[2] + (size_t)j * n3 : nullptr;
  T* const tb3 = blend ? (T*)R.b_target[2] + (size_t)j * O : nullptr;

  stage_params<T, NT>(L, w1, b1, w2, b2, w3, b3, D, O, t);

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
    T* const m_w1 = (T*)R.m_w[0] + (size_t)j * n1; T* const v_w1 = (T*)R.v_w[0] + (size_t)j * n1;
    T* const m_w2 = (T*)R.m_w[1] + (size_t)j * n2; T* const v_w2 = (T*)R.v_w[1] + (size_t)j * n2;
    T* const m_w3 = (T*)R.m_w[2] + (size_t)j * n3; T* const v_w3 = (T*)R.v_w[2] + (size_t)j * n3;
    T* const m_b1 = (T*)R.m_b[0] + (size_t)j * kH; T* const v_b1 = (T*)R.v_b[0] + (size_t)j * kH;
    T* const m_b2 = (T*)R.m_b[1] + (size_t)j * kH; T* const v_b2 = (T*)R.v_b[1] + (size_t)j * kH;
    T* const m_b3 = (T*)R.m_b[2] + (size_t)j * O;  T* const v_b3 = (T*)R.v_b[2] + (size_t)j * O;
    // (the inputs are requested BEFORE the optimizer state: loads return in order, and the forward
    //  pass must not queue behind 3 x 59 KB it does not need)
    load_inputs<T, NT>(L.x, R.in_table,
                   R.in_table ? R.in_index + (size_t)(j / R.in_div) * kB : nullptr,
                   R.in_table ? nullptr : (const T*)R.in_dense + (size_t)j * kB * D, D, t);
    // second layer: wave w owns the gradient tiles (rows j = 16 (w % 4) .., columns k = 16 kt ..
    // for kt = 2 (w / 4), 2 (w / 4) + 1) and the delta tile (samples 16 (w / 4) .., columns k =
    // 16 (w % 4) ..)
    const int lane2 = t & 63, wave2 = t >> 6, jt2 = (wave2 & 3) * 16, li2 = lane2 & 15;
    const int kt0 = (wave2 >> 2) * 2, st2 = (wave2 >> 2) * 16;
    constexpr int kU3 = (kMaxO * kH + NT - 1) / NT, kU1 = (kMaxD * kH + NT - 1) / NT;
    adam_slot<T> s2[2][4], s3[kU3], s1[kU1], sb1, sb2, sb3;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        s2[kt][v] = slot_load<T>(m_w2, v_w2, tw2,
                                 (size_t)(jt2 + mfma_acc<T>::row(lane2, v)) * kH + 16 * (kt0 + kt) + li2,
                                 blend);
#pragma unroll
    for (int u = 0; u < kU3; ++u) {
      const int e = t + NT * u;
      s3[u].m = s3[u].v = s3[u].target = (T)0;
      if (e < O * kH) s3[u] = slot_load<T>(m_w3, v_w3, tw3, (size_t)e, blend);
    }
#pragma unroll
    for (int u = 0; u < kU1; ++u) {
      const int e = t + NT * u;
      s1[u].m = s1[u].v = s1[u].target = (T)0;
      if (e < kH * D) s1[u] = slot_load<T>(m_w1, v_w1, tw1, (size_t)e, blend);
    }
    sb1.m = sb1.v = sb1.target = sb2.m = sb2.v = sb2.target = sb3.m = sb3.v = sb3.target = (T)0;
    if (t < kH) {
      sb1 = slot_load<T>(m_b1, v_b1, tb1, (size_t)t, blend);
      sb2 = slot_load<T>(m_b2, v_b2, tb2, (size_t)t, blend);
    }
    if (t < O) sb3 = slot_load<T>(m_b3, v_b3, tb3, (size_t)t, blend);

    lds_barrier();
    forward32<T, NT>(L, L.q, D, O, t);
    if (R.debug_stage == 1) return;

    // ---- loss gradient at the output ----------------------------------------------------------
    // mean over the marked samples and the O outputs of (out - target)^2: 2 (out - y) / (count O)
    const uint8_t* const mask = R.sample_mask ? R.sample_mask + (size_t)j * kB : nullptr;
    int count = kB;
    if (mask) {
      count = 0;
      for (int s = 0; s < kB; ++s) count += mask[s] ? 1 : 0;
      count = count > 0 ? count : 1;
    }
    const T scale = (T)1 / (T)(count * O);
    const T* const y = (const T*)R.targets + (size_t)(j / R.tgt_div) * kB * O;
    staged_copy<(kB * kMaxO + NT - 1) / NT, NT, T>(y, kB * O, t, [&](int e, T yv) {
      const int s = e / O;
      const T d = L.q[e] - yv;
      L.q[e] = (!mask || mask[s]) ? ((T)2 * d) * scale : (T)0;   // delta3
    });
    lds_barrier();

    // ---- Adam constants of this network (its own step count) ------------------------------------
    adam_consts<T> c;
    {
      const double st = R.steps[j] + 1.0;
      const T bc1 = (T)(1.0 - pow(R.beta1, st));
      c.bc2_sqrt = (T)sqrt(1.0 - pow(R.beta2, st));
      c.step_size = (T)R.lr / bc1;
      c.one_m_b1 = (T)(1.0 - R.beta1);
      c.b2 = (T)R.beta2;
      c.one_m_b2 = (T)(1.0 - R.beta2);
      c.eps = (T)R.eps;
      c.wd = (T)R.weight_decay;
      c.has_wd = R.weight_decay != 0.0;
      c.tau = (T)R.tau;
      c.blend = blend;
    }

    // ---- output layer: dW3[a][k] = sum_s delta3[s][a] h2[s][k], db3[a] = sum_s delta3[s][a] ----
    // (the new weights go to memory now and into LDS once delta2 has used the old ones)
    T new_w3[kU3];
#pragma unroll
    for (int u = 0; u < kU3; ++u) {
      const int e = t + NT * u;
      new_w3[u] = (T)0;
      if (e < O * kH) {
        const int a = e >> 6, k = e & 63;
        T g = (T)0;
#pragma unroll 8
        for (int s = 0; s < kB; ++s) g = fma_t<T>(L.q[s * O + a], L.h2[s * kRow + k], g);
        new_w3[u] = adam_apply<T>(w3, m_w3, v_w3, tw3, (size_t)e, L.w3[e], g, s3[u], c);
      }
    }
    T new_b3 = (T)0;
    if (t < O) {
      T gb = (T)0;
      for (int s = 0; s < kB; ++s) gb = gb + L.q[s * O + t];
      new_b3 = adam_apply<T>(b3, m_b3, v_b3, tb3, (size_t)t, L.b3[t], gb, sb3, c);
    }
    lds_barrier();
    // delta2[s][k] = (sum_a W3[a][k] delta3[s][a]) * (h2[s][k] > 0), in place over h2
    for (int e = t; e < kB * kH; e += NT) {
      const int s = e >> 6, k = e & 63;
      T d = (T)0;
      for (int a = 0; a < O; ++a) d = fma_t<T>(L.w3[a * kH + k], L.q[s * O + a], d);
      const T h = L.h2[s * kRow + k];
      L.h2[s * kRow + k] = h > (T)0 ? d : (T)0;
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < kU3; ++u) {
      const int e = t + NT * u;
      if (e < O * kH) L.w3[e] = new_w3[u];
    }
    if (t < O) L.b3[t] = new_b3;
    if (R.debug_stage == 2) return;

    // ---- second layer: dW2[j][k] = sum_s delta2[s][j] h1[s][k] (MFMA) ---------------------------
    T new_w2[2][4];
    {
      typedef typename mfma_acc<T>::type acc_t;
      const int lq = lane2 >> 4;
      acc_t g2[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) g2[a] = acc_t{(T)0, (T)0, (T)0, (T)0};
#pragma unroll 2
      for (int s0 = 0; s0 < kB; s0 += 4) {
        const T a = L.h2[(s0 + lq) * kRow + jt2 + li2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          g2[kt] = mfma(a, L.h1[(s0 + lq) * kRow + 16 * (kt0 + kt) + li2], g2[kt]);
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int jj = jt2 + mfma_acc<T>::row(lane2, v), k = 16 * (kt0 + kt) + li2;
          new_w2[kt][v] = adam_apply<T>(w2, m_w2, v_w2, tw2, (size_t)jj * kH + k,
                                        L.wt2[k * kRow + jj], g2[kt][v], s2[kt][v], c);
        }
      if (t < kH) {
        T gb = (T)0;
        for (int s = 0; s < kB; ++s) gb = gb + L.h2[s * kRow + t];
        L.b2[t] = adam_apply<T>(b2, m_b2, v_b2, tb2, (size_t)t, L.b2[t], gb, sb2, c);
      }
    }
    if (R.debug_stage == 3) return;
    lds_barrier();
    // delta1[s][k] = (sum_j delta2[s][j] W2[j][k]) * (h1[s][k] > 0), in place over h1, from the
    // weights this step started from (LDS still holds them)
    {
      typedef typename mfma_acc<T>::type acc_t;
      const int lq = lane2 >> 4;
      acc_t d0 = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll 4
      for (int j0 = 0; j0 < kH; j0 += 4)
        d0 = mfma(L.h2[(st2 + li2) * kRow + j0 + lq], L.wt2[(jt2 + li2) * kRow + j0 + lq], d0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = st2 + mfma_acc<T>::row(lane2, v), k = jt2 + li2;
        const T h0 = L.h1[r * kRow + k];
        L.h1[r * kRow + k] = h0 > (T)0 ? d0[v] : (T)0;
      }
    }
    lds_barrier();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        L.wt2[(16 * (kt0 + kt) + li2) * kRow + jt2 + mfma_acc<T>::row(lane2, v)] = new_w2[kt][v];

    if (R.debug_stage == 4) return;
    // ---- first layer: dW1[j][d] = sum_s delta1[s][j] x[s][d] ------------------------------------
#pragma unroll
    for (int u = 0; u < kU1; ++u) {
      const int e = t + NT * u;
      if (e < kH * D) {
        const int jj = e / D, d = e - jj * D;
        T g = (T)0;
#pragma unroll 8
        for (int s = 0; s < kB; ++s) g = fma_t<T>(L.h1[s * kRow + jj], L.x[s * D + d], g);
        L.wt1[d * kH + jj] =
            adam_apply<T>(w1, m_w1, v_w1, tw1, (size_t)e, L.wt1[d * kH + jj], g, s1[u], c);
      }
    }
    if (t < kH) {
      T gb = (T)0;
      for (int s = 0; s < kB; ++s) gb = gb + L.h1[s * kRow + t];
      L.b1[t] = adam_apply<T>(b1, m_b1, v_b1, tb1, (size_t)t, L.b1[t], gb, sb1, c);
    }
    if (t == 0) R.steps[j] = R.steps[j] + 1.0;
  }

  // ---- outputs of the (updated) network for the extra rows -------------------------------------
  if (R.ep_out && R.ep_rows > 0) {
    lds_barrier();   // LDS holds the current parameters; x / h1 / h2 are free
    const int E = R.ep_rows;
    for (int e = t; e < E * D; e += NT) {
      const int r = e / D, d = e - r * D;
      L.x[e] = R.ep_table ? (T)R.ep_table[(size_t)R.ep_index[j / R.ep_div] * D + d]
                          : ((const T*)R.ep_dense)[((size_t)j * E + r) * D + d];
    }
    lds_barrier();
    {
      const int r = t >> 6, k = t & 63;   // rows x 64 neurons (waves beyond the rows idle)
      if (r < E) {
        T acc = L.b1[k];
        for (int d = 0; d < D; ++d) acc = fma_t<T>(L.wt1[d * kH + k], L.x[r * D + d], acc);
        L.h1[r * kRow + k] = acc > (T)0 ? acc : (T)0;
      }
      lds_barrier();
      if (r < E) {
        T acc = L.b2[k];
#pragma unroll 8
        for (int kk = 0; kk < kH; ++kk) acc = fma_t<T>(L.wt2[kk * kRow + k], L.h1[r * kRow + kk], acc);
        L.h2[r * kRow + k] = acc > (T)0 ? acc : (T)0;
      }
      lds_barrier();
      for (int e = t; e < E * O; e += NT) {
        const int rr = e / O, a = e - rr * O;
        T acc = L.b3[a];
#pragma unroll 8
        for (int kk = 0; kk < kH; ++kk) acc = fma_t<T>(L.w3[a * kH + kk], L.h2[rr * kRow + kk], acc);
        ((T*)R.ep_out)[((size_t)j * E + rr) * O + a] = acc;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kFitThreads) void k_mlp_fit(const fit_args A) {
  mlp_fit_body<T>(A);
}

int shape_ok(int32_t D, int32_t O, const char* who) {
  COBEL_REQUIRE(D >= 1 && D <= kMaxD && O >= 1 && O <= kMaxO, COBEL_E_UNSUPPORTED,
                "%s: Linear(D <= %d, 64)-ReLU-Linear(64, 64)-ReLU-Linear(64, O <= %d) on batches "
                "of 32 is covered (got D %d, O %d)", who, kMaxD, kMaxO, D, O);
  return COBEL_OK;
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
    *lds_bytes = (int32_t)(net_lds_elems(n_inputs, n_outputs) * (is_float64 ? 8 : 4));
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
  if (r.n == 0) return COBEL_OK;
  fwd_args A;
  A.r = r;
  hipStream_t st = (hipStream_t)stream;
  if (r.is_float64) {
    if (int rc = raise_lds(&k_mlp_forward<double>, lds)) return rc;
    hipLaunchKernelGGL(k_mlp_forward<double>, dim3(r.n), dim3(256), lds, st, A);
  } else {
    hipLaunchKernelGGL(k_mlp_forward<float>, dim3(r.n), dim3(256), lds, st, A);
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
  if (r.n == 0) return COBEL_OK;
  fit_args A;
  A.r = r;
  hipStream_t st = (hipStream_t)stream;
  lds += cobel_debug_lds_pad(lds, 160 * 1024);   // (occupancy experiments only)
  if (r.is_float64) {
    if (int rc = raise_lds(&k_mlp_fit<double>, lds)) return rc;
    hipLaunchKernelGGL(k_mlp_fit<double>, dim3(r.n), dim3(kFitThreads), lds, st, A);
  } else {
    hipLaunchKernelGGL(k_mlp_fit<float>, dim3(r.n), dim3(kFitThreads), lds, st, A);
  }
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
