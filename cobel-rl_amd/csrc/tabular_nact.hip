// Q-learning on worlds whose action count is not four — one WAVEFRONT per instance.
//
// The wavefront kernels of tabular.hip are laid out for four actions (16-byte Q rows, packed
// world records, 48 CDF thresholds).  A Topology's action space is the neighbour count of its
// start node (interface/topology.py:110-112): six on the hexagonal graphs of
// misc/topology_tools.py:175-272.  Such runs used to take k_tab_general — one LANE per instance,
// every table in HBM: each of a step's 1 + B TD updates two dependent trips to memory for 4-24
// useful bytes of a 128-byte line (4.3e8 env-steps/s on the six-action bench leg, 7 % of the
// byte roofline of SURVEY 8d).  Here the tables of an instance live in LDS for the whole call:
//   * Q rows padded to eight floats (32 B; the pad cells hold -inf, so the row maximum and the tie
//     pattern need no action count), the world as u16 next[S][8] + {reward, terminal}[S];
//   * epsilon-greedy (policy/greedy.py:40-88) without floating point, as in k_tab_wpi: for every
//     tie pattern of the A values the A - 1 thresholds ceil(cdf_k 2^53) of the normalised float64
//     CDF, worked out once per launch by the workgroup itself (sequential float64 cumulative sum,
//     correctly rounded division — what np.cumsum and Generator.choice compute) into an LDS table;
//     a step reads its pattern's row and counts the thresholds the 53-bit draw has passed;
//   * lane k < A evaluates successor k of the current state ahead of the action draw (next state,
//     its reward / terminal record, the maximum of its Q row): the step pays one LDS round trip,
//     the chosen successor is a readlane away;
//   * the replay batch (agent/q.py:344-354: B indices into the experience log, one vector draw) is
//     one lane per update; the logged records are gathered from HBM one step ahead (the memory
//     stream is counter based) and patched with the two records the gather cannot have seen; the
//     B sequential float32 TD updates run as speculative rounds, the lanes to hold back found in
//     the Q table itself (tags, round 6), as in k_tab_wpi / k_tab_pwg.
// Same streams, counters, arithmetic and order of effects as k_tab_general: identical Q tables,
// logs, counters and monitors (tests/test_gpu_general.py, scripts/fuzz_topology.py).
//
// Round 5: nine to 32 actions on the same kernel with rows of 16 / 32 values (template W; up to
// 1 024 / 512 states).  No threshold table
// there (2^A tie patterns): the wave works the selection's float64 CDF out itself, in the order
// cobel_eps_greedy_select_n (cobel_policy.h) states it — lane k holds value k, the cumulative sum is
// ONE chain of W additions every lane runs, lane k keeps entry k, divides by the last entry and
// compares with the draw — and action masks (an LDS copy) and per-instance parameter sets take part
// in it (such runs on rows of 8 take this path too); the
// replayed updates read a row's maximum with W / 4 LDS reads (and once more, as unsigned integers,
// when a cell changed: the tag test).
// Masked twelve-action QAgent on a 256-node graph, B 32: 8.1e8 env-steps/s against the 1.9e8 of
// k_tab_general (bench.py general_wide_q / general_wide_q_lane).
//
// Reference behaviour restated: agent/q.py:160-228 (train), :289-315 (update_q), :344-354 (replay);
// interface/topology.py:126-157 (step); policy/greedy.py:40-88.
#include <math.h>
#include <stdlib.h>

#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

struct nact_args {
  const uint16_t* next_n;      // [W][S][A]
  const float* reward_s;       // [W][S]
  const uint8_t* terminal_s;   // [W][S]
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds, A;
  int32_t wpg;                 // wavefronts (instances) per workgroup
  int32_t shared_world;        // one world: the workgroup keeps ONE copy of it
  cobel_tab_run_t r;
  float alpha_f, gamma_f;
};


__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float max8(const float4 a, const float4 b) {
  return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
}

// QAgent replay record of the general kernel (general.hip, log_pack_n): lo = f32 reward,
// hi = s | ns << 14 | action << 28 | nonterminal << 30 up to four actions, << 31 beyond
__device__ __forceinline__ uint64_t log_pack8(float r, uint32_t s, uint32_t a, uint32_t ns,
                                              uint32_t nt, uint32_t nt_shift) {
  return (uint64_t)fbits(r) | ((uint64_t)(s | (ns << 14) | (a << 28) | (nt << nt_shift)) << 32);
}

__host__ __device__ inline size_t nact_thr_words(int A) {
  return A > 1 ? ((size_t)1 << A) * (size_t)(A - 1) : 1;
}
// the row width an action count is served with: rows of 8, 16 or 32 values (pad cells: -inf)
__host__ __device__ inline int nact_width(int A) { return A <= 8 ? 8 : (A <= 16 ? 16 : 32); }
// LDS of a workgroup of `wpg` instances (bytes): thresholds (rows of 8) or the action masks (wider
// rows) | worlds | per instance Q + hash
__host__ __device__ inline size_t nact_lds_bytes(int S, int A, int wpg, bool shared, bool masked) {
  const int W = nact_width(A);
  const size_t thr = (W == 8 && !masked) ? (nact_thr_words(A) * 8 + 15) & ~(size_t)15
                                         : (((size_t)S * 4 + 15) & ~(size_t)15);
  const size_t world = (size_t)S * (2 * W + 8);
  return thr + (shared ? world : world * wpg) + (size_t)wpg * ((size_t)S * W * 4);
}
// QAgent replay record of nine to 32 actions (general.hip): hi = s | ns << 13 | action << 26 | nt << 31
__device__ __forceinline__ uint64_t log_pack32(float r, uint32_t s, uint32_t a, uint32_t ns, uint32_t nt) {
  return (uint64_t)fbits(r) | ((uint64_t)(s | (ns << 13) | (a << 26) | (nt << 31)) << 32);
}

// PLAIN: training with a replay batch on a world of more than one action — what the step's
// wave-uniform tests (learn, B > 0, A > 1) are compiled away for (as k_tab_lpi's EXTRA: a taken
// branch is an instruction-buffer refill the wave waits for).
// W: the row width (8: one to eight actions, thresholds from the table; 16 / 32: nine to 32 actions —
// round 5 —, the selection's float64 CDF worked out by the wave in the reference's order, action
// masks from an LDS copy).
// CDF: the selection worked out by the wave (always on the wider rows; on rows of 8 for runs with
// an action mask — the threshold table knows tie patterns only).
template <bool PLAIN, int W, bool CDF = (W > 8)>
__global__ __launch_bounds__(512) void k_tab_wqn(const nact_args G) {
  static_assert(CDF || W == 8, "the threshold table serves rows of eight");
  constexpr uint32_t WU = (uint32_t)W;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int S = G.S, A = PLAIN ? max(G.A, 2) : G.A;
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int i = (int)blockIdx.x * G.wpg + wave;
  const bool present = i < G.r.n;
  const uint32_t g = G.r.instance_base + (uint32_t)(present ? i : 0);
  const int world = (int)(g % (uint32_t)G.n_worlds);
  const size_t wbase = (size_t)world * S;

  // ---- LDS carve-up ---------------------------------------------------------------------------
  unsigned long long* const thr = reinterpret_cast<unsigned long long*>(lds_raw);
  uint32_t* const maskL = reinterpret_cast<uint32_t*>(lds_raw);          // CDF: [S] allowed actions
  size_t off = !CDF ? (nact_thr_words(A) * 8 + 15) & ~(size_t)15 : (((size_t)S * 4 + 15) & ~(size_t)15);
  const size_t wbytes = (size_t)S * (2 * W + 8);
  unsigned char* const wl = lds_raw + off + (G.shared_world ? 0 : (size_t)wave * wbytes);
  off += G.shared_world ? wbytes : wbytes * (size_t)G.wpg;
  uint16_t* const nextL = reinterpret_cast<uint16_t*>(wl);               // [S][W]
  uint2* const RT = reinterpret_cast<uint2*>(wl + (size_t)S * 2 * W);    // [S] {reward bits, terminal}
  unsigned char* const mine = lds_raw + off + (size_t)wave * ((size_t)S * W * 4);
  float4* const Qs = reinterpret_cast<float4*>(mine);                    // [S][W / 4]
  float* const Qf = reinterpret_cast<float*>(mine);

  // ---- the threshold table (whole workgroup) ----------------------------------------------------
  // thr[t * (A - 1) + k] = ceil(cdf_k * 2^53) of the tie pattern t (bit a set: action a attains
  // the maximum): probs = eps / A + ((1 - eps) * tie) / n_ties, sequential cumulative sum, divided
  // by its last entry (greedy.py:83-86, Generator.choice) — cobel_make_eps_consts for A values.
  // (action masks: one byte per state up to eight actions, one 32-bit word beyond)
  const uint8_t* const amask_g = (CDF && (G.r.flags & COBEL_F_MASK_ACTIONS)) ? G.r.action_mask : nullptr;
  if (CDF) {
    const uint32_t all = A >= 32 ? 0xffffffffu : ((1u << A) - 1u);
    for (int s = (int)threadIdx.x; s < S; s += (int)blockDim.x) {
      uint32_t m = all;
      if (amask_g) m &= W == 8 ? (uint32_t)amask_g[s] : reinterpret_cast<const uint32_t*>(amask_g)[s];
      maskL[s] = m;
    }
  }
  if (!CDF && A > 1) {
    const double eps = G.r.epsilon;
    for (int t = (int)threadIdx.x; t < (1 << A); t += (int)blockDim.x) {
      const int nt = __popc((unsigned)t);
      const double base = eps / (double)A;
      const double bonus = nt ? ((1.0 - eps) * 1.0) / (double)nt : 0.0;
      double cum[8];
      double run = 0.0;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const double p = a < A ? base + (((t >> a) & 1) ? bonus : 0.0) : 0.0;
        run = a == 0 ? p : run + p;
        cum[a] = run;
      }
      double total = cum[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) total = a == A - 1 ? cum[a] : total;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        if (k < A - 1) {
          const double cdf = nt ? cum[k] / total : 2.0;   // t = 0 cannot occur
          const double scaled = ceil(ldexp(cdf, 53));
          thr[(size_t)t * (A - 1) + k] =
              scaled >= 18446744073709551615.0 ? ~0ull : (unsigned long long)scaled;
        }
      }
    }
  }
  // ---- the world(s) ---------------------------------------------------------------------------
  {
    const int nthr = G.shared_world ? (int)blockDim.x : 64;
    const int tid = G.shared_world ? (int)threadIdx.x : lane;
    if (G.shared_world || present) {
      for (int e = tid; e < S * W; e += nthr) {
        const int s = e / W, a = e % W;
        nextL[e] = a < A ? G.next_n[(wbase + s) * A + a] : (uint16_t)s;
      }
      for (int s = tid; s < S; s += nthr)
        RT[s] = make_uint2(fbits(G.reward_s[wbase + s]), (uint32_t)G.terminal_s[wbase + s]);
    }
  }
  // ---- this instance's Q table (pad cells: -inf) -------------------------------------------------
  float* const Qg = G.r.q + (size_t)(present ? i : 0) * S * A;
  if (present) {
    for (int e0 = 0; e0 < S * W; e0 += 512) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * 64 + lane;
        const int s = e / W, a = e % W;
        const bool ok = e < S * W && a < A;
        v[j] = Qg[ok ? (size_t)s * A + a : 0];
        if (!ok) v[j] = -__builtin_huge_valf();
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * 64 + lane;
        if (e < S * W) Qf[e] = v[j];
      }
    }
  }
  __syncthreads();
  if (!present) return;

  // ---- scalar state -----------------------------------------------------------------------------
  int32_t* const inst = G.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t cm = (uint32_t)inst[COBEL_I_CTR_MEMORY];
  uint32_t loglen = (uint32_t)inst[COBEL_I_LOG_LEN];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO + (lane & 0));
  asm volatile("" : "+v"(trew));

  const uint32_t flags = G.r.flags;
  const bool learn = PLAIN || (flags & COBEL_F_LEARN) != 0;
  uint64_t* const rlog = G.r.replay_log ? G.r.replay_log + (size_t)i * G.r.log_cap : nullptr;
  const uint32_t cap = rlog ? (uint32_t)G.r.log_cap : 0u;
  // (BT = the batch; B = the updates one pass of the wavefront takes — lane j < B owns update j —,
  //  batches beyond COBEL_MAX_BATCH run as further passes behind the first: see the replay below)
  const int BT = PLAIN ? max(G.r.batch, 1)
                       : ((learn && !(flags & COBEL_F_NO_REPLAY) && rlog) ? G.r.batch : 0);
  const int B = BT > COBEL_MAX_BATCH ? COBEL_MAX_BATCH : BT;
  const uint32_t pol_stream =
      (flags & COBEL_F_TEST_STREAM) ? COBEL_STREAM_POLICY_TEST : COBEL_STREAM_POLICY;
  uint64_t seed = G.r.seed;
  asm volatile("" : "+v"(seed));
  const int start_lo = G.start_off[world];
  const uint32_t start_cnt = (uint32_t)(G.start_off[world + 1] - start_lo);
  // hyper-parameters: launch-wide, or this instance's parameter set (grid-search fan-out; the
  // selection is then the wave's own CDF — the threshold table is one epsilon's)
  float alpha_f = G.alpha_f, gamma_f = G.gamma_f;
  double eps_sel = G.r.epsilon;
  if (CDF && G.r.param_index) {
    const int kset = (int)G.r.param_index[i];
    const cobel_param_set_t* const P =
        G.r.param_sets + (kset < G.r.n_param_sets ? kset : G.r.n_param_sets - 1);
    alpha_f = P->alpha_f;
    gamma_f = P->gamma_f;
    eps_sel = P->epsilon;
  }
  asm volatile("" : "+v"(alpha_f), "+v"(gamma_f));
  const uint32_t nt_shift = A <= 4 ? 30u : 31u, a_mask = A <= 4 ? 3u : 7u;
  // the maximum of a row of W values (pad cells hold -inf)
  auto row_max = [&](uint32_t row) -> float {
    float4 v = Qs[row * (WU / 4u)];
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
    for (uint32_t j = 1; j < WU / 4u; ++j) {
      v = Qs[row * (WU / 4u) + j];
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    return m;
  };

  // Cached Philox output: lanes < B hold memory block mb_idx (four consecutive batches), lanes 62
  // and 63 the policy blocks 2 pb_idx and 2 pb_idx + 1 (action draws 4 pb_idx .. 4 pb_idx + 3).
  // The memory block is the one of the NEXT batch (cm + 1): its records are gathered a step ahead.
  cobel_u4 blk = {0, 0, 0, 0};
  uint32_t mb_idx = ~0u, pb_idx = ~0u;
  auto refresh = [&]() {
    const uint32_t pq = cp >> 2, mi = (cm + 1u) >> 2;
    if (__builtin_expect(pq == pb_idx && (B == 0 || mi == mb_idx), 1)) return;
    const bool p0 = lane == 62, p1 = lane == 63;
    blk = cobel_philox(p1 ? 2u * pq + 1u : (p0 ? 2u * pq : mi), (p0 || p1) ? 0u : (uint32_t)lane,
                       g, (p0 || p1) ? pol_stream : COBEL_STREAM_MEMORY, seed);
    pb_idx = pq;
    mb_idx = mi;
  };
  // the record batch `c` replays in lane j from a log of `len` entries, or 0 (gather: entries that
  // exist in memory now, i.e. below `have`)
  auto gather = [&](uint32_t word, uint32_t len, uint32_t have, uint32_t& idx) -> uint64_t {
    uint64_t rec = 0;
    idx = 0u;
    if (lane < B && len > 0u) {
      idx = cobel_bounded(word, len);
      if (idx < have) rec = rlog[idx];
    }
    return rec;
  };

  // ---- B sequential float32 TD updates (q.py:305-313), speculative rounds ------------------------
  // (nb = the updates of this pass: B, fewer in the last of several passes)
  auto run_batch = [&](uint64_t rec, const int nb) {
    const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
    const uint32_t s = W == 8 ? hi & 0x3fffu : hi & 0x1fffu;
    const uint32_t ns = W == 8 ? (hi >> 14) & 0x3fffu : (hi >> 13) & 0x1fffu;
    const uint32_t a = W == 8 ? (hi >> 28) & a_mask : (hi >> 26) & 31u;
    const uint32_t nt = W == 8 ? (hi >> nt_shift) & 1u : hi >> 31;
    const float r = __builtin_bit_cast(float, lo);
    const uint32_t p = s * WU + a;
    const bool on = lane < nb;
    auto td_of = [&](float q) -> float {
      const float m = W == 8 ? max8(Qs[ns * 2u], Qs[ns * 2u + 1u]) : row_max(ns);
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * m;
      td = td - q;
      return q + alpha_f * td;
    };
    // Who is held back is found IN the table (round 6, as in k_tab_pwg / k_tab_wpi; until then two
    // tables of lane masks beside every Q table): a lane that changes its cell raises it to the TAG
    // ~lane with ds_max_u32 (bit patterns 0xffffffc0 .. 0xffffffff: above every float, the pad
    // cells' -inf included, that is not a NaN of exactly that payload) — the cell then holds the
    // tag of the EARLIEST lane that writes it; every lane reads its row and its cell again and is
    // held back iff one of them is a tag above its own.  The earliest writer of a cell stores the
    // new value (committed) or puts the old one back (held back).
    uint32_t* const Qu = reinterpret_cast<uint32_t*>(Qf);
    const uint4* const Qs4u = reinterpret_cast<const uint4*>(Qs);
    const uint32_t tag_mine = ~(uint32_t)lane;
    // (`lo` = the first lane of the round: committed regardless — only a table that held a tag
    //  pattern to begin with could hold it back —, so every round ends one lane further)
    auto tag_round = [&](bool act, bool ch, float q, float qn, int lo) -> int {
      if (ch) atomicMax(&Qu[p], tag_mine);
      __builtin_amdgcn_wave_barrier();
      uint32_t t = 0u, c2 = 0u;
      if (act) {
        c2 = Qu[p];
        t = c2;
#pragma unroll
        for (uint32_t j = 0; j < WU / 4u; ++j) {
          const uint4 v = Qs4u[ns * (WU / 4u) + j];
          t = max(max(max(v.x, v.y), max(v.z, v.w)), t);
        }
      }
      const unsigned long long blocked = __builtin_amdgcn_ballot_w64(act && t > tag_mine);
      const int stop = max(blocked ? __ffsll((long long)blocked) - 1 : nb, lo + 1);
      if (ch && c2 == tag_mine) Qf[p] = lane < stop ? qn : q;
      __builtin_amdgcn_wave_barrier();
      return stop;
    };
    // (the first round on its own, in front of the loop over the rounds — most batches end with
    //  it —, as in k_tab_pwg)
    int first;
    {
      float q = 0.0f, qn = 0.0f;
      if (on) {
        q = Qf[p];
        qn = td_of(q);
      }
      const bool ch = on && fbits(qn) != fbits(q);
      if (!__builtin_amdgcn_ballot_w64(ch)) return;
      first = tag_round(on, ch, q, qn, 0);
    }
    while (first < nb) {
      const bool act = on && lane >= first;
      float q = 0.0f, qn = 0.0f;
      if (act) {
        q = Qf[p];
        qn = td_of(q);
      }
      const bool ch = act && fbits(qn) != fbits(q);
      if (!__ballot(ch)) return;
      first = tag_round(act, ch, q, qn, first);
    }
  };

  // what depends only on the state being entered: lane k < W evaluates successor k
  uint32_t sn = 0, srw = 0, ste = 0;
  float smax = 0.0f;
  auto enter_state = [&](int s) {
    sn = nextL[(uint32_t)s * WU + (uint32_t)(lane & (W - 1))];
    const uint2 rt = RT[sn];
    srw = rt.x;
    ste = rt.y;
  };
  auto begin_trial = [&]() -> bool {
    if (trial >= G.r.trials_target) return false;
    state = (int)G.starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed,
                                                             start_cnt)];
    ce += 1u;
    step = 0;
    trew = 0.0;
    asm volatile("" : "+v"(trew));
    iflags |= 1u;
    return true;
  };

  // ---- prologue -----------------------------------------------------------------------------------
  bool live = (iflags & 1u) ? true : begin_trial();
  // the batch of the first step: drawn over the log as it will be after that step's append
  uint32_t idx_cur = 0u;
  uint64_t rec_cur = 0;
  uint64_t fresh_prev = 0;          // the latest record this launch appended before the current step
  bool have_prev = false;
  if (B > 0 && live) {
    const uint32_t len1 = loglen + ((learn && loglen < cap) ? 1u : 0u);
    const cobel_u4 b0 = cobel_philox(cm >> 2, (uint32_t)lane, g, COBEL_STREAM_MEMORY, seed);
    rec_cur = gather(cobel_word(b0, cm & 3u), len1, loglen, idx_cur);
  }
  const size_t mstripe = cobel_mon_offset(G.r.mon_stripes, G.r.trial_cap);
  int budget = G.r.step_budget > 0 ? G.r.step_budget : 0x7fffffff;
  unsigned long long executed = 0ull;
  uint32_t batches = 0u;

  while (live) {
    if (budget == 0) break;
    budget -= 1;
    refresh();
    // ---- successors of the current state, and its own row -------------------------------------------
    enter_state(state);
    // (lane l holds value l % 8 of the padded row Q[state]: the row's maximum is three fused DPP
    //  steps — two quad permutes and a half-row mirror —, the tie pattern the low eight bits of ONE
    //  ballot and Q[s][a] a readlane, where eight compares, fifteen scalar instructions and seven
    //  selects stood; as in k_tab_pwg / k_tab_wpi)
    const float qc = Qf[(uint32_t)state * WU + (uint32_t)(lane & (W - 1))];
    smax = W == 8 ? max8(Qs[sn * 2u], Qs[sn * 2u + 1u]) : row_max(sn);
    // ---- select --------------------------------------------------------------------------------------
    const int src_lane = 62 + (int)((cp >> 1) & 1u);
    const uint32_t w0 = rl((cp & 1u) ? blk.z : blk.x, src_lane);
    const uint32_t w1 = rl((cp & 1u) ? blk.w : blk.y, src_lane);
    cp += 1u;
    int a = 0;
    if (CDF) {
      // policy/greedy.py:77-86 + Generator.choice as cobel_eps_greedy_select_n states them, by the
      // wave: lane k holds value k of the row; float64 probabilities eps / n + (tie ? (1 - eps) / n_ties
      // : 0) of the allowed actions, their cumulative sum IN ORDER (one add per action, every lane
      // the same chain, lane k keeps entry k), normalised by the last entry, searchsorted(side='right')
      const uint32_t k = (uint32_t)(lane & (W - 1));
      const uint32_t allowed = rfl(maskL[state]);
      const bool ok = ((allowed >> k) & 1u) != 0u;
      const float q_ok = ok ? qc : -__builtin_huge_valf();
      float m;
      // (the maximum over a row of 16 lanes: two quad permutes, a half-row and a row mirror, each
      //  fused into the maximum; every lane of the wave is active here)
      asm("s_nop 1\n\t"
          "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
          : "=&v"(m)
          : "v"(q_ok));
      if (W > 8) {
        asm("s_nop 1\n\t"
            "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
            : "+v"(m));
      }
      if (W > 16) m = fmaxf(m, __shfl_xor(m, 16));
      const uint32_t ties = (uint32_t)__ballot(ok && qc == m) & (W >= 32 ? 0xffffffffu : ((1u << W) - 1u));
      const int n = __popc(allowed), n_ties = __popc(ties);
      const double eps = eps_sel;
      const double base = eps / (double)n;
      const double bonus = ((1.0 - eps) * 1.0) / (double)n_ties;
      const double both = base + bonus;
      double run = 0.0, ck = 0.0;
#pragma unroll
      for (uint32_t j = 0; j < WU; ++j) {
        const double pj = ((allowed >> j) & 1u) ? (((ties >> j) & 1u) ? both : base) : 0.0;
        run = j == 0u ? pj : run + pj;
        ck = k == j ? run : ck;
      }
      const unsigned long long cb = __builtin_bit_cast(unsigned long long, ck);
      const double total = __builtin_bit_cast(
          double, ((unsigned long long)rl((uint32_t)(cb >> 32), A - 1) << 32) | rl((uint32_t)cb, A - 1));
      const double u = cobel_u01(w0, w1);
      const bool pass = (int)k < A - 1 && ck / total <= u;
      a = __popc((uint32_t)__ballot(pass) & (W >= 32 ? 0xffffffffu : ((1u << W) - 1u)));
    } else if (A > 1) {
      float m;
      // (two wait states between a VALU write of a register and a DPP read of it: the compiler does
      //  not look for hazards inside an asm block.  Every lane of the wave is active here — a
      //  permute that reads a disabled lane leaves its destination unwritten)
      asm("s_nop 1\n\t"
          "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
          : "=&v"(m)
          : "v"(qc));
      const uint32_t t = (uint32_t)__ballot(qc == m) & 255u;
      const uint64_t K = cobel_u53(w0, w1);
      const unsigned long long T = lane < A - 1 ? thr[(size_t)t * (A - 1) + lane] : ~0ull;
      a = __popcll(__ballot(lane < A - 1 && T <= K));
    }
    // ---- env.step (interface/topology.py:126-157) ------------------------------------------------------
    const int ns = (int)rl(sn, a);
    const uint32_t r_bits = rl(srw, a);
    const float r = __builtin_bit_cast(float, r_bits);
    const uint32_t end = rl(ste, a);
    const uint32_t nt = 1u - end;
    const float ns_max = __builtin_bit_cast(float, rl(fbits(smax), a));
    const uint32_t p_sa = (uint32_t)state * WU + (uint32_t)a;
    const bool trial_over = end || (step + 1 >= G.r.steps_per_trial);
    // ---- online TD (q.py:305-313, float32) and the log (q.py:213) ----------------------------------------
    uint64_t fresh_cur = 0;
    bool appended = false;
    if (learn) {
      const float q_sa = __builtin_bit_cast(float, rl(fbits(qc), a));
      const float gnt = nt ? gamma_f : 0.0f;
      float td = r + gnt * ns_max;
      td = td - q_sa;
      const float qn = q_sa + alpha_f * td;
      fresh_cur = W == 8 ? log_pack8(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt, nt_shift)
                         : log_pack32(r, (uint32_t)state, (uint32_t)a, (uint32_t)ns, nt);
      appended = loglen < cap;      // (a full log takes no record)
      if (lane == 0) {
        Qf[p_sa] = qn;
        if (appended) rlog[loglen] = fresh_cur;
      }
      if (appended) loglen += 1u;
      __builtin_amdgcn_wave_barrier();
    }
    trew += (double)r;
    executed += 1ull;
    if (!PLAIN && G.r.occupancy && lane == 0) atomicAdd(G.r.occupancy + wbase + ns, 1ull);
    state = ns;
    // ---- replay: this step's batch (gathered a step ago), the next one's gather ----------------------------
    if (B > 0) {
      uint64_t rec_next = 0;
      uint32_t idx_next = 0u;
      {
        const uint32_t len2 = loglen + ((loglen < cap) ? 1u : 0u);
        rec_next = gather(cobel_word(blk, (cm + 1u) & 3u), len2, loglen, idx_next);
      }
      if (loglen > 0u) {
        batches += 1u;
        // what the gather of a step ago cannot have seen: the record this step appended (index
        // loglen - 1) and, its store possibly still in flight then, the one before it
        uint64_t rec = rec_cur;
        if (appended) {
          if (idx_cur + 1u == loglen) rec = fresh_cur;
          else if (idx_cur + 2u == loglen && have_prev) rec = fresh_prev;
        } else if (idx_cur + 1u == loglen && have_prev) {
          rec = fresh_prev;
        }
        run_batch(rec, B);
        // updates COBEL_MAX_BATCH .. BT - 1 of this step's batch (agent/q.py:344-354 draws ONE vector of
        // indices: element j comes from sub-stream j), gathered here — the log as it stands, this
        // step's own record from registers — and applied pass by pass, in order
        if (__builtin_expect(BT > COBEL_MAX_BATCH, 0)) {
          for (int j0 = COBEL_MAX_BATCH; j0 < BT; j0 += COBEL_MAX_BATCH) {
            const int nb = BT - j0 < COBEL_MAX_BATCH ? BT - j0 : COBEL_MAX_BATCH;
            const cobel_u4 b = cobel_philox(cm >> 2, (uint32_t)(j0 + lane), g, COBEL_STREAM_MEMORY, seed);
            uint64_t rec2 = 0;
            if (lane < nb) {
              const uint32_t idx2 = cobel_bounded(cobel_word(b, cm & 3u), loglen);
              rec2 = (appended && idx2 + 1u == loglen) ? fresh_cur : rlog[idx2];
            }
            run_batch(rec2, nb);
          }
        }
      }
      cm += 1u;
      rec_cur = rec_next;
      idx_cur = idx_next;
      if (appended) {
        fresh_prev = fresh_cur;
        have_prev = true;
      }
    }
    if (__builtin_expect(trial_over, 0)) {
      if (lane == 0 && trial >= 0 && trial < G.r.trial_cap) {
        const size_t mo = mstripe + (size_t)trial;
        if (G.r.lat_sum) atomicAdd(G.r.lat_sum + mo, (unsigned long long)step);
        if (G.r.lat_cnt) atomicAdd(G.r.lat_cnt + mo, 1ull);
        if (G.r.reward_sum) atomicAdd(G.r.reward_sum + mo, trew);
        if (G.r.resp_cnt && trew > 0.0) atomicAdd(G.r.resp_cnt + mo, 1ull);
        if (G.r.lat_trace) G.r.lat_trace[(size_t)i * G.r.trial_cap + trial] = step;
      }
      trial += 1;
      iflags &= ~1u;
      if (!begin_trial()) break;
    } else {
      step += 1;
    }
  }

  // ---- write back ---------------------------------------------------------------------------------
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < S * W; e += 64) {
    const int s = e / W, a = e % W;
    if (a < A) Qg[(size_t)s * A + a] = Qf[e];
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
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) += executed;
    if (G.r.steps_done && executed) atomicAdd(G.r.steps_done, executed);
    if (G.r.batches_done && batches) atomicAdd(G.r.batches_done, (unsigned long long)batches);
  }
}

// waves per workgroup and the LDS they take; false: the run is not covered
bool nact_plan(const cobel_world* world, const cobel_tab_run_t& r, int* wpg_out, size_t* lds_out) {
  const int S = world->n_states, A = world->n_actions;
  const int W = nact_width(A);
  const bool masked = (r.flags & COBEL_F_MASK_ACTIONS) != 0;
  // (the packed log record holds 14-bit states on rows of 8, 13-bit states beyond; the table sizes
  //  below are the tested ones)
  if (r.agent != COBEL_AGENT_Q || A == 4 || A < 1 || A > 32 || !world->next_n || world->succ_off ||
      r.last_exp || (r.param_index && !r.param_sets) ||
      (masked && (!r.action_mask || (W > 8 && ((uintptr_t)r.action_mask & 3u)))) ||
      (r.flags & (COBEL_F_TAB_GENERAL | COBEL_F_EPISODIC)) ||
      S > 1024 || (size_t)S * W > (W == 8 ? 8192u : 16384u) || r.n < 1)
    return false;
  const bool shared = world->n_worlds == 1;
  int n_cu = 0;
  size_t lds_cu = 0;
  if (cobel_device_limits(world->device, &n_cu, &lds_cu) != COBEL_OK) return false;
  // Instances per workgroup where they share the world's copy: the count (1, 2, 4 or 8) that puts the
  // most wavefronts on a CU — up to four unless eight win (rows of 16 on 256 states: 17.5 KB per
  // instance + 11 KB of world and masks, eight in ONE workgroup where two workgroups of four miss
  // the 160 KiB by 2 KB and three of two make six)
  int wpg = 1;
  if (shared) {
    size_t best = 0;
    for (int w = 1; w <= 8; w <<= 1) {
      const size_t need = nact_lds_bytes(S, A, w, shared, masked || r.param_index != nullptr);
      if (need > lds_cu) break;
      size_t waves = (lds_cu / need) * (size_t)w;
      if (waves > 16) waves = 16;
      if (waves > best || (waves == best && w <= 4)) {
        best = waves;
        wpg = w;
      }
    }
  }
  // (few instances: rather a workgroup on every CU than full workgroups on a few of them)
  while (wpg > 1 && (r.n + wpg - 1) / wpg < n_cu) wpg >>= 1;
  const size_t lds = nact_lds_bytes(S, A, wpg, shared, masked || r.param_index != nullptr);
  if (lds > lds_cu) return false;
  *wpg_out = wpg;
  *lds_out = lds;
  return true;
}

}  // namespace

bool cobel_tab_nact_covers(const cobel_world* world, const cobel_tab_run_t& r, size_t* lds_bytes,
                           int* instances_per_workgroup) {
  int wpg = 0;
  size_t lds = 0;
  if (!nact_plan(world, r, &wpg, &lds)) return false;
  if (lds_bytes) *lds_bytes = lds;
  if (instances_per_workgroup) *instances_per_workgroup = wpg;
  return true;
}

int cobel_tab_nact_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st) {
  int wpg = 0;
  size_t lds = 0;
  if (!nact_plan(world, r, &wpg, &lds))
    return cobel_fail(COBEL_E_UNSUPPORTED, "cobel_tab_nact_launch: run not covered");
  nact_args G;
  G.next_n = world->next_n;
  G.reward_s = world->reward_s;
  G.terminal_s = world->terminal_s;
  G.starts = world->starts;
  G.start_off = world->start_off;
  G.S = world->n_states;
  G.n_worlds = world->n_worlds;
  G.A = world->n_actions;
  G.wpg = wpg;
  G.shared_world = world->n_worlds == 1 ? 1 : 0;
  G.r = r;
  G.alpha_f = (float)r.alpha;
  G.gamma_f = (float)r.gamma;
  const bool plain = (r.flags & COBEL_F_LEARN) && !(r.flags & COBEL_F_NO_REPLAY) && r.replay_log &&
                     r.batch > 0 && G.A > 1 && !r.occupancy;
  const int W = nact_width(G.A);
  const dim3 grid((unsigned)((r.n + wpg - 1) / wpg)), block(64 * wpg);
#define COBEL_WQN(PLAIN, W, CDF)                                                                  \
  do {                                                                                            \
    if (lds > 64 * 1024)                                                                          \
      COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tab_wqn<PLAIN, W, CDF>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
    hipLaunchKernelGGL((k_tab_wqn<PLAIN, W, CDF>), grid, block, lds, st, G);                      \
  } while (0)
  const bool masked = (r.flags & COBEL_F_MASK_ACTIONS) != 0;
  if (W == 8 && (masked || r.param_index)) {
    if (plain) COBEL_WQN(true, 8, true);
    else COBEL_WQN(false, 8, true);
  } else if (W == 8) {
    if (plain) COBEL_WQN(true, 8, false);
    else COBEL_WQN(false, 8, false);
  } else if (W == 16) {
    if (plain) COBEL_WQN(true, 16, true);
    else COBEL_WQN(false, 16, true);
  } else {
    if (plain) COBEL_WQN(true, 32, true);
    else COBEL_WQN(false, 32, true);
  }
#undef COBEL_WQN
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
