// Dyna-Q planning kernel, persistent-workgroup form (plain training runs on worlds whose Q table
// limits the LDS-resident kernel of tabular.hip to fewer instances per CU than the register file
// would hold).
//
// k_tab_wpi keeps an instance's Q table (16 B per state) in LDS, so at 32 x 32 states nine
// instances fit on a CU (17 408 B each).  The step is a chain of dependent work, not a stream:
// measured on trained agents, the same kernel on 24 x 24 mazes padded to 9 / 10 / 11 / 12 / 14 / 16
// instances per CU takes 16.75 / 15.45 / 14.37 / 13.43 / 12.05 / 11.03 ms per launch
// (scripts/experiments/exp_occ_trained.py) — every further resident wave is throughput.  LDS is full at nine;
// registers allow sixteen.  So here ONE workgroup of sixteen wavefronts owns a CU for the whole
// launch: `nl` of its waves keep their instance's Q table in LDS exactly as k_tab_wpi does, the
// other `ng` waves work on the caller's Q table where it lies, in global memory (L2 resident while
// the instance is in flight: 16 KiB).  All waves take instances from one atomic counter until it
// runs out, so the two kinds may run at different speeds.
//
// Q in global memory (QG) without a memory round trip per planning round:
//   * everything a step reads from Q is requested at the top of the step, after the previous
//     step's stores have been acknowledged: Q[s], the rows of the four possible successors, and for
//     the planning lanes the row Q[ns_j] and the cell Q[s_j][a_j] of the pair each will replay
//     (the pairs are known one step ahead, as in k_tab_wpi);
//   * the one cell the online TD update writes before planning reads is patched in registers;
//   * the speculative rounds of k_tab_wpi (agent/dyna_q.py:327-330 applies the B updates one after
//     another) run on those registers: lanes whose inputs were changed by an earlier lane of the
//     round take the new values from the writers' registers (ds_bpermute over their exact conflict
//     set) instead of re-reading the table.  Only changed cells are stored.
// Same draws, same arithmetic, same order of effects as k_tab_wpi: identical tables, digests,
// counters and monitors (tests/test_gpu_pwg.py compares the two kernels and the oracle).
//
// Reference behaviour restated: agent/dyna_q.py:164-215 (train loop), :290-299 (TD), :327-330
// (replay); memory/dyna_q.py:92-96 (store), :137-155 (retrieve_batch).
#include <stdlib.h>
#include <string.h>

#include "cobel_common.h"
#include "cobel_policy.h"

namespace {

constexpr int kHashBytes = 128 * 8;   // two 64-bucket tables of lane masks per global-memory wave
constexpr int kMaxSlices = 8;
constexpr uint32_t kWholeTicket = 0x40000000u;
constexpr uint32_t kWholeThenSlices = 0x80000000u;   // pwg_args::whole: see the ticket draw
constexpr uint32_t kWholeSlice = 0xffu;   // slice number of an instance that runs all its steps as one ticket
// word of the scratch header a sliced launch raises when a wave gives up waiting for a ring entry
// (COBEL_TAB_SCRATCH_ABORT_WORD in cobel_hip.h), and the polls (~1-2 us each) before it does
constexpr uint32_t kAbortWord = COBEL_TAB_SCRATCH_ABORT_WORD;
constexpr uint32_t kRingSpinLimit = 1u << 22;
struct pwg_args {
  const cobel_wrec* rec;
  const uint16_t* starts;
  const int32_t* start_off;
  int32_t S, n_worlds;
  cobel_tab_run_t r;
  uint64_t thr[16][3];   // integer CDF thresholds of the epsilon-greedy tie patterns
  float alpha_f, gamma_f, model_lr_f;
  int32_t nl, ng;        // waves per workgroup with Q in LDS / with Q in global memory
  // tickets (all counters zeroed before the launch)
  uint32_t* queue;       // eight heads, 32 B apart: word 0 = next (sliced) ticket of each queue, word 1 =
                         // next of its whole-instance tickets (sliced launches with `whole` > 0)
  uint32_t reserve;      // tickets per queue the global-memory waves leave to the others
  // slices (n_slices > 1): an instance's steps of this call are cut into n_slices tickets
  int32_t n_slices;
  int32_t slice_steps[kMaxSlices];
  uint32_t whole;        // sliced launches: instances per queue — its first ones — that run unsliced,
                         // ONE ticket each (second counter of the queue), on the global-memory waves
  uint32_t* owner;       // [8] XCD (id + 1) that serves each queue, 0 = nobody yet
  uint32_t* tail;        // eight counters, 32 B apart: entries pushed onto each queue's ring
  uint32_t* ring;        // eight rings of ring_stride words: the tickets after a queue's first nq
  uint32_t ring_stride;
  uint32_t xcc_limit;    // 7; tests: the workgroups of XCDs beyond it sit a sliced launch out, so
                         // that their queues have no XCD of their own
};

#if defined(COBEL_PWG_NT)
#define NT_LD(p) __builtin_nontemporal_load(p)
#define NT_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define NT_LD(p) (*(p))
#define NT_ST(v, p) (*(p) = (v))
#endif

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float max4(const float4 v) {
  float m;
  asm("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4"
      : "=&v"(m)
      : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  return m;
}
__device__ __forceinline__ uint32_t next_of(uint32_t w0, uint32_t w1, int a) {
  const uint32_t w = (a & 2) ? w1 : w0;
  return (a & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
// this wave's earlier global stores are complete before anything after this point is issued
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#if defined(COBEL_PWG_STAMPS)
// (timing experiments, scripts/experiments/exp_pwg_stamps.py: cycles per phase of a ticket, summed per wave
//  into the scratch area behind the counters; the previous stamp waits in a spare LDS word)
__device__ __forceinline__ void pwg_stamp(uint32_t* stamps, uint32_t* prev, int k, int lane) {
  if (lane == 0) {
    const uint32_t now = (uint32_t)__builtin_readcyclecounter();
    atomicAdd(stamps + k, now - *prev);
    *prev = now;
  }
}
#define PWG_STAMP(k) pwg_stamp(stamps, stamp_prev, k, lane)
#else
#define PWG_STAMP(k)
#endif

// The kernel arguments as they lie in the kernarg segment, through a pointer the optimizer cannot
// see through: what only trial ends and the final save read through it (monitor arrays, trial
// counts, start lists) is loaded where it is used instead of being held — and spilled — in scalar
// registers across the step loop.
// (kept in the constant address space: what is read through it is a scalar load and wave-uniform
//  by construction — through a generic pointer the loads are per-lane to the compiler, and a loop
//  whose exit depends on one runs under an exec mask)
typedef const __attribute__((address_space(4))) pwg_args* pwg_kargs;
__device__ __forceinline__ pwg_kargs rare_args() {
#if defined(__HIP_DEVICE_COMPILE__)
  pwg_kargs p = (pwg_kargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
#else
  return nullptr;
#endif
}

// One instance, all the steps of the call.  QG: the Q table stays in global memory.  POW2: the
// state count is a power of two — a bounded draw mulhi32(x, 4 S) is then a shift (v_mul_hi_u32
// holds a SIMD four times as long as a shift).
template <bool QG, bool POW2>
__device__ __forceinline__ void pwg_instance(const pwg_args& A, const int i, const int step_budget,
                                             unsigned char* const lds, const int lane,
                                             const uint64_t thr_mine, const uint32_t stripe
#if defined(COBEL_PWG_STAMPS)
                                             , uint32_t* stamps, uint32_t* stamp_prev
#endif
                                             ) {
  const int S = A.S;
  float4* const Qs = reinterpret_cast<float4*>(lds);
  float* const Qf = reinterpret_cast<float*>(lds);
  unsigned long long* const H = reinterpret_cast<unsigned long long*>(lds);   // (QG only)
  const uint32_t g = A.r.instance_base + (uint32_t)i;
  const int world = (int)(g % (uint32_t)A.n_worlds);
  const uint4* const W4 = reinterpret_cast<const uint4*>(A.rec + (size_t)world * S);
  float4* const Qg = reinterpret_cast<float4*>(A.r.q) + (size_t)i * S;
  float* const Qgf = reinterpret_cast<float*>(Qg);
  uint64_t* const model = A.r.model + (size_t)i * S * 4;
  const uint32_t* const model32 = reinterpret_cast<const uint32_t*>(model);
  const uint32_t SA = (uint32_t)S * 4u;
  uint16_t* const Mg = A.r.model_index + (size_t)i * SA;
  // Lane constants in vector registers where the compiler would keep lane masks in scalar pairs
  // (the step loop is short of those: masks it spilled were reloaded in every step): lane l reads
  // successor l % 4 of a packed record pair as bfe(bfi(sel, w1, w0), sh, 16).
  uint32_t succ_sel = (lane & 2) ? 0xffffffffu : 0u, succ_sh = (uint32_t)(lane & 1) * 16u;
  uint32_t SAv = POW2 ? (uint32_t)__builtin_clz(SA) + 1u : SA;   // (POW2: the shift)
  uint32_t qlane = (uint32_t)(lane & 3);
  asm volatile("" : "+v"(succ_sel), "+v"(succ_sh), "+v"(SAv), "+v"(qlane));
  auto succ_of = [&](uint32_t w0, uint32_t w1) -> uint32_t {
    return (((w1 & succ_sel) | (w0 & ~succ_sel)) >> succ_sh) & 0xffffu;
  };

  // ---- stage ---------------------------------------------------------------------------------
  // (the instance's scalar state is requested first and the table in batches of eight rows per
  //  lane, every request of a batch before the first use: staged row by row — a memory round trip
  //  per row — the 16 rows of a 32 x 32 table were 20 us of an instance's ~500, and of every slice)
  int32_t* const inst = A.r.inst + (size_t)i * COBEL_I_WORDS;
  int state = inst[COBEL_I_STATE];
  int step = inst[COBEL_I_STEP];
  int trial = inst[COBEL_I_TRIAL];
  uint32_t ce = (uint32_t)inst[COBEL_I_CTR_ENV];
  uint32_t cp = (uint32_t)inst[COBEL_I_CTR_POLICY];
  uint32_t cm = (uint32_t)inst[COBEL_I_CTR_MEMORY];
  uint32_t iflags = (uint32_t)inst[COBEL_I_FLAGS];
  double trew = *reinterpret_cast<const double*>(inst + COBEL_I_REWARD_LO + (lane & 0));
  asm volatile("" : "+v"(trew));
  // The table.  LDS waves: straight into LDS (global_load_lds_dwordx4: lane l of request j
  // supplies row 64 j + l, the 64 rows land side by side — the layout of Qs), sixteen requests in
  // flight behind the scalar state, no register in between; read back for the all-zero test once
  // they have landed.  Global-memory waves only look at it (eight rows per lane and trip).
  uint32_t nonzero = 0u;
  if (QG) {
    for (int s0 = 0; s0 < S; s0 += 512) {
      float4 qv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + j * 64 + lane;
        qv[j] = Qg[s < S ? s : S - 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        nonzero |= fbits(qv[j].x) | fbits(qv[j].y) | fbits(qv[j].z) | fbits(qv[j].w);
    }
  } else {
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      if (s < S)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(Qg + s),
            (__attribute__((address_space(3))) void*)(Qs + s0), 16, 0, 0);
    }
  }

  const int B = A.r.batch;
  // The Philox key schedule (twenty values derived from the seed) is formed by the scalar unit
  // where a block is evaluated — once in four steps — instead of living in twenty vector registers
  // across the step loop: the seed is handed out through an opaque scalar move at every use.
#if defined(COBEL_PWG_SSEED)
  auto seed_now = [&]() -> uint64_t {
    uint64_t sd = A.r.seed;
    asm volatile("" : "+s"(sd));
    return sd;
  };
#else
  uint64_t seed_v = A.r.seed;
  asm volatile("" : "+v"(seed_v));
  auto seed_now = [&]() -> uint64_t { return seed_v; };
#endif
  double alpha = A.r.alpha, gamma = A.r.gamma;
  float alpha_f = A.alpha_f, gamma_f = A.gamma_f, mlr_f = A.model_lr_f;
  asm volatile("" : "+v"(alpha), "+v"(gamma), "+v"(alpha_f), "+v"(gamma_f), "+v"(mlr_f));

  // ---- values carried from one step to the next (as in k_tab_wpi, MIDX) -----------------------
  uint32_t cw0 = 0, cw1 = 0;
  uint4 cand = {0, 0, 0, 0};
  cobel_u4 blk = {0, 0, 0, 0};
  int refresh_in = 0;
  uint2 mdig = {0u, 0u};   // the four digest entries of the current state (requested a step ahead)
  uint32_t fix_sa = ~0u;
  float fix_r = 0.0f;
  uint32_t idx_cur = 0, mg_cur = 0;

  auto draw_m = [&](uint32_t counter) -> uint32_t {
    const cobel_u4 b = cobel_philox(counter >> 2, (uint32_t)lane, g, COBEL_STREAM_MEMORY, seed_now());
    return cobel_word(b, counter & 3u);
  };
  auto refresh_draws = [&](uint32_t pq) {
    const uint32_t mi = (cm + 1u) >> 2;
    const bool p0 = lane == 62, p1 = lane == 63;
    blk = cobel_philox(p1 ? 2u * pq + 1u : (p0 ? 2u * pq : mi), (p0 || p1) ? 0u : (uint32_t)lane,
                       g, (p0 || p1) ? COBEL_STREAM_POLICY : COBEL_STREAM_MEMORY, seed_now());
  };
  auto enter_state = [&](int s) {
    const uint4 c = W4[s];
    cw0 = rfl(c.x);
    cw1 = rfl(c.y);
    cand = W4[succ_of(cw0, cw1)];   // (every lane: successor lane % 4; four distinct lines)
    mdig = *reinterpret_cast<const uint2*>(&Mg[(uint32_t)s * 4u]);
  };
  auto begin_trial = [&]() -> bool {
    const pwg_kargs R = rare_args();
    if (trial >= R->r.trials_target) return false;
    const int start_lo = R->start_off[world];
    const uint32_t start_cnt = (uint32_t)(R->start_off[world + 1] - start_lo);
    state = (int)R->starts[start_lo + (int)cobel_draw_bounded(ce, 0u, g, COBEL_STREAM_ENV, seed_now(),
                                                              start_cnt)];
    ce += 1u;
    step = 0;
    trew = 0.0;
    asm volatile("" : "+v"(trew));
    iflags |= 1u;
    enter_state(state);
    return true;
  };
  // the float64 planning update (NumPy promotion of dyna_q.py:290-299 with a float32 table)
  auto plan_td = [&](float q, float m, float r, uint32_t nt) -> float {
    const double gnt = gamma * (double)nt;
    double td = (double)r + gnt * (double)m;
    td = td - (double)q;
    return (float)((double)q + alpha * td);
  };
  // (global-memory waves)
  // exact conflict sets (see k_tab_wpi): all EARLIER lanes that write a cell this lane reads
  auto conflict_sets = [&](uint32_t idx, uint32_t ns) -> unsigned long long {
    unsigned long long* const H1 = H;
    unsigned long long* const H2 = H + 64;
    const uint32_t h1 = idx & 63u, h2 = idx >> 6;
    const unsigned long long bit = 1ull << lane;
    atomicOr(&H1[h1], bit);
    atomicOr(&H2[h2], bit);
    __builtin_amdgcn_wave_barrier();
    const ulonglong2* const r4 = reinterpret_cast<const ulonglong2*>(&H1[(ns & 15u) * 4u]);
    const ulonglong2 ra = r4[0], rb = r4[1];
    const unsigned long long cell = H1[h1] & H2[h2];
    const unsigned long long row = ((ra.x | ra.y) | (rb.x | rb.y)) & H2[ns >> 4];
    const unsigned long long cnd = (cell | row) & (bit - 1ull);
    __builtin_amdgcn_wave_barrier();
    H1[h1] = 0ull;
    H2[h2] = 0ull;
    return cnd;
  };
  // ---- one batch, Q in LDS (called under lane < B) -------------------------------------------
  // Which lanes must wait is found IN the table (round 6; until then two tables of lane masks per
  // wave, 1 KiB next to every Q table — the tenth table did not fit the CU's 160 KiB beside them):
  // a lane that changes its cell raises the cell to the TAG ~lane with ds_max_u32 — tags are the
  // bit patterns 0xffffffc0 .. 0xffffffff, above every float that is not a NaN of exactly that
  // payload, so the cell then holds the tag of the EARLIEST lane that writes it; every lane reads
  // its five inputs again and is held back iff one of them is a tag above its own — an earlier
  // writer of a cell it reads, the very set the lane masks gave.  The earliest writer of a cell
  // then stores the new value (committed) or puts the old one back (held back): one store per
  // cell.  1 + 2 + 1 LDS instructions and a v_max3 pair for 9 and ~35 vector instructions.
  uint32_t* const Qu = reinterpret_cast<uint32_t*>(lds);
  const uint4* const Qs4u = reinterpret_cast<const uint4*>(lds);
  uint32_t tag_mine = ~(uint32_t)lane;
  // (`lo` = the first lane of the round: nothing can hold it back — tags are raised by lanes of the
  //  round only — unless the table held one of the tag patterns to begin with, a NaN no arithmetic
  //  produces; the round then commits that lane regardless: garbage in, garbage out, but every
  //  round ends one lane further and the batch ends)
  auto tag_round = [&](uint32_t idx, uint32_t ns, bool act, bool ch, float q, float qn, int lo) -> int {
    if (ch) atomicMax(&Qu[idx], tag_mine);
    __builtin_amdgcn_wave_barrier();
    uint32_t t = 0u, c2 = 0u;
    if (act) {
      const uint4 r2 = Qs4u[ns];
      c2 = Qu[idx];
      t = max(max(max(r2.x, r2.y), r2.z), max(r2.w, c2));
    }
    const unsigned long long blocked = __builtin_amdgcn_ballot_w64(act && t > tag_mine);
    const int stop = max(blocked ? __ffsll((long long)blocked) - 1 : B, lo + 1);
    if (ch && c2 == tag_mine) Qf[idx] = lane < stop ? qn : q;
    __builtin_amdgcn_wave_barrier();
    return stop;
  };
  auto run_batch_lds = [&](uint32_t idx, uint32_t ns, uint32_t nt, float r) {
    // The first round, every lane of the batch, written out on its own: 91 % of the batches end
    // here on trained agents (0 / 1 / 2 / 3 / 4 writers in 7 / 17 / 23 / 21 / 15 % of them,
    // scripts/experiments/exp_pwg_hist.py), and as the first trip of one loop over the rounds it
    // carried that loop's `act` masks and round state: 12.27 -> 12.06 ms per C3 launch.
    int first;
    {
      const float4 row = Qs[ns];
      const float q = Qf[idx];
      const float qn = plan_td(q, max4(row), r, nt);
      const bool ch = fbits(qn) != fbits(q);
      if (!__builtin_amdgcn_ballot_w64(ch)) return;
      first = tag_round(idx, ns, true, ch, q, qn, 0);
    }
    // Later rounds (a lane was held back): the lanes from `first` on.
    while (first < B) {
      const bool act = lane >= first;
      float q = 0.0f, qn = 0.0f;
      if (act) {
        const float4 row = Qs[ns];
        q = Qf[idx];
        qn = plan_td(q, max4(row), r, nt);
      }
      const bool ch = act && fbits(qn) != fbits(q);
      if (!__ballot(ch)) return;
      first = tag_round(idx, ns, act, ch, q, qn, first);
    }
  };
  // ---- one batch, Q in global memory: inputs (row, q) already in registers -------------------
  auto run_batch_glb = [&](uint32_t idx, uint32_t ns, uint32_t nt, float r, float4 row, float q) {
    int first = B;
    bool have_conf = false;
    unsigned long long conf = 0ull;
    // the lanes that go again take what the committed writers wrote, in lane order (a later writer
    // of the same cell wins, as in the table): `pend` = the committed writers of a lane's conflict set
    auto take_writes = [&](unsigned long long pend, float qn) {
      while (__ballot(pend != 0ull)) {
        const int e = pend ? (__ffsll((long long)pend) - 1) : 0;
        const uint32_t ie = (uint32_t)__shfl((int)idx, e);
        const float ve = __shfl(qn, e);
        if (pend) {
          if (ie == idx) q = ve;
          if ((ie >> 2) == ns) {
            const uint32_t c = ie & 3u;
            row.x = c == 0u ? ve : row.x;
            row.y = c == 1u ? ve : row.y;
            row.z = c == 2u ? ve : row.z;
            row.w = c == 3u ? ve : row.w;
          }
          pend &= pend - 1ull;
        }
      }
    };
    {
      const float qn = plan_td(q, max4(row), r, nt);
      const bool ch = fbits(qn) != fbits(q);
      const unsigned long long changed = __builtin_amdgcn_ballot_w64(ch);
      if (!changed) return;
      conf = conflict_sets(idx, ns);
      have_conf = true;
      const unsigned long long blocked = __builtin_amdgcn_ballot_w64((conf & changed) != 0ull);
      if (blocked) first = __ffsll((long long)blocked) - 1;
      if (ch && lane < first) Qgf[idx] = qn;
      if (first < B) {
        const unsigned long long committed = changed & ((1ull << first) - 1ull);
        take_writes(lane >= first ? (conf & committed) : 0ull, qn);
      }
    }
    while (first < B) {
      const bool act = lane >= first;
      float qn = q;
      if (act) qn = plan_td(q, max4(row), r, nt);
      const bool ch = act && fbits(qn) != fbits(q);
      const unsigned long long changed = __ballot(ch);
      int stop = B;
      if (changed) {
        if (!have_conf) {
          conf = conflict_sets(idx, ns);
          have_conf = true;
        }
        const unsigned long long blocked = __ballot(act && (conf & changed) != 0ull);
        if (blocked) stop = __ffsll((long long)blocked) - 1;
        if (ch && lane < stop) Qgf[idx] = qn;
        if (stop < B) {
          const unsigned long long committed = changed & ((1ull << stop) - 1ull);
          take_writes(lane >= stop ? (conf & committed) : 0ull, qn);
        }
      }
      first = stop;
    }
  };

  // ---- prologue -------------------------------------------------------------------------------
  // What a launch (or a slice) has to fetch before its first step, overlapped: the state's world
  // record and, behind it, the successors' records and the first batch's digest entries travel
  // while the table is landing (three trips to memory in a row; row by row the table alone was
  // sixteen).
  // (one request per cache line of the instance's digest: the planning lanes of the first steps
  //  gather from all over it, and an instance comes back to a wave long after its lines left the L2)
  uint32_t warm;
  {
    const uint32_t off = (uint32_t)lane * 128u, last = SA * 2u - 4u;
    warm = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(Mg) + (off < last ? off : last));
  }
  bool live = true;
  if (iflags & 1u) enter_state(state);
  else live = begin_trial();
  refresh_draws(cp >> 2);
  {
    const int jp = 3 - (int)(cp & 3u), jm = 4 - (int)((cm + 1u) & 3u);
    refresh_in = jp < jm ? jp : jm;
  }
  idx_cur = lane < B ? cobel_bounded(draw_m(cm), SA) : 0u;
  mg_cur = lane < B ? (uint32_t)Mg[idx_cur] : 0u;
  asm volatile("" ::"v"(warm));
  if (!QG) {
    stores_done();   // (vmcnt(0): the table has landed)
    for (int s = lane; s < S; s += 64) {
      const float4 qv = Qs[s];
      nonzero |= fbits(qv.x) | fbits(qv.y) | fbits(qv.z) | fbits(qv.w);
    }
  }
  if (QG) {
    for (int b = lane; b < 128; b += 64) H[b] = 0ull;
    __builtin_amdgcn_wave_barrier();
  }
  // COBEL_IF_NONZERO (bit 1, launch-local): see k_tab_wpi
  if (!__ballot(nonzero != 0u)) {
    // (the digest, a state's four entries per request, eight requests in flight)
    const uint2* const M2 = reinterpret_cast<const uint2*>(Mg);
    for (int s0 = 0; s0 < S; s0 += 512) {
      uint2 mv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + j * 64 + lane;
        mv[j] = M2[s < S ? s : S - 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) nonzero |= (mv[j].x | mv[j].y) & 0x80008000u;
    }
  }
  if (__ballot(nonzero != 0u)) iflags |= 2u;

  PWG_STAMP(1);   // prologue
  const int budget0 = step_budget > 0 ? step_budget : 0x7fffffff;
  int budget = budget0;
  uint32_t batches = 0u;
  asm volatile("" : "+v"(batches));

  while (live) {
    if (budget == 0) break;
    budget -= 1;

    // ---- draws ----------------------------------------------------------------------------------
    const int src_lane = 62 + (int)((cp >> 1) & 1u);
    const uint32_t w0 = rl((cp & 1u) ? blk.z : blk.x, src_lane);
    const uint32_t w1 = rl((cp & 1u) ? blk.w : blk.y, src_lane);
    if (__builtin_expect(refresh_in == 0, 0)) {
      refresh_draws((cp + 1u) >> 2);
      const int jp = 4 - (int)((cp + 1u) & 3u), jm = 4 - (int)((cm + 1u) & 3u);
      refresh_in = jp < jm ? jp : jm;
    }
    refresh_in -= 1;
    cp += 1u;

    // ---- everything this step reads from Q ----------------------------------------------------------
    const uint32_t succ = succ_of(cw0, cw1);
    float4 srow;
    float qc;   // component lane % 4 of Q[state]: ONE compare gives the tie pattern, the maximum is two
                // quad permutes away, and the chosen action's value a readlane
    float4 prow = {0.0f, 0.0f, 0.0f, 0.0f};   // QG, lane j < B: Q[ns_j] ...
    float pq = 0.0f;                           // ... and Q[s_j][a_j] of the pair it replays
    if (QG) {
      stores_done();   // (the previous step's planning stores, by other lanes of this wave)
      qc = Qgf[(uint32_t)state * 4u + qlane];
      srow = Qg[succ];
      if (lane < B && (iflags & 2u)) {
        prow = Qg[mg_cur & 0x3fffu];
        pq = Qgf[idx_cur];
      }
    } else {
      qc = Qf[(uint32_t)state * 4u + qlane];
      srow = Qs[succ];
    }
    const float smax = max4(srow);

    // ---- select (policy/greedy.py:40-88): integer thresholds of the tie pattern's CDF ---------------
    int a;
    {
      // (lanes 4k .. 4k + 3 hold the four values: maximum over the quad, tie pattern = the low four
      //  bits of one ballot)
      // (the permute fused into the maximum: the compiler keeps v_mov_b32_dpp and v_max_f32 apart)
      float m;
      // (two wait states between a VALU write of a register and a DPP read of it: the compiler does
      //  not look for hazards inside an asm block.  Every lane of the wave is active here — a
      //  permute that reads a disabled lane leaves its destination unwritten)
      asm("s_nop 1\n\t"
          "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
          : "=&v"(m)
          : "v"(qc));
      const int t = (int)((uint32_t)__ballot(qc == m) & 15u);
      const uint64_t K = cobel_u53(w0, w1);
      const unsigned long long passed = __ballot(thr_mine <= K);
      a = __popcll((passed >> (t * 3)) & 7ull);
    }
    // ---- env.step (interface/gridworld.py:115-126) --------------------------------------------------
    const int ns = (int)next_of(cw0, cw1, a);
    const uint32_t nw0 = rl(cand.x, a), nw1 = rl(cand.y, a);
    const uint32_t r_bits = rl(cand.z, a);
    const float r = __builtin_bit_cast(float, r_bits);
    if (r_bits != 0u) iflags |= 2u;
    const uint32_t end = rl(cand.w, a);
    const float ns_max = __builtin_bit_cast(float, rl(fbits(smax), a));
    const uint32_t nt = 1u - end;
    const uint32_t sa = (uint32_t)state * 4u + (uint32_t)a;
    const bool trial_over = end || (step + 1 >= A.r.steps_per_trial);
    // (everything the previous step requested is consumed BEFORE this step's requests go out: the
    //  wait for a register loaded across the loop edge is a wait for every load in flight)
    // The reward estimate of (state, a): +0.0f unless its digest entry is flagged — then, rarely
    // (pairs that lead into a rewarded state), it is fetched from the packed record.  The model
    // records themselves are never read otherwise: their lines stay out of the L2.
    const uint32_t dpair = rfl((a & 2) ? mdig.y : mdig.x);
    const uint32_t dold = (a & 1) ? (dpair >> 16) : (dpair & 0xffffu);
    float R = 0.0f;
    if (__builtin_expect((dold & 0x8000u) != 0u, 0)) {
      R = __builtin_bit_cast(float, rfl(model32[2u * sa]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // (unconditional: at a trial's end the two requests are wasted — begin_trial asks again — but a
    //  request under a condition merges with the old value, and the merge is a copy behind a wait
    //  for the answer, a memory round trip per step of the global-memory waves: 25.7 -> 32 ms per
    //  launch of those alone in one build of round 6)
    cand = W4[succ_of(nw0, nw1)];
    const uint2 mdig_next = *reinterpret_cast<const uint2*>(&Mg[(uint32_t)ns * 4u]);

    // ---- model store (memory/dyna_q.py:92-96) and online TD (agent/dyna_q.py:290-299), float32 ------
    if (sa == fix_sa) R = fix_r;
    const float d = r - R;
    const float Rn = R + mlr_f * d;
    const uint32_t fresh_idx = sa;
    const float fresh_r = Rn;
    const uint32_t fresh_m = (uint32_t)ns | (nt << 14) | (fbits(Rn) ? 0x8000u : 0u);
    fix_sa = sa;
    fix_r = Rn;
    const float q_sa = __builtin_bit_cast(float, rl(fbits(qc), a));
    const float gnt = nt ? gamma_f : 0.0f;
    float td = r + gnt * ns_max;
    td = td - q_sa;
    const float qn_online = q_sa + alpha_f * td;
    if (lane == 0) {
      if (QG) Qgf[sa] = qn_online;
      else Qf[sa] = qn_online;
      NT_ST(cobel_model_pack(fresh_r, (uint32_t)ns, nt), &model[sa]);
      Mg[sa] = (uint16_t)fresh_m;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- bookkeeping ----------------------------------------------------------------------------------
    trew += (double)r;
    const int s_prev = state;
    state = ns;
    cw0 = nw0;
    cw1 = nw1;

    // ---- planning: next step's batch is drawn and its digest entries requested, then this batch runs -
    {
      uint32_t idx_next = 0u, mg_next = 0u;
      const uint32_t m = (idx_cur == fresh_idx) ? fresh_m : mg_cur;
      if (lane < B) {
        float rj = 0.0f;
        idx_next = POW2 ? cobel_word(blk, (cm + 1u) & 3u) >> SAv
                        : cobel_bounded(cobel_word(blk, (cm + 1u) & 3u), SAv);
        mg_next = (uint32_t)Mg[idx_next];
        if (iflags & 2u) {
          batches += 1u;
          if (__builtin_expect((m & 0x8000u) != 0u, 0)) {
            rj = __builtin_bit_cast(float, model32[2u * idx_cur]);
            // (consumed here: a wait for this rare load after the branches have joined would be a
            //  wait for every gather this step has just sent out)
            asm volatile("" : "+v"(rj));
          }
          if (idx_cur == fresh_idx) rj = fresh_r;
          const uint32_t nsj = m & 0x3fffu;
          if (QG) {
            // a pair that replays the entry this very step stored plans towards the state just
            // entered: its row is the successor row read above
            if (__ballot(idx_cur == fresh_idx)) {
              const float4 nrow = {__builtin_bit_cast(float, rl(fbits(srow.x), a)),
                                   __builtin_bit_cast(float, rl(fbits(srow.y), a)),
                                   __builtin_bit_cast(float, rl(fbits(srow.z), a)),
                                   __builtin_bit_cast(float, rl(fbits(srow.w), a))};
              if (idx_cur == fresh_idx) prow = nrow;
            }
            // the online update wrote Q[s][a] after these registers were loaded
            if (idx_cur == sa) pq = qn_online;
            if (nsj == (uint32_t)s_prev) {
              prow.x = a == 0 ? qn_online : prow.x;
              prow.y = a == 1 ? qn_online : prow.y;
              prow.z = a == 2 ? qn_online : prow.z;
              prow.w = a == 3 ? qn_online : prow.w;
            }
            run_batch_glb(idx_cur, nsj, (m >> 14) & 1u, rj, prow, pq);
          } else {
            run_batch_lds(idx_cur, nsj, (m >> 14) & 1u, rj);
          }
        }
        if (idx_next == fresh_idx) mg_next = fresh_m;   // the gather may have passed the store
      }
      idx_cur = idx_next;
      mg_cur = mg_next;
      mdig = mdig_next;
      cm += 1u;
    }

    if (__builtin_expect(trial_over, 0)) {
      // agent/dyna_q.py:207-212: current_trial += 1; logs['steps'] = step (0-based)
      const pwg_kargs R = rare_args();
#define rr (R->r)
      if (lane == 0 && trial >= 0 && trial < rr.trial_cap) {
        const size_t mo = (size_t)stripe * (size_t)rr.trial_cap + (size_t)trial;
        if (rr.lat_sum) atomicAdd(rr.lat_sum + mo, (unsigned long long)step);
        if (rr.lat_cnt) atomicAdd(rr.lat_cnt + mo, 1ull);
        if (rr.reward_sum) atomicAdd(rr.reward_sum + mo, trew);
        if (rr.resp_cnt && trew > 0.0) atomicAdd(rr.resp_cnt + mo, 1ull);
        if (rr.lat_trace) rr.lat_trace[(size_t)i * rr.trial_cap + trial] = step;
      }
#undef rr
      trial += 1;
      iflags &= ~1u;
      if (!begin_trial()) break;
    } else {
      step += 1;
    }
  }

  // ---- write back -------------------------------------------------------------------------------
  __builtin_amdgcn_wave_barrier();
  PWG_STAMP(2);   // steps
  if (!QG) {
    for (int s0 = 0; s0 < S; s0 += 256) {
      float4 qv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s0 + j * 64 + lane;
        qv[j] = Qs[s < S ? s : S - 1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s0 + j * 64 + lane;
        if (s < S) Qg[s] = qv[j];
      }
    }
  }
  if (lane == 0) {
    inst[COBEL_I_STATE] = state;
    inst[COBEL_I_STEP] = step;
    inst[COBEL_I_TRIAL] = trial;
    inst[COBEL_I_CTR_ENV] = (int32_t)ce;
    inst[COBEL_I_CTR_POLICY] = (int32_t)cp;
    inst[COBEL_I_CTR_MEMORY] = (int32_t)cm;
    inst[COBEL_I_FLAGS] = (int32_t)(iflags & ~2u);
    const unsigned long long executed = (unsigned long long)(budget0 - budget);
    *reinterpret_cast<double*>(inst + COBEL_I_REWARD_LO) = trew;
    *reinterpret_cast<unsigned long long*>(inst + COBEL_I_STEPS_LO) += executed;
    const pwg_kargs R = rare_args();
    if (R->r.steps_done && executed) atomicAdd(R->r.steps_done, executed);
    if (R->r.batches_done && batches) atomicAdd(R->r.batches_done, (unsigned long long)batches);
  }
  // (the next instance of this wave reuses the LDS slice: its staging stores follow these loads in
  //  the wave's own program order)
  __builtin_amdgcn_wave_barrier();
}

// SLICED: the launch's tickets are slices (n_slices > 1).  Its own instantiation: what the ticket
// loop carries for slices and whole-instance tickets costs the step loop of the UNSLICED launch —
// the headline — a per cent through the register allocation alone (11.55 -> 11.66 ms, same box).
template <bool SLICED, bool POW2>
__global__ __launch_bounds__(1024) void k_tab_pwg(const pwg_args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int waves = A.nl + A.ng;
  const size_t slice_l = (size_t)A.S * 16;
  const bool qg = wave >= A.nl;
  unsigned char* const lds =
      lds_raw + (qg ? (size_t)A.nl * slice_l + (size_t)(wave - A.nl) * kHashBytes
                    : (size_t)wave * slice_l);
  const uint64_t thr_mine = A.thr[(lane % 48) / 3][lane % 3];
  const uint32_t wave_id = (uint32_t)blockIdx.x * (uint32_t)waves + (uint32_t)wave;
  const uint32_t stripe = A.r.mon_stripes > 1 ? wave_id % (uint32_t)A.r.mon_stripes : 0u;
  // Work is handed out as tickets from eight queues, one per XCD: instance i belongs to queue i % 8,
  // and a workgroup serves the queue of the XCD it runs on first.  The world of an instance is
  // (global id) % n_worlds, so an XCD's L2 sees an eighth of the worlds' records and the same
  // instances launch after launch — what the round-robin placement of one workgroup per instance
  // gives k_tab_wpi for free.
  //   * n_slices == 1: a ticket is an instance with all its steps of this call; a wave whose queue
  //     is empty takes tickets of the others until all eight are (speed only: any wave may run any
  //     instance — nothing an instance touches is handed from one wave to another inside a launch).
  //   * n_slices > 1 (few instances per wave slot: the shard of a batch split over several GPUs):
  //     an instance's steps are cut into slices — long ones first, short ones last — and a slice is
  //     what a launch with that step budget would be: tables staged, stepped, written back.  All
  //     instances then advance at about the same pace and the last round of the launch consists of
  //     short tickets (an instance's 512 steps as ONE unit of work leave 8 192 instances on 3 328
  //     wave slots with a launch of four rounds for 2.5 rounds of work).  Tickets 0 .. nq - 1 of a
  //     queue are the first slices of its nq instances; the wave that finishes slice k of an
  //     instance appends (instance, k + 1) to the queue's ring, and tickets nq, nq + 1, ... are the
  //     ring's entries in that order: a ticket exists when its predecessor is done, a wave that
  //     draws a ticket not yet written waits for the entry (somebody is running the slice that
  //     will write it — the waiting waves hold no ticket, so there is no cycle).  The tables travel
  //     from wave to wave through the XCD's L2 — per-XCD L2s are not coherent with each other, and
  //     writing one back per hand-off would turn every dirty line of the model tables into HBM
  //     traffic — so here a queue is served by the waves of ONE XCD only: the first wave to ask
  //     claims it for its XCD (`owner`), its own XCD's queue first; an XCD whose queue is empty
  //     claims queues nobody serves (none on a chip whose eight XCDs all run workgroups).  The
  //     producer's stores are complete in L2 (vmcnt(0)) before the entry is written; the consumer's
  //     CU drops its L1 (agent-scope acquire) before it reads.
#if defined(COBEL_PWG_STAMPS)
  uint32_t* const stamps = A.ring + (size_t)8 * A.ring_stride + (size_t)wave_id * 8u;
  uint32_t* const stamp_prev = reinterpret_cast<uint32_t*>(
      lds_raw + (size_t)A.nl * slice_l + (size_t)A.ng * kHashBytes + (size_t)wave * 8);
  if (lane == 0) {
    *stamp_prev = (uint32_t)__builtin_readcyclecounter();
    stamps[7] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  }
#endif
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  if (SLICED && xcc > A.xcc_limit) return;
  // (what the ticket loop carries across an instance — XCD, queues tried, slice — is ONE scalar
  //  register: the step loop has none to spare, and a value it had to spill for the ticket loop's
  //  sake was reloaded in every step)
  uint32_t tstate = xcc << 8;   // bits 0-7: queues (xcc + k) % 8, k < k0, are empty (or another XCD's)
  uint32_t t_next = 0xffffffffu;   // a ticket of the current queue drawn ahead (with the last append)
  // (s_setprio for either kind of wave was measured and loses: LDS waves at priority 1 / 3 12.64 ms
  //  per launch against 12.50, global-memory waves at 3 13.1 ms)
  for (;;) {
    int i = -1;
    {
      constexpr bool sliced = SLICED;
      uint32_t k0 = tstate & 0xffu;
      const uint32_t x = (tstate >> 8) & 0xffu;
      uint32_t sl = 0u;
      while (k0 < 8u) {
        const uint32_t q = (x + k0) & 7u;
        const uint32_t nq = ((uint32_t)A.r.n + 7u - q) / 8u;
        // (sliced launches: the queue's first nw instances run whole — tickets of counter 1 —, the
        //  other ns in slices — counter 0: first slices, then the ring)
        const uint32_t nw = sliced ? min(nq, A.whole & 0xffffffu) : 0u;
        const uint32_t ns = nq - nw;
        const uint32_t total = SLICED ? ns * (uint32_t)A.n_slices : ns;
        uint32_t t = t_next;
        t_next = 0xffffffffu;
        if (t >= total) t = 0xffffffffu;   // (drawn ahead, beyond the sliced tickets: the whole ones are left)
        if (t == 0xffffffffu) {
          t = 0x10000000u;
          if (lane == 0 && nq) {
            bool serve = true;
            if (sliced) {
              uint32_t o = __hip_atomic_load(A.owner + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (o == 0u) {
                o = atomicCAS(A.owner + q, 0u, x + 1u);
                if (o == 0u) o = x + 1u;
              }
              serve = o == x + 1u;
            }
            if (serve) {
              // A wave with Q in global memory needs two to three times as long for a ticket.  With
              // whole instances set aside (nw > 0) those are all it draws; otherwise it leaves the
              // last `reserve` tickets of a queue to the LDS waves, which finish them sooner than
              // it would finish one.  The LDS waves draw slices first, left-over whole instances last.
              if (qg && nw) {
                const uint32_t b = atomicAdd(A.queue + q * 8u + 1u, 1u);
                if (b < nw) t = kWholeTicket | b;
              }
              // (kWholeThenSlices: a global-memory wave whose whole instances are gone goes on with slices)
              if (t == 0x10000000u && !(qg && nw && !(A.whole & kWholeThenSlices)) &&
                  !(qg && __hip_atomic_load(A.queue + q * 8u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT) + A.reserve >= total)) {
                t = atomicAdd(A.queue + q * 8u, 1u);
                if (t >= total) {
                  t = 0x10000000u;
                  if (nw) {
                    const uint32_t b = atomicAdd(A.queue + q * 8u + 1u, 1u);
                    if (b < nw) t = kWholeTicket | b;
                  }
                }
              }
            }
          }
          t = rfl(t);
        }
        if (SLICED && (t & kWholeTicket)) {
          i = (int)((t & 0xffffffu) * 8u + q);
          sl = kWholeSlice;
          break;
        }
        if (t < total) {
          if (t < ns) {
            i = (int)((nw + t) * 8u + q);
          } else {
            const uint32_t* const slot = A.ring + (size_t)q * A.ring_stride + (t - ns);
            // The entry is written by the wave that runs the predecessor slice — a few
            // milliseconds at most.  Should it never come (a producer that faulted or was
            // killed), the wait gives up after kRingSpinLimit polls (seconds), raises the
            // launch's abort word and leaves; the other waiting waves see the word and follow,
            // so the grid drains and cobel_tab_scratch_check reports COBEL_E_HIP instead of the
            // whole GPU hanging.
            uint32_t e, spins = 0u;
            while ((e = rfl(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) == 0u) {
              __builtin_amdgcn_s_sleep(32);
              ++spins;
              if (spins >= kRingSpinLimit ||
                  ((spins & 1023u) == 0u &&
                   rfl(__hip_atomic_load(A.queue + kAbortWord, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT)) != 0u)) {
                if (lane == 0)
                  __hip_atomic_store(A.queue + kAbortWord, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                return;
              }
            }
            // (this CU's L1 is dropped; the wave's own loads behind the fence need no wait for it)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sl = e >> 24;
            i = (int)(((e & 0xffffffu) - 1u) * 8u + q);
          }
          break;
        }
        k0 += 1u;
      }
      tstate = k0 | (x << 8) | (sl << 16);
    }
#if defined(COBEL_PWG_STAMPS)
    if (i < 0 && lane == 0) {
      stamps[5] = tstate >> 8 & 0xffu;
      stamps[6] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    }
#endif
    if (i < 0) break;
    int budget = A.r.step_budget;
    if (SLICED) {
      const int slice = (int)(tstate >> 16);   // (kWholeSlice: all steps of the call)
#pragma unroll
      for (int j = 0; j < kMaxSlices; ++j)
        if (j == slice) budget = A.slice_steps[j];
    }
#if defined(COBEL_PWG_STAMPS)
    PWG_STAMP(0);   // ticket
    if (lane == 0) atomicAdd(stamps + 4, 1u);
    if (qg) pwg_instance<true, POW2>(A, i, budget, lds, lane, thr_mine, stripe, stamps, stamp_prev);
    else pwg_instance<false, POW2>(A, i, budget, lds, lane, thr_mine, stripe, stamps, stamp_prev);
    PWG_STAMP(3);   // write-back issued
#else
    if (qg) pwg_instance<true, POW2>(A, i, budget, lds, lane, thr_mine, stripe);
    else pwg_instance<false, POW2>(A, i, budget, lds, lane, thr_mine, stripe);
#endif
    if (SLICED) {
      // the instance's next slice becomes a ticket; this wave's next ticket is drawn in the same
      // trip to memory (an LDS wave: the global-memory waves look at the queue's length first)
      const uint32_t q = (uint32_t)i & 7u;
      const uint32_t sl1 = (tstate >> 16) + 1u;
      const bool more = sl1 < (uint32_t)A.n_slices;   // (kWholeSlice + 1 is not)
      uint32_t pos = 0u, t = 0xffffffffu;
      if (lane == 0) {
        if (more) pos = atomicAdd(A.tail + q * 8u, 1u);
        if (!qg) t = atomicAdd(A.queue + q * 8u, 1u);
      }
      // Release: the instance's tables are complete in this XCD's L2 before the entry that
      // publishes them.  Workgroup scope on purpose — its code is the wait for this wave's stores
      // (vmcnt(0)) without the L2 write-back an agent-scope release adds: the consumer is a wave
      // of the SAME XCD (a queue has one owner), so the tables reach it through this L2, and
      // writing every dirty model line back to HBM per hand-off is what slicing must not cost.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      stores_done();
      if (lane == 0 && more)
        __hip_atomic_store(A.ring + (size_t)q * A.ring_stride + pos,
                           (((uint32_t)i >> 3) + 1u) | (sl1 << 24), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      t_next = rfl(t);
    }
  }
}

}  // namespace

// Does this run qualify, and with how many waves of each kind?  (plain Dyna-Q training with the
// digest in HBM, exact dependency lookup, no visit counters, no parameter sets)
bool cobel_tab_pwg_plan(const cobel_world* world, const cobel_tab_run_t& r, int* nl_out, int* ng_out,
                        size_t* lds_out) {
  static const char* const force = cobel_debug_env("COBEL_DEBUG_PWG");   // "nl,ng" (experiments)
  const int S = world->n_states;
  if (r.agent != COBEL_AGENT_DYNAQ || !r.model_index || r.occupancy || r.param_index ||
      r.last_exp || S * 4 > 4096 || S <= 256 || r.batch < 1 || r.batch > COBEL_MAX_BATCH)
    return false;
  const bool scratch = r.scratch && r.scratch_bytes >= COBEL_TAB_SCRATCH_BYTES(r.n);
  if (!scratch && !world->queue) return false;
  int n_cu = 0;
  size_t total = 0;
  if (cobel_device_limits(world->device, &n_cu, &total) != COBEL_OK) return false;
  const size_t slice_l = (size_t)S * 16;
#if defined(COBEL_PWG_STAMPS)
  total -= 1024;   // (the timing build keeps a word per wave in LDS)
#endif
  int nl = (int)(total / slice_l);
  if (nl > 16) nl = 16;
  // LDS waves only (round 6).  While every Q table had 1 KiB of lane masks beside it nine fitted a
  // CU at 32 x 32 and four global-memory waves on top paid (9 + 0 / 9 + 4: 12.7 / 11.5 ms per C3
  // launch); with the dependencies found in the table itself ten fit, and a global-memory wave next
  // to ten costs the others more in L2 misses on their digests than it adds (10 + 0 / 1 / 2 / 3:
  // 10.9 / 11.2 / 11.1 / 11.3 ms; scripts/experiments/pwg_r06/).  Global-memory waves remain for
  // COBEL_F_PWG_GLOBAL and forced mixes (tests, experiments).
  int ng = 0;
  if (force) {
    int a = 0, b = 0;
    if (sscanf(force, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0 && a + b >= 1 && a + b <= 16 &&
        (size_t)a * slice_l + (size_t)b * kHashBytes <= total) {
      nl = a;
      ng = b;
    }
  } else if (r.flags & COBEL_F_PWG_GLOBAL) {
    nl = 0;
    ng = 16;
  } else if (128 / ((slice_l + 1279) / 1280) >= 16) {
    // (LDS comes in blocks of 1 280 B, 128 per CU: tabular.hip lds_workgroups_per_cu)
    return false;   // LDS is not what limits the resident instances: k_tab_wpi as it is
  }
  *nl_out = nl;
  *ng_out = ng;
  *lds_out = (size_t)nl * slice_l + (size_t)ng * kHashBytes;
  return true;
}

// The slices of a launch (see k_tab_pwg).  A slice boundary costs an instance ~19 us on C3 (9.6 us
// of ticket, prologue and write-back, the rest in steps that run slower while more tables are on
// the move; scripts/experiments/exp_pwg_stamps.py), a short last slice saves waiting at the end of the launch.
// Measured on one MI355X, C3, TEN LDS waves on each of 256 CUs (round 6; ms per launch of 512
// steps, profiles/r06_sweeps.txt): 8 192 instances 1.61 whole / 1.48 with 384 + 128 (1.51 with 320 +
// 128 + 64, more slices: slower); 16 384: 3.00 / 2.85 with 448 + 64 (2.88 with 384 + 128, three
// slices 2.94); 32 768: 5.64 whichever way; 65 536: slices cost 1.5 %.  With waves that keep Q in
// global memory (forced mixes only since round 6: cobel_tab_pwg_plan) between 1.75 and 3.5 instances
// per wave slot those waves are kept out of the slices: each runs exactly ONE whole instance, and the
// LDS waves share the others in slices (round 5: such a wave needs 1.0 of a 1.6 ms launch for one).
// A staggered split — every wave slot starts with the HEAD of a split instance, of a length spread
// over (0, budget), the tails are handed out last, longest first: one boundary per wave slot instead
// of two per instance — was built and measured in round 6 (scripts/experiments/pwg_r06/
// staggered_split.patch): 65 536 instances 11.01 -> 10.89 ms, but 16 384 / 8 192 only 2.96 / 1.62.
// The waves of a SIMD do not run at one speed — the arbiter serves the oldest first: 358 / 431 /
// 559 us per ticket for the first / second / third wave of a SIMD (s_setprio turns the order
// round, the sum stays) — so a long tail drawn late by a third wave ends the launch; even slices
// keep all instances at the same pace whatever the wave.
static int plan_slices(const cobel_tab_run_t& r, int grid, int n_cu, int nl, int ng, bool scratch,
                       int32_t* steps /* [kMaxSlices] */, uint32_t* whole) {
  const char* const forced = cobel_debug_env("COBEL_DEBUG_PWG_SLICES");   // "320,128,64" (tests, experiments)
  for (int j = 0; j < kMaxSlices; ++j) steps[j] = 0;
  steps[0] = r.step_budget;
  *whole = 0u;
  if (!scratch || grid < n_cu || r.step_budget < 64) return 1;
  if (forced) {
    int v[kMaxSlices] = {0}, k = 0, sum = 0;
    const char* c = forced;
    while (k < kMaxSlices && *c) {
      v[k] = atoi(c);
      if (v[k] <= 0) break;
      sum += v[k++];
      while (*c && *c != ',') ++c;
      if (*c == ',') ++c;
    }
    if (k >= 1 && sum == r.step_budget) {
      for (int j = 0; j < k; ++j) steps[j] = v[j];
      return k;
    }
    return 1;
  }
  const double rounds = (double)r.n / ((double)grid * (nl + ng));
  const int b = r.step_budget;
  if (rounds < 1.0) return 1;
  if (ng == 0) {
    if (rounds >= 10.0) return 1;
    steps[1] = rounds >= 5.0 ? b / 8 : b / 4;
    steps[0] = b - steps[1];
    return 2;
  }
  // (mixes with global-memory waves: round 5's plan)
  if (rounds >= 14.0) return 1;
  if (rounds >= 7.0) {
    steps[1] = b / 8;
    steps[0] = b - steps[1];
    return 2;
  }
  if (rounds >= 3.5) {
    // (from 4.5 per slot: two whole instances per global-memory wave first, then slices — 16 384 /
    //  20 000 instances 3.12 -> 3.09 / 3.78 -> 3.75 ms; at 12 000, 3.6 per slot, it costs 3 %)
    if (rounds >= 4.5 && nl > 0)
      *whole = (uint32_t)(2 * ((ng * grid + 7) / 8)) | kWholeThenSlices;
    steps[1] = b / 4;
    steps[0] = b - steps[1];
    return 2;
  }
  if (rounds >= 1.75 && nl > 0) {
    // one whole instance per global-memory wave (a queue's share of them, rounded up)
    *whole = (uint32_t)((ng * grid + 7) / 8);
    steps[1] = b / 4;
    steps[0] = b - steps[1];
    return 2;
  }
  steps[2] = b / 8;
  steps[1] = b / 4;
  steps[0] = b - steps[1] - steps[2];
  return 3;
}

int cobel_tab_pwg_launch(const cobel_world* world, const cobel_tab_run_t& r, hipStream_t st) {
  int nl = 0, ng = 0;
  size_t lds = 0;
  if (!cobel_tab_pwg_plan(world, r, &nl, &ng, &lds))
    return cobel_fail(COBEL_E_UNSUPPORTED, "cobel_tab_pwg_launch: run not covered");
  pwg_args A;
  A.rec = world->rec;
  A.starts = world->starts;
  A.start_off = world->start_off;
  A.S = world->n_states;
  A.n_worlds = world->n_worlds;
  A.r = r;
  const cobel_eps_consts eps = cobel_make_eps_consts(r.epsilon);
  memcpy(A.thr, eps.thr, sizeof(A.thr));
  A.alpha_f = (float)r.alpha;
  A.gamma_f = (float)r.gamma;
  A.model_lr_f = (float)r.model_lr;
  A.nl = nl;
  A.ng = ng;
  int n_cu = 0;
  size_t lds_cu = 0;
  if (cobel_device_limits(world->device, &n_cu, &lds_cu) != COBEL_OK) return COBEL_E_HIP;
  const int waves = nl + ng;
  int grid = (r.n + waves - 1) / waves;
  if (grid > n_cu) grid = n_cu;
  // the counters of the launch: in the caller's scratch area when there is one (any number of
  // calls in flight), else in the 256 bytes of the world handle (one call per handle at a time)
  const bool scratch = r.scratch && r.scratch_bytes >= COBEL_TAB_SCRATCH_BYTES(r.n);
  uint32_t* const base = scratch ? static_cast<uint32_t*>(r.scratch) : world->queue;
  A.queue = base;
  A.owner = scratch ? base + 64 : base;
  A.tail = scratch ? base + 128 : base;
  A.ring = scratch ? base + 256 : base;
  A.n_slices = plan_slices(r, grid, n_cu, nl, ng, scratch, A.slice_steps, &A.whole);
  A.ring_stride = (uint32_t)((r.n + 7) / 8) * (uint32_t)(A.n_slices - 1);
  {
    const char* const w_env = cobel_debug_env("COBEL_DEBUG_PWG_WHOLE");   // (experiments, tests)
    if (w_env) A.whole = (uint32_t)strtoul(w_env, nullptr, 0);
    if (nl == 0 || A.n_slices < 2) A.whole = 0u;   // (nobody else would draw the slices)
  }
  {
    const char* const m_env = cobel_debug_env("COBEL_DEBUG_PWG_XCCLIMIT");   // (tests)
    A.xcc_limit = m_env ? (uint32_t)atoi(m_env) & 7u : 7u;
  }
  {
    static const char* const k_env = cobel_debug_env("COBEL_DEBUG_PWG_RESERVE");   // (experiments)
    // (2.0: within 0.5 % of the best at 32 768 and 65 536 instances per GPU, 3-5 % ahead of 2.5 at the
    //  8 192 / 16 384 an eight- / four-way split leaves each GPU)
    const double k = k_env ? atof(k_env) : 2.0;
    A.reserve = nl ? (uint32_t)((double)nl * grid * k / 8.0) : 0u;
  }
#if defined(COBEL_PWG_STAMPS)
  lds += 1024;
  COBEL_HIP_TRY(hipMemsetAsync(base, 0, (256 + (size_t)8 * A.ring_stride + (size_t)8 * grid * waves) * 4, st));
#else
  // (the abort word — word 255 of a scratch area — is STICKY: no launch clears it, so a wave that
  //  gave up is reported by every later cobel_tab_scratch_check, whichever launches followed)
  COBEL_HIP_TRY(hipMemsetAsync(base, 0, scratch ? (size_t)kAbortWord * 4 : (size_t)256, st));
  if (A.n_slices > 1)
    COBEL_HIP_TRY(hipMemsetAsync(A.ring, 0, (size_t)8 * A.ring_stride * 4, st));
#endif
  const bool pow2 = (A.S & (A.S - 1)) == 0;
  void (*const kernel)(const pwg_args) =
      A.n_slices > 1 ? (pow2 ? &k_tab_pwg<true, true> : &k_tab_pwg<true, false>)
                     : (pow2 ? &k_tab_pwg<false, true> : &k_tab_pwg<false, false>);
  if (lds > 64 * 1024)
    COBEL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), lds, st, A);
  COBEL_HIP_TRY(hipGetLastError());
  return COBEL_OK;
}
