/*
 * TEST INFRASTRUCTURE — not part of the product path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 *
 * Plain-C restatement of the reference's hot loop for MANY independent instances, one after the
 * other on one core: Gridworld step/reset, epsilon-greedy, tabular Q / Dyna-Q (model store, TD
 * update, planning replays), QAgent replay over the experience log, and the successor-
 * representation agent.  It follows the same reference lines as oracle/ref_loop.py (which is
 * pinned bit-exactly against golden vectors captured from the real reference) and exists because
 * the NumPy restatement is too slow to check thousands of instances:
 *
 *   interface/gridworld.py:115-126,142   policy/greedy.py:58,77-86   memory/dyna_q.py:92-96,137-155
 *   agent/dyna_q.py:164-215,290-299,327-330   agent/q.py:183-228,305-313,353-354
 *   agent/sr.py:155-197,267-284,302-308   (paths relative to /root/reference/src/cobel)
 *
 * Parity pinned: tests/test_oracle_golden.py runs this file against every fixture in
 * tests/golden/ (float64 mode == the reference as shipped; float32 mode == the reference with its
 * tables cast to float32, the numerics the HIP kernels implement).
 *
 * Arithmetic model.  Tables are held in double.  In float32 mode every operation the reference
 * performs in float32 is computed in double and rounded to float immediately; for + - * of two
 * floats this equals the float32 operation exactly (53 >= 2*24 + 2 bits, no double-rounding
 * error).  Operations the reference performs in float64 (Dyna-Q planning TD, SR row TD) stay in
 * double and are rounded once where the reference stores into its float32 table.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define STREAM_ENV 0u
#define STREAM_POLICY 1u
#define STREAM_MEMORY 2u
#define STREAM_POLICY_TEST 3u

#define F_LEARN 1u
#define F_NO_REPLAY 2u
#define F_EPISODIC 4u
#define F_MASK 8u
#define F_TEST_STREAM 16u

enum { AG_Q = 0, AG_DYNAQ = 1 };

typedef struct {
  int32_t n_states, n_worlds;
  const uint16_t* next;      /* [W][S][4] */
  const double* reward;      /* [W][S]    */
  const uint8_t* terminal;   /* [W][S]    */
  const uint16_t* starts;    /* concatenated */
  const int32_t* start_off;  /* [W+1]     */
} orc_world;

/* per-instance scalar state, mirrors the kernel's inst record */
typedef struct {
  int32_t state, step, trial;
  uint32_t ctr_env, ctr_policy, ctr_memory, log_len, flags;
  double trial_reward;
  uint64_t steps;
} orc_inst;

typedef struct {
  int32_t n, agent, f32, batch, trials_target, steps_per_trial, step_budget, trial_cap, log_cap;
  uint32_t instance_base, flags;
  double alpha, gamma, epsilon, model_lr;
  uint64_t seed;
} orc_cfg;

/* ------------------------------------------------------------------------------------------- */
static void philox(uint32_t index, uint32_t sub, uint32_t instance, uint32_t stream, uint64_t seed,
                   uint32_t out[4]) {
  uint32_t c0 = index, c1 = sub, c2 = instance, c3 = stream;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static uint32_t bounded(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 32); }
static double u01(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
/* draw counter c -> value (see oracle/philox.py for the stream layout) */
static uint32_t draw_bounded(uint32_t c, uint32_t sub, uint32_t g, uint32_t stream, uint64_t seed,
                             uint32_t n) {
  uint32_t x[4];
  philox(c >> 2, sub, g, stream, seed, x);
  return bounded(x[c & 3u], n);
}
static double draw_u01(uint32_t c, uint32_t g, uint32_t stream, uint64_t seed) {
  uint32_t x[4];
  philox(c >> 1, 0, g, stream, seed, x);
  return (c & 1u) ? u01(x[2], x[3]) : u01(x[0], x[1]);
}
void orc_philox(uint32_t index, uint32_t sub, uint32_t instance, uint32_t stream, uint64_t seed,
                uint32_t* out) { philox(index, sub, instance, stream, seed, out); }

static double rnd(double x, int f32) { return f32 ? (double)(float)x : x; }

/* policy/greedy.py:77-86 + Generator.choice: float64 probabilities, exact-equality ties */
int orc_eps_greedy(const double* v, uint32_t mask, double eps, double u, double* probs) {
  int n = 0, nt = 0;
  double m = -INFINITY;
  for (int a = 0; a < 4; ++a)
    if (mask >> a & 1) { ++n; if (v[a] > m) m = v[a]; }
  for (int a = 0; a < 4; ++a)
    if ((mask >> a & 1) && v[a] == m) ++nt;
  double p[4], c[4];
  for (int a = 0; a < 4; ++a) {
    p[a] = 0.0;
    if (mask >> a & 1) {
      p[a] = eps / (double)n;
      p[a] += ((1.0 - eps) * (v[a] == m ? 1.0 : 0.0)) / (double)nt;
    }
    if (probs) probs[a] = p[a];
  }
  c[0] = p[0];
  for (int a = 1; a < 4; ++a) c[a] = c[a - 1] + p[a];
  int act = 0;
  for (int a = 0; a < 4; ++a)
    if (c[a] / c[3] <= u) ++act;
  return act;
}

static double max4(const double* q) {
  double m = q[0];
  for (int a = 1; a < 4; ++a) if (q[a] > m) m = q[a];
  return m;
}

/* agent/dyna_q.py:290-299 evaluated in the table dtype (online update, QAgent replay) */
static double td_table_dtype(double* Q, int s, int a, double r, int ns, int nt, double alpha,
                             double gamma, int f32) {
  const double g = f32 ? (double)(float)(gamma * nt) : gamma * nt;
  const double al = f32 ? (double)(float)alpha : alpha;
  double td = rnd(rnd(r, f32) + rnd(g * max4(Q + 4 * ns), f32), f32);
  td = rnd(td - Q[4 * s + a], f32);
  Q[4 * s + a] = rnd(Q[4 * s + a] + rnd(al * td, f32), f32);
  return td;
}
/* ... and as the reference evaluates it for planning replays: float64, one rounding on store */
static void td_planning(double* Q, int s, int a, double r, int ns, int nt, double alpha,
                        double gamma, int f32) {
  double td = r + (gamma * (double)nt) * max4(Q + 4 * ns);
  td = td - Q[4 * s + a];
  Q[4 * s + a] = rnd(Q[4 * s + a] + alpha * td, f32);
}

/* Tables per instance: Q[S][4]; Dyna-Q: MR[S][4], MS[S][4], MT[S][4]; QAgent: log arrays.
 * trace (optional, instance trace_inst only): rows of {s, a, r, ns, nt, td}. */
int orc_tab_run(const orc_world* w, const orc_cfg* c, orc_inst* inst, double* Q, double* MR,
                int32_t* MS, int32_t* MT, int32_t* LS, int32_t* LA, double* LR, int32_t* LNS,
                int32_t* LNT, const uint8_t* action_mask, int32_t* lat_trace,
                uint64_t* lat_sum, uint64_t* lat_cnt, double* reward_sum, uint64_t* resp_cnt,
                uint64_t* occupancy,
                int32_t trace_inst, double* trace, int64_t trace_cap, int64_t* trace_len) {
  const int S = w->n_states, f32 = c->f32;
  const int learn = (c->flags & F_LEARN) != 0;
  const int episodic = c->agent == AG_DYNAQ && (c->flags & F_EPISODIC);
  const int B = (learn && !(c->flags & F_NO_REPLAY) && (c->agent == AG_DYNAQ || LS)) ? c->batch : 0;
  const uint32_t pol_stream = (c->flags & F_TEST_STREAM) ? STREAM_POLICY_TEST : STREAM_POLICY;
  const double mlr = f32 ? (double)(float)c->model_lr : c->model_lr;
  if (trace_len) *trace_len = 0;
  for (int i = 0; i < c->n; ++i) {
    const uint32_t g = c->instance_base + (uint32_t)i;
    const int wi = (int)(g % (uint32_t)w->n_worlds);
    const uint16_t* next = w->next + (size_t)wi * S * 4;
    const double* reward = w->reward + (size_t)wi * S;
    const uint8_t* terminal = w->terminal + (size_t)wi * S;
    const uint16_t* starts = w->starts + w->start_off[wi];
    const uint32_t n_starts = (uint32_t)(w->start_off[wi + 1] - w->start_off[wi]);
    orc_inst* in = inst + i;
    double* q = Q + (size_t)i * S * 4;
    double* mr = MR ? MR + (size_t)i * S * 4 : NULL;
    int32_t* ms = MS ? MS + (size_t)i * S * 4 : NULL;
    int32_t* mt = MT ? MT + (size_t)i * S * 4 : NULL;
    const size_t lo = (size_t)i * (size_t)c->log_cap;
    int budget = c->step_budget > 0 ? c->step_budget : 0x7fffffff;
    for (;;) {
      if (!(in->flags & 1u)) {
        if (in->trial >= c->trials_target) break;
        in->state = starts[draw_bounded(in->ctr_env++, 0, g, STREAM_ENV, c->seed, n_starts)];
        in->step = 0;
        in->trial_reward = 0.0;
        in->flags |= 1u;
      }
      if (budget == 0) break;
      --budget;
      const int s = in->state;
      const double u = draw_u01(in->ctr_policy++, g, pol_stream, c->seed);
      const uint32_t mask = (c->flags & F_MASK) ? (action_mask[s] & 15u) : 15u;
      const int a = orc_eps_greedy(q + 4 * s, mask, c->epsilon, u, NULL);
      const int ns = next[4 * s + a];
      const double r = rnd(reward[ns], f32);
      const int end = terminal[ns] != 0, nt = 1 - end;
      double td = 0.0;
      if (learn) {
        if (c->agent == AG_DYNAQ) { /* memory/dyna_q.py:92-96 */
          const double R = mr[4 * s + a];
          mr[4 * s + a] = rnd(R + rnd(mlr * rnd(r - R, f32), f32), f32);
          ms[4 * s + a] = ns;
          mt[4 * s + a] = nt;
        } else if (LS && in->log_len < (uint32_t)c->log_cap) {
          const size_t k = lo + in->log_len++;
          LS[k] = s; LA[k] = a; LR[k] = r; LNS[k] = ns; LNT[k] = nt;
        }
        td = td_table_dtype(q, s, a, r, ns, nt, c->alpha, c->gamma, f32);
      }
      if (trace && i == trace_inst && *trace_len < trace_cap) {
        double* t = trace + 6 * (*trace_len)++;
        t[0] = s; t[1] = a; t[2] = r; t[3] = ns; t[4] = nt; t[5] = td;
      }
      in->trial_reward += r;
      in->steps += 1;
      if (occupancy) occupancy[(size_t)wi * S + ns] += 1;
      const int over = end || (in->step + 1 >= c->steps_per_trial);
      in->state = ns;
      if (B > 0 && !episodic) {
        if (c->agent == AG_DYNAQ) {
          for (int j = 0; j < B; ++j) {
            const uint32_t idx =
                draw_bounded(in->ctr_memory, (uint32_t)j, g, STREAM_MEMORY, c->seed, (uint32_t)S * 4u);
            td_planning(q, (int)(idx >> 2), (int)(idx & 3u), mr[idx], ms[idx], mt[idx], c->alpha,
                        c->gamma, f32);
          }
        } else if (in->log_len > 0) {
          for (int j = 0; j < B; ++j) {
            const size_t k =
                lo + draw_bounded(in->ctr_memory, (uint32_t)j, g, STREAM_MEMORY, c->seed, in->log_len);
            td_table_dtype(q, LS[k], LA[k], LR[k], LNS[k], LNT[k], c->alpha, c->gamma, f32);
          }
        }
        in->ctr_memory += 1;
      }
      if (over) {
        const int t = in->trial;
        if (t >= 0 && t < c->trial_cap) {
          if (lat_sum) lat_sum[t] += (uint64_t)in->step;
          if (lat_cnt) lat_cnt[t] += 1;
          if (reward_sum) reward_sum[t] += in->trial_reward;
          /* ResponseMonitor's default response: int(trial_reward > 0) (monitor/behavior.py:289) */
          if (resp_cnt && in->trial_reward > 0.0) resp_cnt[t] += 1;
          if (lat_trace) lat_trace[(size_t)i * c->trial_cap + t] = in->step;
        }
        in->trial += 1;
        in->flags &= ~1u;
        if (episodic && B > 0) {
          for (int j = 0; j < B; ++j) {
            const uint32_t idx =
                draw_bounded(in->ctr_memory, (uint32_t)j, g, STREAM_MEMORY, c->seed, (uint32_t)S * 4u);
            td_planning(q, (int)(idx >> 2), (int)(idx & 3u), mr[idx], ms[idx], mt[idx], c->alpha,
                        c->gamma, f32);
          }
          in->ctr_memory += 1;
        }
      } else {
        in->step += 1;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum) of the
 * elementwise products a[k] * b[k], each product and each add rounded to the table dtype. */
static double pw_dot(const double* a, const double* b, int n, int f32) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res = rnd(res + rnd(a[i] * b[i], f32), f32);
    return res;
  }
  if (n <= 128) {
    double r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = rnd(a[i] * b[i], f32);
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] = rnd(r[k] + rnd(a[i + k] * b[i + k], f32), f32);
    double res = rnd(rnd(rnd(r[0] + r[1], f32) + rnd(r[2] + r[3], f32), f32) +
                     rnd(rnd(r[4] + r[5], f32) + rnd(r[6] + r[7], f32), f32), f32);
    for (; i < n; ++i) res = rnd(res + rnd(a[i] * b[i], f32), f32);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return rnd(pw_dot(a, b, n2, f32) + pw_dot(a + n2, b + n2, n - n2, f32), f32);
}
double orc_pairwise_dot(const double* a, const double* b, int n, int f32) {
  return pw_dot(a, b, n, f32);
}

/* SR[N][S][S], T[N][S][4], RW[N][S];  qtrace (optional): the 4 q-values seen at every step of
 * instance trace_inst. */
int orc_sr_run(const orc_world* w, const orc_cfg* c, orc_inst* inst, double* SR, int32_t* T,
               double* RW, const uint8_t* action_mask, int32_t* lat_trace, uint64_t* lat_sum,
               uint64_t* lat_cnt, double* reward_sum, uint64_t* resp_cnt, uint64_t* occupancy,
               int32_t trace_inst,
               double* trace, double* qtrace, int64_t trace_cap, int64_t* trace_len) {
  const int S = w->n_states, f32 = c->f32;
  const int learn = (c->flags & F_LEARN) != 0;
  const uint32_t pol_stream = (c->flags & F_TEST_STREAM) ? STREAM_POLICY_TEST : STREAM_POLICY;
  const double al_t = f32 ? (double)(float)c->alpha : c->alpha;   /* table-dtype alpha */
  const double ga_t = f32 ? (double)(float)c->gamma : c->gamma;
  if (trace_len) *trace_len = 0;
  for (int i = 0; i < c->n; ++i) {
    const uint32_t g = c->instance_base + (uint32_t)i;
    const int wi = (int)(g % (uint32_t)w->n_worlds);
    const uint16_t* next = w->next + (size_t)wi * S * 4;
    const double* reward = w->reward + (size_t)wi * S;
    const uint8_t* terminal = w->terminal + (size_t)wi * S;
    const uint16_t* starts = w->starts + w->start_off[wi];
    const uint32_t n_starts = (uint32_t)(w->start_off[wi + 1] - w->start_off[wi]);
    orc_inst* in = inst + i;
    double* sr = SR + (size_t)i * S * S;
    int32_t* tt = T + (size_t)i * S * 4;
    double* rw = RW + (size_t)i * S;
    int budget = c->step_budget > 0 ? c->step_budget : 0x7fffffff;
    for (;;) {
      if (!(in->flags & 1u)) {
        if (in->trial >= c->trials_target) break;
        in->state = starts[draw_bounded(in->ctr_env++, 0, g, STREAM_ENV, c->seed, n_starts)];
        in->step = 0;
        in->trial_reward = 0.0;
        in->flags |= 1u;
      }
      if (budget == 0) break;
      --budget;
      const int s = in->state;
      double q[4];
      for (int a = 0; a < 4; ++a) q[a] = pw_dot(sr + (size_t)tt[4 * s + a] * S, rw, S, f32);
      const double u = draw_u01(in->ctr_policy++, g, pol_stream, c->seed);
      const uint32_t mask = (c->flags & F_MASK) ? (action_mask[s] & 15u) : 15u;
      const int a = orc_eps_greedy(q, mask, c->epsilon, u, NULL);
      const int ns = next[4 * s + a];
      const double r = rnd(reward[ns], f32);
      const int end = terminal[ns] != 0, nt = 1 - end;
      if (learn) { /* sr.py:267-284 */
        const double d = rnd(r - rw[ns], f32);
        rw[ns] = rnd(rw[ns] + rnd(d * al_t, f32), f32);
        tt[4 * s + a] = ns;
        double* row_s = sr + (size_t)s * S;
        const double* row_n = sr + (size_t)ns * S;
        for (int e = 0; e < S; ++e) {
          double td = (e == s) ? 1.0 : 0.0;
          if (nt > 0) td = td + rnd(ga_t * row_n[e], f32);   /* float32 product in f32 mode */
          else td = td + c->gamma * ((e == ns) ? 1.0 : 0.0);
          td = td - row_s[e];
          /* row_n may alias row_s (ns == s): the reference copies SR[ns] first, and every element
             only depends on the same element, so in-place evaluation is equivalent */
          row_s[e] = rnd(row_s[e] + c->alpha * td, f32);
        }
      }
      if (trace && i == trace_inst && *trace_len < trace_cap) {
        double* t = trace + 6 * (*trace_len);
        t[0] = s; t[1] = a; t[2] = r; t[3] = ns; t[4] = nt; t[5] = 0.0;
        if (qtrace) memcpy(qtrace + 4 * (*trace_len), q, sizeof(q));
        ++*trace_len;
      }
      in->trial_reward += r;
      in->steps += 1;
      if (occupancy) occupancy[(size_t)wi * S + ns] += 1;
      const int over = end || (in->step + 1 >= c->steps_per_trial);
      in->state = ns;
      if (over) {
        const int t = in->trial;
        if (t >= 0 && t < c->trial_cap) {
          if (lat_sum) lat_sum[t] += (uint64_t)in->step;
          if (lat_cnt) lat_cnt[t] += 1;
          if (reward_sum) reward_sum[t] += in->trial_reward;
          /* ResponseMonitor's default response: int(trial_reward > 0) (monitor/behavior.py:289) */
          if (resp_cnt && in->trial_reward > 0.0) resp_cnt[t] += 1;
          if (lat_trace) lat_trace[(size_t)i * c->trial_cap + t] = in->step;
        }
        in->trial += 1;
        in->flags &= ~1u;
      } else {
        in->step += 1;
      }
    }
  }
  return 0;
}
