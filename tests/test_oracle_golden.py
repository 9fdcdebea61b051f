"""CPU tests: pin the oracle (NumPy restatement oracle/ref_loop.py and C restatement
oracle/cobel_oracle.c) against every golden vector captured from the real reference
(tests/golden/gen_golden.py).  float64 mode == the reference as shipped; float32 mode == the
reference with its tables cast to float32, which is what the HIP kernels implement."""
import numpy as np
import pytest

from conftest import SEED, cases, coinciding_trials
from oracle import c_oracle, philox, ref_loop
from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    def h(c, k):
        return [int(v) for v in philox.philox4x32(c, k)]
    assert h([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert h([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert h([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_c_philox_matches_numpy():
    import ctypes as C
    out = (C.c_uint32 * 4)()
    rng = np.random.default_rng(0)
    for _ in range(200):
        idx, sub, inst, stream = [int(x) for x in rng.integers(0, 2**32, 4)]
        seed = int(rng.integers(0, 2**63))
        c_oracle.lib().orc_philox(C.c_uint32(idx), C.c_uint32(sub), C.c_uint32(inst),
                                  C.c_uint32(stream), C.c_uint64(seed), out)
        ref = philox.philox4x32([idx, sub, inst, stream], [seed & 0xffffffff, seed >> 32])
        assert list(out) == [int(v) for v in ref]


def test_tape_rng_reproduces_generator_choice():
    """TapeRNG.choice == numpy Generator.choice when fed the same underlying draws."""
    for seed in range(100):
        g1, g2 = np.random.default_rng(seed), np.random.default_rng(seed)
        p = np.random.default_rng(seed + 1000).random(4)
        p /= p.sum()
        t = TapeRNG(0, 0, 0)
        t.random = g2.random
        assert g1.choice(np.arange(4), p=p) == t.choice(np.arange(4), p=p)
        t.integers = lambda lo, hi=None, size=None: g2.integers(lo, hi, size)
        a = np.arange(10, 30)
        assert g1.choice(a) == t.choice(a)
        assert np.array_equal(g1.choice(7, 5), t.choice(7, 5))


def test_eps_greedy_kat(golden):
    rows = golden('eps_greedy_kat')['rows']
    for r in rows:
        is32, eps, v, bits, u, act, p = r[0], r[1], r[2:6], int(r[6]), r[7], int(r[8]), r[9:13]
        v = v.astype(np.float32 if is32 else np.float64)
        a_c, p_c = c_oracle.eps_greedy(v, bits, eps, u)
        assert a_c == act and np.array_equal(p_c, p)
    # the NumPy restatement, on a subsample (it is slow)
    for r in rows[::37]:
        pol = ref_loop.RefEpsilonGreedy(r[1], None)
        mask = np.array([(int(r[6]) >> i) & 1 for i in range(4)], dtype=bool)
        v = r[2:6].astype(np.float32 if r[0] else np.float64)
        assert np.array_equal(pol.get_action_probs(v, mask), r[9:13])


@pytest.mark.parametrize('S', [5, 7, 8, 9, 25, 64, 100, 128, 129, 200, 1000, 1024, 1025, 4096])
def test_pairwise_dot_is_numpy_sum(S):
    """orc_pairwise_dot == np.sum(a * b) (NumPy pairwise summation) in float32 and float64."""
    rng = np.random.default_rng(S)
    for _ in range(5):
        a = rng.standard_normal(S) * rng.random(S) ** 8
        b = rng.standard_normal(S)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        assert c_oracle.pairwise_dot(a32, b32, True) == float(np.sum((a32 * b32)[None], axis=1)[0])
        assert c_oracle.pairwise_dot(a, b, False) == float(np.sum((a * b)[None], axis=1)[0])


def _world(golden_worlds, name):
    t = golden_worlds(name)
    return {k: t[k] for k in ('next', 'reward', 'terminal', 'starts')}


def test_gridworld_kat(golden, golden_worlds):
    """unit_tests/test_gridworld.py:16-41 through the oracle env."""
    k = golden('gridworld_kat')
    env = ref_loop.RefGridworld(_world(golden_worlds, 'kat_5x5'), TapeRNG(SEED, 0, STREAM_ENV))
    s, _ = env.reset()
    assert s == 24 == int(k['start'])
    got = [env.step(a)[:3] for a in k['actions']]
    assert [g[0] for g in got] == list(k['states'])
    assert [float(g[1]) for g in got] == list(k['rewards'])
    assert [g[2] for g in got] == list(k['terminals'])


DQ = None


def _dq(golden):
    global DQ
    if DQ is None:
        DQ = golden('dynaq_traces')
    return DQ


@pytest.mark.parametrize('name', ['maze32_b50_f32', 'open5_b32_f32', 'open5_b32_f64',
                                  'open5_b50_f32_i7', 'open5_episodic_f32', 'open5_noreplay_f32',
                                  'walls8_b8_f32', 'walls8_b8_f64', 'walls8_mask_f32',
                                  'walls8_traintest_f32', 'walls8_b130_f32'])
def test_dynaq_golden(golden, golden_worlds, name):
    D = _dq(golden)
    inst, f32, trials, steps, B, norep, epi, mask, tt, nts = [int(x) for x in D[name + '/cfg']]
    tab = _world(golden_worlds, str(D[name + '/world']))
    S = tab['next'].shape[0]
    am = D[name + '/action_mask'] if mask else None
    # --- C oracle ---
    w = c_oracle.OracleWorld([tab])
    o = c_oracle.TabOracle(w, 1, c_oracle.AG_DYNAQ, SEED, bool(f32), instance_base=inst,
                           trial_cap=trials + tt, action_mask=am)
    flags = c_oracle.F_LEARN | (c_oracle.F_NO_REPLAY if norep else 0) | \
        (c_oracle.F_EPISODIC if epi else 0)
    tr = o.run(trials, steps, B, flags, trace_inst=0, trace_cap=100000)
    if tt:
        tr = np.concatenate([tr, o.run(trials + tt, steps, 0, 0, trace_inst=0, trace_cap=100000)])
    assert np.array_equal(tr[:, 0], D[name + '/state'])
    assert np.array_equal(tr[:, 1], D[name + '/action'])
    assert np.array_equal(tr[:, 2], D[name + '/reward'])
    assert np.array_equal(tr[:, 3], D[name + '/next_state'])
    assert np.array_equal(tr[:, 4], D[name + '/nonterminal'])
    assert np.array_equal(tr[:nts, 5], D[name + '/td'][:nts])
    assert np.array_equal(o.lat_trace[0], D[name + '/steps'])
    assert np.array_equal(o.reward_sum, D[name + '/trial_reward'])
    assert np.array_equal(o.Q[0], D[name + '/Q'])
    assert np.array_equal(o.MR[0], D[name + '/M_rewards'])
    assert np.array_equal(o.MS[0], D[name + '/M_states'])
    assert np.array_equal(o.MT[0], D[name + '/M_terminals'])
    # --- NumPy restatement (skip the long maze run) ---
    if name.startswith('maze32'):
        return
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, inst, STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    ag = ref_loop.RefDynaQ(S, 4, pol, TapeRNG(SEED, inst, STREAM_MEMORY),
                           dtype=np.float32 if f32 else np.float64)
    if mask:
        ag.mask_actions, ag.action_mask = True, am
    ag.episodic_replay = bool(epi)
    t2 = ref_loop.new_trace(True)
    ag.train(env, trials, steps, B, bool(norep), trace=t2)
    if tt:
        ag.test(env, tt, steps, trace=t2)
    a = np.array(t2['sarsn'])
    assert np.array_equal(a, tr[:, :5])
    assert np.array_equal(ag.Q.astype(np.float64), D[name + '/Q'])
    assert np.array_equal(np.array(t2['Q_trial'], dtype=np.float64), D[name + '/Q_trial'])
    assert np.array_equal(ag.M.rewards.astype(np.float64), D[name + '/M_rewards'])


@pytest.mark.parametrize('name', ['open5_b0_f32', 'open5_b0_f64', 'open5_b8_f32',
                                  'walls8_b16_f32', 'walls8_b16_f64', 'walls8_b100_f32'])
def test_qagent_golden(golden, golden_worlds, name):
    D = golden('qagent_traces')
    inst, f32, trials, steps, B = [int(x) for x in D[name + '/cfg']]
    tab = _world(golden_worlds, str(D[name + '/world']))
    w = c_oracle.OracleWorld([tab])
    o = c_oracle.TabOracle(w, 1, c_oracle.AG_Q, SEED, bool(f32), instance_base=inst, alpha=0.9,
                           gamma=0.8, trial_cap=trials, log_cap=trials * steps if B else 0)
    tr = o.run(trials, steps, B, trace_inst=0, trace_cap=100000)
    assert np.array_equal(tr[:, 0], D[name + '/state'])
    assert np.array_equal(tr[:, 1], D[name + '/action'])
    assert np.array_equal(tr[:, 3], D[name + '/next_state'])
    assert np.array_equal(o.lat_trace[0], D[name + '/steps'])
    assert np.array_equal(o.Q[0], D[name + '/Q'])
    if B:
        assert int(o.inst['log_len'][0]) == int(D[name + '/log_len'])
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, inst, STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    ag = ref_loop.RefQAgent(tab['next'].shape[0], 4, pol, TapeRNG(SEED, inst, STREAM_MEMORY),
                            dtype=np.float32 if f32 else np.float64)
    ag.train(env, trials, steps, B)
    assert np.array_equal(ag.Q.astype(np.float64), D[name + '/Q'])


@pytest.mark.parametrize('name', ['open5_f32', 'open5_f64', 'walls8_f32', 'walls8_f64',
                                  'walls8_mask_f32'])
def test_sr_golden(golden, golden_worlds, name):
    D = golden('sr_traces')
    inst, f32, trials, steps, mask = [int(x) for x in D[name + '/cfg']]
    tab = _world(golden_worlds, str(D[name + '/world']))
    am = D[name + '/action_mask'] if mask else None
    w = c_oracle.OracleWorld([tab])
    o = c_oracle.SROracle(w, 1, SEED, bool(f32), instance_base=inst, trial_cap=trials,
                          action_mask=am)
    tr, q = o.run(trials, steps, trace_inst=0, trace_cap=100000)
    assert np.array_equal(tr[:, 0], D[name + '/state'])
    assert np.array_equal(tr[:, 1], D[name + '/action'])
    assert np.array_equal(tr[:, 3], D[name + '/next_state'])
    assert np.array_equal(q, D[name + '/q'])
    assert np.array_equal(o.lat_trace[0], D[name + '/steps'])
    assert np.array_equal(o.SR[0], D[name + '/SR'])
    assert np.array_equal(o.RW[0], D[name + '/rewards'])
    assert np.array_equal(o.T[0], D[name + '/T'])
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, inst, STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    ag = ref_loop.RefSR(tab['next'].shape[0], 4, pol, dtype=np.float32 if f32 else np.float64)
    if mask:
        ag.mask_actions, ag.action_mask = True, am
    ag.train(env, trials, steps)
    assert np.array_equal(ag.SR.astype(np.float64), D[name + '/SR'])
    assert np.array_equal(ag.T, D[name + '/T'])


def _sr32_full(D, name):
    sr = np.eye(1024, dtype=np.float32)
    sr[D[name + '/SR_rows'].astype(int)] = D[name + '/SR_values']
    return sr


@pytest.mark.parametrize('name', ['open32_near_f32', 'open32_dense_f32', 'open32_r20_f32'])
def test_sr_golden_at_32x32(golden, golden_worlds, name):
    """Config C4's own size: 1 024-term row sums (`retrieve_q`, agent/sr.py:288-308) of the REAL
    reference — one, twenty and 1 024 non-zero reward estimates — against the C restatement's
    pairwise order and the NumPy restatement, bit for bit."""
    D = golden('sr32_traces')
    inst, f32, trials, steps, _ = [int(x) for x in D[name + '/cfg']]
    tab = _world(golden_worlds, str(D[name + '/world']))
    rw0 = D[name + '/rewards0'] if name + '/rewards0' in D.files else None
    w = c_oracle.OracleWorld([tab])
    o = c_oracle.SROracle(w, 1, SEED, True, instance_base=inst, trial_cap=trials)
    if rw0 is not None:
        o.RW[0] = rw0
    tr, q = o.run(trials, steps, trace_inst=0, trace_cap=100000)
    assert np.array_equal(tr[:, 0], D[name + '/state'])
    assert np.array_equal(tr[:, 1], D[name + '/action'])
    assert np.array_equal(tr[:, 3], D[name + '/next_state'])
    assert np.array_equal(q, D[name + '/q'])
    assert np.array_equal(o.lat_trace[0], D[name + '/steps'])
    assert np.array_equal(o.SR[0].astype(np.float32), _sr32_full(D, name))
    assert np.array_equal(o.RW[0].astype(np.float32), D[name + '/rewards'])
    assert np.array_equal(o.T[0], D[name + '/T'])
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, inst, STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    ag = ref_loop.RefSR(1024, 4, pol, dtype=np.float32)
    if rw0 is not None:
        ag.rewards = rw0.copy()
    trace = {'sarsn': [], 'q': [], 'steps': [], 'trial_reward': []}
    ag.train(env, trials, steps, trace)
    assert np.array_equal(np.array(trace['q']), D[name + '/q'])
    assert np.array_equal(ag.SR, _sr32_full(D, name)) and np.array_equal(ag.T, D[name + '/T'])
    assert np.array_equal(ag.rewards, D[name + '/rewards'])


def test_float32_tables_track_float64_reference(golden):
    """North-star tolerance, on the reference's own two runs: with identical draws the float32 run
    stays within 1e-6 of the float64 run on the 5x5 config for as long as the trajectories
    coincide (they fork when a float32 tie is not a float64 tie)."""
    D = _dq(golden)
    same = coinciding_trials(D, 'open5_b32_f32', 'open5_b32_f64')
    assert same >= 5
    d = np.abs(D['open5_b32_f32/Q_trial'][:same] - D['open5_b32_f64/Q_trial'][:same])
    assert d.max() <= 1e-6


def test_monitor_kat(golden):
    k = golden('monitor_kat')
    lat = np.full(len(k['steps']), np.nan)
    lat[k['order']] = k['steps'][k['order']]
    assert np.array_equal(lat, k['latency'], equal_nan=True)
    # the reference fills avg in arrival order; trials arrive in increasing order in the fixture
    avg = ref_loop.escape_latency_avg(lat, int(k['max_steps']))
    assert np.array_equal(avg, k['latency_avg'], equal_nan=True)


def test_many_instances_are_independent_of_batching(golden_worlds):
    """C oracle: running instances one by one, all at once, or in budgeted chunks is identical."""
    tab = _world(golden_worlds, 'walls_8x8')
    w = c_oracle.OracleWorld([tab])
    whole = c_oracle.TabOracle(w, 6, c_oracle.AG_DYNAQ, 7, True, trial_cap=5)
    whole.run(5, 40, 12)
    chunk = c_oracle.TabOracle(w, 6, c_oracle.AG_DYNAQ, 7, True, trial_cap=5)
    for _ in range(100):
        chunk.run(5, 40, 12, step_budget=3)
    assert np.array_equal(whole.Q, chunk.Q) and np.array_equal(whole.lat_trace, chunk.lat_trace)
    for i in range(6):
        one = c_oracle.TabOracle(w, 1, c_oracle.AG_DYNAQ, 7, True, instance_base=i, trial_cap=5)
        one.run(5, 40, 12)
        assert np.array_equal(one.Q[0], whole.Q[i])


@pytest.mark.parametrize('name', ['f32', 'f64'])
def test_dynaq_memory_kat(golden, name):
    """oracle RefDynaQMemory == the reference's DynaQMemory (memory/dyna_q.py:62-157) over a
    sequence of stores interleaved with retrieve_batch draws, float32 and float64 tables."""
    from oracle import philox, ref_loop
    k = golden('dynaq_memory_kat')
    inst = int(k[name + '/instance'])
    M = ref_loop.RefDynaQMemory(25, 4, philox.TapeRNG(SEED, inst, philox.STREAM_MEMORY),
                                dtype=np.float32 if name == 'f32' else np.float64)
    stores, batches = iter(k[name + '/stores']), k[name + '/batches']
    got = []
    for op in k[name + '/ops']:
        if op == 0:
            s, a, r, ns, nt = next(stores)
            M.store(int(s), int(a), float(r), int(ns), int(nt))
        else:
            got.extend(M.sample(int(op))[0])
    assert np.array_equal(np.array(got, dtype=np.float64), batches)
    assert np.array_equal(M.rewards.astype(np.float64), k[name + '/rewards'])
    assert np.array_equal(M.states, k[name + '/states'])
    assert np.array_equal(M.terminals, k[name + '/terminals'])


def test_eps_greedy_six_actions_kat(golden):
    """policy/greedy.py:40-88 over six values (a hexagonal Topology's action space), masks and
    draws at the CDF edges: the NumPy restatement against the reference's rows."""
    rows = golden('eps_greedy_kat')['rows6']
    assert len(rows) > 2000
    for r in rows[::7]:
        eps, v, bits, u, act, p = r[0], r[1:7].astype(np.float32), int(r[7]), r[8], int(r[9]), r[10:16]
        mask = np.array([(bits >> i) & 1 for i in range(6)], dtype=bool)

        class _U:
            def random(self, size=None, u=u):
                return u
        pol = ref_loop.RefEpsilonGreedy(eps, _U())
        assert np.array_equal(pol.get_action_probs(v, mask), p)
        assert pol.select_action(v, mask) == act


@pytest.mark.parametrize('name', ['hex5_b0_f32', 'hex5_b8_f32', 'hex4_b70_f32'])
def test_qagent_on_hexagonal_topology_golden(golden, name):
    """QAgent on the six-action hexagonal graphs (misc/topology_tools.py:175-272; the action space
    is the neighbour count, interface/topology.py:110-112): the NumPy restatement against the
    reference's float32 run — trajectory, escape latencies, Q table, replay log length; one case
    with a replay batch (70) beyond what one wavefront plans."""
    D, K = golden('qagent_topology_traces'), golden('topology_kat')
    inst, f32, trials, steps, B = [int(x) for x in D[name + '/cfg']]
    graph = {'hex5': 'hex_5_goal7', 'hex4': 'hex_4'}[name[:4]]
    tab = dict(next=K[graph + '/nbr'], reward=K[graph + '/reward'],
               terminal=K[graph + '/terminal'], starts=K[graph + '/starts'])
    assert tab['next'].shape[1] == 6
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, inst, STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    ag = ref_loop.RefQAgent(tab['next'].shape[0], 6, pol, TapeRNG(SEED, inst, STREAM_MEMORY),
                            dtype=np.float32 if f32 else np.float64)
    tr = ref_loop.new_trace()
    ag.train(env, trials, steps, B, trace=tr)
    got = np.array(tr['sarsn'], dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(got[:, col], D['%s/%s' % (name, key)]), key
    assert np.array_equal(tr['steps'], D[name + '/steps'])
    assert np.array_equal(ag.Q.astype(np.float64), D[name + '/Q'])
    assert len(ag.M) == int(D[name + '/log_len'])


# ---------------------------------------------------------------------------------------------
# Worlds whose transition rows are distributions (interface/gridworld.py:119-123: the successor is
# drawn from sas[s][a]); fixtures from the reference with its sas edited, tests/golden/gen_golden.py
# gen_stochastic.
def stochastic_world(Z, wname):
    return {'next': Z[wname + '/next'], 'reward': Z[wname + '/reward'],
            'terminal': Z[wname + '/terminal'], 'starts': Z[wname + '/starts'],
            'sas': Z[wname + '/sas']}


@pytest.mark.parametrize('wname', ['slip_4x4', 'slip_5x6_wind'])
def test_stochastic_step_known_answers(golden, wname):
    """(state, action, uniform) -> successor, reward, end as the reference's step() returns them,
    including uniforms on and next to the edges of the cumulative distribution."""
    Z = golden('stochastic_traces')
    tab = stochastic_world(Z, wname)
    assert (np.count_nonzero(tab['sas'], axis=2) > 1).any()
    env = ref_loop.RefGridworld(tab, TapeRNG(SEED, 0, STREAM_ENV, double_sub=1))
    for s, a, u, ns, r, end in Z[wname + '/step_kat']:
        env.current_state = int(s)
        env.rng.random = lambda size=None, u=u: u
        got = env.step(int(a))
        assert (got[0], float(got[1]), got[2]) == (int(ns), float(r), bool(end))


@pytest.mark.parametrize('name', ['slip4_dynaq_b8', 'slip4_q_b4', 'slip4_q_b0', 'slip56_dynaq_b70'])
def test_stochastic_world_runs_golden(golden, name):
    """Dyna-Q and Q-learning on the slippery worlds: the restatement against the reference's
    float32 run — trajectory, TD errors, latencies, tables, and the number of draws the env stream
    handed out (one integer per trial start, one double per step)."""
    Z = golden('stochastic_traces')
    inst, trials, steps, B = [int(x) for x in Z[name + '/cfg']]
    tab = stochastic_world(Z, str(Z[name + '/world']))
    S = tab['next'].shape[0]
    erng = TapeRNG(SEED, inst, STREAM_ENV, double_sub=1)
    env = ref_loop.RefGridworld(tab, erng)
    pol = ref_loop.RefEpsilonGreedy(0.1, TapeRNG(SEED, inst, STREAM_POLICY))
    tr = ref_loop.new_trace()
    if str(Z[name + '/agent']) == 'dynaq':
        ag = ref_loop.RefDynaQ(S, 4, pol, TapeRNG(SEED, inst, STREAM_MEMORY), dtype=np.float32)
        ag.train(env, trials, steps, B, trace=tr)
        assert np.array_equal(ag.M.rewards.astype(np.float64), Z[name + '/M_rewards'])
        assert np.array_equal(ag.M.states, Z[name + '/M_states'])
        assert np.array_equal(ag.M.terminals, Z[name + '/M_terminals'])
    else:
        ag = ref_loop.RefQAgent(S, 4, pol, TapeRNG(SEED, inst, STREAM_MEMORY), dtype=np.float32)
        ag.train(env, trials, steps, B, trace=tr)
        assert len(ag.M) == int(Z[name + '/log_len'])
    got = np.array(tr['sarsn'], dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(got[:, col], Z['%s/%s' % (name, key)]), key
    assert np.array_equal(tr['steps'], Z[name + '/steps'])
    assert np.array_equal(ag.Q.astype(np.float64), Z[name + '/Q'])
    assert erng.index == int(Z[name + '/env_draws'])
    if B == 0 or str(Z[name + '/agent']) == 'dynaq':
        assert np.array_equal(np.array(tr['td']), Z[name + '/td'])
    # the walk is not the one of the argmax table
    assert (got[:, 3] != tab['next'][got[:, 0].astype(int), got[:, 1].astype(int)]).any()
