"""GPU parity tests proper: the HIP path (through the C ABI / host classes) against the golden
vectors captured from the real reference, and against the oracle on the same seeded inputs.

Bar: bit-exact for states / actions / rewards / terminals / step counts and for float32 tables
against the reference run with float32 tables; <= 1e-6 against the float64 reference for as long
as the two trajectories coincide (an exact-equality tie in float32 that is not a tie in float64
legitimately forks them — SURVEY.md §8c)."""
import os
import numpy as np
import pytest

from conftest import SEED, as_world, cases, coinciding_trials

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch


# ---------------------------------------------------------------------------------------------
def test_rng_streams_match_oracle(torch_cuda):
    """Device Philox streams == oracle/philox.py for scattered (instance, index, sub)."""
    torch = torch_cuda
    from cobel_amd import _lib
    from oracle import philox
    dev = torch.device('cuda', 0)
    n, base = 1000, 123456
    idx = torch.arange(n, dtype=torch.int32, device=dev) * 7919 + 5
    u = torch.empty(n, dtype=torch.float64, device=dev)
    for stream in (0, 1, 2, 3):
        _lib.check(_lib.lib().cobel_rng_uniform(_lib.ptr(idx), SEED, stream, base, _lib.ptr(u), n,
                                                0, None))
        ref = philox.draw_double(SEED, base + np.arange(n), idx.cpu().numpy().astype(np.uint32),
                                 0, stream)
        assert np.array_equal(u.cpu().numpy(), ref)
    out = torch.empty((n, 50), dtype=torch.int32, device=dev)
    for bound in (4, 100, 4096, 2**31 + 12345):
        _lib.check(_lib.lib().cobel_rng_bounded(_lib.ptr(idx), SEED, 2, base, bound, _lib.ptr(out),
                                                n, 50, 0, None))
        ref = philox.draw_bounded(SEED, (base + np.arange(n))[:, None],
                                  idx.cpu().numpy().astype(np.uint32)[:, None],
                                  np.arange(50)[None, :], 2, bound)
        got = out.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert np.array_equal(got, ref)
    before = idx.clone()
    _lib.check(_lib.lib().cobel_rng_uniform(_lib.ptr(idx), SEED, 1, base, _lib.ptr(u), n, 1, None))
    assert torch.equal(idx, before + 1)


def test_gridworld_kat(torch_cuda, golden):
    """unit_tests/test_gridworld.py:16-41 verbatim against the HIP env."""
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.spaces import Discrete
    k = golden('gridworld_kat')
    world = make_gridworld(5, 5, [0], np.array([[0, 10.]]), starting_states=[24])
    env = Gridworld(world)
    assert isinstance(env.observation_space, Discrete) and isinstance(env.action_space, Discrete)
    assert env.observation_space.n == 25 and env.action_space.n == 4
    state, _ = env.reset()
    assert state == 24 == int(k['start'])
    states, rewards, terminals = [], [], []
    for action in [0, 0, 0, 0, 0, 1, 1, 1, 1]:
        state, reward, terminal, trunc, info = env.step(action)
        assert type(state) is int and trunc is False and info == {}
        states.append(state), rewards.append(reward), terminals.append(terminal)
    assert states == [23, 22, 21, 20, 20, 15, 10, 5, 0] == list(k['states'])
    assert rewards == [0] * 8 + [10.] and terminals == [False] * 8 + [True]
    assert np.array_equal(env.get_position(), world['coordinates'][0])


def test_env_vectorised_matches_tables(torch_cuda, golden_worlds):
    """N instances over two different mazes: every transition equals the golden tables."""
    torch = torch_cuda
    from cobel_amd.interface import Gridworld
    tabs = [golden_worlds('maze_32x32_1234'), golden_worlds('maze_32x32_1235')]
    env = Gridworld([as_world(t) for t in tabs], n_envs=4096, seed=SEED, instance_base=3)
    rng = np.random.default_rng(0)
    s0 = env.state.cpu().numpy()
    w = (3 + np.arange(4096)) % 2
    for t, ww in zip(tabs, (0, 1)):
        assert np.isin(s0[w == ww], t['starts']).all()
    from oracle import philox
    for ww in (0, 1):   # constructor consumed draw 0, so the current state came from draw 0
        sel = np.flatnonzero(w == ww)
        k = philox.draw_bounded(SEED, 3 + sel, 0, 0, 0, len(tabs[ww]['starts']))
        assert np.array_equal(s0[sel], tabs[ww]['starts'][k])
    for _ in range(20):
        a = rng.integers(0, 4, 4096)
        s = env.state.cpu().numpy()
        ns, r, d, _, _ = env.step(torch.as_tensor(a))
        exp_ns = np.where(w == 0, tabs[0]['next'][s, a], tabs[1]['next'][s, a])
        assert np.array_equal(ns.cpu().numpy(), exp_ns)
        exp_r = np.where(w == 0, tabs[0]['reward'][exp_ns], tabs[1]['reward'][exp_ns])
        assert np.array_equal(r.cpu().numpy(), exp_r.astype(np.float32))
        exp_d = np.where(w == 0, tabs[0]['terminal'][exp_ns], tabs[1]['terminal'][exp_ns])
        assert np.array_equal(d.cpu().numpy(), exp_d.astype(bool))
    mask = rng.integers(0, 2, 4096).astype(bool)
    before = env.state.cpu().numpy().copy()
    env.reset(mask)
    after = env.state.cpu().numpy()
    assert np.array_equal(after[~mask], before[~mask])


def test_eps_greedy_kat(torch_cuda, golden):
    """policy/greedy.py get_action_probs + injected-u select_action: exact actions, exact probs."""
    from cobel_amd.policy import EpsilonGreedy
    rows = golden('eps_greedy_kat')['rows']
    rows = rows[rows[:, 0] == 1]      # the float32-valued rows are the build's dtype
    for eps in np.unique(rows[:, 1]):
        r = rows[rows[:, 1] == eps]
        pol = EpsilonGreedy(float(eps))
        v = r[:, 2:6].astype(np.float32)
        bits = r[:, 6].astype(int)
        mask = np.stack([(bits >> i) & 1 for i in range(4)], axis=1).astype(bool)
        act = pol.select_action(v, mask, u=r[:, 7]).cpu().numpy()
        assert np.array_equal(act, r[:, 8].astype(int))
        probs = pol.get_action_probs(v, mask)
        assert np.array_equal(probs, r[:, 9:13])
    one = EpsilonGreedy(0.1)
    assert one.select_action(np.array([0., 1., 0., 0.]), None, u=0.5) == 1
    assert np.allclose(one.get_action_probs(np.zeros(4)), 0.25)
    with pytest.raises(AssertionError):
        one.select_action(np.zeros(4), np.zeros(4, dtype=bool), u=0.1)


# ---------------------------------------------------------------------------------------------
class Spy:
    def __init__(self):
        self.sarsn, self.td, self.steps, self.reward, self.q = [], [], [], [], []

    def step_end(self, logs):
        self.sarsn.append((logs['state'], logs['action'], logs['reward'], logs['next_state'],
                           logs['terminal']))
        self.td.append(logs.get('td', 0.0))

    def trial_end(self, logs):
        self.steps.append(logs['steps'])
        self.reward.append(logs['trial_reward'])
        q = logs['agent'].Q
        self.q.append(np.array(q, dtype=np.float64))


def _dynaq(golden, golden_worlds, name, n_envs, base, callbacks=None):
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dynaq_traces')
    inst, f32, trials, steps, B, norep, epi, mask, tt, nts = [int(x) for x in D[name + '/cfg']]
    env = Gridworld(as_world(golden_worlds(str(D[name + '/world']))), n_envs=n_envs, seed=SEED,
                    instance_base=base)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                  custom_callbacks=callbacks)
    agent.track_instances = True
    if mask:
        agent.mask_actions = True
        agent.action_mask = D[name + '/action_mask']
    agent.episodic_replay = bool(epi)
    agent.train(env, trials, steps, B, bool(norep))
    if tt:
        agent.test(env, tt, steps)
    return D, agent, env, inst


DYNAQ_F32 = ['open5_b32_f32', 'open5_b50_f32_i7', 'open5_noreplay_f32', 'open5_episodic_f32',
             'walls8_b8_f32', 'walls8_mask_f32', 'walls8_traintest_f32', 'maze32_b50_f32',
             'walls8_b130_f32']


@pytest.mark.parametrize('name', DYNAQ_F32)
def test_dynaq_golden_vectorised(torch_cuda, golden, golden_worlds, name):
    """8 instances in one launch; the instance the fixture was recorded for must reproduce the
    reference's float32 run bit for bit: steps per trial, Q, model tables."""
    D, agent, env, inst = _dynaq(golden, golden_worlds, name, 8, 0)
    steps = agent.monitors.lat_trace[inst].cpu().numpy()
    assert np.array_equal(steps[: len(D[name + '/steps'])], D[name + '/steps'])
    assert np.array_equal(agent.Q[inst].cpu().numpy().astype(np.float64), D[name + '/Q'])
    assert np.array_equal(agent.M.rewards[inst].astype(np.float64), D[name + '/M_rewards'])
    assert np.array_equal(agent.M.states[inst], D[name + '/M_states'])
    assert np.array_equal(agent.M.terminals[inst], D[name + '/M_terminals'])
    total = int(D[name + '/steps'].astype(np.int64).sum() + len(D[name + '/steps']))
    assert int(agent.inst[inst, 10].item()) == total   # lifetime env steps of that instance


@pytest.mark.parametrize('name', ['open5_b32_f32', 'walls8_traintest_f32', 'walls8_mask_f32'])
def test_dynaq_golden_per_step_callbacks(torch_cuda, golden, golden_worlds, name):
    """n_envs = 1 with the reference's per-step callbacks: the full (s, a, r, s', nt, td) stream,
    trial rewards and the Q table after every trial match the float32 reference run."""
    spy = Spy()
    D = golden('dynaq_traces')
    inst = int(D[name + '/cfg'][0])
    cbs = {'on_step_end': [spy.step_end], 'on_trial_end': [spy.trial_end]}
    D, agent, env, _ = _dynaq(golden, golden_worlds, name, 1, inst, cbs)
    a = np.array(spy.sarsn, dtype=np.float64)
    assert np.array_equal(a[:, 0], D[name + '/state'])
    assert np.array_equal(a[:, 1], D[name + '/action'])
    assert np.array_equal(a[:, 2], D[name + '/reward'])
    assert np.array_equal(a[:, 3], D[name + '/next_state'])
    assert np.array_equal(a[:, 4], D[name + '/nonterminal'])
    nts = int(D[name + '/cfg'][9])
    assert np.array_equal(np.array(spy.td)[:nts], D[name + '/td'][:nts])
    assert np.array_equal(spy.steps, D[name + '/steps'])
    assert np.array_equal(spy.reward, D[name + '/trial_reward'])
    if name + '/Q_trial' in D.files:
        assert np.array_equal(np.array(spy.q), D[name + '/Q_trial'])


@pytest.mark.parametrize('pair', [('open5_b32_f32', 'open5_b32_f64', 1e-6),
                                  ('walls8_b8_f32', 'walls8_b8_f64', 2e-6)])
def test_dynaq_float32_vs_float64_reference(torch_cuda, golden, golden_worlds, pair):
    """Against the float64 reference: identical trajectory and |dQ| <= 1e-6 (north-star tolerance,
    on the 5x5 config; 2e-6 relative to max(1, |Q|) on the 8x8 world, whose re-collectable reward
    lets Q grow past 10 so that 25 trials of float32 rounding reach 1.1e-6 relative — this is the
    reference's own float32-vs-float64 drift, the kernel being bit-identical to its float32 run) up
    to the first trial in which a float32 tie forks the two runs (a few trials must coincide)."""
    f32, f64, tol = pair
    spy = Spy()
    D = golden('dynaq_traces')
    inst = int(D[f32 + '/cfg'][0])
    _dynaq(golden, golden_worlds, f32, 1, inst, {'on_trial_end': [spy.trial_end]})
    ref_q = D[f64 + '/Q_trial']
    assert np.array_equal(spy.steps, D[f32 + '/steps'])      # the kernel IS the float32 run
    same = coinciding_trials(D, f32, f64)
    assert same >= 3
    for t in range(same):
        assert np.max(np.abs(spy.q[t] - ref_q[t]) / np.maximum(1.0, np.abs(ref_q[t]))) <= tol


def test_dynaq_chunking_and_sharding_invariance(torch_cuda, golden, golden_worlds):
    """Results do not depend on how a run is cut into launches (step_budget) or how instances are
    split over devices (instance_base): draws are a function of the global instance id only."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('walls_8x8'))

    def run(n, base, budget):
        env = Gridworld(world, n_envs=n, seed=99, instance_base=base)
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False)
        ag.monitors.reserve(6, n, True)
        for _ in range(200):
            ag._launch(env, ag.policy, flags, 6, 40, budget, 20)
            if int(ag.inst[:, _lib.I_TRIAL].min().item()) >= 6:
                break
        return ag

    whole = run(16, 0, 0)
    chunked = run(16, 0, 7)
    assert torch.equal(whole._q, chunked._q) and torch.equal(whole.M.table, chunked.M.table)
    assert torch.equal(whole.monitors.lat_trace, chunked.monitors.lat_trace)
    assert torch.equal(whole.inst[:, :8], chunked.inst[:, :8])
    lo, hi = run(8, 0, 0), run(8, 8, 0)
    assert torch.equal(whole._q, torch.cat([lo._q, hi._q]))
    assert torch.equal(whole.monitors.lat_sum, lo.monitors.lat_sum + hi.monitors.lat_sum)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['open5_b0_f32', 'open5_b8_f32', 'walls8_b16_f32', 'walls8_b100_f32'])
def test_qagent_golden(torch_cuda, golden, golden_worlds, name):
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    D = golden('qagent_traces')
    inst, f32, trials, steps, B = [int(x) for x in D[name + '/cfg']]
    env = Gridworld(as_world(golden_worlds(str(D[name + '/world']))), n_envs=4, seed=SEED,
                    instance_base=0)
    agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    agent.track_instances = True
    agent.train(env, trials, steps, B)
    assert np.array_equal(agent.monitors.lat_trace[inst].cpu().numpy(), D[name + '/steps'])
    assert np.array_equal(agent.Q[inst].cpu().numpy().astype(np.float64), D[name + '/Q'])
    # (the reference appends to M whether it replays or not: q.py:213 — also at batch_size 0)
    assert int(agent.inst[inst, 6].item()) == int(D[name + '/log_len'])
    if inst == 0:
        assert len(agent.M) == int(D[name + '/log_len'])


@pytest.mark.parametrize('name', ['open5_f32', 'walls8_f32', 'walls8_mask_f32'])
def test_sr_golden(torch_cuda, golden, golden_worlds, name):
    """SR tables, learned transitions, reward estimates and the step counts against the float32
    reference run — bit for bit (float64 row TD, NumPy pairwise summation order)."""
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    D = golden('sr_traces')
    inst, f32, trials, steps, mask = [int(x) for x in D[name + '/cfg']]
    env = Gridworld(as_world(golden_worlds(str(D[name + '/world']))), n_envs=4, seed=SEED,
                    instance_base=0)
    agent = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    agent.track_instances = True
    if mask:
        agent.mask_actions = True
        agent.action_mask = D[name + '/action_mask']
    agent.train(env, trials, steps)
    assert np.array_equal(agent.monitors.lat_trace[inst].cpu().numpy(), D[name + '/steps'])
    assert np.array_equal(agent.T[inst].cpu().numpy(), D[name + '/T'])
    assert np.array_equal(agent.rewards[inst].cpu().numpy().astype(np.float64), D[name + '/rewards'])
    assert np.array_equal(agent.SR[inst].cpu().numpy().astype(np.float64), D[name + '/SR'])


@pytest.mark.parametrize('stream_rows', [False, True])
@pytest.mark.parametrize('name', ['open32_near_f32', 'open32_dense_f32', 'open32_r20_f32'])
def test_sr_golden_at_32x32(torch_cuda, golden, golden_worlds, name, stream_rows):
    """Config C4's own size against the REAL reference (tests/golden/gen_golden.py gen_sr32): one,
    twenty and 1 024 non-zero reward estimates — 1 024-term row sums in NumPy's pairwise order
    (agent/sr.py:288-308) — on the sparse-reward kernel where it serves (k_sr_wave: up to 32
    rewarded states) and on the row-streaming kernel (k_sr), bit for bit."""
    torch = torch_cuda
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    D = golden('sr32_traces')
    inst, f32, trials, steps, _ = [int(x) for x in D[name + '/cfg']]
    env = Gridworld(as_world(golden_worlds(str(D[name + '/world']))), n_envs=3, seed=SEED,
                    instance_base=0)
    agent = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    agent.track_instances = True
    agent.stream_rows = stream_rows
    if name + '/rewards0' in D.files:
        agent._bind(env)
        agent._rw.copy_(torch.as_tensor(D[name + '/rewards0'], device='cuda')[None].expand(3, -1))
    agent.train(env, trials, steps)
    sr = np.eye(1024, dtype=np.float32)
    sr[D[name + '/SR_rows'].astype(int)] = D[name + '/SR_values']
    assert np.array_equal(agent.monitors.lat_trace[inst].cpu().numpy(), D[name + '/steps'])
    assert np.array_equal(agent.T[inst].cpu().numpy(), D[name + '/T'])
    assert np.array_equal(agent.rewards[inst].cpu().numpy(), D[name + '/rewards'])
    assert np.array_equal(agent.SR[inst].cpu().numpy(), sr)
    # (only the sparse-reward kernel counts its traffic.  It serves this world — one rewarded state —
    #  whatever the estimates hold: pre-loaded non-zero entries elsewhere send the instance to its
    #  full sums from memory, csrc/sr_wave.hip `dense`)
    assert (int(agent.traffic[1].item()) > 0) == (not stream_rows)


# ---------------------------------------------------------------------------------------------
# Parity against the C oracle on the benchmark workloads themselves (same seeded inputs, sizes the
# oracle finishes in seconds), launched exactly like bench.py launches them.
def _bench_like(torch, cfg_name, n, launches, budget, base=0):
    import bench
    cfg = dict(bench.CONFIGS[cfg_name], instances=n, env_steps_per_launch=budget)
    env, agent = bench.build_agent(cfg_name, cfg, n, 0, torch.device('cuda', 0))
    env.instance_base = base
    env.env_ctr.zero_()
    env.reset()
    agent.track_occupancy = True
    agent.track_responses = True
    runner = bench.Runner(cfg, env, agent)
    for _ in range(launches):
        runner.launch()
    torch.cuda.synchronize()
    return cfg, env, agent


def _oracle_world(worlds):
    from oracle import c_oracle
    return c_oracle.OracleWorld([dict(next=w['next'], reward=w['rewards'],
                                      terminal=w['terminals'], starts=w['starting_states'])
                                 for w in worlds])


def _inst_fields(agent):
    a = agent.inst.cpu().numpy()
    return {'state': a[:, 0], 'step': a[:, 1], 'trial': a[:, 2], 'ctr_env': a[:, 3],
            'ctr_policy': a[:, 4], 'ctr_memory': a[:, 5], 'log_len': a[:, 6], 'flags': a[:, 7]}


def _cmp_inst(agent, oracle, keys):
    got = _inst_fields(agent)
    for k in keys:
        assert np.array_equal(got[k].astype(np.int64), oracle.inst[k].astype(np.int64)), k
    rew = agent.inst[:, 8:10].contiguous().view(torch_mod().float64).reshape(-1).cpu().numpy()
    assert np.array_equal(rew, oracle.inst['trial_reward'])


def torch_mod():
    import torch
    return torch


def test_c3_dynaq_mazes_vs_oracle(torch_cuda):
    """C3: 64 obstacle mazes, Dyna-Q with 50 planning updates per step, 3 launches of 40 steps."""
    from oracle import c_oracle
    n, launches, budget = 512, 3, 40
    cfg, env, agent = _bench_like(torch_cuda, 'C3', n, launches, budget, base=1000)
    o = c_oracle.TabOracle(_oracle_world(env.worlds), n, c_oracle.AG_DYNAQ, env.seed, True,
                           instance_base=1000, trial_cap=64, occupancy=True)
    for _ in range(launches):
        o.run(0x7fffffff, cfg['steps_per_trial'], cfg['batch'], step_budget=budget)
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(agent.M.rewards.astype(np.float64), o.MR)
    assert np.array_equal(agent.M.states, o.MS) and np.array_equal(agent.M.terminals, o.MT)
    _cmp_inst(agent, o, ['state', 'step', 'trial', 'ctr_env', 'ctr_policy', 'ctr_memory', 'flags'])
    assert np.array_equal(agent.monitors.lat_sum.cpu().numpy()[:64], o.lat_sum.astype(np.int64))
    assert np.array_equal(agent.monitors.lat_cnt.cpu().numpy()[:64], o.lat_cnt.astype(np.int64))
    assert np.array_equal(agent.monitors.resp_cnt.cpu().numpy()[:64], o.resp_cnt.astype(np.int64))
    assert np.array_equal(agent.monitors.occupancy.cpu().numpy(), o.occupancy.astype(np.int64))
    assert agent.env_steps() == n * launches * budget == int(o.inst['steps'].sum())


def test_c2_qlearning_vs_oracle(torch_cuda):
    """C2: 5x5 open field, online Q-learning only, many short trials with auto-reset."""
    from oracle import c_oracle
    n, launches, budget = 4096, 2, 300
    cfg, env, agent = _bench_like(torch_cuda, 'C2', n, launches, budget)
    o = c_oracle.TabOracle(_oracle_world(env.worlds), n, c_oracle.AG_Q, env.seed, True,
                           alpha=0.9, gamma=0.8, trial_cap=4096, occupancy=True)
    for _ in range(launches):
        o.run(0x7fffffff, cfg['steps_per_trial'], 0, step_budget=budget)
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)
    _cmp_inst(agent, o, ['state', 'step', 'trial', 'ctr_env', 'ctr_policy', 'flags'])
    assert np.array_equal(agent.monitors.lat_sum.cpu().numpy(), o.lat_sum.astype(np.int64))
    assert np.array_equal(agent.monitors.lat_cnt.cpu().numpy(), o.lat_cnt.astype(np.int64))
    assert np.array_equal(agent.monitors.resp_cnt.cpu().numpy(), o.resp_cnt.astype(np.int64))
    assert o.resp_cnt.sum() > 0 and (o.resp_cnt <= o.lat_cnt).all()
    from cobel_amd.monitor import ResponseMonitor
    rm = ResponseMonitor(4096)
    rm.update_from_device(agent.monitors, reduce=False)
    done = o.lat_cnt > 0
    assert np.array_equal(rm.responses[done], o.resp_cnt[done] / o.lat_cnt[done])
    assert np.isnan(rm.responses[~done]).all()
    assert np.allclose(agent.monitors.reward_sum.cpu().numpy(), o.reward_sum, rtol=0, atol=1e-9)
    assert np.array_equal(agent.monitors.occupancy.cpu().numpy(), o.occupancy.astype(np.int64))


def test_c4_sr_32x32_vs_oracle(torch_cuda):
    """C4: 32x32 open field, successor representation (4 MiB SR per instance)."""
    from oracle import c_oracle
    n, launches, budget = 24, 2, 150
    cfg, env, agent = _bench_like(torch_cuda, 'C4', n, launches, budget, base=5)
    o = c_oracle.SROracle(_oracle_world(env.worlds), n, env.seed, True, instance_base=5,
                          trial_cap=16, occupancy=True)
    for _ in range(launches):
        o.run(0x7fffffff, cfg['steps_per_trial'], step_budget=budget)
    assert np.array_equal(agent._T.cpu().numpy().astype(np.int64), o.T)
    assert np.array_equal(agent._rw.cpu().numpy().astype(np.float64), o.RW)
    assert np.array_equal(agent._sr.cpu().numpy().astype(np.float64), o.SR)
    _cmp_inst(agent, o, ['state', 'step', 'trial', 'ctr_env', 'ctr_policy', 'flags'])
    assert np.array_equal(agent.monitors.occupancy.cpu().numpy(), o.occupancy.astype(np.int64))
    assert np.array_equal(agent.monitors.resp_cnt.cpu().numpy()[:16], o.resp_cnt.astype(np.int64))
    assert np.array_equal(agent.monitors.lat_cnt.cpu().numpy()[:16], o.lat_cnt.astype(np.int64))


def test_sr_retrieve_q_matches_numpy_sum(torch_cuda):
    """retrieve_q / predict_on_batch: V = np.sum(SR * rewards, axis=1) bit for bit on a table with
    many non-zero reward estimates (the summation ORDER matters here), S = 100 (not a multiple of
    the 128-element leaf) and S = 1024."""
    torch = torch_cuda
    from cobel_amd import _lib
    for S in (100, 1024, 200):
        rng = np.random.default_rng(S)
        n = 3
        sr = (rng.standard_normal((n, S, S)) * rng.random((n, S, S)) ** 4).astype(np.float32)
        rw = rng.standard_normal((n, S)).astype(np.float32)
        T = rng.integers(0, S, (n, S, 4)).astype(np.int16)
        st = rng.integers(0, S, n).astype(np.int32)
        d = lambda a: torch.as_tensor(a, device='cuda').contiguous()  # noqa: E731
        dsr, drw, dT, dst = d(sr), d(rw), d(T), d(st)
        q = torch.empty((n, 4), dtype=torch.float32, device='cuda')
        _lib.check(_lib.lib().cobel_sr_retrieve_q(_lib.ptr(dsr), _lib.ptr(dT), _lib.ptr(drw),
                                                  _lib.ptr(dst), _lib.ptr(q), n, S, None))
        for i in range(n):
            v = np.sum(sr[i] * rw[i], axis=1)
            assert np.array_equal(q[i].cpu().numpy(), v[T[i, st[i]]])


def test_qagent_replay_vs_oracle(torch_cuda, golden_worlds):
    """QAgent with replay over the experience log (B = 24), 256 instances, two train() calls."""
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = as_world(golden_worlds('walls_8x8'))
    n = 256
    env = Gridworld(world, n_envs=n, seed=4242)
    agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.2))
    agent.track_instances = True
    agent.train(env, 6, 30, 24)
    agent.train(env, 4, 30, 24)
    o = c_oracle.TabOracle(_oracle_world([world]), n, c_oracle.AG_Q, 4242, True, alpha=0.9,
                           gamma=0.8, epsilon=0.2, trial_cap=10, log_cap=300)
    o.run(6, 30, 24)
    o.run(10, 30, 24)
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(agent.monitors.lat_trace.cpu().numpy(), o.lat_trace)
    assert np.array_equal(agent.inst[:, 6].cpu().numpy(), o.inst['log_len'].astype(np.int32))
    assert len(agent.M) == int(o.inst['log_len'][0])


MAZE_TEMPLATES = {
    't_maze': lambda gt: gt.make_t_maze(3, 2, 'right', 1.0),
    'double_t_maze': lambda gt: gt.make_double_t_maze(3, 2, 'left-right', 1.5),
    'two_sided_t_maze': lambda gt: gt.make_two_sided_t_maze(4, 3, 'left-left', 2.0),
    'two_choice_t_maze': lambda gt: gt.make_two_choice_t_maze(5, 7, 3, 'left', 'left'),
    '8_maze': lambda gt: gt.make_8_maze(4, 3, 'left', 0.5),
    'detour_maze': lambda gt: gt.make_detour_maze(2, 2, 4, 3, 1.0),
    'cross_maze': lambda gt: gt.make_cross_maze(2, 2, 'left'),
}


@pytest.mark.parametrize('template', sorted(MAZE_TEMPLATES))
def test_maze_templates_dynaq_and_sr_vs_oracle(torch_cuda, template):
    """Dyna-Q (20 planning updates per step) and SR on every maze template of gridworld_tools
    (the tables themselves are pinned to the reference in test_host_cpu), 192 instances each,
    against the C oracle: Q / SR tables, world models and per-trial latencies."""
    from cobel_amd.agent import SR, DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc import gridworld_tools as gt
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = MAZE_TEMPLATES[template](gt)
    n = 192
    env = Gridworld(world, n_envs=n, seed=2024)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.15))
    agent.track_instances = True
    agent.train(env, 12, 60, 20)
    o = c_oracle.TabOracle(_oracle_world([world]), n, c_oracle.AG_DYNAQ, 2024, True,
                           epsilon=0.15, trial_cap=12)
    o.run(12, 60, 20)
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(agent.M.states, o.MS) and np.array_equal(agent.M.terminals, o.MT)
    assert np.array_equal(agent.monitors.lat_trace.cpu().numpy(), o.lat_trace)
    assert (o.lat_trace[:, -1] < 59).any()          # some instances do reach the goal

    env = Gridworld(world, n_envs=n, seed=2025)
    sr = SR(env.observation_space, env.action_space, EpsilonGreedy(0.15))
    sr.track_instances = True
    sr.train(env, 8, 60)
    so = c_oracle.SROracle(_oracle_world([world]), n, 2025, True, epsilon=0.15, trial_cap=8)
    so.run(8, 60)
    assert np.array_equal(sr._sr.cpu().numpy().astype(np.float64), so.SR)
    assert np.array_equal(sr._T.cpu().numpy().astype(np.int64), so.T)
    assert np.array_equal(sr.monitors.lat_trace.cpu().numpy(), so.lat_trace)


# ---------------------------------------------------------------------------------------------
# Per-instance hyper-parameters (parameter sets): the reference runs one simulation per
# combination (optimizer/grid_search.py:173-262); here combinations ride on the instance axis.
# Streams are keyed by the global instance id, so instance i of a mixed launch must equal
# instance i of a launch in which EVERY instance uses i's combination — and those uniform
# launches are the ones pinned to the reference / the C oracle above.
PSET_COMBOS = [(0.99, 0.99, 0.1, 0.9), (0.5, 0.9, 0.3, 0.9), (0.99, 0.8, 0.0, 0.5),
               (0.1, 0.99, 1.0, 0.9), (0.5, 0.9, 0.3, 0.25)]


def _pset_arrays(n):
    which = (np.arange(n) * 7 + 3) % len(PSET_COMBOS)
    cols = np.array(PSET_COMBOS)[which]
    return which, cols[:, 0].copy(), cols[:, 1].copy(), cols[:, 2].copy(), cols[:, 3].copy()


@pytest.mark.parametrize('variant', ['dynaq_fast', 'dynaq_masked', 'dynaq_lds_model', 'qagent_replay',
                                     'qagent_online'])
def test_param_sets_tabular_equal_uniform_runs(torch_cuda, golden_worlds, variant):
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.memory import DynaQMemory
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('walls_8x8'))
    n = 160
    which, alpha, gamma, eps, mlr = _pset_arrays(n)

    def run(a, g, e, m):
        env = Gridworld(world, n_envs=n, seed=777, instance_base=40)
        if variant.startswith('dynaq'):
            ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(e), learning_rate=a,
                       gamma=g, memory=DynaQMemory(64, 4, m))
            if variant == 'dynaq_masked':
                ag.mask_actions = True
                ag.action_mask[:, 2] = False
                ag.action_mask[5] = True
        else:
            ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(e), learning_rate=a,
                        gamma=g)
        ag.track_instances = True
        if variant == 'dynaq_lds_model':
            ag._bind(env)
            ag._env_in(env)
            flags = _lib.F_LEARN | _lib.F_FORCE_LDS_MODEL | ag._policy_in(ag.policy, env, False)
            ag.monitors.reserve(8, n, True)
            ag._launch(env, ag.policy, flags, 8, 30, 0, 24)
        elif variant == 'qagent_online':
            ag.train(env, 8, 30, 0)
        else:
            ag.train(env, 8, 30, 24)
        torch.cuda.synchronize()
        return ag

    mixed = run(alpha, gamma, eps, mlr)
    for k, (a, g, e, m) in enumerate(PSET_COMBOS):
        uni = run(a, g, e, m)
        sel = torch.as_tensor(np.flatnonzero(which == k), device='cuda')
        assert len(sel) > 0
        assert torch.equal(mixed._q[sel], uni._q[sel]), (variant, k)
        assert torch.equal(mixed.inst[sel], uni.inst[sel]), (variant, k)
        assert torch.equal(mixed.monitors.lat_trace[sel], uni.monitors.lat_trace[sel])
        if variant.startswith('dynaq'):
            assert torch.equal(mixed.M.table[sel], uni.M.table[sel]), (variant, k)
            assert torch.equal(mixed.M.index[sel], uni.M.index[sel]), (variant, k)
    # the combinations do differ from one another (the test is not vacuous)
    a0 = torch.as_tensor(np.flatnonzero(which == 0)[:1], device='cuda')
    a1 = torch.as_tensor(np.flatnonzero(which == 1)[:1], device='cuda')
    assert not torch.equal(mixed._q[a0], mixed._q[a1])


def test_param_sets_sr_equal_uniform_runs(torch_cuda, golden_worlds):
    torch = torch_cuda
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('walls_8x8'))
    n = 96
    which, alpha, gamma, eps, _ = _pset_arrays(n)

    def run(a, g, e):
        env = Gridworld(world, n_envs=n, seed=4711, instance_base=3)
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(e), learning_rate=a, gamma=g)
        ag.track_instances = True
        ag.train(env, 6, 30)
        torch.cuda.synchronize()
        return ag

    mixed = run(alpha, gamma, eps)
    for k, (a, g, e, _) in enumerate(PSET_COMBOS):
        uni = run(a, g, e)
        sel = torch.as_tensor(np.flatnonzero(which == k), device='cuda')
        assert torch.equal(mixed._sr[sel], uni._sr[sel]), k
        assert torch.equal(mixed._T[sel], uni._T[sel]) and torch.equal(mixed._rw[sel], uni._rw[sel])
        assert torch.equal(mixed.inst[sel], uni.inst[sel]), k


@pytest.mark.parametrize('agent_name', ['dynaq', 'q'])
def test_param_sets_tabular_against_the_oracle_instance_by_instance(torch_cuda, golden_worlds, agent_name):
    """A mixed launch of Dyna-Q / QAgent, instance by instance against the C restatement
    (agent/dyna_q.py:275-330, agent/q.py:305-354) run with THAT instance's parameters."""
    torch = torch_cuda
    from oracle import c_oracle
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.memory import DynaQMemory
    from cobel_amd.policy import EpsilonGreedy
    tab = golden_worlds('walls_8x8')
    n, trials, steps, B = 24, 6, 30, 24
    _, alpha, gamma, eps, mlr = _pset_arrays(n)
    env = Gridworld(as_world(tab), n_envs=n, seed=777, instance_base=40)
    if agent_name == 'dynaq':
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha,
                   gamma=gamma, memory=DynaQMemory(64, 4, mlr))
    else:
        ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha,
                    gamma=gamma)
    ag.track_instances = True
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    w = c_oracle.OracleWorld([tab])
    for i in range(n):
        o = c_oracle.TabOracle(w, 1, c_oracle.AG_DYNAQ if agent_name == 'dynaq' else c_oracle.AG_Q, 777,
                               True, instance_base=40 + i, alpha=float(alpha[i]), gamma=float(gamma[i]),
                               epsilon=float(eps[i]), model_lr=float(mlr[i]), trial_cap=trials,
                               log_cap=0 if agent_name == 'dynaq' else trials * steps)
        o.run(trials, steps, B)
        assert np.array_equal(ag._q[i].cpu().numpy().astype(np.float64), o.Q[0]), i
        assert np.array_equal(ag.monitors.lat_trace[i, :trials].cpu().numpy(), o.lat_trace[0]), i
        if agent_name == 'dynaq':
            assert np.array_equal(ag.M.rewards[i].astype(np.float64), o.MR[0]), i
            assert np.array_equal(ag.M.states[i], o.MS[0]) and np.array_equal(ag.M.terminals[i], o.MT[0])


def test_param_sets_sr_against_the_oracle_instance_by_instance(torch_cuda, golden_worlds):
    """The same mixed launch, instance by instance against the C restatement of agent/sr.py:255-308
    run with THAT instance's learning rate, discount and epsilon (not launch against launch)."""
    torch = torch_cuda
    from oracle import c_oracle
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    tab = golden_worlds('walls_8x8')
    n, trials, steps = 24, 5, 30
    _, alpha, gamma, eps, _ = _pset_arrays(n)
    for stream_rows in (False, True):
        env = Gridworld(as_world(tab), n_envs=n, seed=4711, instance_base=3)
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha,
                gamma=gamma)
        ag.track_instances = True
        ag.stream_rows = stream_rows
        ag.train(env, trials, steps)
        torch.cuda.synchronize()
        w = c_oracle.OracleWorld([tab])
        for i in range(n):
            o = c_oracle.SROracle(w, 1, 4711, True, instance_base=3 + i, alpha=float(alpha[i]),
                                  gamma=float(gamma[i]), epsilon=float(eps[i]), trial_cap=trials)
            o.run(trials, steps)
            assert np.array_equal(ag._sr[i].cpu().numpy().astype(np.float64), o.SR[0]), (stream_rows, i)
            assert np.array_equal(ag._T[i].cpu().numpy(), o.T[0]), (stream_rows, i)
            assert np.array_equal(ag._rw[i].cpu().numpy().astype(np.float64), o.RW[0]), (stream_rows, i)
            assert np.array_equal(ag.monitors.lat_trace[i, :trials].cpu().numpy(), o.lat_trace[0])
    assert len({(float(a), float(g), float(e)) for a, g, e in zip(alpha, gamma, eps)}) > 2


def test_grid_search_vectorised_equals_sequential(torch_cuda, golden_worlds, tmp_path):
    """GridSearchOptimizer.fit_vectorised: every (combination, run) is one instance of a single
    Dyna-Q launch with per-instance learning rate / epsilon; the fit equals the reference-style
    sequential fit() in which each combination is simulated on its own (same instance ids)."""
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.optimizer import GridSearchOptimizer, spread_over_instances
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('open_5x5'))
    grid = {'learning_rate': [0.2, 0.6, 0.99], 'epsilon': np.array([0.3, 0.05])}
    tasks = {'short': {'trials': 12, 'steps': 20}, 'long': {'trials': 6, 'steps': 40}}
    data = {'short': 6.0, 'long': 8.0}
    runs, launches = 4, []

    def simulate(task, lr, eps, n, base):
        env = Gridworld(world, n_envs=n, seed=99, instance_base=base)
        agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(eps),
                      learning_rate=lr)
        agent.track_instances = True
        agent.train(env, task['trials'], task['steps'], 16)
        launches.append(n)
        return agent.monitors.lat_trace[:, -3:].double().mean(dim=1).cpu().numpy()

    def batched(task, combos, nb_runs):
        arrays, which = spread_over_instances(combos, nb_runs)
        lat = simulate(task, arrays['learning_rate'], arrays['epsilon'], len(which), 0)
        return [list(lat[which == c]) for c in range(len(combos))]

    def loss(sim, exp):
        return float(np.mean([(np.mean(sim[t]) - exp[t]) ** 2 for t in sim]))

    d1, d2 = tmp_path / 'vec', tmp_path / 'seq'
    d1.mkdir()
    d2.mkdir()
    vec = GridSearchOptimizer(str(d1) + '/', grid, nb_runs=runs)
    fit_v = vec.fit_vectorised(batched, tasks, data, loss)
    assert launches == [24, 24] and len(fit_v) == 6

    seq = GridSearchOptimizer(str(d2) + '/', grid, nb_runs=1)
    order = {key: c for c, key in enumerate(seq.parameter_combinations)}

    def one(task, params):   # all runs of one combination, at the instance ids they had above
        c = order[(params['learning_rate'], params['epsilon'])]
        return simulate(task, params['learning_rate'], params['epsilon'], runs, c * runs)

    fit_s = seq.fit(one, tasks, data, lambda sim, exp: loss({t: sim[t][0] for t in sim}, exp))
    assert list(fit_s) == list(fit_v)
    assert [fit_s[k] for k in fit_s] == [fit_v[k] for k in fit_v]
    assert len(set(fit_v.values())) > 1


@pytest.mark.parametrize('n,worlds,side', [(4100, 1, 5), (200, 3, 5), (300, 2, 14), (130, 1, 20)])
def test_lane_per_instance_kernel_vs_oracle(torch_cuda, golden_worlds, n, worlds, side):
    """Runs without planning take the lane-per-instance kernel (64 instances per wave; 32 at
    14x14 and 16 at 20x20, where the Q columns of a full wave no longer fit in LDS): Q-learning
    with batch 0 over several budgeted launches (partial last wave, 1 to 3 worlds), then a
    greedy test() phase with an action mask — all against the C oracle, and against the
    wave-per-instance kernel forced on the same inputs."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    S = side * side
    ws = [make_open_field(side, side, g, 1.0 + g) for g in (0, S // 2, S - 1)][:worlds]

    def run(force_wave):
        env = Gridworld(ws, n_envs=n, seed=77, instance_base=9)
        ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1), EpsilonGreedy(0.0))
        ag.track_instances = True
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False) | \
            (_lib.F_FORCE_WAVE if force_wave else 0)
        ag.monitors.reserve(64, n, True)
        for _ in range(3):
            ag._launch(env, ag.policy, flags, 64, 30, 37, 0)
        ag._policy_out(ag.policy)
        ag._env_out(env)
        ag.current_trial = 64
        ag.inst[:, _lib.I_TRIAL] = 64          # start the test phase at a common trial index
        ag.mask_actions = True
        ag.action_mask = np.ones((S, 4), dtype=bool)
        ag.action_mask[:, 0] = False           # never move left
        ag.action_mask[0] = True
        ag.test(env, 3, 20)
        torch.cuda.synchronize()
        return ag

    lpi, wpi = run(False), run(True)
    assert torch.equal(lpi._q, wpi._q) and torch.equal(lpi.inst, wpi.inst)
    assert torch.equal(lpi.monitors.lat_trace, wpi.monitors.lat_trace)
    assert torch.equal(lpi.monitors.lat_sum, wpi.monitors.lat_sum)

    tabs = [dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'],
                 starts=w['starting_states']) for w in ws]
    mask = np.ones((S, 4), dtype=bool)
    mask[:, 0] = False
    mask[0] = True
    o = c_oracle.TabOracle(c_oracle.OracleWorld(tabs), n, c_oracle.AG_Q, 77, True, instance_base=9,
                           alpha=0.9, gamma=0.8, trial_cap=67)
    for _ in range(3):
        o.run(64, 30, 0, step_budget=37)
    assert np.array_equal(lpi._q.cpu().numpy().astype(np.float64), o.Q)
    o.inst['trial'] = 64
    o.inst['flags'] = 0
    o.inst['step'] = 0
    # separate test policy -> its own stream and counter
    saved = o.inst['ctr_policy'].copy()
    o.inst['ctr_policy'] = 0
    o.mask = (mask * np.array([1, 2, 4, 8])).sum(axis=1).astype(np.uint8)
    o.run(67, 20, 0, flags=c_oracle.F_TEST_STREAM, epsilon=0.0)
    o.inst['ctr_policy'] = saved
    assert np.array_equal(lpi.monitors.lat_trace.cpu().numpy()[:, 64:67], o.lat_trace[:, 64:67])
    got = lpi.inst.cpu().numpy()
    assert np.array_equal(got[:, 0], o.inst['state']) and np.array_equal(got[:, 3], o.inst['ctr_env'].astype(np.int32))


def test_lane_per_instance_monitor_window_overflow(torch_cuda):
    """One-step trials: 900 trials per instance in one launch run past the 512-trial window of
    per-wave monitor accumulators (the rest goes to the global arrays directly); striped monitor
    copies (n >= 4096) sum to the same totals as the wave-per-instance kernel's."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    n, trials = 4200, 900

    def run(force_wave):
        env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=5)
        ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        ag.track_responses = True
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False) | \
            (_lib.F_FORCE_WAVE if force_wave else 0)
        ag.monitors.reserve(trials, n, False)
        ag._launch(env, ag.policy, flags, trials, 1, 0, 0)
        torch.cuda.synchronize()
        return ag

    lpi, wpi = run(False), run(True)
    assert lpi.monitors.stripes == 16 and lpi.monitors.raw('lat_cnt').shape == (16, trials)
    assert torch.equal(lpi._q, wpi._q) and torch.equal(lpi.inst, wpi.inst)
    for name in ('lat_sum', 'lat_cnt', 'resp_cnt'):
        assert torch.equal(getattr(lpi.monitors, name), getattr(wpi.monitors, name)), name
    assert torch.allclose(lpi.monitors.reward_sum, wpi.monitors.reward_sum, rtol=1e-12, atol=0)
    assert int(lpi.monitors.lat_cnt.min()) == n and int(lpi.monitors.lat_sum.sum()) == 0


def test_topology_kat_and_walk(torch_cuda, golden):
    """unit_tests/test_topology.py:64-109 (pose observations) and a draw-injected random walk on
    linear_track(10, 2) recorded from the reference: nodes, pose observations, rewards, terminals,
    truncated == end_trial, resets."""
    torch = torch_cuda
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import linear_track, t_maze
    from cobel_amd.spaces import Box, Discrete
    k = golden('topology_kat')
    nodes, starts = t_maze(4, 3, 1)
    env = Topology(nodes, starts)
    assert isinstance(env.observation_space, Box) and isinstance(env.action_space, Discrete)
    assert env.observation_space.shape == (6,) and env.action_space.n == 4
    env.reset()
    assert env.current_node == '10'
    got = []
    for a in k['kat_actions']:
        obs, r, t, trunc, info = env.step(int(a))
        assert trunc == t and info == {} and obs.shape == (6,)
        got.append((int(env.current_node), r, t))
    assert [g[0] for g in got] == list(k['kat_states']) == [9, 8, 7, 3, 3, 4, 5, 6]
    assert [g[1] for g in got] == list(k['kat_rewards']) and [g[2] for g in got] == list(k['kat_terminals'])

    nodes, starts = linear_track(10, 2, 1., 20., 'right')
    env = Topology(nodes, starts, seed=SEED, instance_base=3)
    obs, _ = env.reset()
    assert int(env.current_node) == int(k['walk_first'])
    for i, a in enumerate(k['walk_actions']):
        obs, r, t, trunc, _ = env.step(int(a))
        assert np.array_equal(obs, k['walk_obs'][i]) and r == k['walk_rewards'][i]
        assert t == bool(k['walk_terminals'][i]) and trunc == t
        node = int(env.current_node)
        if t:
            env.reset()
            node = int(env.current_node) + 1000
        assert node == int(k['walk_states'][i])
    assert np.array_equal(env.get_position(), np.array(nodes[env.current_node]['pose']))

    # vectorised: every instance follows its neighbour table, observations are pose rows
    n = 1000
    venv = Topology(nodes, starts, n_envs=n, seed=5)
    nbr, pose = k['linear_10x2/nbr'], k['linear_10x2/pose']
    rng = np.random.default_rng(1)
    for _ in range(10):
        s = venv.state.cpu().numpy()
        a = rng.integers(0, 4, n)
        obs, r, d, tr, _ = venv.step(torch.as_tensor(a))
        ns = nbr[s, a]
        assert np.array_equal(venv.state.cpu().numpy(), ns)
        assert np.array_equal(obs.cpu().numpy(), pose[ns])
        assert np.array_equal(r.cpu().numpy(), k['linear_10x2/reward'][ns].astype(np.float32))
        assert np.array_equal(d.cpu().numpy(), k['linear_10x2/terminal'][ns])
        venv.reset(d)


def _mlp(torch, init):
    from collections import OrderedDict
    net = torch.nn.Sequential(OrderedDict([
        ('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
        ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
        ('output', torch.nn.Linear(64, 4))])).double()
    state = net.state_dict()
    for key, w in zip(state, init):
        state[key] = torch.as_tensor(w)
    net.load_state_dict(state)
    return net


def _module_mlp(torch, init):
    """The same network written the way the reference's demos write it
    (demo/topology/demo_dqn.py:36-60): a custom Module calling functional relu in forward."""
    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layer_dense_1 = torch.nn.Linear(6, 64)
            self.layer_dense_2 = torch.nn.Linear(64, 64)
            self.layer_output = torch.nn.Linear(64, 4)
            self.double()

        def forward(self, layer_input):
            x = torch.reshape(layer_input, (len(layer_input), -1))
            x = torch.nn.functional.relu(self.layer_dense_1(x))
            x = torch.nn.functional.relu(self.layer_dense_2(x))
            return self.layer_output(x)

    net = Model()
    state = net.state_dict()
    for key, w in zip(state, init):
        state[key] = torch.as_tensor(w)
    net.load_state_dict(state)
    return net


@pytest.mark.parametrize('name', ['dqn_i0', 'dqn_i2', 'dqn_two_sessions', 'dqn_two_sessions_cap40'])
def test_dqn_matches_reference_float64(torch_cuda, golden, name):
    """DQN on linear_track(10, 2), float64 6-64-64-4 MLP on PyTorch-ROCm, against the reference run
    on the CPU with the same initial weights and the same injected draws: identical node /
    action / reward sequence; online and target weights and Q(all poses) within 1e-9 relative
    (different GEMM summation order on the GPU, amplified through ~60 Adam steps)."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    inst, trials, steps, batch, ddqn = [int(x) for x in D[name + '/cfg']]
    init = [D['%s/init_%d' % (name, i)] for i in range(6)]
    nodes, starts = linear_track(10, 2, 1., 20., 'right')
    env = Topology(nodes, starts, seed=SEED, instance_base=inst)
    from cobel_amd.memory import DQNMemory
    user_model = TorchNetwork(_mlp(torch, init))
    agent = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3), user_model, gamma=0.8,
                memory=DQNMemory(int(D[name + '/capacity'])))
    # (two train() calls on one agent: the ring has to keep the first session's experiences up to
    #  the memory's capacity, as the reference's growing arrays do)
    for part in D[name + '/sessions']:
        agent.train(env, int(part), steps, batch)
    assert agent.fused_steps > 0, 'the golden must exercise the two-kernel DQN step'
    assert agent.fused_graph_steps > 0, '... replayed from the HIP graph of 16 steps'
    m = agent.M
    size = int(m.size[0].item())
    assert size == len(D[name + '/actions'])
    order = ((int(m.head[0].item()) + np.arange(size)) % m.slots)      # oldest first
    assert np.array_equal(m.actions[0].cpu().numpy()[order], D[name + '/actions'])
    assert np.array_equal(m.rewards[0].cpu().numpy()[order], D[name + '/rewards'])
    pose = np.array([nodes[k]['pose'] for k in nodes])
    seen = m.next_states[0].cpu().numpy()[order]
    assert np.array_equal(seen, pose[D[name + '/nodes'][-size:]])
    lat = agent.monitors.lat_sum.cpu().numpy()
    assert np.array_equal(lat[:trials], D[name + '/steps'])
    for i, w in enumerate(agent._online.get_weights(0)):
        assert np.allclose(w, D['%s/online_%d' % (name, i)], rtol=1e-9, atol=1e-12), i
    for i, w in enumerate(agent._target.get_weights(0)):
        assert np.allclose(w, D['%s/target_%d' % (name, i)], rtol=1e-9, atol=1e-12), i
    assert np.allclose(agent.predict_on_batch(pose), D[name + '/q_all'], rtol=1e-9, atol=1e-12)
    # the attributes the reference's users read hold the trained networks (agent/dqn.py:108-110):
    # model_online / model_target, and the module the user passed in (the target network)
    for i, w in enumerate(agent.model_online.get_weights()):
        assert np.array_equal(w, agent._online.get_weights(0)[i]), i
    for i, w in enumerate(user_model.get_weights()):
        assert np.array_equal(w, agent._target.get_weights(0)[i]), i
    assert np.allclose(agent.model_online.predict_on_batch(pose), D[name + '/q_all'], rtol=1e-9,
                       atol=1e-12)
    # weights assigned to the user-facing networks between runs are adopted by every instance
    agent.model_online.set_weights(init)
    assert not agent._online.matches(agent.model_online)
    agent._adopt_user_weights()
    for i, w in enumerate(agent._online.get_weights(0)):
        assert np.array_equal(w, init[i]), i
    assert agent._target.matches(agent.model_target)


def test_dqn_graph_replay_equals_eager_steps(torch_cuda, golden):
    """Fixed-budget DQN runs replay one captured step from a HIP graph: same transitions, same
    monitors, weights equal to float64 round-off (torch's capturable Adam forms the bias
    corrections on the device) as the eager loop over the same number of steps."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.memory import DQNMemory
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(10, 2, 1., 20., 'right')

    def run(graph):
        env = Topology(nodes, starts, n_envs=64, seed=SEED)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(_mlp(torch, init)), gamma=0.8, memory=DQNMemory(capacity=64))
        ag.use_graph = graph
        ag._run(env, 10 ** 6, 12, 32, True, budget=40)
        torch.cuda.synchronize()
        return ag, env

    (eager, e1), (graph, e2) = run(False), run(True)
    assert eager.graph_replays == 0 and graph.graph_replays == 36
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.env_ctr, e2.env_ctr)
    assert torch.equal(eager.M.actions, graph.M.actions)
    assert torch.equal(eager.M.next_states, graph.M.next_states)
    assert torch.equal(eager.trial, graph.trial)
    assert torch.equal(eager.monitors.lat_sum, graph.monitors.lat_sum)
    assert torch.equal(eager.monitors.lat_cnt, graph.monitors.lat_cnt)
    assert int(eager.monitors.lat_cnt.sum()) > 64            # trials did end and restart
    for i in (0, 31, 63):
        for a, b in zip(eager._online.get_weights(i), graph._online.get_weights(i)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
        for a, b in zip(eager._target.get_weights(i), graph._target.get_weights(i)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_dqn_train_graph_chunks_equal_eager(torch_cuda, golden):
    """train() with many instances replays graph chunks while no instance can finish and steps
    eagerly (with exact freezing) at the tail: same result as the all-eager run."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(10, 2, 1., 20., 'right')

    def run(graph):
        env = Topology(nodes, starts, n_envs=256, seed=SEED)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(_mlp(torch, init)), gamma=0.8)
        ag.use_graph = graph
        ag.train(env, 14, 6, 16)
        torch.cuda.synchronize()
        return ag, env

    (eager, e1), (auto, e2) = run(False), run(None)
    assert eager.graph_replays == 0 and auto.graph_replays >= 8
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.env_ctr, e2.env_ctr)
    assert torch.equal(eager.M.size, auto.M.size) and torch.equal(eager.M.actions, auto.M.actions)
    assert torch.equal(eager.trial, auto.trial) and int(auto.trial.min()) == 14
    assert torch.equal(eager.monitors.lat_sum, auto.monitors.lat_sum)
    assert torch.equal(eager.monitors.lat_cnt, auto.monitors.lat_cnt)
    for i in (0, 100, 255):
        for a, b in zip(eager._online.get_weights(i), auto._online.get_weights(i)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('kind', ['dyna_dqn', 'dyna_dsr'])
def test_dyna_network_agents_graph_replay_equals_eager(torch_cuda, kind):
    """Fixed-budget Dyna-DQN / Dyna-DSR runs from a HIP graph (Dyna-DSR: masked per-network Adam
    and a second optimizer inside the captured step) against the eager loop."""
    torch = torch_cuda
    import bench
    from cobel_amd.agent import DynaDQN, DynaDSR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy

    def run(graph):
        torch.manual_seed(3)
        env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=48, seed=SEED)
        if kind == 'dyna_dqn':
            ag = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                         TorchNetwork(bench._mlp(25, 4)), gamma=0.8)
        else:
            ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                         TorchNetwork(bench._mlp(25, 25)), TorchNetwork(bench._mlp(25, 1)),
                         gamma=0.8)
        ag.use_graph = graph
        ag._run(env, 10 ** 6, 15, 16, True, budget=30)
        torch.cuda.synchronize()
        return ag, env

    (eager, e1), (graph, e2) = run(False), run(True)
    assert eager.graph_replays == 0 and graph.graph_replays == 26
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.env_ctr, e2.env_ctr)
    assert torch.equal(eager.trial, graph.trial)
    assert torch.equal(eager.monitors.lat_sum, graph.monitors.lat_sum)
    assert torch.equal(eager.M.states, graph.M.states)
    nets = [('_online', 0), ('_online', 47)] if kind == 'dyna_dqn' else \
        [('_online', 0), ('_online', 4 * 47 + 3), ('_reward_net', 20)]
    for attr, i in nets:
        for a, b in zip(getattr(eager, attr).get_weights(i), getattr(graph, attr).get_weights(i)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('dtype_name', ['f64', 'f32'])
def test_fused_adam_equals_torch_adam(torch_cuda, dtype_name):
    """cobel_adam_step against torch.optim.Adam on stacked networks: all instances stepping, and
    with an activity mask that changes from step to step (per-instance step counts, frozen
    optimizer state) — parameters and both moment estimates."""
    torch = torch_cuda
    import bench
    from cobel_amd.network import TorchNetwork
    dt = torch.float64 if dtype_name == 'f64' else torch.float32
    tol = dict(rtol=1e-12, atol=1e-14) if dtype_name == 'f64' else dict(rtol=2e-5, atol=1e-7)
    n, B = 37, 16
    gen = torch.Generator(device='cuda').manual_seed(5)

    def build(fused):
        torch.manual_seed(11)
        proto = TorchNetwork(bench._mlp(6, 4, dtype_name), optimizer_params={'lr': 3e-3,
                                                                             'weight_decay': 1e-3})
        proto.set_device(torch.device('cuda', 0))
        net = proto.replicate(n)
        net.fused_adam = fused
        return net

    fused, plain = build(True), build(False)
    fused_target, plain_target = fused.clone(), plain.clone()
    for step in range(12):
        x = torch.rand((n, B, 6), generator=gen, device='cuda', dtype=dt)
        y = torch.rand((n, B, 4), generator=gen, device='cuda', dtype=dt)
        active = None if step < 4 else (torch.rand(n, generator=gen, device='cuda') < 0.6)
        if active is not None and not bool(active.any()):
            active[0] = True
        for net, tgt in ((fused, fused_target), (plain, plain_target)):
            net.train_on_device(x, y, active, blend_into=tgt, tau=0.05)
    assert fused._diverged and hasattr(fused, '_steps')
    for a, b in zip(fused_target.params.values(), plain_target.params.values()):
        assert torch.allclose(a, b, **tol)        # the target blend rides on the same kernel
    for (k, a), b in zip(fused.params.items(), plain.params.values()):
        assert torch.allclose(a, b, **tol), k
        sa, sb = fused.optimizer.state[a], plain.optimizer.state[b]
        assert torch.allclose(sa['exp_avg'], sb['exp_avg'], **tol), k
        assert torch.allclose(sa['exp_avg_sq'], sb['exp_avg_sq'], **tol), k
        assert torch.equal(sa['steps'], sb['steps'])


def test_dqn_vectorised_equals_single_instances(torch_cuda, golden):
    """8 instances in lockstep (stacked networks, per-instance rings and streams) give exactly
    the trajectories of the single-instance runs; weights agree to float64 round-off."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(10, 2, 1., 20., 'right')

    def run(n, base):
        env = Topology(nodes, starts, n_envs=n, seed=SEED, instance_base=base)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(_mlp(torch, init)), gamma=0.8)
        ag.train(env, 3, 25, 32)
        return ag

    vec = run(8, 0)
    assert np.array_equal(vec.M.actions[0, :int(vec.M.size[0])].cpu().numpy(), D['dqn_i0/actions'])
    for i in (2, 5):
        one = run(1, i)
        k = int(one.M.size[0])
        assert int(vec.M.size[i]) == k
        assert torch.equal(vec.M.actions[i, :k], one.M.actions[0, :k])
        assert torch.equal(vec.M.next_states[i, :k], one.M.next_states[0, :k])
        for a, b in zip(vec._online.get_weights(i), one._online.get_weights(0)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    lat = vec.monitors.mean_latency()
    assert not np.isnan(lat[:3]).any()


@pytest.mark.parametrize('offset', [0, 1, 2, 3])
def test_model_index_variants_agree(torch_cuda, golden_worlds, offset):
    """The planning kernel with the model digest in HBM (gathered one step ahead) and with the
    digest in LDS produce identical tables, counters and digests — on a maze with several reward
    sites (flagged reward estimates) and bumping moves (ns == s patches).  `offset` steps without
    planning come first, so that the policy and memory stream counters take every relative
    phase (the cached-draw refresh schedule of the digest-in-HBM kernel depends on it)."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('walls_8x8'))

    def run(extra):
        env = Gridworld(world, n_envs=96, seed=31337)
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.2))
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | extra | ag._policy_in(ag.policy, env, False)
        ag.monitors.reserve(16, 96, True)
        if offset:
            ag._launch(env, ag.policy, flags | _lib.F_NO_REPLAY, 16, 25, offset, 40)
        for _ in range(5):
            ag._launch(env, ag.policy, flags, 16, 25, 33, 40)
        return ag

    a, b = run(0), run(_lib.F_FORCE_LDS_MODEL)
    assert torch.equal(a._q, b._q) and torch.equal(a.M.table, b.M.table)
    assert torch.equal(a.M.index, b.M.index) and torch.equal(a.inst, b.inst)
    fresh = torch.empty_like(a.M.index)
    _lib.check(_lib.lib().cobel_model_index_build(_lib.ptr(a.M.table), _lib.ptr(fresh), 96, 64, None))
    assert torch.equal(fresh, a.M.index)


def test_sr_prefetch_variants_agree(torch_cuda):
    """k_sr with the next step's value rows prefetched into registers (default) and with rows
    loaded at the top of every step give identical SR matrices, transition tables and counters
    (32x32 open field, long enough for revisits, bumps into walls and trial ends)."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    world = make_open_field(32, 32, 0, 1)

    def run(extra):
        env = Gridworld(world, n_envs=12, seed=2718)
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.3))
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | extra | ag._policy_in(ag.policy, env, False)
        ag.monitors.reserve(64, 12, True)
        for _ in range(3):
            ag._launch(env, ag.policy, flags, 0x7fffffff, 60, 170, 0)
        return ag

    a, b = run(0), run(_lib.F_NO_PREFETCH)
    assert torch.equal(a._sr, b._sr) and torch.equal(a._T, b._T) and torch.equal(a._rw, b._rw)
    assert torch.equal(a.inst, b.inst)
    assert torch.equal(a.monitors.lat_trace, b.monitors.lat_trace)


# ---------------------------------------------------------------------------------------------
# Edge cases: empty batches of instances, extreme batch sizes, one-step trials, zero trials,
# large state spaces (LDS near its limit), replay-log overflow.
def test_edge_empty_and_zero_work(torch_cuda):
    torch = torch_cuda
    import ctypes as C
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    lib = _lib.lib()
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=3, seed=1)
    # n = 0 is a no-op for every entry point
    z32 = torch.zeros(1, dtype=torch.int32, device='cuda')
    _lib.check(lib.cobel_env_step(env.handle.ptr, _lib.ptr(z32), _lib.ptr(z32), None, None, 0, 0, None))
    _lib.check(lib.cobel_env_reset(env.handle.ptr, _lib.ptr(z32), None, _lib.ptr(z32), 1, 0, 0, None))
    run = _lib.TabRun()
    run.q, run.inst, run.model = _lib.ptr(z32), _lib.ptr(z32), _lib.ptr(z32)
    run.n, run.agent, run.steps_per_trial, run.epsilon = 0, 1, 5, 0.1
    _lib.check(lib.cobel_tab_run(env.handle.ptr, C.byref(run), None))
    what = (C.c_int32 * 4)(9, 9, 9, 9)
    _lib.check(lib.cobel_tab_describe(env.handle.ptr, C.byref(run), what))
    assert list(what) == [0, 0, 0, 0]
    # ... also for the two kernels of the DQN step
    z64 = torch.zeros(8, dtype=torch.float64, device='cuda')
    rep = _lib.DQNReplay()
    for k in range(3):
        for arr in (rep.w, rep.b, rep.w_target, rep.b_target, rep.m_w, rep.m_b, rep.v_w, rep.v_b):
            arr[k] = _lib.ptr(z64)
    rep.steps = rep.states = rep.next_states = rep.rewards = rep.nonterminal = _lib.ptr(z64)
    rep.actions = _lib.ptr(z64)
    rep.n, rep.n_inputs, rep.n_hidden1, rep.n_hidden2, rep.n_actions, rep.batch = 0, 6, 64, 64, 4, 32
    rep.is_float64 = 1
    _lib.check(lib.cobel_dqn_replay(C.byref(rep), None))
    act = _lib.DQNAct()
    for name in ('state', 'env_ctr', 'obs_table', 'q', 'policy_ctr', 'ring_states',
                 'ring_next_states', 'ring_actions', 'ring_rewards', 'ring_nonterminal',
                 'ring_size', 'ring_head', 'trial', 'step', 'trial_reward', 'active', 'stepped'):
        setattr(act, name, _lib.ptr(z64))
    act.n, act.n_obs, act.slots, act.batch, act.steps_per_trial = 0, 6, 4, 32, 10
    act.epsilon, act.is_float64 = 0.1, 1
    _lib.check(lib.cobel_dqn_act(env.handle.ptr, C.byref(act), None))
    act.n, act.epsilon = 1, 1.5
    with pytest.raises(AssertionError):     # refused before any launch
        _lib.check(lib.cobel_dqn_act(env.handle.ptr, C.byref(act), None))
    assert float(z64.abs().sum()) == 0.0
    # zero trials: nothing moves, nothing is drawn
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    before = env.state.clone()
    agent.train(env, 0, 10, 8)
    assert torch.equal(env.state, before) and agent.env_steps() == 0 and agent.current_trial == 0
    assert float(agent.Q.abs().sum()) == 0.0
    # bad arguments are refused before any launch
    run.n, run.steps_per_trial = 3, 0
    with pytest.raises(IndexError):
        _lib.check(lib.cobel_tab_run(env.handle.ptr, C.byref(run), None))
    run.steps_per_trial, run.batch = 5, -1
    with pytest.raises(IndexError):
        _lib.check(lib.cobel_tab_run(env.handle.ptr, C.byref(run), None))
    # (a batch above COBEL_MAX_BATCH is no longer refused: it takes the general kernel,
    #  tests/test_gpu_general.py; cobel_tab_query still answers for the wavefront kernels)
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_tab_query(25, 1, 63, None, None))
    with pytest.raises(IndexError):   # next state out of range
        bad = make_open_field(3, 3, 0, 1)
        bad['next'] = bad['next'].copy()
        bad['next'][4, 2] = 9
        Gridworld(bad)


@pytest.mark.parametrize('batch,steps', [(1, 1), (62, 7), (33, 200)])
def test_edge_batch_sizes_and_one_step_trials(torch_cuda, golden_worlds, batch, steps):
    """Planning batches of 1 and of the maximum 62, trials of a single step (every step ends a
    trial), against the C oracle."""
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = as_world(golden_worlds('walls_8x8'))
    n, trials = 70, 9
    env = Gridworld(world, n_envs=n, seed=555)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.25))
    agent.track_instances = True
    agent.train(env, trials, steps, batch)
    o = c_oracle.TabOracle(_oracle_world([world]), n, c_oracle.AG_DYNAQ, 555, True, epsilon=0.25,
                           trial_cap=trials)
    o.run(trials, steps, batch)
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(agent.monitors.lat_trace.cpu().numpy(), o.lat_trace)
    assert np.array_equal(agent.M.states, o.MS)


def test_edge_large_state_space(torch_cuda):
    """64x64 = 4096 states: Q alone is 64 KiB of LDS per instance (dynamic LDS above the 64 KiB
    default), model digest in HBM or in LDS (96 KiB)."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = make_obstacle_maze(64, 64, 7, density=0.15)
    n = 20

    def run(extra):
        env = Gridworld(world, n_envs=n, seed=4096)
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | extra | ag._policy_in(ag.policy, env, False)
        ag.monitors.reserve(8, n, True)
        ag._launch(env, ag.policy, flags, 0x7fffffff, 300, 400, 50)
        torch.cuda.synchronize()
        return ag

    a, b = run(0), run(_lib.F_FORCE_LDS_MODEL)
    assert torch.equal(a._q, b._q) and torch.equal(a.M.table, b.M.table)
    o = c_oracle.TabOracle(_oracle_world([world]), n, c_oracle.AG_DYNAQ, 4096, True, trial_cap=8)
    o.run(0x7fffffff, 300, 50, step_budget=400)
    assert np.array_equal(a._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(a.M.states, o.MS)


def test_edge_replay_log_overflow(torch_cuda, golden_worlds):
    """QAgent whose experience log is smaller than the run: logging stops at capacity, replay keeps
    sampling the truncated log (oracle has the same rule)."""
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = as_world(golden_worlds('walls_8x8'))
    n = 40
    env = Gridworld(world, n_envs=n, seed=808)
    agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.3))
    agent._bind(env)
    agent.reserve_replay(37)
    agent._session(env, 5, 20, 10, True)
    o = c_oracle.TabOracle(_oracle_world([world]), n, c_oracle.AG_Q, 808, True, alpha=0.9, gamma=0.8,
                           epsilon=0.3, trial_cap=5, log_cap=37)
    o.run(5, 20, 10)
    assert int(agent.inst[:, 6].max().item()) == 37
    assert np.array_equal(agent.inst[:, 6].cpu().numpy(), o.inst['log_len'].astype(np.int32))
    assert np.array_equal(agent._q.cpu().numpy().astype(np.float64), o.Q)


def test_dqn_idle_instances_are_frozen(torch_cuda, golden):
    """Instances finish their trials at different times (short track, trials end early); an
    instance that is done must stay exactly as its own single-instance run left it — weights,
    optimizer history, replay ring, stream counters — while the others keep training."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(4, 1, 1., 1., 'right')

    def run(n, base):
        env = Topology(nodes, starts, n_envs=n, seed=777, instance_base=base)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.5),
                 TorchNetwork(_mlp(torch, init)), gamma=0.8)
        ag.train(env, 4, 30, 16)
        ag.train(env, 2, 30, 16)        # a second call continues every instance's own history
        return ag, env

    vec, venv = run(6, 0)
    sizes = vec.M.size.cpu().numpy()
    assert len(set(sizes.tolist())) > 1, 'the scenario needs instances of different length'
    for i in (0, 3, 5):
        one, oenv = run(1, i)
        k = int(one.M.size[0])
        assert int(sizes[i]) == k
        assert torch.equal(vec.M.actions[i, :k], one.M.actions[0, :k])
        assert torch.equal(vec.M.next_states[i, :k], one.M.next_states[0, :k])
        assert int(vec.policy.counter[i]) == int(one.policy.counter[0])
        assert int(vec.M.counter[i]) == int(one.M.counter[0])
        assert int(venv.env_ctr[i]) == int(oenv.env_ctr[0])
        for a, b in zip(vec._online.get_weights(i), one._online.get_weights(0)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), np.abs(a - b).max()
        for a, b in zip(vec._target.get_weights(i), one._target.get_weights(0)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), np.abs(a - b).max()


@pytest.mark.parametrize('name', ['ddqn_i0', 'ddqn_i1', 'ddqn_mlp64'])
def test_dyna_dqn_matches_reference(torch_cuda, golden, name):
    """DynaDQN (DQN fed by the tabular Dyna-Q model) on a 4x4 open field, float64, against the
    reference with the same initial weights and injected draws: identical state / action
    sequence and model tables; networks within 1e-9."""
    torch = torch_cuda
    from collections import OrderedDict
    from cobel_amd.agent import DynaDQN
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dyna_dqn_trace')
    inst, trials, steps, B = [int(x) for x in D[name + '/cfg']]
    if name == 'ddqn_mlp64':       # 64-64 ReLU, batch 32: the two-kernel step in world-model mode
        net = torch.nn.Sequential(OrderedDict([
            ('dense_1', torch.nn.Linear(16, 64)), ('relu_1', torch.nn.ReLU()),
            ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
            ('output', torch.nn.Linear(64, 4))])).double()
    else:
        net = torch.nn.Sequential(OrderedDict([
            ('dense_1', torch.nn.Linear(16, 32)), ('relu_1', torch.nn.ReLU()),
            ('output', torch.nn.Linear(32, 4))])).double()
    state = net.state_dict()
    for i, key in enumerate(state):
        state[key] = torch.as_tensor(D['%s/init_%d' % (name, i)])
    net.load_state_dict(state)
    env = Gridworld(make_open_field(4, 4, 0, 1), n_envs=3, seed=SEED, instance_base=inst)
    agent = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.2), TorchNetwork(net),
                    gamma=0.9)
    agent.train(env, trials, steps, B)
    assert (agent.fused_steps > 0) == (name == 'ddqn_mlp64'), \
        'the 64-64 golden has to run on the fused kernels (and only that one can)'
    m = agent.M
    assert np.array_equal(m.states[0].cpu().numpy().reshape(16, 4), D[name + '/M_states'])
    assert np.array_equal(m.terminals[0].cpu().numpy().reshape(16, 4), D[name + '/M_terminals'])
    assert np.array_equal(m.rewards[0].cpu().numpy().reshape(16, 4), D[name + '/M_rewards'])
    for i, w in enumerate(agent._online.get_weights(0)):
        assert np.allclose(w, D['%s/online_%d' % (name, i)], rtol=1e-9, atol=1e-12), i
    for i, w in enumerate(agent._target.get_weights(0)):
        assert np.allclose(w, D['%s/target_%d' % (name, i)], rtol=1e-9, atol=1e-12), i
    q = agent._online.predict_on_device(
        torch.eye(16, dtype=torch.float64, device='cuda')[None].expand(3, 16, 16).contiguous())
    assert np.allclose(q[0].cpu().numpy(), D[name + '/q_all'], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', ['ddsr_default', 'ddsr_switches', 'ddsr_mlp64',
                                  'ddsr_mlp64_switches'])
def test_dyna_dsr_matches_reference(torch_cuda, golden, name):
    """DynaDSR (deep successor representation fed by the tabular Dyna-Q model) on a 4x4 open
    field, float64, against the reference with the same initial weights and injected draws:
    identical state / action sequence; the four online and four target successor networks, the
    reward network and the resulting Q-values within 1e-9.  Instance 0 of three (the others run
    different streams), so the per-network masked training and step counts are exercised."""
    torch = torch_cuda
    from collections import OrderedDict
    from cobel_amd.agent import DynaDSR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dyna_dsr_trace')
    inst, trials, steps, B = [int(x) for x in D[name + '/cfg']]

    mlp64 = 'mlp64' in name     # 64-64 ReLU networks, batch 32: the fused MLP kernels

    def net(sizes, tag):
        if mlp64:
            m = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(sizes[0], 64)), ('relu_1', torch.nn.ReLU()),
                ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
                ('output', torch.nn.Linear(64, sizes[2]))])).double()
        else:
            m = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(sizes[0], sizes[1])), ('relu_1', torch.nn.ReLU()),
                ('output', torch.nn.Linear(sizes[1], sizes[2]))])).double()
        state = m.state_dict()
        for i, key in enumerate(state):
            state[key] = torch.as_tensor(D['%s/init_%s_%d' % (name, tag, i)])
        m.load_state_dict(state)
        return TorchNetwork(m)

    env = Gridworld(make_open_field(4, 4, 0, 1), n_envs=3, seed=SEED, instance_base=inst)
    agent = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.25),
                    net((16, 24, 16), 'sr'), net((16, 12, 1), 'rw'), gamma=0.9)
    if name == 'ddsr_switches':
        agent.use_DR, agent.use_follow_up_state = True, True
        agent.ignore_terminality, agent.target_update = False, 3
    if name == 'ddsr_mlp64_switches':
        agent.use_DR, agent.use_follow_up_state = True, True
        agent.ignore_terminality, agent.target_update = False, 0.05
    seen = []
    agent.callbacks.custom_callbacks = {'on_trial_end': [lambda logs: seen.append(logs['steps'])]}
    agent.track_instances = True
    agent.train(env, trials, steps, B)
    assert (agent.fused_steps > 0) == mlp64, 'the 64-64 goldens run on cobel_mlp_fit / _forward'
    for a in range(4):
        for i, w in enumerate(agent.get_weights(a)):
            assert np.allclose(w, D['%s/online_%d_%d' % (name, a, i)], rtol=1e-9, atol=1e-12), (a, i)
        for i, w in enumerate(agent.get_weights(a, target=True)):
            assert np.allclose(w, D['%s/target_%d_%d' % (name, a, i)], rtol=1e-9, atol=1e-12), (a, i)
    for i, w in enumerate(agent.get_reward_weights()):
        assert np.allclose(w, D['%s/reward_%d' % (name, i)], rtol=1e-9, atol=1e-12), i
    q = agent.predict_on_batch(np.arange(16))
    assert np.allclose(q[0].cpu().numpy(), D[name + '/q_all'], rtol=1e-9, atol=1e-12)
    assert np.allclose(agent.retrieve_q(5)[0].cpu().numpy(), D[name + '/q_all'][5], rtol=1e-9,
                       atol=1e-12)
    # the other two instances ran their own streams and ended up elsewhere
    assert not np.allclose(q[1].cpu().numpy(), q[0].cpu().numpy())


# ---------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: a scattered sample of instances against the oracle plus
# size-independent properties (conservation of steps / trials / visits) over all instances.
def _sample_vs_oracle(agent, env, cfg, agent_kind, ids, launches, budget, **okw):
    from oracle import c_oracle
    w = _oracle_world(env.worlds)
    q = agent._q[ids].cpu().numpy().astype(np.float64)
    inst = agent.inst[ids].cpu().numpy()
    for k, g in enumerate(ids):
        o = c_oracle.TabOracle(w, 1, agent_kind, env.seed, True, instance_base=int(g), **okw)
        for _ in range(launches):
            o.run(0x7fffffff, cfg['steps_per_trial'], cfg['batch'], step_budget=budget)
        assert np.array_equal(q[k], o.Q[0]), g
        assert inst[k, 0] == o.inst['state'][0] and inst[k, 2] == o.inst['trial'][0], g
        assert inst[k, 3] == int(o.inst['ctr_env'][0]) and inst[k, 4] == int(o.inst['ctr_policy'][0])


def test_full_size_c3_sample_and_conservation(torch_cuda):
    """C3 at 65 536 instances x 64 mazes, Dyna-Q with 50 planning updates per step: 192 instances
    spread over the whole range bit-exact against the oracle; every instance executed exactly
    the budgeted steps; trial / visit totals are conserved."""
    from oracle import c_oracle
    n, launches, budget = 65536, 2, 48
    cfg, env, agent = _bench_like(torch_cuda, 'C3', n, launches, budget)
    ids = np.unique(np.concatenate([np.arange(0, n, 911), [1, 63, 64, n - 1]]))[:192]
    _sample_vs_oracle(agent, env, cfg, c_oracle.AG_DYNAQ, ids, launches, budget)
    inst = agent.inst.cpu().numpy()
    steps = inst[:, 10].astype(np.int64)
    assert (steps == launches * budget).all() and agent.env_steps() == n * launches * budget
    mon = agent.monitors
    assert int(mon.lat_cnt.sum().item()) == int(inst[:, 2].astype(np.int64).sum())
    assert int(mon.occupancy.sum().item()) == n * launches * budget
    # steps of finished trials + steps of the running trials == all steps
    finished = int(mon.lat_sum.sum().item()) + int(mon.lat_cnt.sum().item())
    running = int((inst[:, 1].astype(np.int64) * (inst[:, 7] & 1)).sum())
    assert finished + running == n * launches * budget


def test_full_size_c3_trained_agents_vs_oracle(torch_cuda):
    """C3 in the state the headline is TIMED in: bench.py's own launches (65 536 instances, 512 env
    steps each, 50 planning updates per step) until the kernel reports that >= 95 % of the planning
    batches it draws are evaluated (~29 000 steps per instance; young agents skip most batches and
    that is all the two-launch test above sees), then two more launches — 32 instances spread over
    the range against the C oracle run for the same steps: Q, the model's three tables, the
    digest, state / step / trial and the three stream counters, bit for bit
    (agent/dyna_q.py:140-215, :275-330, memory/dyna_q.py:122-157)."""
    torch = torch_cuda
    import bench
    from oracle import c_oracle
    n = 65536
    cfg = dict(bench.CONFIGS['C3'])
    env, agent = bench.build_agent('C3', cfg, n, 0, torch.device('cuda', 0))
    runner = bench.Runner(cfg, env, agent)
    assert runner.describe()['kernel'] == runner._lib.TAB_KERNEL_PWG
    per_launch = n * cfg['env_steps_per_launch']
    launches, frac = 0, 0.0
    while frac < cfg['train_until'] and launches < 96:
        for _ in range(7):
            runner.launch()
        b0 = int(agent.batches_done.item())
        runner.launch()               # (the fraction of ONE launch, as bench.py measures it)
        launches += 8
        frac = (int(agent.batches_done.item()) - b0) / per_launch
    assert frac >= cfg['train_until'], 'pre-training did not reach the full-work state: %.3f' % frac
    for _ in range(2):
        runner.launch()
    launches += 2
    steps = launches * cfg['env_steps_per_launch']
    assert agent.env_steps() == n * steps
    ids = np.unique(np.concatenate([np.arange(0, n, 2179), [1, 63, 64, n - 1]]))[:32]
    assert len(ids) == 32
    w = _oracle_world(env.worlds)
    sel = torch.as_tensor(ids, device='cuda')
    q = agent._q[sel].cpu().numpy().astype(np.float64)
    inst = agent.inst[sel].cpu().numpy()
    raw = agent.M.table[sel].cpu().numpy()
    m_r = (raw & 0xFFFFFFFF).astype(np.uint32).view(np.float32).astype(np.float64)
    m_s = ((raw >> 32) & 0xFFFF).astype(np.int64)
    m_t = ((raw >> 48) & 1).astype(np.int64)
    trained = 0
    for k, g in enumerate(ids):
        o = c_oracle.TabOracle(w, 1, c_oracle.AG_DYNAQ, env.seed, True, instance_base=int(g))
        for _ in range(launches):
            o.run(0x7fffffff, cfg['steps_per_trial'], cfg['batch'],
                  step_budget=cfg['env_steps_per_launch'])
        assert np.array_equal(q[k], o.Q[0]), g
        assert np.array_equal(m_r[k], o.MR[0]) and np.array_equal(m_s[k], o.MS[0]), g
        assert np.array_equal(m_t[k], o.MT[0]), g
        for col, key in ((0, 'state'), (1, 'step'), (2, 'trial'), (3, 'ctr_env'), (4, 'ctr_policy'),
                         (5, 'ctr_memory')):
            assert int(inst[k, col]) == int(o.inst[key][0]), (g, key)
        trained += int(np.count_nonzero(o.Q[0]) > 512)
    assert trained >= 24, 'the sampled instances are not trained agents'
    # the digest the planning lanes read agrees with the table after 30 000 steps of updates
    fresh = torch.empty_like(agent.M.index[sel])
    from cobel_amd import _lib
    _lib.check(_lib.lib().cobel_model_index_build(_lib.ptr(agent.M.table[sel].contiguous()),
                                                  _lib.ptr(fresh), len(ids), 1024, None))
    torch.cuda.synchronize()
    assert torch.equal(fresh, agent.M.index[sel])


def test_full_size_c2_sample_and_conservation(torch_cuda):
    """C2 at 65 536 x 5x5 (lane-per-instance kernel): 1 024 instances bit-exact against the oracle,
    conservation of steps and trials over all of them."""
    torch = torch_cuda
    import bench
    from oracle import c_oracle
    n, launches, budget = 65536, 2, 200
    cfg = dict(bench.CONFIGS['C2'], instances=n, env_steps_per_launch=budget)
    env, agent = bench.build_agent('C2', cfg, n, 0, torch.device('cuda', 0))
    runner = bench.Runner(cfg, env, agent)
    for _ in range(launches):
        runner.launch()
    torch.cuda.synchronize()
    ids = np.arange(0, n, 64)[:1024] + (np.arange(1024) % 64)
    o = c_oracle.TabOracle(_oracle_world(env.worlds), n, c_oracle.AG_Q, env.seed, True, alpha=0.9,
                           gamma=0.8, trial_cap=4096)
    for _ in range(launches):
        o.run(0x7fffffff, cfg['steps_per_trial'], 0, step_budget=budget)
    assert np.array_equal(agent._q[ids].cpu().numpy().astype(np.float64), o.Q[ids])
    inst = agent.inst.cpu().numpy()
    assert np.array_equal(inst[:, 0], o.inst['state']) and np.array_equal(inst[:, 2], o.inst['trial'])
    mon = agent.monitors
    assert np.array_equal(mon.lat_sum.cpu().numpy(), o.lat_sum.astype(np.int64))
    assert np.array_equal(mon.lat_cnt.cpu().numpy(), o.lat_cnt.astype(np.int64))
    assert agent.env_steps() == n * launches * budget == int(inst[:, 10].astype(np.int64).sum())


def test_full_size_c4_sample_and_conservation(torch_cuda):
    """C4 at 16 384 instances x 32x32 (64 GiB of successor matrices): 12 instances spread over
    the range bit-exact against the oracle (SR rows, transition table, reward estimate); every
    instance executed the budgeted steps; no instance has more changed SR rows than steps."""
    torch = torch_cuda
    from oracle import c_oracle
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip('needs 64 GiB of HBM for the successor matrices')
    n, launches, budget = 16384, 2, 40
    cfg, env, agent = _bench_like(torch, 'C4', n, launches, budget)
    ids = [0, 1, 63, 64, 4097, 8191, 8192, 12345, 16000, 16382, 16383, 777]
    w = _oracle_world(env.worlds)
    for g in ids:
        o = c_oracle.SROracle(w, 1, env.seed, True, instance_base=g)
        for _ in range(launches):
            o.run(0x7fffffff, cfg['steps_per_trial'], step_budget=budget)
        assert np.array_equal(agent._sr[g].cpu().numpy().astype(np.float64), o.SR[0]), g
        assert np.array_equal(agent._T[g].cpu().numpy().astype(np.int64), o.T[0]), g
        assert np.array_equal(agent._rw[g].cpu().numpy().astype(np.float64), o.RW[0]), g
    inst = agent.inst.cpu().numpy()
    assert (inst[:, 10].astype(np.int64) == launches * budget).all()
    assert agent.env_steps() == n * launches * budget
    assert int(agent.monitors.occupancy.sum().item()) == n * launches * budget
    # a walk of 80 steps changes at most 80 rows: all others are still rows of the identity
    sr = agent._sr[:128]
    changed = (sr != torch.eye(1024, device='cuda')[None]).any(dim=2).sum(dim=1)
    assert int(changed.max().item()) <= launches * budget and int(changed.min().item()) > 0


def test_describe_reports_the_kernel_a_run_takes(torch_cuda):
    """cobel_tab_describe: the benchmark workloads take the kernels DESIGN.md says they take, with
    the LDS footprint and the workgroups per CU (1 280-byte LDS blocks, 128 per CU) it states;
    describing a run launches nothing."""
    torch = torch_cuda
    import bench
    from cobel_amd import _lib
    dev = torch.device('cuda', 0)
    # (C3: one persistent workgroup per CU — ten wavefronts, each with its instance's Q table in LDS,
    #  16 384 B, which is all of it; F_NO_PWG: one workgroup per instance, nine per CU — LDS comes in blocks of 1 280 B)
    for name, n, kernel, lds, per_cu, per_wg, extra in [
            ('C3', 256, _lib.TAB_KERNEL_PWG, None, 1, None, 0),
            ('C3', 256, _lib.TAB_KERNEL_WPI_INDEX, 1024 * 16, 9, 1, _lib.F_NO_PWG),
            ('C2', 256, _lib.TAB_KERNEL_LPI, None, None, 64, 0)]:
        cfg = dict(bench.CONFIGS[name], instances=n)
        env, ag = bench.build_agent(name, cfg, n, 0, dev)
        ag.extra_flags = extra
        r = bench.Runner(cfg, env, ag)
        got = ag.describe_launch(env, ag.policy, r.flags, 0x7fffffff, cfg['steps_per_trial'],
                                 cfg['env_steps_per_launch'], cfg['batch'])
        torch.cuda.synchronize()
        assert got['kernel'] == kernel, (name, got)
        if per_wg is not None:
            assert got['instances_per_workgroup'] == per_wg, (name, got)
        else:
            waves = got['instances_per_workgroup']
            assert waves == 10 and got['workgroups_per_cu'] == 1
            assert got['lds_bytes'] == 10 * 1024 * 16 == 160 * 1024
        if lds is not None:
            assert (got['lds_bytes'], got['workgroups_per_cu']) == (lds, per_cu), (name, got)
        assert ag.env_steps() == 0 and int(ag.inst[:, _lib.I_STEPS_LO].sum().item()) == 0
    # the generic wave-per-instance kernel (masked actions) keeps the model digest in LDS
    cfg = dict(bench.CONFIGS['C3'], instances=64)
    env, ag = bench.build_agent('C3', cfg, 64, 0, dev)
    ag.mask_actions = True
    r = bench.Runner(cfg, env, ag)
    got = ag.describe_launch(env, ag.policy, r.flags | _lib.F_MASK_ACTIONS, 0x7fffffff, 200, 16, 50)
    assert got['kernel'] == _lib.TAB_KERNEL_WPI and got['lds_bytes'] == 1024 * 24


# ---------------------------------------------------------------------------------------------
# Fused DQN replay step (cobel_dqn_replay) against the PyTorch path it replaces
@pytest.mark.parametrize('dtype_name,n_in,ddqn,kernel', [
    ('f64', 6, False, None), ('f64', 25, True, None), ('f32', 6, False, None), ('f64', 1, False, None),
    ('f64', 6, True, 'stream'), ('f32', 25, False, 'stream'), ('f64', 25, True, 'lds'),
    ('f64', 32, False, 'stream')])
def test_fused_dqn_replay_equals_torch_path(torch_cuda, dtype_name, n_in, ddqn, kernel,
                                            f32_atol=1e-6):
    """targets -> MSE backward -> Adam -> target blend in one kernel == the same step through
    vmap'ed forward passes, autograd and the optimizer kernel: online and target parameters and
    both Adam moments after every one of 10 steps fed with the same batches (duplicated samples,
    terminal transitions, an activity mask that changes from step to step, weight decay).
    kernel: the form of the step cobel_dqn_replay picks by itself (None), or one of the two pinned
    (COBEL_DEBUG_DQN_KERNEL: parameters staged in LDS / weight operands streamed from memory)."""
    if kernel:
        os.environ['COBEL_DEBUG'], os.environ['COBEL_DEBUG_DQN_KERNEL'] = '1', kernel
        try:
            return test_fused_dqn_replay_equals_torch_path(torch_cuda, dtype_name, n_in, ddqn, None,
                                                           f32_atol)
        finally:
            del os.environ['COBEL_DEBUG_DQN_KERNEL'], os.environ['COBEL_DEBUG']
    torch = torch_cuda
    import bench
    from cobel_amd.network import TorchNetwork
    dt = torch.float64 if dtype_name == 'f64' else torch.float32
    # (float32: Adam divides by |g| + 1e-8, so the rounding of gradient entries near 1e-8 shows
    #  up at ~1e-3 of a step — scripts/experiments/exp_f32_adam.py; the shape sweep passes a wider f32_atol)
    tol = dict(rtol=1e-10, atol=1e-13) if dtype_name == 'f64' else dict(rtol=2e-4, atol=f32_atol)
    n, B, gamma, tau = 23, 32, 0.8, 0.01
    gen = torch.Generator(device='cuda').manual_seed(7)

    def build(fused):
        torch.manual_seed(3)
        proto = TorchNetwork(bench._mlp(n_in, 4, dtype_name), optimizer_params={
            'lr': 2e-3, 'weight_decay': 1e-3 if n_in == 25 else 0.0})
        proto.set_device(torch.device('cuda', 0))
        net = proto.replicate(n)
        with torch.no_grad():   # one network per instance, all different
            for p in net.params.values():
                p.add_(0.05 * torch.randn(p.shape, generator=gen, device='cuda', dtype=dt))
        net.fused_mlp = fused
        return net

    gen.manual_seed(7)
    fused = build(True)
    gen.manual_seed(7)
    plain = build(False)
    fused_t, plain_t = fused.clone(), plain.clone()
    with torch.no_grad():
        for a, b in zip(fused_t.params.values(), plain_t.params.values()):
            a.mul_(0.9)
            b.mul_(0.9)
    for step in range(10):
        s = torch.rand((n, B, n_in), generator=gen, device='cuda', dtype=dt) * 2 - 0.5
        ns = torch.rand((n, B, n_in), generator=gen, device='cuda', dtype=dt) * 2 - 0.5
        s[:, 5] = s[:, 4]          # sampling is with replacement: repeated experiences
        a = torch.randint(0, 4, (n, B), generator=gen, device='cuda')
        a[:, 5] = a[:, 4]
        r = torch.rand((n, B), generator=gen, device='cuda', dtype=dt)
        nt = (torch.rand((n, B), generator=gen, device='cuda') < 0.8).to(dt)
        active = None if step < 3 else (torch.rand(n, generator=gen, device='cuda') < 0.7)
        if active is not None and not bool(active.any()):
            active[0] = True
        assert fused.dqn_replay_fused(fused_t, s, a, r, ns, nt, gamma, ddqn, tau, active)
        assert not plain.dqn_replay_fused(plain_t, s, a, r, ns, nt, gamma, ddqn, tau, active)
        with torch.no_grad():      # DQN.replay's PyTorch path (agent/dqn.py of this package)
            targets = plain.forward(s).clone()
            boot = plain_t.forward(ns)
            pick = (plain.forward(ns) if ddqn else boot).argmax(dim=2)
            boot = torch.gather(boot, 2, pick[..., None])[..., 0]
            targets.scatter_(2, a[..., None], (r + boot * nt * gamma)[..., None])
        plain.train_on_device(s, targets, active, blend_into=plain_t, tau=tau)
        for (k, x), y in zip(fused.params.items(), plain.params.values()):
            assert torch.allclose(x, y, **tol), (step, k, float((x - y).abs().max()))
            sx, sy = fused.optimizer.state[x], plain.optimizer.state[y]
            assert torch.allclose(sx['exp_avg'], sy['exp_avg'], **tol), (step, k)
            assert torch.allclose(sx['exp_avg_sq'], sy['exp_avg_sq'], **tol), (step, k)
            assert torch.equal(sx['steps'], sy['steps'])
        for (k, x), y in zip(fused_t.params.items(), plain_t.params.values()):
            assert torch.allclose(x, y, **tol), (step, k)
    # networks of another shape are left to the PyTorch path
    other = TorchNetwork(torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(),
                                             torch.nn.Linear(32, 4)).to(dt))
    other.set_device(torch.device('cuda', 0))
    o = other.replicate(3)
    assert not o.dqn_replay_fused(o.clone(), s[:3], a[:3], r[:3], ns[:3], nt[:3], gamma, False,
                                  tau, None)
    if n_in == 6:   # right shapes, wrong activation: recognised by behaviour, so refused as well
        tanh = TorchNetwork(torch.nn.Sequential(
            torch.nn.Linear(6, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
            torch.nn.Linear(64, 4)).to(dt))
        tanh.set_device(torch.device('cuda', 0))
        o = tanh.replicate(3)
        assert o._mlp3_names() is None
        assert not o.dqn_replay_fused(o.clone(), s[:3], a[:3], r[:3], ns[:3], nt[:3], gamma, False,
                                      tau, None)


@pytest.mark.parametrize('dtype_name,kernel', [('f64', None), ('f32', None), ('f64', 'stream'),
                                               ('f32', 'stream')])
def test_dqn_two_kernel_loop_equals_torch_loop(torch_cuda, golden, monkeypatch, dtype_name, kernel):
    """DQN.train through cobel_dqn_act + cobel_dqn_replay (two launches per lockstep step) against
    the PyTorch loop it replaces: identical transitions in the replay rings, stream counters,
    trial counts and monitors (instances finish at different times; a small ring wraps around;
    a second train() call continues); weights to round-off in float64.  kernel: the form of the
    replay step the library picks for six inputs (parameters staged in LDS) or the streaming form
    pinned — the batch is read through the ring slots in both."""
    torch = torch_cuda
    if kernel:
        monkeypatch.setenv('COBEL_DEBUG', '1')
        monkeypatch.setenv('COBEL_DEBUG_DQN_KERNEL', kernel)
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.memory import DQNMemory
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(6, 2, 1., 5., 'right')

    def run(fused):
        env = Topology(nodes, starts, n_envs=48, seed=4242, instance_base=9)
        # float64: the reference's way of writing the model (a custom Module); float32: Sequential
        net = _module_mlp(torch, init) if dtype_name == 'f64' else _mlp(torch, init).float()
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.4), TorchNetwork(net),
                 gamma=0.8, memory=DQNMemory(capacity=40))
        ag.fused_loop = None if fused else False
        ag.use_graph = False if not fused else None
        ag.train(env, 5, 14, 32)
        ag.train(env, 3, 14, 32)
        torch.cuda.synchronize()
        return ag, env

    (a, ea), (b, eb) = run(True), run(False)
    assert a.fused_steps > 0 and b.fused_steps == 0
    assert torch.equal(a.trial, b.trial) and int(a.trial.min()) == 8
    if dtype_name == 'f32':
        # float32 Q-values of the two paths differ in the last bits (MFMA vs GEMM summation
        # order), and a near-tie may then pick another action: most instances, not necessarily
        # all, walk the same path.  Those that do must agree everywhere else.
        same = (a.M.actions == b.M.actions).all(dim=1) & (a.M.size == b.M.size)
        assert float(same.float().mean()) >= 0.75, float(same.float().mean())
        keep = same.nonzero().flatten()
    else:
        keep = torch.arange(48, device='cuda')
        for k in ('lat_sum', 'lat_cnt', 'reward_sum'):
            assert torch.equal(getattr(a.monitors, k), getattr(b.monitors, k)), k
        assert torch.equal(ea.env_ctr, eb.env_ctr)
    for x, y in ((a.M.size, b.M.size), (a.M.head, b.M.head), (a.M.actions, b.M.actions),
                 (a.M.rewards, b.M.rewards), (a.M.states, b.M.states),
                 (a.M.next_states, b.M.next_states), (a.M.terminals, b.M.terminals),
                 (a.M.counter, b.M.counter), (a.policy.counter, b.policy.counter)):
        assert torch.equal(x[keep], y[keep])
    assert len(set(a.M.size.cpu().numpy().tolist())) > 1 or int(a.M.size.max()) == 40
    tol = dict(rtol=1e-9, atol=1e-12) if dtype_name == 'f64' else dict(rtol=5e-3, atol=1e-4)
    for i in keep.cpu().numpy()[[0, len(keep) // 2, -1]]:
        for x, y in zip(a._online.get_weights(int(i)), b._online.get_weights(int(i))):
            assert np.allclose(x, y, **tol), float(np.abs(x - y).max())
        for x, y in zip(a._target.get_weights(int(i)), b._target.get_weights(int(i))):
            assert np.allclose(x, y, **tol)
    assert a.current_trial == b.current_trial == 8


@pytest.mark.parametrize('dtype_name,ddqn', [('f64', False), ('f64', True), ('f32', False)])
def test_dqn_two_kernel_loop_on_a_hexagonal_topology(torch_cuda, dtype_name, ddqn):
    """demo/topology/demo_dqn.py --env hexagonal: six actions per node.  The two-kernel loop (the
    streaming form of the replay step, selection over six Q-values, the world as neighbour tables)
    against the PyTorch loop: transitions, rings, counters and monitors identical in float64,
    weights to round-off."""
    torch = torch_cuda
    from collections import OrderedDict
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.memory import DQNMemory
    from cobel_amd.misc.topology_tools import hexagonal
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    nodes, starts = hexagonal(5, (0.0, 1.0))

    def run(fused):
        torch.manual_seed(11)
        env = Topology(nodes, starts, n_envs=40, seed=777, instance_base=3)
        assert int(env.action_space.n) == 6
        net = torch.nn.Sequential(OrderedDict([
            ('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
            ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
            ('output', torch.nn.Linear(64, 6))]))
        net = net.double() if dtype_name == 'f64' else net.float()
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.4), TorchNetwork(net),
                 gamma=0.8, memory=DQNMemory(capacity=48))
        ag.DDQN = ddqn
        ag.fused_loop = None if fused else False
        ag.use_graph = False if not fused else None
        ag.train(env, 4, 12, 32)
        ag.train(env, 2, 12, 32)
        torch.cuda.synchronize()
        return ag, env

    (a, ea), (b, eb) = run(True), run(False)
    assert a.fused_steps > 0 and b.fused_steps == 0
    assert torch.equal(a.trial, b.trial) and int(a.trial.min()) == 6
    if dtype_name == 'f32':
        same = (a.M.actions == b.M.actions).all(dim=1) & (a.M.size == b.M.size)
        assert float(same.float().mean()) >= 0.7, float(same.float().mean())
        keep = same.nonzero().flatten()
    else:
        keep = torch.arange(40, device='cuda')
        for k in ('lat_sum', 'lat_cnt', 'reward_sum'):
            assert torch.equal(getattr(a.monitors, k), getattr(b.monitors, k)), k
        assert torch.equal(ea.env_ctr, eb.env_ctr)
    assert int(a.M.actions.max()) == 5
    for x, y in ((a.M.size, b.M.size), (a.M.head, b.M.head), (a.M.actions, b.M.actions),
                 (a.M.rewards, b.M.rewards), (a.M.states, b.M.states),
                 (a.M.next_states, b.M.next_states), (a.M.terminals, b.M.terminals),
                 (a.M.counter, b.M.counter), (a.policy.counter, b.policy.counter)):
        assert torch.equal(x[keep], y[keep])
    tol = dict(rtol=1e-9, atol=1e-12) if dtype_name == 'f64' else dict(rtol=5e-3, atol=1e-4)
    for i in keep.cpu().numpy()[[0, len(keep) // 2, -1]]:
        for x, y in zip(a._online.get_weights(int(i)), b._online.get_weights(int(i))):
            assert np.allclose(x, y, **tol), float(np.abs(x - y).max())
        for x, y in zip(a._target.get_weights(int(i)), b._target.get_weights(int(i))):
            assert np.allclose(x, y, **tol)


def test_full_size_c5_sample_and_conservation(torch_cuda, golden):
    """C5 at its full size — 8 192 linear_track(10, 2) instances, float64 6-64-64-4 networks, the
    two-kernel loop replayed from its HIP graph: eight instances spread over the range equal the
    same global instances run alone through the PyTorch loop (transitions exactly, weights to
    float64 round-off); over all instances every step was stored and counted once."""
    torch = torch_cuda
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.memory import DQNMemory
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    D = golden('dqn_trace')
    init = [D['dqn_i0/init_%d' % i] for i in range(6)]
    nodes, starts = linear_track(10, 2, 1., 20., 'right')
    n, budget = 8192, 40

    def run(n_envs, base, fused):
        env = Topology(nodes, starts, n_envs=n_envs, seed=SEED, instance_base=base)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(_mlp(torch, init)), gamma=0.8, memory=DQNMemory(capacity=256))
        ag.fused_loop = None if fused else False
        ag.use_graph = None if fused else False
        ag._run(env, 10 ** 6, 100, 32, True, budget=budget)
        torch.cuda.synchronize()
        return ag, env

    big, env = run(n, 0, True)
    assert big.fused_steps == budget and big.fused_graph_steps == 32
    size = big.M.size.cpu().numpy()
    assert (size == budget).all(), 'every instance stores one experience per step'
    trials = big.trial.cpu().numpy()
    mon = big.monitors
    assert int(mon.lat_cnt.sum().item()) == int(trials.sum())
    # steps of finished trials + steps of the running ones = the budget, per instance and in total
    lat_total = int(mon.lat_sum.sum().item()) + int(mon.lat_cnt.sum().item())   # logs['steps'] is 0-based
    assert lat_total <= n * budget and lat_total >= n * budget - n * 100
    assert int(big.policy.counter.min().item()) == int(big.policy.counter.max().item()) == budget
    for g in (0, 1, 63, 64, 4097, 8190, 8191, 5000):
        one, _ = run(1, g, False)
        assert one.fused_steps == 0
        assert torch.equal(big.M.actions[g, :budget], one.M.actions[0, :budget]), g
        assert torch.equal(big.M.next_states[g, :budget], one.M.next_states[0, :budget]), g
        assert int(big.trial[g]) == int(one.trial[0])
        for x, y in zip(big._online.get_weights(g), one._online.get_weights(0)):
            assert np.allclose(x, y, rtol=1e-9, atol=1e-12), (g, float(np.abs(x - y).max()))
        for x, y in zip(big._target.get_weights(g), one._target.get_weights(0)):
            assert np.allclose(x, y, rtol=1e-9, atol=1e-12), g


@pytest.mark.parametrize('kernel', [None, 'lds'])
def test_dyna_dqn_two_kernel_loop_equals_torch_loop(torch_cuda, monkeypatch, kernel):
    """DynaDQN.train through cobel_dqn_act in world-model mode + cobel_dqn_replay reading the
    batch as rows of the observation table, against the PyTorch loop: identical model tables,
    stream counters, trial counts and monitors; weights to float64 round-off.  kernel: the form the
    library picks for 20 one-hot inputs in float64 (streaming) or the parameter-staging one pinned."""
    torch = torch_cuda
    if kernel:
        monkeypatch.setenv('COBEL_DEBUG', '1')
        monkeypatch.setenv('COBEL_DEBUG_DQN_KERNEL', kernel)
    import bench
    from cobel_amd.agent import DynaDQN
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    world = make_gridworld(4, 5, terminals=[3], rewards=np.array([[3, 1.0]]), goals=[3],
                           invalid_transitions=[(6, 7), (7, 6)])

    def run(fused):
        torch.manual_seed(5)
        env = Gridworld(world, n_envs=40, seed=991, instance_base=3)
        ag = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                     TorchNetwork(bench._mlp(20, 4)), gamma=0.9)
        ag.fused_loop = None if fused else False
        ag.use_graph = None if fused else False
        ag.train(env, 4, 12, 32)
        ag.train(env, 2, 12, 32)
        torch.cuda.synchronize()
        return ag, env

    (a, ea), (b, eb) = run(True), run(False)
    assert a.fused_steps > 0 and b.fused_steps == 0
    assert torch.equal(a.M.rewards, b.M.rewards) and torch.equal(a.M.states, b.M.states)
    assert torch.equal(a.M.terminals, b.M.terminals) and torch.equal(a.M.counter, b.M.counter)
    assert torch.equal(a.policy.counter, b.policy.counter) and torch.equal(ea.env_ctr, eb.env_ctr)
    assert torch.equal(a.trial, b.trial) and int(a.trial.min()) == 6
    for k in ('lat_sum', 'lat_cnt', 'reward_sum'):
        assert torch.equal(getattr(a.monitors, k), getattr(b.monitors, k)), k
    assert float(a.M.rewards.abs().max()) > 0.0
    for i in (0, 11, 39):
        for x, y in zip(a._online.get_weights(i), b._online.get_weights(i)):
            assert np.allclose(x, y, rtol=1e-9, atol=1e-12), float(np.abs(x - y).max())
        for x, y in zip(a._target.get_weights(i), b._target.get_weights(i)):
            assert np.allclose(x, y, rtol=1e-9, atol=1e-12)
