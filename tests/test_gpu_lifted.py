"""What earlier rounds refused where the reference runs (COBEL_E_UNSUPPORTED): successor
representation beyond 4 096 states (agent/sr.py:109-140 has no size limit), per-instance
parameter sets on worlds whose rows are distributions, action masks on rows of nine to 32 values
(policy/greedy.py:60-88 compacts any mask), replay logs whose states do not fit the packed record
(agent/q.py:213 appends tuples) — each against the oracle or the definition."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


def test_sr_on_a_72x72_world_vs_oracle():
    """5 184 states (132 KiB of LDS for the six rows of an instance): SR matrix rows, transition
    table, reward estimate and counters of every instance against the C oracle, bit for bit."""
    import torch
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    world = make_open_field(72, 72, 0, 1)
    world['starting_states'] = np.array([1, 72, 73, 74, 146, 5183])   # (some trials reach the goal)
    n, trials, steps = 3, 5, 40
    env = Gridworld(world, n_envs=n, seed=SEED)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.3), learning_rate=0.25, gamma=0.9)
    ag.track_instances = True
    ag.train(env, trials, steps)
    torch.cuda.synchronize()
    w = c_oracle.OracleWorld([dict(next=world['next'], reward=world['rewards'],
                                   terminal=world['terminals'], starts=world['starting_states'])])
    o = c_oracle.SROracle(w, n, SEED, True, alpha=0.25, gamma=0.9, epsilon=0.3, trial_cap=trials)
    o.run(trials, steps)
    assert np.array_equal(ag._T.cpu().numpy().astype(np.int64), o.T)
    assert np.array_equal(ag._rw.cpu().numpy().astype(np.float64), o.RW)
    for i in range(n):
        assert np.array_equal(ag._sr[i].cpu().numpy().astype(np.float64), o.SR[i]), i
    inst = ag.inst.cpu().numpy()
    assert np.array_equal(inst[:, 0], o.inst['state']) and np.array_equal(inst[:, 2], o.inst['trial'])
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy()[:, :trials], o.lat_trace[:, :trials])
    assert o.RW.any() and int((o.lat_trace[:, :trials] < steps - 1).sum()) > 0   # (a goal was met)
    # retrieve_q through its own entry point on the same tables
    q = ag.retrieve_q(np.array([73, 1, 5183]))
    assert q.shape[-1] == 4


def test_sr_beyond_the_lds_rows_vs_oracle():
    """81 x 81 = 6 561 states: six rows of them no longer fit one workgroup's LDS, the BIG kernels
    read every row where it lies (same pairwise sums, same float64 row update).  Two instances
    with visit counters against the C oracle, bit for bit; retrieve_q through its entry point
    against the values the oracle's own walk selected by."""
    import torch
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    side = 81
    world = make_open_field(side, side, 0, 1)
    world['starting_states'] = np.array([1, side, side + 1, side + 2, 2 * side + 2, side * side - 1])
    n, trials, steps = 2, 5, 40
    env = Gridworld(world, n_envs=n, seed=SEED)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.3), learning_rate=0.25, gamma=0.9)
    ag.track_instances = True
    ag.track_occupancy = True
    ag.train(env, trials, steps)
    torch.cuda.synchronize()
    w = c_oracle.OracleWorld([dict(next=world['next'], reward=world['rewards'],
                                   terminal=world['terminals'], starts=world['starting_states'])])
    o = c_oracle.SROracle(w, n, SEED, True, alpha=0.25, gamma=0.9, epsilon=0.3, trial_cap=trials,
                          occupancy=True)
    o.run(trials, steps)
    assert np.array_equal(ag._T.cpu().numpy().astype(np.int64), o.T)
    assert np.array_equal(ag._rw.cpu().numpy().astype(np.float64), o.RW)
    for i in range(n):
        assert np.array_equal(ag._sr[i].cpu().numpy().astype(np.float64), o.SR[i]), i
    inst = ag.inst.cpu().numpy()
    assert np.array_equal(inst[:, 0], o.inst['state']) and np.array_equal(inst[:, 2], o.inst['trial'])
    assert np.array_equal(inst[:, 3], o.inst['ctr_env'].astype(np.int32))
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy()[:, :trials], o.lat_trace[:, :trials])
    assert np.array_equal(ag.monitors.occupancy.cpu().numpy().astype(np.uint64), o.occupancy)
    assert o.RW.any() and int((o.lat_trace[:, :trials] < steps - 1).sum()) > 0   # (a goal was met)
    # retrieve_q: V[T[s][a]] = pairwise sum of SR[T[s][a]] * R in float32, NumPy's order
    states = np.array([side + 1, 1])
    q = ag.retrieve_q(states).cpu().numpy()
    for i, s_ in enumerate(states):
        for a in range(4):
            row = o.SR[i][o.T[i][s_, a]].astype(np.float32)
            assert q[i, a] == np.sum(row * o.RW[i].astype(np.float32)), (i, a)


def test_sr_parameter_sets_on_a_world_of_distributions(golden):
    """Per-instance learning rate / gamma / epsilon together with drawn successors (refused until
    round 5): the instances of a parameter set equal a run with that set launch-wide."""
    import torch
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    from test_gpu_parity import PSET_COMBOS, _pset_arrays
    from test_gpu_stochastic import slippery_world
    world = slippery_world(golden('stochastic_traces'), 'slip_5x6_wind')
    n = 80
    which, alpha, gamma, eps, _ = _pset_arrays(n)

    def run(a, g, e):
        env = Gridworld(world, n_envs=n, seed=31, instance_base=5)
        assert env.handle.stochastic
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(e), learning_rate=a, gamma=g)
        ag.track_instances = True
        ag.train(env, 5, 25)
        torch.cuda.synchronize()
        return ag

    mixed = run(alpha, gamma, eps)
    for k, (a, g, e, _) in enumerate(PSET_COMBOS):
        uni = run(a, g, e)
        sel = torch.as_tensor(np.flatnonzero(which == k), device='cuda')
        assert torch.equal(mixed._sr[sel], uni._sr[sel]), k
        assert torch.equal(mixed._T[sel], uni._T[sel]) and torch.equal(mixed._rw[sel], uni._rw[sel])
        assert torch.equal(mixed.inst[sel], uni.inst[sel]), k
    assert not torch.equal(mixed._sr[0], mixed._sr[1])


def _graph(S, A, seed):
    r = np.random.default_rng(seed)
    nbr = r.integers(0, S, (S, A))
    terminal = np.zeros(S, dtype=bool)
    terminal[[7, 23]] = True
    reward = np.zeros(S)
    reward[7], reward[23], reward[11] = 1.0, -0.5, 0.25
    nodes = {str(i): {'id': str(i), 'pose': np.array([float(i % 97), float(i // 97), 0., 0., 0., 0.]),
                      'neighbors': [str(int(j)) for j in nbr[i]], 'reward': float(reward[i]),
                      'terminal': bool(terminal[i])} for i in range(S)}
    return nodes, nbr


@pytest.mark.parametrize('A,S,general', [(12, 40, False), (12, 40, True), (32, 64, False), (32, 64, True),
                                         (17, 256, False), (17, 512, False), (12, 1000, False),
                                         (12, 9000, False), (9, 16384, False)])
def test_masked_wide_action_rows_and_wide_replay_logs(A, S, general):
    """QAgent with log replay on random graphs: a 12- and a 32-action graph with an action mask
    (32-bit mask words beyond eight actions), a 12-action graph of 9 000 nodes and a nine-action one of
    16 384, the most a world handle holds (their states do not fit the packed replay record of a
    wide-action world: two words per logged experience) —
    Q, escape latencies and the decoded replay memory against the restatement of the reference's
    loop; masked epsilon-greedy rows against policy/greedy.py:60-88 in NumPy."""
    import torch
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Topology
    from cobel_amd.policy import EpsilonGreedy
    from oracle import ref_loop
    from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG
    nodes, nbr = _graph(S, A, 7 * A + S)
    r = np.random.default_rng(A + S)
    mask = r.random((S, A)) < 0.6
    mask[np.arange(S), r.integers(0, A, S)] = True          # (never all actions masked)
    n, trials, steps, B = (70, 4, 30, 16) if S < 1000 else (5, 3, 40, 8)
    env = Topology(nodes, None, n_envs=n, seed=SEED)
    assert int(env.action_space.n) == A
    ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.2), learning_rate=0.9, gamma=0.9)
    ag.mask_actions = True
    ag.action_mask = mask.copy()
    ag.track_instances = True
    ag.force_general = general
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    what = ag.describe_launch(env, ag.policy, _lib.F_LEARN | _lib.F_MASK_ACTIONS, trials, steps, 0, B)
    # (the wavefront kernel keys a cell by s W + a in 14 bits: up to 1 024 / 512 states on rows of 16 / 32)
    wave = not general and S * (16 if A <= 16 else 32) <= 16384
    assert what['kernel'] == (_lib.TAB_KERNEL_WQN if wave else _lib.TAB_KERNEL_GENERAL)
    assert ag._log_words() == (2 if S > 8192 else 1)
    w = env.world
    tab = dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'], starts=w['starting_states'])
    Q = ag._q.cpu().numpy()
    lat = ag.monitors.lat_trace.cpu().numpy()
    for i in sorted({0, n // 2, n - 1}):
        renv = ref_loop.RefGridworld(tab, TapeRNG(SEED, i, STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(0.2, TapeRNG(SEED, i, STREAM_POLICY))
        ref = ref_loop.RefQAgent(S, A, pol, TapeRNG(SEED, i, STREAM_MEMORY), 0.9, 0.9, dtype=np.float32)
        ref.mask_actions, ref.action_mask = True, mask
        tr = ref_loop.new_trace()
        ref.train(renv, trials, steps, B, trace=tr)
        assert np.array_equal(lat[i, :trials], tr['steps'])
        assert np.array_equal(Q[i].reshape(S, A), ref.Q)
        assert all(mask[s, a] for s, a, *_ in ref.M)
        if i == 0:   # the replay memory as the reference shows it
            M = ag.M
            assert len(M) == len(ref.M)
            for got, exp in zip(M, ref.M):
                assert got['state'][0] == exp[0] and got['action'] == exp[1] and got['reward'] == exp[2]
                assert got['next_state'][0] == exp[3] and got['terminal'] == exp[4]
    if A <= 8:
        return
    # masked epsilon-greedy rows of A values (policy/greedy.py:77-86)
    v = r.integers(0, 3, (200, A)).astype(np.float32)
    u = r.random(200)
    m = r.random((200, A)) < 0.5
    m[np.arange(200), r.integers(0, A, 200)] = True
    pol = EpsilonGreedy(0.3)
    act = pol.select_action(v, m, u)
    probs = pol.get_action_probs(v, m)
    for k in range(200):
        idx = np.flatnonzero(m[k])
        vals = v[k][idx]
        p = np.full(len(idx), 0.3 / len(idx))
        ties = vals == vals.max()
        p[ties] += ((1.0 - 0.3) * 1.0) / ties.sum()
        full = np.zeros(A)
        full[idx] = p
        assert np.array_equal(np.asarray(probs[k]), full), k
        c = np.cumsum(p)
        assert int(act[k]) == int(idx[np.searchsorted(c / c[-1], u[k], side='right')]), k
