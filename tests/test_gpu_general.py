"""The general tabular path (csrc/general.hip): action counts other than four, batches beyond what
one wavefront plans, state counts beyond the LDS — against the reference's rows (epsilon-greedy
over six values), the C oracle, and the wavefront kernels on runs both can serve."""
import numpy as np
import pytest

from conftest import SEED, as_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch


def test_eps_greedy_six_actions_kat(torch_cuda, golden):
    """cobel_eps_greedy_n on the reference's six-value rows: every action exact, probabilities
    bit for bit; and on four-value rows it equals cobel_eps_greedy."""
    torch = torch_cuda
    from cobel_amd import _lib
    rows = golden('eps_greedy_kat')['rows6']
    n = len(rows)
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), device='cuda').to(dt).contiguous()  # noqa: E731
    v, bits, u = d(rows[:, 1:7], torch.float32), d(rows[:, 7], torch.uint8), d(rows[:, 8], torch.float64)
    act = torch.empty(n, dtype=torch.uint8, device='cuda')
    probs = torch.empty((n, 6), dtype=torch.float64, device='cuda')
    for eps in np.unique(rows[:, 0]):
        sel = torch.as_tensor(np.flatnonzero(rows[:, 0] == eps), device='cuda')
        vv, bb, uu = v[sel].contiguous(), bits[sel].contiguous(), u[sel].contiguous()
        a_out = torch.empty(len(sel), dtype=torch.uint8, device='cuda')
        p_out = torch.empty((len(sel), 6), dtype=torch.float64, device='cuda')
        _lib.check(_lib.lib().cobel_eps_greedy_n(_lib.ptr(vv), _lib.ptr(bb), _lib.ptr(uu), float(eps),
                                                 _lib.ptr(a_out), _lib.ptr(p_out), len(sel), 6, None))
        act[sel], probs[sel] = a_out, p_out
    assert np.array_equal(act.cpu().numpy(), rows[:, 9].astype(np.uint8))
    assert np.array_equal(probs.cpu().numpy(), rows[:, 10:16])
    # four values: the general entry equals the four-action one (float32 and float64 rows)
    r4 = golden('eps_greedy_kat')['rows']
    r4 = r4[r4[:, 1] == 0.3]
    v4, b4, u4 = d(r4[:, 2:6], torch.float32), d(r4[:, 6], torch.uint8), d(r4[:, 7], torch.float64)
    outs = []
    for fn, extra in ((_lib.lib().cobel_eps_greedy, ()), (_lib.lib().cobel_eps_greedy_n, (4,))):
        a_out = torch.empty(len(r4), dtype=torch.uint8, device='cuda')
        p_out = torch.empty((len(r4), 4), dtype=torch.float64, device='cuda')
        _lib.check(fn(_lib.ptr(v4), _lib.ptr(b4), _lib.ptr(u4), 0.3, _lib.ptr(a_out),
                      _lib.ptr(p_out), len(r4), *extra, None))
        outs.append((a_out, p_out))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    v8 = d(r4[:, 2:6], torch.float64)
    a64 = torch.empty(len(r4), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().cobel_eps_greedy_n_f64(_lib.ptr(v8), _lib.ptr(b4), _lib.ptr(u4), 0.3,
                                                 _lib.ptr(a64), None, len(r4), 4, None))
    b64 = torch.empty_like(a64)
    _lib.check(_lib.lib().cobel_eps_greedy_f64(_lib.ptr(v8), _lib.ptr(b4), _lib.ptr(u4), 0.3,
                                               _lib.ptr(b64), None, len(r4), None))
    assert torch.equal(a64, b64)


def test_hexagonal_topology_env_and_policy_surface(torch_cuda, golden):
    """Topology over hexagonal(5): six actions, steps follow the reference's neighbour table,
    the policy facade selects among six values, the four-action entry points refuse the world."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import SR
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import hexagonal
    from cobel_amd.policy import EpsilonGreedy
    from cobel_amd.spaces import Discrete
    K = golden('topology_kat')
    nodes, starts = hexagonal(5, (0.0, 2.0), 3.0, '7')
    env = Topology(nodes, starts, n_envs=64, seed=SEED)
    assert int(env.action_space.n) == 6 and env.handle.n_actions == 6
    nbr, rew, term = K['hex_5_goal7/nbr'], K['hex_5_goal7/reward'], K['hex_5_goal7/terminal']
    rng = np.random.default_rng(0)
    for _ in range(12):
        before = env.state.cpu().numpy()
        act = rng.integers(0, 6, 64).astype(np.uint8)
        _, r, done, trunc, _ = env.step(torch.as_tensor(act, device='cuda'))
        after = env.state.cpu().numpy()
        assert np.array_equal(after, nbr[before, act])
        assert np.array_equal(r.cpu().numpy(), rew[after].astype(np.float32))
        assert np.array_equal(done.cpu().numpy(), term[after].astype(bool))
        env.reset(done)
    pol = EpsilonGreedy(0.0)
    assert pol.select_action(np.array([0., 0., 0., 2., 0., 1.], dtype=np.float32), u=0.5) == 3
    p = pol.get_action_probs(np.array([1., 0., 1., 0., 0., 1.], dtype=np.float32),
                             np.array([1, 1, 0, 1, 1, 1], dtype=bool))
    assert np.array_equal(p, [0.5, 0, 0, 0, 0, 0.5])
    with pytest.raises(AssertionError):      # the SR kernels are four-action kernels
        SR(Discrete(23), env.action_space, EpsilonGreedy(0.1))
    run = _lib.SRRun()
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.lib().cobel_sr_run(env.handle.ptr, run, None))


def _dynaq(torch, world, n, seed, general, eps=0.1, base=0):
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld(world, n_envs=n, seed=seed, instance_base=base)
    ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(eps))
    ag.force_general = general
    ag.track_instances = True
    ag.track_occupancy = True
    ag.track_responses = True
    return env, ag


def _same_tab(torch, a, b):
    assert torch.equal(a._q, b._q), 'Q tables differ'
    assert torch.equal(a.inst, b.inst)
    if hasattr(a.M, 'table'):
        assert torch.equal(a.M.table, b.M.table) and torch.equal(a.M.index, b.M.index)
        assert torch.equal(a.M.counter, b.M.counter)
    for name in ('lat_sum', 'lat_cnt', 'resp_cnt', 'reward_sum'):
        assert torch.equal(getattr(a.monitors, name), getattr(b.monitors, name)), name
    assert torch.equal(a.monitors.lat_trace, b.monitors.lat_trace)
    if a.monitors.occupancy is not None or b.monitors.occupancy is not None:
        assert torch.equal(a.monitors.occupancy, b.monitors.occupancy)


def test_general_kernel_equals_wavefront_kernels(torch_cuda, golden_worlds):
    """Four actions, batch <= 62: the lane-per-instance kernel and the wavefront kernels leave
    identical Q tables, model tables + digests, counters and monitors — Dyna-Q (plain, masked,
    episodic, without replay, train then test) and QAgent (online, with its replay log)."""
    torch = torch_cuda
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    world = as_world(golden_worlds('walls_8x8'))
    nxt = np.asarray(world['next'])
    bump = nxt != np.arange(64)[:, None]
    bump[~bump.any(axis=1)] = True
    # (batch 24: one pass of the wavefront kernel; 70 and 130: two and three passes)
    for kind, batch in (('plain', 24), ('mask', 24), ('episodic', 24), ('no_replay', 24),
                        ('train_test', 24), ('plain', 70), ('mask', 130), ('episodic', 70)):
        out = []
        for general in (False, True):
            env, ag = _dynaq(torch, world, 96, 17, general, eps=0.2, base=3)
            if kind == 'mask':
                ag.mask_actions, ag.action_mask = True, bump
            ag.episodic_replay = kind == 'episodic'
            ag.train(env, 7, 30, batch, no_replay=(kind == 'no_replay'))
            if kind == 'train_test':
                ag.test(env, 4, 30)
            torch.cuda.synchronize()
            out.append(ag)
        _same_tab(torch, out[0], out[1])
    for B in (0, 24):
        out = []
        for general in (False, True):
            env = Gridworld(world, n_envs=80, seed=23)
            ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.15))
            ag.force_general = general
            ag.track_instances = ag.track_occupancy = ag.track_responses = True
            ag.train(env, 6, 25, B)
            ag.train(env, 3, 25, B)
            torch.cuda.synchronize()
            out.append(ag)
        _same_tab(torch, out[0], out[1])
        if B:
            assert torch.equal(out[0]._log, out[1]._log)
            assert out[0].M[:5] == out[1].M[:5]


@pytest.mark.parametrize('agent_name', ['dynaq', 'q'])
def test_batch_of_100_updates_vs_oracle(torch_cuda, golden_worlds, agent_name):
    """batch_size = 100 (the reference has no limit; one wavefront plans at most 62 per pass):
    Dyna-Q and — since round 6 — QAgent in two passes of the wavefront kernel, against the C
    oracle — Q, model tables, replay counters, latencies."""
    torch = torch_cuda
    from oracle import c_oracle
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    tab = golden_worlds('walls_8x8')
    world = as_world(tab)
    n, trials, steps, B = 20, 5, 30, 100
    w = c_oracle.OracleWorld([tab])
    if agent_name == 'dynaq':
        env, ag = _dynaq(torch, world, n, SEED, False)
        ag.train(env, trials, steps, B)
        # Dyna-Q: two passes (62 + 38 updates) of the wavefront kernel with the digest in HBM ...
        assert ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)['kernel'] == \
            _lib.TAB_KERNEL_WPI_INDEX
        # ... which leave what the lane-per-instance kernel leaves
        env2, ag2 = _dynaq(torch, world, n, SEED, True)
        ag2.train(env2, trials, steps, B)
        _same_tab(torch, ag, ag2)
        o = c_oracle.TabOracle(w, n, c_oracle.AG_DYNAQ, SEED, True, trial_cap=trials)
        o.run(trials, steps, B)
        assert np.array_equal(ag.M.rewards.astype(np.float64), o.MR)
        assert np.array_equal(ag.M.states, o.MS) and np.array_equal(ag.M.terminals, o.MT)
        assert np.array_equal(ag.M.counter.cpu().numpy(), o.inst['ctr_memory'].astype(np.int32))
    else:
        env = Gridworld(world, n_envs=n, seed=SEED)
        ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        ag.track_instances = True
        ag.train(env, trials, steps, B)
        assert ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)['kernel'] == \
            _lib.TAB_KERNEL_WPI
        o = c_oracle.TabOracle(w, n, c_oracle.AG_Q, SEED, True, alpha=0.9, gamma=0.8,
                               trial_cap=trials, log_cap=trials * steps)
        o.run(trials, steps, B)
        assert np.array_equal(ag.inst[:, _lib.I_LOG_LEN].cpu().numpy(), o.inst['log_len'].astype(np.int32))
    torch.cuda.synchronize()
    assert np.array_equal(ag._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy()[:, :trials], o.lat_trace)
    assert np.array_equal(ag.inst[:, _lib.I_CTR_POLICY].cpu().numpy(),
                          o.inst['ctr_policy'].astype(np.int32))


def test_state_count_beyond_lds_runs_on_the_general_kernel(torch_cuda):
    """90 x 90 = 8 100 states: Q + model digest of one instance exceed 160 KiB of LDS, the run
    takes the general kernel and matches the C oracle."""
    torch = torch_cuda
    from oracle import c_oracle
    from cobel_amd import _lib
    from cobel_amd.misc.gridworld_tools import make_open_field
    world = make_open_field(90, 90, 4000, 1)
    world['starting_states'] = np.array([3999, 4001, 3910, 4090, 3820])
    n, trials, steps, B = 6, 4, 40, 16
    env, ag = _dynaq(torch, world, n, 5, False)
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    assert ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)['kernel'] == \
        _lib.TAB_KERNEL_GENERAL
    w = c_oracle.OracleWorld([dict(next=world['next'], reward=world['rewards'],
                                   terminal=world['terminals'], starts=world['starting_states'])])
    o = c_oracle.TabOracle(w, n, c_oracle.AG_DYNAQ, 5, True, trial_cap=trials)
    o.run(trials, steps, B)
    assert np.array_equal(ag._q.cpu().numpy().astype(np.float64), o.Q)
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy()[:, :trials], o.lat_trace)
    assert (ag.monitors.lat_trace.cpu().numpy()[:, :trials] < steps - 1).any(), 'goal never reached'


def test_dqn_on_hexagonal_topology(torch_cuda):
    """DQN over six actions (the two-kernel loop since late round 4: the streaming form of the
    replay step serves 1 .. 8 actions): runs, selects all six actions, and instance i of a batch
    equals the same instance run alone."""
    torch = torch_cuda
    from collections import OrderedDict
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.misc.topology_tools import hexagonal
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    nodes, starts = hexagonal(4)

    def run(n, base):
        torch.manual_seed(3)
        net = torch.nn.Sequential(OrderedDict([
            ('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
            ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
            ('output', torch.nn.Linear(64, 6))])).double()
        env = Topology(nodes, starts, n_envs=n, seed=SEED, instance_base=base)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3), TorchNetwork(net),
                 gamma=0.8)
        ag.train(env, 3, 15, 32)
        torch.cuda.synchronize()
        return ag
    vec, one = run(4, 0), run(1, 2)
    assert vec.fused_steps > 0
    size = int(one.M.size[0].item())
    assert size == int(vec.M.size[2].item()) and size > 0
    assert torch.equal(vec.M.actions[2, :size], one.M.actions[0, :size])
    assert int(vec.M.actions.max().item()) == 5
    for a, b in zip(vec._online.get_weights(2), one._online.get_weights(0)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_batches_done_counts_what_planning_evaluates(torch_cuda):
    """cobel_tab_run_t.batches_done: a Dyna-Q batch is drawn in every learning step but evaluated
    only once the instance's Q or its model's reward estimates hold something other than +0.0f.
    In a world without rewards nothing is ever evaluated (and Q stays zero); with a reward, the
    evaluated batches are the steps from the first rewarded one on; the general kernel and
    Q-learning's log replay evaluate every batch."""
    torch = torch_cuda
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    bare = make_gridworld(4, 4, terminals=[15], rewards=np.array([[15, 0.0]]), goals=[15])
    env = Gridworld(bare, n_envs=70, seed=1)
    ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.3))
    ag.train(env, 3, 20, 16)
    steps = int(ag.monitors.steps_done.item())
    assert steps > 0 and int(ag.batches_done.item()) == 0 and float(ag._q.abs().max()) == 0.0
    ag.force_general = True                          # the general kernel plans unconditionally
    ag.train(env, 1, 20, 16)
    assert int(ag.batches_done.item()) == int(ag.monitors.steps_done.item()) - steps
    paid = make_gridworld(4, 4, terminals=[15], rewards=np.array([[15, 1.0]]), goals=[15])
    env = Gridworld(paid, n_envs=70, seed=1)
    ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.3))
    ag.track_instances = True
    ag.train(env, 6, 30, 16)
    steps, evaluated = int(ag.monitors.steps_done.item()), int(ag.batches_done.item())
    assert 0 < evaluated < steps
    # an instance evaluates its batches from the step that first pays on: at least one batch per
    # instance that has been rewarded, at most its steps after the first trial's first step
    rewarded = int((ag._q.abs().amax(dim=(1, 2)) > 0).sum())
    assert rewarded > 0 and evaluated >= rewarded
    never = ag._q.abs().amax(dim=(1, 2)) == 0
    if bool(never.any()):     # instances that never reached the goal contributed nothing
        lat = ag.monitors.lat_trace[:, :6] + 1
        assert evaluated <= steps - int(lat[never].sum())
    env = Gridworld(paid, n_envs=3, seed=2)
    qa = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.3))
    qa.train(env, 2, 10, 8)
    assert int(qa.batches_done.item()) == int(qa.monitors.steps_done.item())


@pytest.mark.parametrize('A,n_worlds,batch,masked',
                         [(6, 1, 32, False), (6, 1, 0, False), (6, 3, 62, False), (5, 1, 8, False),
                          (7, 2, 24, False), (8, 1, 32, False), (3, 1, 16, False), (2, 2, 5, False),
                          (1, 1, 4, False), (6, 1, 32, True), (8, 1, 20, True), (3, 1, 0, True),
                          (12, 1, 32, True), (20, 1, 16, False), (12, 2, 16, True), (30, 3, 9, False),
                          (6, 1, 32, 'psets'), (12, 1, 16, 'psets'), (6, 2, 8, 'occupancy'),
                          (12, 1, 32, 'occupancy'),
                          # batches above 62: further passes of the wavefront kernel (round 6)
                          (6, 1, 100, False), (12, 1, 70, True), (5, 2, 130, False)])
def test_wavefront_kernel_for_other_action_counts_equals_the_general_kernel(torch_cuda, A, n_worlds, batch,
                                                                            masked):
    """Q-learning on worlds of 1..32 (not four) actions, with an action mask or without, runs one
    wavefront per instance with its tables in LDS (csrc/tabular_nact.hip: rows of 8 / 16 / 32
    values); the lane-per-instance general kernel executes the
    reference's loop literally: same Q, logs, counters, monitors — training in two sessions with a
    test phase in between, launches cut at odd step counts."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Topology
    from cobel_amd.policy import EpsilonGreedy
    rng = np.random.default_rng(100 * A + n_worlds)
    S = 37

    def graph(k):
        nbr = rng.integers(0, S, (S, A))
        nbr = np.where(rng.random((S, A)) < 0.2, np.arange(S)[:, None], nbr)
        term = rng.random(S) < 0.08
        term[0] = False
        rew = np.where(rng.random(S) < 0.2, rng.choice([1.0, -1.0, 0.5], S), 0.0)
        rew[term] = 2.0
        nodes = {str(i): {'id': str(i), 'pose': np.array([float(i), float(k), 0., 0., 0., 0.]),
                          'neighbors': [str(int(j)) for j in nbr[i]], 'reward': float(rew[i]),
                          'terminal': bool(term[i])} for i in range(S)}
        return nodes
    worlds = [graph(k) for k in range(n_worlds)]
    outs = []
    for general in (False, True):
        env = Topology(worlds[0], None, n_envs=150, seed=SEED, instance_base=5)
        if n_worlds > 1:    # (several graphs in one handle: instance g walks graph g % n_worlds)
            from cobel_amd.interface.gridworld import WorldHandle
            env.handle = WorldHandle([Topology(w, None, seed=1).world for w in worlds], env.device)
        if masked == 'psets':   # (per-instance hyper-parameters: the grid search's fan-out; the CDF path)
            pr = np.random.default_rng(5)
            ag = QAgent(env.observation_space, env.action_space,
                        EpsilonGreedy(pr.choice([0.05, 0.2, 0.5], 150)),
                        learning_rate=pr.choice([0.3, 0.7, 0.9], 150), gamma=pr.choice([0.8, 0.95], 150))
        else:
            ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.2), learning_rate=0.7,
                        gamma=0.9)
        ag.force_general = general
        ag.track_instances = True
        ag.track_responses = True
        ag.track_occupancy = masked == 'occupancy'
        if masked is True:   # (masked rows: the wave works the selection's CDF out itself — no threshold table)
            m = np.random.default_rng(7 * A).random((S, A)) < 0.6
            m[np.arange(S), np.random.default_rng(A).integers(0, A, S)] = True
            ag.mask_actions, ag.action_mask = True, m
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False) | (_lib.F_MASK_ACTIONS if masked is True else 0)
        ag.monitors.reserve(64, 150, True)
        if batch:
            ag.reserve_replay(400)
        kinds = set()
        for steps in (2, 7, 33, 64, 5):      # (a one-step launch carries per-step host logs: general kernel)
            kinds.add(ag.describe_launch(env, ag.policy, flags, 40, 11, steps, batch)['kernel'])
            ag._launch(env, ag.policy, flags, 40, 11, steps, batch)
        ag.test(env, 3, 9)
        ag.train(env, 4, 13, batch)
        torch.cuda.synchronize()
        outs.append((ag, kinds))
    (a, ka), (b, kb) = outs
    assert ka == {_lib.TAB_KERNEL_WQN} and kb == {_lib.TAB_KERNEL_GENERAL}
    assert a._q.abs().sum() > 0 and int(a.inst[:, _lib.I_LOG_LEN].sum()) > 0
    _same_tab(torch, a, b)
    if batch:
        assert torch.equal(a._log, b._log)
    assert int(a.batches_done.item()) == int(b.batches_done.item())


@pytest.mark.parametrize('general', [False, True])
@pytest.mark.parametrize('A', [6, 12])
def test_parameter_sets_and_visit_counts_against_the_oracle(torch_cuda, A, general):
    """Per-instance hyper-parameters (the grid search's fan-out) and the visit counters on worlds of
    six / twelve actions, on the wavefront kernel (k_tab_wqn) and the lane-per-instance kernel
    (k_tab_general), EACH against the restatement of the reference's loop (agent/q.py:305-354,
    policy/greedy.py:60-88) run instance by instance with that instance's parameters — not kernel
    against kernel."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Topology
    from cobel_amd.policy import EpsilonGreedy
    from oracle import ref_loop
    from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG
    r = np.random.default_rng(900 + A)
    S = 31
    nbr = r.integers(0, S, (S, A))
    nbr = np.where(r.random((S, A)) < 0.15, np.arange(S)[:, None], nbr)
    terminal = np.zeros(S, dtype=bool)
    terminal[[5, 19]] = True
    reward = np.zeros(S)
    reward[5], reward[19], reward[12] = 1.0, -0.5, 0.25
    nodes = {str(i): {'id': str(i), 'pose': np.array([float(i % 6), float(i // 6), 0., 0., 0., 0.]),
                      'neighbors': [str(int(j)) for j in nbr[i]], 'reward': float(reward[i]),
                      'terminal': bool(terminal[i])} for i in range(S)}
    n, trials, steps, B = 36, 4, 25, 12
    pr = np.random.default_rng(5)
    eps = pr.choice([0.05, 0.2, 0.5], n)
    alpha = pr.choice([0.3, 0.7, 0.9], n)
    gamma = pr.choice([0.8, 0.95], n)
    env = Topology(nodes, None, n_envs=n, seed=SEED, instance_base=11)
    ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha,
                gamma=gamma)
    ag.track_instances = True
    ag.track_occupancy = True
    ag.force_general = general
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    what = ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)
    assert what['kernel'] == (_lib.TAB_KERNEL_GENERAL if general else _lib.TAB_KERNEL_WQN)
    w = env.world
    tab = dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'], starts=w['starting_states'])
    Q = ag._q.cpu().numpy()
    lat = ag.monitors.lat_trace.cpu().numpy()
    visits = np.zeros(S, dtype=np.int64)
    for i in range(n):
        g = 11 + i
        renv = ref_loop.RefGridworld(tab, TapeRNG(SEED, g, STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(float(eps[i]), TapeRNG(SEED, g, STREAM_POLICY))
        ref = ref_loop.RefQAgent(S, A, pol, TapeRNG(SEED, g, STREAM_MEMORY), float(alpha[i]),
                                 float(gamma[i]), dtype=np.float32)
        tr = ref_loop.new_trace()
        ref.train(renv, trials, steps, B, trace=tr)
        assert np.array_equal(lat[i, :trials], tr['steps']), i
        assert np.array_equal(Q[i].reshape(S, A), ref.Q), i
        visits += np.bincount(np.array(tr['sarsn'])[:, 3].astype(int), minlength=S)
    assert np.array_equal(ag.monitors.occupancy.cpu().numpy().reshape(-1)[:S].astype(np.int64), visits)
    assert visits.sum() == ag.env_steps() and len(set(zip(eps, alpha, gamma))) > 4


@pytest.mark.parametrize('general', [False, True])
@pytest.mark.parametrize('A', [9, 12, 17, 32])
def test_more_than_eight_actions(torch_cuda, A, general):
    """Nine to 32 neighbours per node (interface/topology.py:110-112 takes any count): the wide
    rows of the wavefront kernel (k_tab_wqn<., 16 / 32>) and, forced, of the general kernel — QAgent
    with log replay against the restatement of the reference's loop, the decoded replay memory,
    epsilon-greedy rows against NumPy."""
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Topology
    from cobel_amd.policy import EpsilonGreedy
    from oracle import ref_loop
    from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG
    r = np.random.default_rng(100 + A)
    S = 40
    nbr = r.integers(0, S, (S, A))
    terminal = np.zeros(S, dtype=bool)
    terminal[[7, 23]] = True
    reward = np.zeros(S)
    reward[7], reward[23], reward[11] = 1.0, -0.5, 0.25
    nodes = {str(i): {'id': str(i), 'pose': np.array([float(i % 7), float(i // 7), 0., 0., 0., 0.]),
                      'neighbors': [str(int(j)) for j in nbr[i]], 'reward': float(reward[i]),
                      'terminal': bool(terminal[i])} for i in range(S)}
    n, trials, steps, B = 70, 4, 30, 16
    env = Topology(nodes, None, n_envs=n, seed=SEED)
    assert int(env.action_space.n) == A
    ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.2), learning_rate=0.9, gamma=0.9)
    ag.track_instances = True
    ag.force_general = general
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    what = ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)
    assert what['kernel'] == (_lib.TAB_KERNEL_GENERAL if general else _lib.TAB_KERNEL_WQN)
    w = env.world
    tab = dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'], starts=w['starting_states'])
    Q = ag._q.cpu().numpy()
    lat = ag.monitors.lat_trace.cpu().numpy()
    for i in (0, 33, 69):
        renv = ref_loop.RefGridworld(tab, TapeRNG(SEED, i, STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(0.2, TapeRNG(SEED, i, STREAM_POLICY))
        ref = ref_loop.RefQAgent(S, A, pol, TapeRNG(SEED, i, STREAM_MEMORY), 0.9, 0.9, dtype=np.float32)
        tr = ref_loop.new_trace()
        ref.train(renv, trials, steps, B, trace=tr)
        assert np.array_equal(lat[i, :trials], tr['steps'])
        assert np.array_equal(Q[i].reshape(S, A), ref.Q)
        if i == 0:   # the replay memory as the reference shows it (five action bits, 13-bit states)
            M = ag.M
            assert len(M) == len(ref.M)
            for got, exp in zip(M, ref.M):
                assert got['state'][0] == exp[0] and got['action'] == exp[1] and got['reward'] == exp[2]
                assert got['next_state'][0] == exp[3] and got['terminal'] == exp[4]
    # epsilon-greedy rows of A values against policy/greedy.py:77-86 in NumPy
    v = r.integers(0, 3, (200, A)).astype(np.float32)
    u = r.random(200)
    dv = torch.as_tensor(v, device='cuda')
    du = torch.as_tensor(u, device='cuda')
    act = torch.empty(200, dtype=torch.uint8, device='cuda')
    probs = torch.empty((200, A), dtype=torch.float64, device='cuda')
    _lib.check(_lib.lib().cobel_eps_greedy_n(_lib.ptr(dv), None, _lib.ptr(du), 0.3, _lib.ptr(act),
                                             _lib.ptr(probs), 200, A, None))
    for k in range(200):
        p = np.full(A, 0.3 / A)
        ties = v[k] == v[k].max()
        p[ties] += ((1.0 - 0.3) * 1.0) / ties.sum()
        c = np.cumsum(p)
        assert np.array_equal(probs[k].cpu().numpy(), p)
        assert int(act[k].item()) == int(np.searchsorted(c / c[-1], u[k], side='right'))
    # beyond eight actions the masks are 32-bit words (round 5; tests/test_gpu_lifted.py): all ones
    # changes nothing, a misaligned pointer is refused
    mask = torch.full((201,), -1, dtype=torch.int32, device='cuda')
    act2 = torch.empty(200, dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().cobel_eps_greedy_n(_lib.ptr(dv), _lib.ptr(mask), _lib.ptr(du), 0.3,
                                             _lib.ptr(act2), None, 200, A, None))
    assert torch.equal(act, act2)
    rc = _lib.lib().cobel_eps_greedy_n(_lib.ptr(dv), mask.data_ptr() + 1, _lib.ptr(du), 0.3,
                                       _lib.ptr(act2), None, 200, A, None)
    assert rc == _lib.E_ARG
