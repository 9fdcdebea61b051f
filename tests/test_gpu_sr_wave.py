"""k_sr_wave (csrc/sr_wave.hip) — the sparse-reward form of the successor-representation step, one
wavefront per instance — against the C oracle (the reference's full row sums, sr.py:302-306) and
against the row-streaming kernel k_sr on the same seeded inputs: SR matrices, transition tables,
reward estimates, counters and monitors bit for bit.

Covered: the three state counts the kernel serves (16x16, 16x32, 32x32), worlds with one and two
rewarded states (also a negative reward), goals that ARE reached (reward estimates become non-zero
in the middle of a launch and the values stop being all-tie), runs cut into launches at odd
budgets, action masks, Agent.test(), per-instance hyper-parameters, occupancy, and the fallback
for hand-edited reward estimates with more than two non-zeros."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch


def _worlds():
    from cobel_amd.misc.gridworld_tools import make_gridworld, make_open_field
    two = make_gridworld(16, 32, terminals=[0, 511], rewards=np.array([[0, 1.0], [511, -0.5]]),
                         goals=[0], starting_states=[17, 40, 250, 300, 480],
                         invalid_states=[100, 101, 102, 200, 232, 264],
                         invalid_transitions=[(33, 34), (34, 33)])
    near = make_gridworld(16, 16, terminals=[0], rewards=np.array([[0, 2.0], [20, 0.25]]),
                          goals=[0], starting_states=[17, 18, 34, 50])
    return {'open_32x32': make_open_field(32, 32, 0, 1),
            'two_rewards_16x32': two,
            'near_goal_16x16': near,     # a rewarded NON-terminal state beside the goal
            'open_16x16': make_open_field(16, 16, 5, 1)}


def _oracle_world(world):
    from oracle import c_oracle
    return c_oracle.OracleWorld([dict(next=world['next'], reward=world['rewards'],
                                      terminal=world['terminals'],
                                      starts=world['starting_states'])])


def _agent(torch, world, n, seed, eps=0.2, stream_rows=False, base=0, mask=None, alpha=0.1,
           gamma=0.99):
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld(world, n_envs=n, seed=seed, instance_base=base)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha,
            gamma=gamma)
    ag.stream_rows = stream_rows
    ag.track_occupancy = True
    ag.track_responses = True
    ag.track_instances = True
    if mask is not None:
        ag.mask_actions = True
        ag.action_mask = mask
    return env, ag


def _launches(torch, env, ag, budgets, steps_per_trial, learn=True, trial_cap=64):
    from cobel_amd import _lib
    ag._bind(env)
    ag._env_in(env)
    pol = ag.policy if learn else ag.policy_test
    flags = (_lib.F_LEARN if learn else 0) | ag._policy_in(pol, env, not learn)
    if ag.mask_actions:
        flags |= _lib.F_MASK_ACTIONS
    ag.monitors.reserve(trial_cap, ag.n_envs, True)
    for b in budgets:
        ag._launch(env, pol, flags, 0x7fffffff, steps_per_trial, b, 0)
    torch.cuda.synchronize()


def _same(torch, a, b):
    assert torch.equal(a._sr, b._sr), 'SR matrices differ'
    assert torch.equal(a._T, b._T) and torch.equal(a._rw, b._rw)
    assert torch.equal(a.inst, b.inst)
    for name in ('lat_sum', 'lat_cnt', 'resp_cnt', 'reward_sum'):
        assert torch.equal(getattr(a.monitors, name), getattr(b.monitors, name)), name
    assert torch.equal(a.monitors.lat_trace, b.monitors.lat_trace)
    assert torch.equal(a.monitors.occupancy, b.monitors.occupancy)
    assert a.env_steps() == b.env_steps()


@pytest.mark.parametrize('name,n,budgets,spt', [
    ('open_32x32', 10, (150, 1, 77), 200),
    ('two_rewards_16x32', 16, (97, 160, 3, 140), 60),
    ('near_goal_16x16', 24, (211, 190), 25),
    ('open_16x16', 12, (64, 64, 64, 64, 64), 40),
])
def test_wave_kernel_vs_oracle_and_row_streaming(torch_cuda, name, n, budgets, spt):
    torch = torch_cuda
    from oracle import c_oracle
    world = _worlds()[name]
    env, ag = _agent(torch, world, n, 31337, base=7)
    _launches(torch, env, ag, budgets, spt)
    env2, ref = _agent(torch, world, n, 31337, base=7, stream_rows=True)
    _launches(torch, env2, ref, budgets, spt)
    _same(torch, ag, ref)
    traffic = ag.traffic.cpu().numpy()
    steps = n * sum(budgets)
    assert traffic[1] == steps, 'one SR row written per env step'
    assert 0 < traffic[0] <= steps + int(ag.inst[:, 2].sum().item()) + n * len(budgets)
    assert traffic[3] == 0 and ref.traffic.sum().item() == 0
    # the goals were reached: the value gathers did real work for part of the run
    rw = ag._rw.cpu().numpy()
    found = int((rw != 0).any(axis=1).sum())
    need = {'near_goal_16x16': n // 2, 'two_rewards_16x32': 2, 'open_16x16': 2}.get(name, 1)
    assert found >= need, 'rewards were estimated by %d instances only' % found
    assert traffic[2] > 0

    o = c_oracle.SROracle(_oracle_world(world), n, env.seed, True, instance_base=7, epsilon=0.2,
                          trial_cap=64, occupancy=True)
    for b in budgets:
        o.run(0x7fffffff, spt, step_budget=b)
    assert np.array_equal(ag._T.cpu().numpy().astype(np.int64), o.T)
    assert np.array_equal(ag._rw.cpu().numpy().astype(np.float64), o.RW)
    assert np.array_equal(ag._sr.cpu().numpy().astype(np.float64), o.SR)
    inst = ag.inst.cpu().numpy()
    for col, key in ((0, 'state'), (1, 'step'), (2, 'trial'), (3, 'ctr_env'), (4, 'ctr_policy')):
        assert np.array_equal(inst[:, col], o.inst[key].astype(np.int32)), key
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy(), o.lat_trace)
    assert np.array_equal(ag.monitors.occupancy.cpu().numpy(), o.occupancy.astype(np.int64))
    assert np.array_equal(ag.monitors.lat_cnt.cpu().numpy(), o.lat_cnt.astype(np.int64))
    assert np.allclose(ag.monitors.reward_sum.cpu().numpy(), o.reward_sum, rtol=0, atol=1e-9)


def test_wave_kernel_masks_and_test_runs(torch_cuda):
    """Action masks (float64 selection path) while learning, then Agent.test() on the learned
    tables (no row traffic at all: values are gathers), both equal to the row-streaming kernel."""
    torch = torch_cuda
    world = _worlds()['near_goal_16x16']
    nxt = np.asarray(world['next'])
    mask = nxt != np.arange(256)[:, None]          # forbid moves that bump into the border
    mask[~mask.any(axis=1)] = True
    runs = []
    for stream_rows in (False, True):
        env, ag = _agent(torch, world, 20, 99, eps=0.3, stream_rows=stream_rows, mask=mask)
        _launches(torch, env, ag, (120, 120), 30)
        before = ag.traffic.clone()
        ag._env_out(env)
        ag._policy_out(ag.policy)
        ag.test(env, 5, 30)
        torch.cuda.synchronize()
        runs.append((ag, before))
    (a, before), (b, _) = runs
    _same(torch, a, b)
    moved = (a.traffic - before).cpu().numpy()
    assert moved[0] == 0 and moved[1] == 0 and moved[2] > 0, 'test runs read values only'
    occ = a.monitors.occupancy.cpu().numpy()[0]
    assert occ.sum() == a.env_steps()


def test_wave_kernel_hand_edited_rewards_take_full_row_sums(torch_cuda):
    """More than two non-zero reward estimates (only a caller's edit can produce them in a world
    with two rewarded states): the kernel evaluates NumPy's pairwise sums from memory; results
    equal the row-streaming kernel and the oracle started from the same estimates."""
    torch = torch_cuda
    from oracle import c_oracle
    world = _worlds()['two_rewards_16x32']
    n, budgets, spt = 9, (50, 41), 45
    rng = np.random.default_rng(5)
    edit = np.zeros((n, 512), dtype=np.float32)
    for i in range(n):
        k = rng.choice(512, size=3 + i, replace=False)
        edit[i, k] = rng.standard_normal(3 + i).astype(np.float32)
    edit[0] = 0
    edit[0, [3, 77]] = [0.5, -0.25]               # instance 0 stays on the sparse path
    out = []
    for stream_rows in (False, True):
        env, ag = _agent(torch, world, n, 2024, eps=0.1, stream_rows=stream_rows)
        ag._bind(env)
        ag._rw.copy_(torch.as_tensor(edit, device='cuda'))
        _launches(torch, env, ag, budgets, spt)
        out.append(ag)
    _same(torch, out[0], out[1])
    assert out[0].traffic[3].item() >= (n - 1) * len(budgets)
    o = c_oracle.SROracle(_oracle_world(world), n, 2024, True, epsilon=0.1, trial_cap=64)
    o.RW[:] = edit
    for b in budgets:
        o.run(0x7fffffff, spt, step_budget=b)
    assert np.array_equal(out[0]._sr.cpu().numpy().astype(np.float64), o.SR)
    assert np.array_equal(out[0]._rw.cpu().numpy().astype(np.float64), o.RW)
    assert np.array_equal(out[0]._T.cpu().numpy().astype(np.int64), o.T)


def test_wave_kernel_third_estimate_mid_launch(torch_cuda):
    """An instance that starts with two non-zero estimates and earns a third one during the launch
    switches to full row sums in the same step (no value may be computed from a stale set)."""
    torch = torch_cuda
    world = _worlds()['near_goal_16x16']          # rewards at 0 (terminal) and 20
    n = 16
    edit = np.zeros((n, 256), dtype=np.float32)
    edit[:, 200] = 0.75                            # a state that is NOT rewarded in the world
    out = []
    for stream_rows in (False, True):
        env, ag = _agent(torch, world, n, 4, eps=0.2, stream_rows=stream_rows)
        ag._bind(env)
        ag._rw.copy_(torch.as_tensor(edit, device='cuda'))
        _launches(torch, env, ag, (400,), 25)
        out.append(ag)
    _same(torch, out[0], out[1])
    nonzeros = (out[0]._rw != 0).sum(dim=1)
    assert int((nonzeros >= 3).sum().item()) >= n // 2
    assert out[0].traffic[3].item() > 0


def test_wave_kernel_param_sets(torch_cuda):
    """Per-instance learning rate / gamma / epsilon: instance i of a mixed launch equals instance
    i of a launch in which every instance uses i's combination (and the row-streaming kernel)."""
    torch = torch_cuda
    world = _worlds()['open_16x16']
    n = 24
    combos = [(0.1, 0.99, 0.1), (0.5, 0.9, 0.3), (0.05, 0.5, 0.0)]
    which = np.arange(n) % 3
    alpha = np.array([combos[k][0] for k in which])
    gamma = np.array([combos[k][1] for k in which])
    eps = np.array([combos[k][2] for k in which])
    env, mixed = _agent(torch, world, n, 8, eps=eps, alpha=alpha, gamma=gamma)
    _launches(torch, env, mixed, (90, 90), 40)
    env, mixed_rows = _agent(torch, world, n, 8, eps=eps, alpha=alpha, gamma=gamma,
                             stream_rows=True)
    _launches(torch, env, mixed_rows, (90, 90), 40)
    _same(torch, mixed, mixed_rows)
    for k, (a, g, e) in enumerate(combos):
        env, uni = _agent(torch, world, n, 8, eps=e, alpha=a, gamma=g)
        _launches(torch, env, uni, (90, 90), 40)
        sel = torch.as_tensor(np.flatnonzero(which == k), device='cuda')
        assert torch.equal(mixed._sr[sel], uni._sr[sel]), k
        assert torch.equal(mixed._T[sel], uni._T[sel]) and torch.equal(mixed.inst[sel], uni.inst[sel])


@pytest.mark.parametrize('h,w', [(2, 2), (4, 5), (10, 10), (12, 13), (20, 20), (24, 24), (30, 30),
                                 (1, 3), (5, 5), (7, 9), (15, 17), (3, 7), (17, 17), (23, 25), (25, 25)])
def test_wave_kernel_any_multiple_of_four_states(torch_cuda, h, w):
    """State counts other than 256 / 512 / 1 024 — multiples of four up to 1 024, anything below
    256 (rows moved element by element) — take the bounds-checked instantiations of k_sr_wave: identical tables, counters and monitors to the row-streaming
    kernel on the same run, and the kernel-counted traffic says which one ran."""
    torch = torch_cuda
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    S = h * w
    world = make_gridworld(h, w, terminals=[S - 1], rewards=np.array([[S - 1, 1.0], [1, -0.5]]),
                           goals=[S - 1], invalid_transitions=[(0, 1), (1, 0)] if w > 2 else [])

    def run(stream_rows):
        env = Gridworld(world, n_envs=96, seed=11, instance_base=5)
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.2))
        ag.track_instances = True
        ag.track_occupancy = True
        ag.stream_rows = stream_rows
        ag.train(env, 5, 40)
        ag.train(env, 2, 40)
        torch.cuda.synchronize()
        return ag, env

    (a, ea), (b, eb) = run(False), run(True)
    assert int(a.traffic[1]) > 0 and int(b.traffic[1]) == 0     # wave kernel / row-streaming kernel
    assert torch.equal(a._sr, b._sr) and torch.equal(a._T, b._T) and torch.equal(a._rw, b._rw)
    assert torch.equal(a.inst, b.inst) and torch.equal(ea.env_ctr, eb.env_ctr)
    assert torch.equal(a.monitors.lat_trace, b.monitors.lat_trace)
    assert torch.equal(a.monitors.occupancy, b.monitors.occupancy)


def test_wave_kernel_any_size_with_many_reward_estimates(torch_cuda):
    """More than two non-zero reward estimates (set by the caller; the world itself has two
    rewarded states) on sizes without the 128-element leaf layout: the kernel's pairwise sums (the
    leaves of NumPy's recursion spread over groups of eight lanes; 1 015 and 1 023 states have nine
    leaves and take the second pass) equal the row-streaming kernel's values, step for step."""
    torch = torch_cuda
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    for h, w in ((10, 10), (18, 22), (3, 4), (5, 5), (9, 11), (21, 27), (29, 35), (31, 33), (12, 43)):
        S = h * w
        world = make_gridworld(h, w, terminals=[S - 1], rewards=np.array([[S - 1, 1.0]]), goals=[S - 1])

        def run(stream_rows):
            env = Gridworld(world, n_envs=40, seed=3)
            ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
            ag.track_instances = True
            ag.stream_rows = stream_rows
            ag.train(env, 1, 3)                       # binds the tables
            gen = torch.Generator(device='cuda').manual_seed(5)
            ag._rw.copy_(torch.randn(ag._rw.shape, generator=gen, device='cuda') *
                         (torch.rand(ag._rw.shape, generator=gen, device='cuda') < 0.3))
            ag._sr.add_(torch.rand(ag._sr.shape, generator=gen, device='cuda') * 0.01)
            ag.train(env, 4, 30)
            torch.cuda.synchronize()
            return ag

        a, b = run(False), run(True)
        assert int(a.traffic[3]) > 0                  # the dense path ran in the wave kernel
        assert torch.equal(a._sr, b._sr) and torch.equal(a._T, b._T) and torch.equal(a._rw, b._rw)
        assert torch.equal(a.inst, b.inst)
        assert torch.equal(a.monitors.lat_trace, b.monitors.lat_trace)


def _multi_reward_worlds():
    from cobel_amd.misc.gridworld_tools import make_gridworld
    four = make_gridworld(32, 32, terminals=[0, 1023], goals=[0],
                          rewards=np.array([[0, 1.0], [1023, 0.5], [517, 0.25], [130, -0.125]]),
                          starting_states=[33, 500, 700, 990, 520, 131])
    eight = make_gridworld(32, 32, terminals=[5], goals=[5],
                           rewards=np.array([[5, 1.0], [6, 0.5], [37, -0.25], [38, 0.125], [100, 2.0],
                                             [255, 0.75], [256, -1.5], [900, 0.3]]),
                           starting_states=[4, 7, 36, 39, 69, 101, 257])
    three_odd = make_gridworld(13, 17, terminals=[0], goals=[0],       # 221 states: rows move by element
                               rewards=np.array([[0, 1.0], [1, 0.5], [18, 0.25]]),
                               starting_states=[2, 19, 35, 36])
    five_any = make_gridworld(20, 24, terminals=[10], goals=[10],      # 480 states: float4 rows, bounds
                              rewards=np.array([[10, 1.0], [11, 0.5], [34, 0.25], [35, -0.5], [58, 0.1]]),
                              starting_states=[9, 12, 33, 36, 59, 82])
    # 961 and 841 states: rows that are neither 16-byte aligned nor short (lane-strided registers)
    one_31 = make_gridworld(31, 31, terminals=[0], goals=[0], rewards=np.array([[0, 1.0]]),
                            starting_states=[1, 31, 32, 63, 500])
    three_29 = make_gridworld(29, 29, terminals=[5], goals=[5],
                              rewards=np.array([[5, 1.0], [6, -0.5], [34, 0.25]]),
                              starting_states=[4, 7, 33, 35, 64])
    # nine to 32 rewarded states (the 32-slot form of the sparse-reward kernel): clusters around the
    # starts so that a short run meets many of them
    def cluster(w, cells, term):
        vals = [1.0, 0.5, -0.25, 0.125, 2.0, 0.75, -1.5, 0.3, 0.0625, -0.5, 1.25, 0.2]
        return np.array([[c, 3.0 if c == term else vals[k % len(vals)]] for k, c in enumerate(cells)])
    c16 = [5, 6, 37, 38, 70, 100, 101, 133, 165, 166, 198, 230, 255, 256, 600, 900]
    sixteen = make_gridworld(32, 32, terminals=[5], goals=[5], rewards=cluster(32, c16, 5),
                             starting_states=[4, 7, 36, 39, 69, 102, 134, 167, 199])
    c32 = sorted(set(c16 + [8, 9, 40, 41, 72, 73, 104, 136, 168, 200, 231, 232, 263, 264, 295, 1000]))
    thirtytwo = make_gridworld(32, 32, terminals=[5], goals=[5], rewards=cluster(32, c32, 5),
                               starting_states=[4, 7, 36, 39, 69, 102, 134, 167, 199, 10, 42])
    c12 = [0, 1, 18, 19, 35, 36, 52, 53, 69, 70, 100, 220]
    twelve_odd = make_gridworld(13, 17, terminals=[0], goals=[0], rewards=cluster(17, c12, 0),
                                starting_states=[2, 20, 37, 54, 71])
    c20 = [10, 11, 34, 35, 58, 59, 82, 83, 106, 107, 130, 131, 154, 155, 178, 179, 202, 203, 300, 479]
    twenty_any = make_gridworld(20, 24, terminals=[10], goals=[10], rewards=cluster(24, c20, 10),
                                starting_states=[9, 12, 33, 36, 60, 84, 108, 132])
    return {'four_32x32': four, 'eight_32x32': eight, 'three_13x17': three_odd,
            'five_20x24': five_any, 'one_31x31': one_31, 'three_29x29': three_29,
            'sixteen_32x32': sixteen, 'thirtytwo_32x32': thirtytwo, 'twelve_13x17': twelve_odd,
            'twenty_20x24': twenty_any}


@pytest.mark.parametrize('name,n,budgets,spt', [
    ('four_32x32', 12, (180, 33, 120), 120),
    ('eight_32x32', 20, (150, 150, 7), 40),
    ('three_13x17', 16, (90, 111), 30),
    ('five_20x24', 16, (128, 128), 35),
    ('one_31x31', 10, (100, 77), 40),
    ('three_29x29', 12, (90, 90), 30),
    ('sixteen_32x32', 16, (150, 150, 9), 40),
    ('thirtytwo_32x32', 16, (160, 140), 40),
    ('twelve_13x17', 16, (90, 111), 30),
    ('twenty_20x24', 16, (128, 128), 35),
])
def test_three_to_eight_rewarded_states_take_the_wave_kernel(torch_cuda, name, n, budgets, spt):
    """Worlds with three to 32 rewarded states (make_gridworld(rewards=...)): the sparse-reward
    kernel adds the up-to-eight (one lane each in groups of eight) or up-to-32 (two per lane in
    groups of sixteen) products of a value row in NumPy's pairwise grouping
    (cobel_pairwise_order) — SR, transition tables, reward estimates, counters and monitors equal
    the row-streaming kernel's and the C oracle's full row sums, bit for bit."""
    torch = torch_cuda
    from oracle import c_oracle
    world = _multi_reward_worlds()[name]
    env, ag = _agent(torch, world, n, 777, base=3, eps=0.25)
    _launches(torch, env, ag, budgets, spt)
    env2, ref = _agent(torch, world, n, 777, base=3, eps=0.25, stream_rows=True)
    _launches(torch, env2, ref, budgets, spt)
    _same(torch, ag, ref)
    traffic = ag.traffic.cpu().numpy()
    assert traffic[1] == n * sum(budgets) and traffic[3] == 0, 'the wave kernel ran, sparse path'
    assert ref.traffic.sum().item() == 0
    rw = ag._rw.cpu().numpy()
    if name != 'one_31x31':
        assert int(((rw != 0).sum(axis=1) >= 2).sum()) >= 2, 'several estimates became non-zero'
    else:
        assert (rw != 0).any()
    o = c_oracle.SROracle(_oracle_world(world), n, env.seed, True, instance_base=3, epsilon=0.25,
                          trial_cap=64, occupancy=True)
    for b in budgets:
        o.run(0x7fffffff, spt, step_budget=b)
    assert np.array_equal(ag._T.cpu().numpy().astype(np.int64), o.T)
    assert np.array_equal(ag._rw.cpu().numpy().astype(np.float64), o.RW)
    assert np.array_equal(ag._sr.cpu().numpy().astype(np.float64), o.SR)
    assert np.array_equal(ag.monitors.lat_trace.cpu().numpy(), o.lat_trace)


def test_multi_reward_world_with_a_foreign_estimate_falls_back_to_full_sums(torch_cuda):
    """A non-zero estimate at a state the world does not reward (a caller's edit): full pairwise
    sums from memory, equal to the row-streaming kernel."""
    torch = torch_cuda
    world = _multi_reward_worlds()['four_32x32']
    n = 6
    edit = np.zeros((n, 1024), dtype=np.float32)
    edit[1:, 444] = 0.5
    edit[:, 517] = 0.25
    out = []
    for stream_rows in (False, True):
        env, ag = _agent(torch, world, n, 11, eps=0.2, stream_rows=stream_rows)
        ag._bind(env)
        ag._rw.copy_(torch.as_tensor(edit, device='cuda'))
        _launches(torch, env, ag, (60, 45), 50)
        out.append(ag)
    _same(torch, out[0], out[1])
    assert out[0].traffic[3].item() == (n - 1) * 2, 'five instances on the dense path, per launch'
