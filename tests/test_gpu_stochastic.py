"""Worlds whose transition rows are distributions — the reference's Gridworld.step then DRAWS the
successor from sas[s][a] (interface/gridworld.py:119-123) — through cobel_world_set_transitions,
cobel_env_step_draw and the general kernel of cobel_tab_run, against golden runs of the reference
on slippery worlds (tests/golden/gen_golden.py gen_stochastic) and the Philox restatement."""
import numpy as np
import pytest

from conftest import SEED, as_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def Z(golden):
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return golden('stochastic_traces')


def slippery_world(Z, wname):
    tab = {k: Z['%s/%s' % (wname, k)] for k in ('height', 'width', 'next', 'reward', 'terminal',
                                                'starts', 'coordinates')}
    w = as_world(tab)
    w['sas'] = np.array(Z[wname + '/sas'])
    w['deterministic'] = False
    return w


@pytest.mark.parametrize('wname', ['slip_4x4', 'slip_5x6_wind'])
def test_env_step_draws_the_successor(Z, wname):
    """Gridworld.step on 4 096 instances: every successor is the one Generator.choice returns for
    the uniform of the instance's env stream (oracle/philox.py), counters advance by one per step
    and by one per reset, rewards / ends belong to the drawn state."""
    import torch
    from cobel_amd.interface import Gridworld
    from oracle import philox
    world = slippery_world(Z, wname)
    sas = np.asarray(world['sas'])
    n, base = 4096, 77
    env = Gridworld(world, n_envs=n, seed=SEED, instance_base=base)
    assert env.handle.stochastic
    rng = np.random.default_rng(1)
    inst = base + np.arange(n)
    moved = 0
    for it in range(6):
        before, ctr = env.state.cpu().numpy(), env.env_ctr.cpu().numpy().astype(np.int64)
        act = rng.integers(0, 4, n).astype(np.uint8)
        _, r, done, _, _ = env.step(torch.as_tensor(act, device='cuda'))
        after = env.state.cpu().numpy()
        u = philox.draw_double(SEED, inst, ctr, 1, philox.STREAM_ENV)
        cdf = np.cumsum(sas[before, act], axis=1)
        cdf /= cdf[:, -1:]
        want = (cdf <= u[:, None]).sum(axis=1)      # searchsorted(u, side='right') per row
        assert np.array_equal(after, want)
        assert np.array_equal(env.env_ctr.cpu().numpy(), ctr + 1)
        assert np.array_equal(r.cpu().numpy(), np.asarray(world['rewards'], dtype=np.float32)[after])
        assert np.array_equal(done.cpu().numpy(), np.asarray(world['terminals'])[after] != 0)
        moved += int((after != np.asarray(world['next'])[before, act]).sum())
        env.reset(done)
    assert moved > 0.05 * 6 * n                      # the walk is not the one of the argmax table


@pytest.mark.parametrize('name', ['slip4_dynaq_b8', 'slip4_q_b4', 'slip4_q_b0', 'slip56_dynaq_b70'])
def test_stochastic_world_runs_match_reference(Z, name):
    """Dyna-Q / Q-learning on the slippery worlds: per-step logs of a single instance, escape
    latencies, tables and the env stream's draw count against the reference's float32 run; the same
    instance inside a vectorised launch; cobel_tab_run takes its general kernel."""
    import torch
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    inst, trials, steps, B = [int(x) for x in Z[name + '/cfg']]
    world = slippery_world(Z, str(Z[name + '/world']))
    dyna = str(Z[name + '/agent']) == 'dynaq'
    cls = DynaQ if dyna else QAgent
    g = lambda k: Z['%s/%s' % (name, k)]     # noqa: E731

    def check_tables(ag, i):
        q = ag._q[i].cpu().numpy().astype(np.float64)
        assert np.array_equal(q, g('Q'))
        if dyna:
            M = ag.M
            sq = (lambda a: np.asarray(a) if ag.n_envs == 1 else np.asarray(a)[i])
            assert np.array_equal(sq(M.states), g('M_states'))
            assert np.array_equal(sq(M.terminals), g('M_terminals'))
            assert np.array_equal(np.asarray(sq(M.rewards), dtype=np.float64), g('M_rewards'))
        else:
            assert int(ag.inst[i, _lib.I_LOG_LEN].item()) == int(g('log_len'))

    # one instance, the reference's callbacks
    sarsn, tds, steps_log = [], [], []
    cbs = {'on_step_end': [lambda l: (sarsn.append((l['state'], l['action'], l['reward'],
                                                    l['next_state'], l['terminal'])),
                                      tds.append(l['td']))],
           'on_trial_end': [lambda l: steps_log.append(l['steps'])]}
    env = Gridworld(world, seed=SEED, instance_base=inst)
    ag = cls(env.observation_space, env.action_space, EpsilonGreedy(0.1), custom_callbacks=cbs)
    ag.train(env, trials, steps, B)
    arr = np.array(sarsn, dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(arr[:, col], g(key)), key
    assert np.array_equal(steps_log, g('steps'))
    if dyna or B == 0:
        assert np.array_equal(np.array(tds, dtype=np.float64), g('td'))
    check_tables(ag, 0)
    assert int(env.env_ctr[0].item()) == int(g('env_draws'))
    # vectorised: instance `inst` of one launch; the dispatcher names the generic wavefront kernel
    # (no replayed updates, or QAgent's log replay beyond the wavefront kernels' batch limit: the
    #  lane-per-instance general kernel)
    env = Gridworld(world, n_envs=inst + 70, seed=SEED)
    ag = cls(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    ag.track_instances = True
    ag.train(env, trials, steps, B)
    torch.cuda.synchronize()
    assert np.array_equal(ag.monitors.lat_trace[inst].cpu().numpy()[:trials], g('steps'))
    check_tables(ag, inst)
    assert int(env.env_ctr[inst].item()) == int(g('env_draws'))
    what = ag.describe_launch(env, ag.policy, _lib.F_LEARN, trials, steps, 0, B)
    assert what['kernel'] == (_lib.TAB_KERNEL_GENERAL if (B == 0 or (not dyna and B > _lib.MAX_BATCH))
                              else _lib.TAB_KERNEL_WPI)


@pytest.mark.parametrize('name', ['slip4_sr', 'slip56_sr'])
def test_sr_on_stochastic_worlds_matches_reference(Z, name):
    """SR.train only calls interface.step (agent/sr.py:170-182), so on a world whose rows are
    distributions the successor is drawn (gridworld.py:119-123): SR matrix, learned transitions,
    reward estimates, escape latencies and the env stream's draw count of the reference's float32
    run — one instance alone with per-step logs, and inside a vectorised launch with visit counts."""
    import torch
    from cobel_amd.agent import SR
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    inst, trials, steps, _ = [int(x) for x in Z[name + '/cfg']]
    world = slippery_world(Z, str(Z[name + '/world']))
    g = lambda k: Z['%s/%s' % (name, k)]     # noqa: E731
    sarsn, steps_log = [], []
    cbs = {'on_step_end': [lambda l: sarsn.append((l['state'], l['action'], l['reward'],
                                                   l['next_state'], l['terminal']))],
           'on_trial_end': [lambda l: steps_log.append(l['steps'])]}
    env = Gridworld(world, seed=SEED, instance_base=inst)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1), custom_callbacks=cbs)
    ag.train(env, trials, steps)
    arr = np.array(sarsn, dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(arr[:, col], g(key)), key
    assert np.array_equal(steps_log, g('steps'))
    assert int(env.env_ctr[0].item()) == int(g('env_draws'))

    def tables(a, i):
        assert np.array_equal(a._T[i].cpu().numpy(), g('T'))
        assert np.array_equal(a._rw[i].cpu().numpy().astype(np.float64), g('rewards'))
        assert np.array_equal(a._sr[i].cpu().numpy().astype(np.float64), g('SR'))
    tables(ag, 0)
    env = Gridworld(world, n_envs=inst + 40, seed=SEED)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    ag.track_instances = True
    ag.track_occupancy = True
    ag.train(env, trials, steps)
    torch.cuda.synchronize()
    assert np.array_equal(ag.monitors.lat_trace[inst].cpu().numpy()[:trials], g('steps'))
    tables(ag, inst)
    assert int(env.env_ctr[inst].item()) == int(g('env_draws'))
    assert int(ag.monitors.occupancy.sum().item()) == ag.env_steps()


@pytest.mark.parametrize('opts,mode,B', [({}, 'reverse', 16), ({'recency': True, 'dynamic': True}, 'default', 24)])
def test_sfma_on_a_stochastic_world_vs_oracle(Z, opts, mode, B):
    """SFMA.train steps the interface (agent/sfma.py:262-264): on a world of distributions the
    successor is drawn.  Latencies, every reactivation, Q and the memory's tables against the NumPy
    restatement of the reference's SFMA (oracle/sfma_loop.py) walking the same slippery world
    (its env is the restatement pinned by the stochastic Dyna-Q / SR fixtures above)."""
    import test_gpu_sfma as ts
    from cobel_amd.memory.utils import SR as SRMetric
    from oracle import sfma_loop
    world = slippery_world(Z, 'slip_5x6_wind')
    D = SRMetric(np.asarray(world['next']), 0.9).D
    o = dict(opts, mode=mode)
    env, agent = ts.build(world, D, o, 24, 500, made=True)
    assert env.handle.stochastic
    trials, steps = 6, 30
    ts.run_schedule(env, agent, o, trials, steps, B)
    ow = dict(next=np.asarray(world['next']), reward=np.asarray(world['rewards'], dtype=np.float64),
              terminal=np.asarray(world['terminals']), starts=np.asarray(world['starting_states']),
              sas=np.asarray(world['sas']))
    moved = 0
    for i in (0, 7, 23):
        ag, oenv = sfma_loop.run_case(ow, D, SEED, 500 + i, True, mode, o, trials, steps, B)
        assert np.array_equal(agent.monitors.lat_trace[i].cpu().numpy()[:trials], ag.steps), i
        rp = np.array(ag.replayed, dtype=np.float64).reshape(-1, 8)
        ts.check_events(ts.events_of(agent, i), rp)
        assert np.array_equal(agent.Q[i].cpu().numpy(), ag.Q), i
        assert np.array_equal(agent.M.states[i], ag.M.states), i
        assert np.array_equal(agent.M.C[i], ag.M.C), i
        assert int(env.env_ctr[i].item()) == oenv.rng.index
        moved += int((np.asarray(ag.M.states) != np.asarray(world['next'])).sum())
    assert moved > 0         # (the model learned successors the argmax table does not hold)


def test_stochastic_worlds_and_the_other_entry_points(Z):
    """The kernels that step transition tables refuse a world of distributions loudly; the network
    agents run it through their PyTorch loops (env.step draws); a world of the same rows with the
    reference's `deterministic` flag SET is the argmax table again."""
    import torch
    import bench
    from cobel_amd import _lib
    from cobel_amd.agent import SR, DynaDQN, DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    world = slippery_world(Z, 'slip_4x4')
    env = Gridworld(world, n_envs=8, seed=3)
    sr = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    sr.train(env, 1, 5)          # (served since round 4: test_sr_on_stochastic_worlds below)
    state = torch.zeros(8, dtype=torch.int32, device='cuda')
    act = torch.zeros(8, dtype=torch.uint8, device='cuda')
    with pytest.raises(NotImplementedError):     # the counter-less entry cannot draw
        _lib.check(_lib.lib().cobel_env_step(env.handle.ptr, _lib.ptr(state), _lib.ptr(act), None,
                                             None, 8, 0, None))
    torch.manual_seed(2)
    env = Gridworld(world, n_envs=6, seed=4)
    ag = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(bench._mlp(16, 4)), gamma=0.9)
    ag.train(env, 2, 8, 32)
    assert ag.fused_steps == 0 and int(env.env_ctr.min().item()) > 2     # steps drew doubles
    # flag set: the reference takes the argmax of every row, so does the build
    det = slippery_world(Z, 'slip_4x4')
    det['deterministic'] = True
    env = Gridworld(det, n_envs=64, seed=5)
    assert not env.handle.stochastic
    ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(1.0))
    ag.train(env, 2, 20, 8)
    M = ag.M
    seen = np.asarray(M.terminals) != 0
    assert (np.asarray(M.states)[seen] == np.broadcast_to(np.asarray(det['next']), (64, 16, 4))[seen]).all()
