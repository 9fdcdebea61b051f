"""The persistent-workgroup Dyna-Q kernel (csrc/tabular_pwg.hip: sixteen wavefronts per CU, part of
them with Q in LDS, the others working on Q in global memory) against the one-workgroup-per-instance
kernel it replaces on large worlds, and against the NumPy oracle: same tables, digests, counters,
monitors — bit for bit, whichever kind of wave an instance happens to get."""
import numpy as np
import pytest
import torch

from conftest import SEED

pytestmark = pytest.mark.gpu


def _run(extra_flags, n, side, seeds, launches, steps, batch=50, spt=60, alpha=0.99, eps=0.1):
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze
    from cobel_amd.policy import EpsilonGreedy
    worlds = [make_obstacle_maze(side, side, s) for s in seeds]
    env = Gridworld(worlds, n_envs=n, seed=SEED, device=torch.device('cuda', 0))
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(eps), learning_rate=alpha)
    agent.extra_flags = extra_flags
    agent.track_instances = True
    agent._bind(env)
    agent._env_in(env)
    flags = _lib.F_LEARN | agent._policy_in(agent.policy, env, False)
    agent.monitors.reserve(2048, n, True)
    kinds = set()
    for _ in range(launches):
        kinds.add(agent.describe_launch(env, agent.policy, flags, 0x7fffffff, spt, steps, batch)['kernel'])
        agent._launch(env, agent.policy, flags, 0x7fffffff, spt, steps, batch)
    torch.cuda.synchronize()
    mon = agent.monitors
    return {'q': agent._q.cpu().numpy(), 'model': agent.M.table.cpu().numpy(),
            'index': agent.M.index.cpu().numpy(), 'inst': agent.inst.cpu().numpy(),
            'lat_sum': mon.lat_sum.cpu().numpy(), 'lat_cnt': mon.lat_cnt.cpu().numpy(),
            'reward_sum': mon.reward_sum.cpu().numpy(), 'lat_trace': mon.lat_trace.cpu().numpy(),
            'batches': int(agent.batches_done.item()), 'steps': agent.env_steps(), 'kinds': kinds,
            'worlds': worlds}


def _same(a, b):
    for key in ('q', 'model', 'index', 'inst', 'lat_sum', 'lat_cnt', 'reward_sum', 'lat_trace'):
        assert np.array_equal(a[key], b[key]), key
    assert a['batches'] == b['batches'] and a['steps'] == b['steps']


@pytest.mark.parametrize('side,n', [(32, 700), (20, 300), (31, 130)])
def test_persistent_workgroups_equal_the_wave_per_instance_kernel(side, n):
    from cobel_amd import _lib
    seeds = [1234, 1235, 1236]
    args = dict(n=n, side=side, seeds=seeds, launches=3, steps=700)
    plain = _run(_lib.F_NO_PWG, **args)
    mixed = _run(0, **args)
    glob = _run(_lib.F_PWG_GLOBAL, **args)
    assert plain['kinds'] == {_lib.TAB_KERNEL_WPI_INDEX}
    assert glob['kinds'] == {_lib.TAB_KERNEL_PWG}
    # (20 x 20: sixteen Q tables fit in LDS, the plain kernel stays)
    assert mixed['kinds'] == ({_lib.TAB_KERNEL_PWG} if side > 24 else {_lib.TAB_KERNEL_WPI_INDEX})
    assert plain['q'].any() and plain['batches'] > 0 and plain['lat_cnt'].sum() > 0
    _same(plain, mixed)
    _same(plain, glob)


def test_global_q_waves_against_the_oracle():
    """Every wave with Q in global memory, young and trained instances, a small learning rate so
    that most planning updates change their cell (many speculative rounds)."""
    from cobel_amd import _lib
    from oracle import philox, ref_loop
    n, side, spt, steps, batch = 40, 32, 40, 300, 62
    out = _run(_lib.F_PWG_GLOBAL, n=n, side=side, seeds=[77], launches=2, steps=steps, batch=batch,
               spt=spt, alpha=0.5, eps=0.3)
    assert out['kinds'] == {_lib.TAB_KERNEL_PWG}
    tabs = out['worlds'][0].compact()
    for i in (0, 7, 39):
        env = ref_loop.RefGridworld(tabs, philox.TapeRNG(SEED, i, philox.STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(0.3, philox.TapeRNG(SEED, i, philox.STREAM_POLICY))
        ref = ref_loop.RefDynaQ(side * side, 4, pol, philox.TapeRNG(SEED, i, philox.STREAM_MEMORY),
                                dtype=np.float32, learning_rate=0.5)
        trials = int(out['inst'][i, 2])
        tr = ref_loop.new_trace()
        ref.train(env, trials, spt, batch, trace=tr)
        done = sum(s + 1 for s in tr['steps'])
        # the oracle runs whole trials: continue into the running one for the remaining steps
        left = 2 * steps - done
        assert 0 <= left <= spt
        if left:
            ref.train(env, 1, left, batch, trace=tr)
        assert np.array_equal(out['q'][i], ref.Q), i


def _sliced(monkeypatch, slices, limit=None, whole=None, **args):
    monkeypatch.setenv('COBEL_DEBUG', '1')
    if whole is not None:
        monkeypatch.setenv('COBEL_DEBUG_PWG_WHOLE', str(whole))
    if slices is not None:
        monkeypatch.setenv('COBEL_DEBUG_PWG_SLICES', slices)
    if limit is not None:
        monkeypatch.setenv('COBEL_DEBUG_PWG_XCCLIMIT', str(limit))
    try:
        return _run(0, **args)
    finally:
        monkeypatch.delenv('COBEL_DEBUG_PWG_SLICES', raising=False)
        monkeypatch.delenv('COBEL_DEBUG_PWG_XCCLIMIT', raising=False)
        monkeypatch.delenv('COBEL_DEBUG_PWG_WHOLE', raising=False)
        monkeypatch.delenv('COBEL_DEBUG', raising=False)


def test_slices_of_an_instances_steps_equal_whole_instances(monkeypatch):
    """A launch of few instances per wave slot is handed out as slices of an instance's steps
    (tickets, csrc/tabular_pwg.hip): the slices of an instance run in order on whichever waves of
    its XCD draw them, and leave exactly what one wave running the whole launch leaves."""
    from cobel_amd import _lib
    # (one workgroup per CU needs >= 256 x 13 instances before the launch is sliced at all)
    args = dict(n=3400, side=32, seeds=[1234, 1235, 1236, 1237, 1238], launches=2, steps=512, spt=90)
    plain = _run(_lib.F_NO_PWG, **args)
    auto = _run(0, **args)                                     # 320 + 128 + 64 by the plan
    odd = _sliced(monkeypatch, '300,7,100,64,41', **args)      # uneven slices
    one = _sliced(monkeypatch, '512', **args)                  # whole instances
    # (a queue's first instances whole, on the global-memory waves — 425 = all of them, more than those
    #  waves get through: the LDS waves take what is left —, the rest in slices)
    mixed = [_sliced(monkeypatch, '300,7,100,64,41', whole=w, **args)
             for w in (1, 60, 150, 424, 425, 0x80000000 + 30, 0x80000000 + 200)]   # (flag: ... then slices)
    assert auto['kinds'] == {_lib.TAB_KERNEL_PWG} and plain['kinds'] == {_lib.TAB_KERNEL_WPI_INDEX}
    assert plain['q'].any() and plain['batches'] > 0 and plain['lat_cnt'].sum() > 0
    _same(plain, auto)
    _same(plain, odd)
    _same(plain, one)
    for m in mixed:
        _same(plain, m)


def test_whole_instances_for_the_global_memory_waves_by_the_plan():
    """1.5 .. 3.5 instances per wave slot (the shard an eight-way split of C3 leaves a GPU): by the
    plan each global-memory wave runs one whole instance and the LDS waves share the rest in slices
    (plan_slices, csrc/tabular_pwg.hip).  Same results as one wave per instance."""
    from cobel_amd import _lib
    args = dict(n=7001, side=32, seeds=[1234, 1235, 1236], launches=2, steps=256, spt=60)
    plain = _run(_lib.F_NO_PWG, **args)
    auto = _run(0, **args)
    assert auto['kinds'] == {_lib.TAB_KERNEL_PWG} and plain['batches'] > 0
    _same(plain, auto)


def test_queues_without_an_xcd_of_their_own_are_claimed(monkeypatch):
    """Sliced launches bind a queue to one XCD (its L2 carries an instance from slice to slice).
    With the workgroups of XCDs 4 ... 7 (or of all but XCD 0) sitting the launch out, their queues
    have no XCD of their own: the XCDs that finish their queue claim them."""
    from cobel_amd import _lib
    args = dict(n=3400, side=32, seeds=[77, 78], launches=1, steps=256, spt=70)
    plain = _run(_lib.F_NO_PWG, **args)
    masked = _sliced(monkeypatch, '128,64,64', limit=3, **args)
    single = _sliced(monkeypatch, '128,64,64', limit=0, **args)
    _same(plain, masked)
    _same(plain, single)


def test_experiment_variables_need_the_master_switch(monkeypatch):
    """A stray COBEL_DEBUG_* variable in a production environment changes nothing: the library
    reads them only under COBEL_DEBUG=1 (the suite itself runs without it, tests/conftest.py)."""
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld([make_obstacle_maze(32, 32, 1234)], n_envs=700, seed=SEED,
                    device=torch.device('cuda', 0))
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    agent._bind(env)
    agent._env_in(env)
    flags = _lib.F_LEARN | agent._policy_in(agent.policy, env, False)
    agent.monitors.reserve(64, 700, False)

    def describe():
        return agent.describe_launch(env, agent.policy, flags, 0x7fffffff, 60, 64, 50)

    assert 'COBEL_DEBUG' not in __import__('os').environ
    plain = describe()
    assert plain['kernel'] == _lib.TAB_KERNEL_PWG
    monkeypatch.setenv('COBEL_DEBUG_LDS_PAD', '1024')
    monkeypatch.setenv('COBEL_DEBUG_PWG_SLICES', '1,63')
    assert describe() == plain                       # ignored without the master switch
    monkeypatch.setenv('COBEL_DEBUG', '1')
    padded = describe()                              # (an occupancy experiment: the padded kernel)
    assert padded['kernel'] == _lib.TAB_KERNEL_WPI_INDEX and padded != plain
    monkeypatch.setenv('COBEL_DEBUG', 'yes')         # only exactly "1" switches them on
    assert describe() == plain


def test_scratch_check_reports_a_raised_abort_word():
    """cobel_tab_scratch_check: COBEL_OK on the area of a finished run, COBEL_E_HIP (CobelHipError)
    once the abort word — what a sliced launch raises when a wave gives up waiting for a ring entry
    — is set; a NULL / short area is never an error."""
    from cobel_amd import _lib
    out_agent = {}

    def run():
        from cobel_amd.agent import DynaQ
        from cobel_amd.interface import Gridworld
        from cobel_amd.misc.gridworld_tools import make_obstacle_maze
        from cobel_amd.policy import EpsilonGreedy
        env = Gridworld([make_obstacle_maze(32, 32, 1234)], n_envs=3400, seed=SEED,
                        device=torch.device('cuda', 0))
        agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        agent.train(env, 1, 40, 50)       # (3 400 instances: a sliced launch of the PWG kernel)
        out_agent['a'] = agent
        return agent

    agent = run()
    assert 0 < agent.env_steps() <= 3400 * 40      # (env_steps() checks the area on the way)
    agent.check_launches()
    lib = _lib.lib()
    st = _lib.current_stream(agent.device)
    assert lib.cobel_tab_scratch_check(None, 0, st) == _lib.OK
    assert lib.cobel_tab_scratch_check(_lib.ptr(agent._scratch), 16, st) == _lib.OK
    agent._scratch[255] = 1                        # COBEL_TAB_SCRATCH_ABORT_WORD
    with pytest.raises(Exception) as err:
        agent.check_launches()
    assert 'sliced launch' in str(err.value)
    agent._scratch[255] = 0
    agent.check_launches()


@pytest.mark.parametrize('side,flags_name', [(8, None), (32, None), (32, 'F_NO_PWG')])
def test_tables_full_of_tag_patterns_end_the_batch(side, flags_name):
    """The dependency test of the planning batch raises cells to tags 0xffffffc0 .. 0xffffffff — NaN
    payloads no arithmetic produces.  A caller's table that HOLDS such patterns (uninitialised
    memory, 0xff fill) is garbage in, garbage out, but it must not hold a lane back for ever: the
    first lane of a round is committed regardless, so every batch ends (k_tab_wpi at 8 x 8,
    k_tab_pwg and k_tab_wpi at 32 x 32)."""
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld([make_obstacle_maze(side, side, 1234)], n_envs=96, seed=SEED,
                    device=torch.device('cuda', 0))
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    if flags_name:
        agent.extra_flags = getattr(_lib, flags_name)
    agent._bind(env)
    agent._q.view(torch.int32).fill_(-1)                       # every cell 0xffffffff
    agent._q.view(torch.int32)[:, ::3, :] = -64                # ... or 0xffffffc0
    agent.train(env, 2, 30, 50)
    torch.cuda.synchronize()
    assert 0 < agent.env_steps() <= 96 * 60
