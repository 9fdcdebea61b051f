"""The boundary's stream and thread contract (include/cobel_hip.h): every entry point
is ordered on the `hipStream_t` it is handed and keeps no state outside the caller's buffers, its
handles and the launch's scratch area — sessions on different streams, driven by different host
threads, interleave freely and leave what they leave alone; an (immutable) world handle may serve
several sessions at once.  The sessions here are the two the code would notice it on: a SLICED
`k_tab_pwg` launch (the shard an eight-way split of C3 leaves a GPU: tickets, rings and `owner`
claims in the scratch area, csrc/tabular_pwg.hip) and `k_sr_wave`; each is compared bit for bit
with the same session run alone on the default stream, and spot-checked against the oracle."""
import threading

import numpy as np
import pytest
import torch

from conftest import SEED

pytestmark = pytest.mark.gpu


def _dynaq_session(n, launches, steps, handle_of=None, seeds=(1234, 1235, 1236, 1237)):
    """`launches` launches of plain Dyna-Q training on 32 x 32 mazes on the CURRENT stream."""
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze
    from cobel_amd.policy import EpsilonGreedy
    worlds = [make_obstacle_maze(32, 32, s) for s in seeds]
    env = Gridworld(worlds, n_envs=n, seed=SEED, device=torch.device('cuda', 0))
    if handle_of is not None:
        env.handle = handle_of.handle      # (the same device tables: one handle, two sessions)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1), learning_rate=0.99)
    agent.track_instances = True
    agent._bind(env)
    agent._env_in(env)
    flags = _lib.F_LEARN | agent._policy_in(agent.policy, env, False)
    agent.monitors.reserve(2048, n, True)
    kinds = set()
    for _ in range(launches):
        kinds.add(agent.describe_launch(env, agent.policy, flags, 0x7fffffff, 90, steps, 50)['kernel'])
        agent._launch(env, agent.policy, flags, 0x7fffffff, 90, steps, 50)
    return env, agent, kinds, worlds


def _dynaq_tables(agent):
    mon = agent.monitors
    return {'q': agent._q.cpu().numpy(), 'model': agent.M.table.cpu().numpy(),
            'index': agent.M.index.cpu().numpy(), 'inst': agent.inst.cpu().numpy(),
            'lat_sum': mon.lat_sum.cpu().numpy(), 'lat_cnt': mon.lat_cnt.cpu().numpy(),
            'reward_sum': mon.reward_sum.cpu().numpy(), 'lat_trace': mon.lat_trace.cpu().numpy()}


def _sr_session(n, launches, steps):
    """`launches` launches of SR training on the open 32 x 32 field (config C4's world)."""
    import bench
    cfg = dict(bench.CONFIGS['C4'], instances=n, env_steps_per_launch=steps)
    env, agent = bench.build_agent('C4', cfg, n, 0, torch.device('cuda', 0))
    runner = bench.Runner(cfg, env, agent)
    for _ in range(launches):
        runner.launch()
    return env, agent


def _sr_tables(agent):
    return {'sr': agent._sr.cpu().numpy(), 'T': agent._T.cpu().numpy(), 'rw': agent._rw.cpu().numpy(),
            'inst': agent.inst.cpu().numpy()}


def _same(a, b):
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key


N_DQ, L_DQ, STEPS_DQ = 8192, 3, 256       # (3.2 instances per wave slot: a sliced launch)
N_SR, L_SR, STEPS_SR = 96, 6, 48


def test_two_threads_two_streams_leave_what_each_leaves_alone():
    from cobel_amd import _lib
    # each session alone, on the default stream
    _, a0, kinds, _ = _dynaq_session(N_DQ, L_DQ, STEPS_DQ)
    torch.cuda.synchronize()
    alone_dq = _dynaq_tables(a0)
    assert kinds == {_lib.TAB_KERNEL_PWG}
    _, s0 = _sr_session(N_SR, L_SR, STEPS_SR)
    torch.cuda.synchronize()
    alone_sr = _sr_tables(s0)
    assert int(s0.traffic[1].item()) > 0          # (the sparse-reward kernel served)
    del a0, s0

    # the two sessions at once: a thread and a non-default stream each
    out, errors = {}, []
    go = threading.Barrier(2)

    def worker(name, fn):
        try:
            stream = torch.cuda.Stream(device=0)
            with torch.cuda.stream(stream):
                assert torch.cuda.current_stream(0).cuda_stream == stream.cuda_stream != 0
                go.wait(timeout=120)
                out[name] = fn()
                stream.synchronize()
        except BaseException as e:      # noqa: BLE001 (reported by the main thread)
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=('dq', lambda: _dynaq_session(N_DQ, L_DQ, STEPS_DQ))),
               threading.Thread(target=worker, args=('sr', lambda: _sr_session(N_SR, L_SR, STEPS_SR)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    torch.cuda.synchronize()
    out['dq'][1].check_launches()                  # (no wave gave up waiting for a ring entry)
    _same(alone_dq, _dynaq_tables(out['dq'][1]))
    _same(alone_sr, _sr_tables(out['sr'][1]))

    # ... and a few instances of the concurrent Dyna-Q session against the oracle
    from oracle import c_oracle
    w = c_oracle.OracleWorld([dict(next=x['next'], reward=x['rewards'], terminal=x['terminals'],
                                   starts=x['starting_states']) for x in out['dq'][0].worlds])
    got = _dynaq_tables(out['dq'][1])
    for i in (0, 5, 4099, 8191):
        o = c_oracle.TabOracle(w, 1, c_oracle.AG_DYNAQ, SEED, True, instance_base=i, alpha=0.99,
                               trial_cap=64)
        for _ in range(L_DQ):
            o.run(0x7fffffff, 90, 50, step_budget=STEPS_DQ)
        assert np.array_equal(o.Q[0].astype(np.float32), got['q'][i]), i
        assert int(o.inst['state'][0]) == int(got['inst'][i, 0]) and \
            int(o.inst['trial'][0]) == int(got['inst'][i, 2]), i


def test_interleaved_launches_on_two_streams_share_one_world_handle():
    """One host thread, two streams, the launches of two sessions alternating — the second session
    works on the FIRST one's world handle (device tables of the worlds, immutable)."""
    from cobel_amd import _lib
    n, steps, launches = 3400, 128, 4
    _, a0, kinds, _ = _dynaq_session(n, launches, steps)
    torch.cuda.synchronize()
    alone = _dynaq_tables(a0)
    assert kinds == {_lib.TAB_KERNEL_PWG}
    del a0
    s1, s2 = torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)
    with torch.cuda.stream(s1):
        env1, ag1, _, _ = _dynaq_session(n, 0, steps)
    s1.synchronize()
    with torch.cuda.stream(s2):
        env2, ag2, _, _ = _dynaq_session(n, 0, steps, handle_of=env1)
    s2.synchronize()
    assert env2.handle.ptr == env1.handle.ptr
    flags = _lib.F_LEARN | ag1._policy_in(ag1.policy, env1, False)
    for _ in range(launches):
        with torch.cuda.stream(s1):
            ag1._launch(env1, ag1.policy, flags, 0x7fffffff, 90, steps, 50)
        with torch.cuda.stream(s2):
            ag2._launch(env2, ag2.policy, flags, 0x7fffffff, 90, steps, 50)
    torch.cuda.synchronize()
    ag1.check_launches()
    ag2.check_launches()
    _same(alone, _dynaq_tables(ag1))
    _same(alone, _dynaq_tables(ag2))
