"""CPU tests of the host logic and of the C-ABI surface (no compute calls: there is no GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, free_port


def test_library_loads_and_exports_every_declared_symbol():
    """libcobel_hip.so loads on a GPU-less host and exports exactly what include/cobel_hip.h
    declares."""
    from cobel_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'cobel_hip.h')).read()
    declared = set(re.findall(r'COBEL_API\s+[\w\s\*]+?\b(cobel_\w+)\s*\(', header))
    assert declared, 'no declarations found'
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), 'missing export ' + name
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert lib.cobel_abi_version() == 1017
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r'\bT (cobel_\w+)', out))
    assert exported == declared


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the run structs have the size the C compiler gives them."""
    from cobel_amd import _lib
    src = ('#include "cobel_hip.h"\n#include <stdio.h>\n#include <stddef.h>\n'
           'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(cobel_tab_run_t), '
           'sizeof(cobel_sr_run_t), sizeof(cobel_param_set_t), sizeof(cobel_sfma_run_t), '
           'offsetof(cobel_sfma_run_t, alpha), offsetof(cobel_sfma_run_t, seed), '
           'sizeof(cobel_sfma_event_t));}')
    exe = '/tmp/cobel_sizeof_%d' % os.getpid()
    subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe],
                   input=src.encode(), check=True)
    a, b, c, d, e, f, g = [int(x) for x in subprocess.check_output([exe]).split()]
    os.remove(exe)
    assert a == C.sizeof(_lib.TabRun) and b == C.sizeof(_lib.SRRun)
    assert c == C.sizeof(_lib.ParamSet) == 512
    assert d == C.sizeof(_lib.SFMARun) and e == _lib.SFMARun.alpha.offset
    assert f == _lib.SFMARun.seed.offset and g == _lib.SFMA_EVENT_BYTES
    from cobel_amd.agent.sfma import EVENT
    assert EVENT.itemsize == g
    src = ('#include "cobel_hip.h"\n#include <stdio.h>\n#include <stddef.h>\n'
           'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(cobel_dqn_replay_t), '
           'offsetof(cobel_dqn_replay_t, gamma), offsetof(cobel_dqn_replay_t, q_out), '
           'sizeof(cobel_dqn_act_t), offsetof(cobel_dqn_act_t, epsilon), '
           'offsetof(cobel_dqn_act_t, seed));}')
    subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe],
                   input=src.encode(), check=True)
    a, b, c, d, e, f = [int(x) for x in subprocess.check_output([exe]).split()]
    os.remove(exe)
    assert a == C.sizeof(_lib.DQNReplay) and b == _lib.DQNReplay.gamma.offset
    assert c == _lib.DQNReplay.q_out.offset and d == C.sizeof(_lib.DQNAct)
    assert e == _lib.DQNAct.epsilon.offset and f == _lib.DQNAct.seed.offset
    # the limits the Python layer asserts against are the header's
    src = ('#include "cobel_hip.h"\n#include <stdio.h>\n'
           'int main(){printf("%d %d %lld\\n", COBEL_MAX_ACTIONS, COBEL_MAX_BATCH, '
           '(long long)COBEL_TAB_SCRATCH_BYTES(1000));}')
    subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe],
                   input=src.encode(), check=True)
    a, b, c = [int(x) for x in subprocess.check_output([exe]).split()]
    os.remove(exe)
    assert a == _lib.MAX_ACTIONS == 32 and b == _lib.MAX_BATCH and c == _lib.tab_scratch_bytes(1000)


def test_sfma_metrics_match_reference(golden):
    """cobel_amd.memory.utils (host NumPy, as in the reference): Euclidean / SR / DR equal the
    matrices the reference's classes produce, from the dense sas tensor and from the compact
    successor table alike."""
    from cobel_amd.memory.utils import DR, SR, Euclidean, Metric
    Z = golden('sfma_traces')
    for wname in ('sfma_5x5', 'sfma_6x7'):
        nxt = Z['world/%s/next' % wname].astype(np.int64)
        W, H = int(Z['world/%s/width' % wname]), int(Z['world/%s/height' % wname])
        inv = [tuple(t) for t in Z['world/%s/invalid_transitions' % wname]]
        S = W * H
        sas = np.zeros((S, 4, S))
        sas[np.arange(S)[:, None], np.arange(4)[None, :], nxt] = 1.0
        for table in (nxt, sas):
            eu, sr, dr = Euclidean(W, H), SR(table, 0.9), DR(W, H, table, 0.9, inv)
            assert all(isinstance(m, Metric) for m in (eu, sr, dr))
            assert np.array_equal(eu.D, Z['metric/%s/Euclidean' % wname])
            np.testing.assert_allclose(sr.D, Z['metric/%s/SR' % wname], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(dr.D, Z['metric/%s/DR' % wname], rtol=1e-12, atol=1e-14)
        dr.update_transitions()
        np.testing.assert_allclose(dr.D, Z['metric/%s/DR' % wname], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('eps', [0.0, 0.1, 0.3, 1.0])
def test_param_set_constants_follow_the_reference_policy(eps):
    """cobel_param_set_fill (host code, no GPU): the epsilon-greedy constants of a parameter set
    are the reference's own float64 expressions (policy/greedy.py:83-86) — eps / n,
    (1 - eps) / n_ties, and per tie pattern the integer thresholds ceil(cdf * 2^53) of the
    normalised CDF that oracle/ref_loop.RefEpsilonGreedy (pinned to the reference's KAT) forms."""
    import math
    from cobel_amd import _lib
    from oracle.ref_loop import RefEpsilonGreedy
    ps = _lib.ParamSet()
    _lib.check(_lib.lib().cobel_param_set_fill(0.7, 0.95, eps, 0.8, C.byref(ps)))
    assert (ps.alpha, ps.gamma, ps.epsilon, ps.model_lr) == (0.7, 0.95, eps, 0.8)
    assert ps.alpha_f == np.float32(0.7) and ps.gamma_f == np.float32(0.95)
    assert ps.model_lr_f == np.float32(0.8)
    for n in range(1, 5):
        assert ps.eps_base[n] == eps / n and ps.eps_bonus[n] == ((1.0 - eps) * 1.0) / n
    pol = RefEpsilonGreedy(eps, None)
    for t in range(1, 16):
        v = np.array([1.0 if (t >> a) & 1 else 0.0 for a in range(4)], dtype=np.float32)
        cdf = np.cumsum(pol.get_action_probs(v))
        cdf /= cdf[-1]
        for k in range(3):
            want = math.ceil(math.ldexp(float(cdf[k]), 53))
            assert ps.eps_thr[t][k] == min(want, 2 ** 64 - 1), (t, k)
    with pytest.raises(AssertionError):
        _lib.check(_lib.lib().cobel_param_set_fill(0.5, 0.5, 1.5, 0.9, C.byref(ps)))


def test_argument_errors_map_to_reference_exceptions():
    """Bad arguments are rejected before any HIP call, with the exception the reference raises."""
    from cobel_amd import _lib
    lib = _lib.lib()
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_world_create(None, None, None, None, None, 25, 1, 0, None))
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_tab_run(None, None, None))
    out = (C.c_int32 * 4)(7, 7, 7, 7)
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_tab_describe(None, None, out))
    assert list(out) == [0, 0, 0, 0]
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_tab_describe(None, None, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_tab_query(1000000, 1, 10, None, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_tab_query(25, 1, 63, None, None))
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_dqn_replay(None, None))
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_dqn_act(None, None, None))
    with pytest.raises(NotImplementedError):      # only 64-64 networks with 4 outputs, batch 32
        _lib.check(lib.cobel_dqn_replay_query(6, 32, 64, 4, 32, 1, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_dqn_replay_query(33, 64, 64, 4, 32, 1, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_dqn_replay_query(6, 64, 64, 4, 16, 0, None))
    lds = C.c_int32()
    _lib.check(lib.cobel_dqn_replay_query(6, 64, 64, 4, 32, 1, C.byref(lds)))
    assert 70000 < lds.value <= 80 * 1024          # parameter-staging kernel: two workgroups per CU
    _lib.check(lib.cobel_dqn_replay_query(25, 64, 64, 4, 32, 1, C.byref(lds)))
    assert 50000 < lds.value <= 53 * 1024          # streaming kernel (activations only) beyond 7 inputs
    _lib.check(lib.cobel_dqn_replay_query(6, 64, 64, 4, 32, 0, C.byref(lds)))
    assert lds.value <= 40 * 1024                  # four in float32
    run = _lib.DQNReplay()
    run.n_inputs, run.n_hidden1, run.n_hidden2, run.n_actions, run.batch = 6, 64, 64, 4, 32
    with pytest.raises(AssertionError):            # NULL tensors are refused before any launch
        _lib.check(lib.cobel_dqn_replay(C.byref(run), None))
    _lib.check(lib.cobel_tab_query(1024, 1, 50, C.byref(lds), None))
    assert lds.value == 1024 * (16 + 8 + 4)   # Q, compact model, visit counters
    r, s, t = C.c_float(), C.c_uint16(), C.c_uint8()
    rec = lib.cobel_pack_model(C.c_float(0.75), 321, 1)
    lib.cobel_unpack_model(rec, C.byref(r), C.byref(s), C.byref(t))
    assert (r.value, s.value, t.value) == (0.75, 321, 1)


def test_world_builders_match_reference_tables(golden_worlds):
    """cobel_amd.misc.gridworld_tools == tables derived from the reference's make_* builders."""
    from cobel_amd.misc import gridworld_tools as gt
    built = {
        'open_5x5': gt.make_open_field(5, 5, 0, 1),
        'kat_5x5': gt.make_gridworld(5, 5, [0], np.array([[0, 10.]]), starting_states=[24]),
        'open_4x7_goal9': gt.make_open_field(4, 7, 9, 2.5),
        'empty_3x3': gt.make_empty_field(3, 3),
        'walls_8x8': gt.make_gridworld(
            8, 8, terminals=[7, 56], rewards=np.array([[7, 1.0], [56, -0.5], [30, 0.25]]),
            goals=[7], invalid_states=[10, 11, 12, 20, 28, 36, 44, 45, 50],
            invalid_transitions=[(0, 1), (1, 0), (62, 63), (63, 62), (33, 34)],
            starting_states=[63, 32, 3, 27]),
        'windy_7x10_up': gt.make_windy_gridworld(
            7, 10, np.array([0, 0, 0, 1, 1, 1, 2, 2, 1, 0]), 37, 1.0, 'up'),
        'windy_5x6_down': gt.make_windy_gridworld(5, 6, np.array([0, 1, 2, 1, 0, 3]), 3, 1.0, 'down'),
        'open_32x32': gt.make_open_field(32, 32, 0, 1),
        'maze_32x32_1234': gt.make_obstacle_maze(32, 32, 1234),
        'maze_32x32_1235': gt.make_obstacle_maze(32, 32, 1235),
        't_maze_3_2_right': gt.make_t_maze(3, 2, 'right', 1.0),
        't_maze_2_4_left': gt.make_t_maze(2, 4, 'left', 2.0),
        'double_t_maze_3_2_lr': gt.make_double_t_maze(3, 2, 'left-right', 1.5),
        'two_sided_t_maze_4_3_ll': gt.make_two_sided_t_maze(4, 3, 'left-left', 2.0),
        'two_choice_t_maze_5_7_3_ll': gt.make_two_choice_t_maze(5, 7, 3, 'left', 'left'),
        'two_choice_t_maze_4_5_2_rr': gt.make_two_choice_t_maze(4, 5, 2, 'right', 'right'),
        '8_maze_3_2_right': gt.make_8_maze(3, 2, 'right', 1.0),
        '8_maze_4_3_left': gt.make_8_maze(4, 3, 'left', 0.5),
        'detour_maze_2_2_4_3': gt.make_detour_maze(2, 2, 4, 3, 1.0),
        'cross_maze_2_2_left': gt.make_cross_maze(2, 2, 'left'),
        'cross_maze_3_1_bottom': gt.make_cross_maze(3, 1, 'bottom', 4.0),
        'double_t_maze_2_1': gt.load_world(os.path.join(ROOT, 'tests', 'golden',
                                                        'double_t_maze_2_1.pkl')),
    }
    for name, w in built.items():
        g = golden_worlds(name)
        assert np.array_equal(w['next'], g['next']), name
        assert np.array_equal(w['rewards'], g['reward']), name
        assert np.array_equal(w['terminals'], g['terminal']), name
        assert np.array_equal(w['starting_states'], g['starts']), name
        assert np.array_equal(w['coordinates'], g['coordinates']), name
        c = w.compact()
        assert c['next'].dtype == np.uint16 and c['reward'].dtype == np.float32
    sas = built['open_5x5']['sas']          # dense form is still there for whoever asks
    assert sas.shape == (25, 4, 25) and np.array_equal(np.argmax(sas, axis=2),
                                                       built['open_5x5']['next'])
    assert np.array_equal(sas.sum(axis=2), np.ones((25, 4)))


def test_monitors_match_reference(golden):
    from cobel_amd.monitor import EscapeLatencyMonitor, RewardMonitor
    k = golden('monitor_kat')
    mon = EscapeLatencyMonitor(len(k['steps']), int(k['max_steps']))
    for t in k['order']:
        mon.update({'trial': int(t), 'steps': int(k['steps'][t])})
    assert np.array_equal(mon.latency_trace, k['latency'], equal_nan=True)
    assert np.array_equal(mon.latency_trace_avg, k['latency_avg'], equal_nan=True)
    assert mon.get_trace() is mon.latency_trace
    # RewardMonitor / ResponseMonitor (monitor/behavior.py:174-199, :269-290)
    from cobel_amd.monitor import QMonitor, ResponseMonitor, RewardMonitor, TrajectoryMonitor
    n = len(k['steps'])
    rew, ra, rb = RewardMonitor(n, (-0.5, 1.5)), ResponseMonitor(n), ResponseMonitor(n)
    for t in k['order']:
        t = int(t)
        rew.update({'trial': t, 'trial_reward': float(k['rewards'][t])})
        ra.update({'trial': t, 'trial_reward': float(k['rewards'][t])})
        rb.update({'trial': t, 'trial_reward': float(k['rewards'][t]),
                   'response': int(k['responses'][t])})
    assert np.array_equal(rew.reward_trace, k['reward_trace'], equal_nan=True)
    assert np.array_equal(rew.reward_trace_avg, k['reward_avg'], equal_nan=True)
    assert np.array_equal(ra.responses, k['resp_default'], equal_nan=True)
    assert np.array_equal(ra.CRC, k['crc_default'], equal_nan=True)
    assert np.array_equal(rb.responses, k['resp_given'], equal_nan=True)
    assert np.array_equal(rb.CRC, k['crc_given'], equal_nan=True)

    class Env:           # behavior.py:359-372: a new list whenever logs['trial_session'] changes
        pos = 0

        def get_position(self):
            self.pos += 1
            return np.array([self.pos, 0.0])
    tm = TrajectoryMonitor(5, Env())
    for session in (0, 0, 0, 1, 1, 0):
        tm.update({'trial_session': session})
    assert [len(t) for t in tm.get_trace()] == [3, 2, 1] and tm.get_trace()[1][0][0] == 4

    class Agent:         # behavior.py:441-451
        def predict_on_batch(self, obs):
            return np.asarray(obs)[:, None] * np.ones(4)
    qm = QMonitor(5, np.arange(3))
    qm.update({'agent': Agent()})
    qm.update({'agent': Agent()})
    assert len(qm.get_trace()) == 2 and qm.get_trace()[0].shape == (3, 4)


def test_occupancy_map_matches_reference(golden, golden_worlds):
    from cobel_amd.analysis import get_occupancy_map, occupancy_from_counts
    k = golden('monitor_kat')
    coords = golden_worlds('walls_8x8')['coordinates']
    cuts = np.cumsum(k['traj_len'])[:-1]
    trajs = [coords[s] for s in np.split(k['traj_states'], cuts)]
    for m in ('expand', 'include', 'ignore'):
        assert np.array_equal(get_occupancy_map(trajs, 8, 8, 1.0, m), k['occ_' + m])
    assert np.array_equal(get_occupancy_map(trajs, 8, 8, 2.0, 'expand'), k['occ_bin2'])
    counts = np.bincount(k['traj_states'], minlength=64)
    assert np.array_equal(occupancy_from_counts(counts, coords, 8, 8, 1.0), k['occ_expand'])
    assert np.array_equal(occupancy_from_counts(counts, coords, 8, 8, 2.0), k['occ_bin2'])
    with pytest.raises(AssertionError):
        get_occupancy_map(trajs, 8, 8, 0.0)


def test_callbacks_semantics():
    """agent/agent.py:145-243: shallow copy, 'agent' key, dict returns are merged."""
    from cobel_amd.agent import Callbacks
    seen = []
    cb = Callbacks('AGENT', {'on_trial_end': [lambda logs: seen.append(dict(logs)) or {'x': 1},
                                              lambda logs: None]})
    logs = {'trial': 3}
    out = cb.on_trial_end(logs)
    assert out == {'trial': 3, 'agent': 'AGENT', 'x': 1} and logs == {'trial': 3}
    assert seen[0]['agent'] == 'AGENT'
    assert cb.has('on_trial_end') and not cb.has('on_step_end')
    assert cb.on_step_begin({'a': 1}) == {'a': 1, 'agent': 'AGENT'}


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'cobel-rl_amd'))
import numpy as np, torch, torch.distributed as dist
from cobel_amd.agent.agent import DeviceMonitors
from cobel_amd.monitor import EscapeLatencyMonitor, RewardMonitor
rank = int(os.environ['RANK'])
dist.init_process_group('gloo')
# each rank owns half of 8 instances: per-trial latencies are rank-local partial sums
lat = np.array([[3, 5, 7], [4, 6, 8], [1, 1, 1], [9, 9, 9], [2, 4, 6], [0, 0, 0], [5, 5, 5], [7, 8, 9]])
mine = lat[rank * 4:(rank + 1) * 4]
mon = DeviceMonitors(torch.device('cpu'), 1, 4, occupancy=True)
mon.reserve(3)
mon.lat_sum += torch.as_tensor(mine.sum(axis=0)); mon.lat_cnt += 4
mon.reward_sum += float(rank + 1); mon.occupancy += rank + 1; mon.steps_done += 100 * (rank + 1)
# first reporting interval, TWO monitors: each asks for the global sums; the rank-local
# accumulators must stay rank-local, every call is one collective
el, rw = EscapeLatencyMonitor(6, 10), RewardMonitor(6)
el.update_from_device(mon)
rw.update_from_device(mon)
assert mon.collectives == 2
assert np.allclose(el.latency_trace[:3], lat.mean(axis=0)), el.latency_trace
assert np.allclose(rw.reward_trace[:3], 3.0 / 8.0), rw.reward_trace
assert int(mon.steps_done.item()) == 100 * (rank + 1), 'local accumulator was overwritten'
assert np.array_equal(mon.lat_sum.numpy(), mine.sum(axis=0))
g = mon.all_reduce()
assert g.ranks == 2 and g.steps_done == 300 and int(g.occupancy.sum()) == 12
assert np.allclose(g.reward_sum, 3.0) and np.array_equal(g.lat_cnt, [8, 8, 8])
# second interval: three more trials on every rank (the kernels keep adding into the same buffers)
mon.reserve(6)
more = (lat + 1)[rank * 4:(rank + 1) * 4]
mon._raw['lat_sum'][0, 3:6] += torch.as_tensor(more.sum(axis=0)); mon._raw['lat_cnt'][0, 3:6] += 4
mon._raw['reward_sum'][0, 3:6] += 0.5 * (rank + 1); mon.steps_done += 7
el.update_from_device(mon)
rw.update_from_device(mon)
assert np.allclose(el.latency_trace, np.concatenate([lat.mean(axis=0), (lat + 1).mean(axis=0)]))
assert np.allclose(rw.reward_trace, [3 / 8.] * 3 + [1.5 / 8.] * 3), rw.reward_trace
g2 = mon.all_reduce()
assert g2.steps_done == 314 and np.array_equal(g2.lat_cnt, [8] * 6)
assert np.array_equal(g2.lat_sum[:3], lat.sum(axis=0))        # earlier trials counted once
# the float64 sums are bit-identical on all ranks (fixed summation order)
bits = torch.as_tensor(np.ascontiguousarray(g2.reward_sum).view(np.int64))
both = [torch.zeros_like(bits) for _ in range(2)]
dist.all_gather(both, bits)
assert torch.equal(both[0], both[1])
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_monitor_allreduce_two_ranks_gloo(tmp_path):
    """The N > 1 path's only collective (monitor reduction), world_size 2 on the gloo backend."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('ok' in o for o in outs)


WORKER8 = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'cobel-rl_amd'))
import numpy as np, torch, torch.distributed as dist
from cobel_amd.agent.agent import DeviceMonitors
from cobel_amd.misc.sharding import shard_instances
rank, world, total, trials = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), 1003, 5
dist.init_process_group('gloo')
# what every instance of the whole job reports (a function of its GLOBAL id), and this rank's share
gid = np.arange(total)
lat = (gid[:, None] * 7 + np.arange(trials)[None, :] * 3) % 41
rew = ((gid[:, None] + np.arange(trials)[None, :]) % 5 == 0).astype(np.float64)
base, count = shard_instances(total, world, rank)
assert count in (total // world, total // world + 1)
mon = DeviceMonitors(torch.device('cpu'), 1, 4, occupancy=True)
mon.reserve(trials)
mine = slice(base, base + count)
mon.lat_sum += torch.as_tensor(lat[mine].sum(axis=0)); mon.lat_cnt += count
mon.reward_sum += torch.as_tensor(rew[mine].sum(axis=0)); mon.steps_done += int(lat[mine].sum()) + count * trials
mon.occupancy += count
g = mon.all_reduce()
assert g.ranks == world and mon.collectives == 1
assert np.array_equal(g.lat_sum[:trials], lat.sum(axis=0)) and np.array_equal(g.lat_cnt[:trials], [total] * trials)
assert np.array_equal(g.reward_sum[:trials], rew.sum(axis=0))
assert g.steps_done == int(lat.sum()) + total * trials and int(g.occupancy.sum()) == 4 * total
bits = torch.as_tensor(np.ascontiguousarray(g.reward_sum).view(np.int64))
every = [torch.zeros_like(bits) for _ in range(world)]
dist.all_gather(every, bits)
assert all(torch.equal(every[0], e) for e in every)
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_monitor_allreduce_eight_ranks_uneven_split_gloo(tmp_path):
    """The split BASELINE config 3 names — eight ranks — rehearsed on the CPU with a total that does
    not divide (1 003 instances: shards of 126 and 125): every rank ends with the sums of the
    whole job, bit-identical everywhere, after ONE collective."""
    script = tmp_path / 'worker8.py'
    script.write_text(WORKER8)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE='8',
               OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(8)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('ok' in o for o in outs)


def test_instance_shards_are_disjoint_and_complete():
    """Global instance ids of the ranks of a node: contiguous, disjoint, complete, balanced —
    for the BASELINE split of C3's 65 536 instances over 1/2/4/8 GPUs and for ragged totals."""
    from cobel_amd.misc.sharding import shard_instances
    for total in (65536, 16384, 8192, 1000, 7, 0):
        for world in (1, 2, 4, 8):
            ids, sizes = [], []
            for rank in range(world):
                base, count = shard_instances(total, world, rank)
                ids.extend(range(base, base + count))
                sizes.append(count)
            assert ids == list(range(total)), (total, world)
            assert max(sizes) - min(sizes) <= 1
    assert [shard_instances(65536, 8, r) for r in (0, 7)] == [(0, 8192), (57344, 8192)]


def test_topology_builders_match_reference(golden):
    """cobel_amd.misc.topology_tools == the reference's linear_track / t_maze / grid / cross as
    tables."""
    from cobel_amd.misc import topology_tools as tt
    k = golden('topology_kat')
    built = {'linear_10x2': tt.linear_track(10, 2, 1., 20., 'right'),
             'linear_5x1_left': tt.linear_track(5, 1, 0.5, 2., 'left'),
             't_maze_4_3_1': tt.t_maze(4, 3, 1),
             't_maze_3_2_2_left': tt.t_maze(3, 2, 2, 2.0, 3.0, 'left'),
             'grid_4x3': tt.grid((4, 3)), 'grid_5': tt.grid(5, (0.0, 2.0), 7.0, '12'),
             'cross_2_1_rot30': tt.cross(2, 1, 0.5, 30.0), 'cross_3_2': tt.cross(3, 2, 2.0),
             'hex_4': tt.hexagonal(4), 'hex_5_goal7': tt.hexagonal(5, (0.0, 2.0), 3.0, '7')}
    for name, (nodes, starts) in built.items():
        ids = list(nodes.keys())
        assert ids == [str(i) for i in range(len(ids))]
        idx = {kk: i for i, kk in enumerate(ids)}
        nbr = np.array([[idx[m] for m in nodes[kk]['neighbors']] for kk in ids])
        assert np.array_equal(nbr, k[name + '/nbr']), name
        assert np.array_equal(np.array([nodes[kk]['pose'] for kk in ids]), k[name + '/pose']), name
        assert np.array_equal([nodes[kk]['reward'] for kk in ids], k[name + '/reward']), name
        assert np.array_equal([bool(nodes[kk]['terminal']) for kk in ids], k[name + '/terminal'])
        assert np.array_equal([idx[kk] for kk in starts], k[name + '/starts']), name
        assert all(nodes[kk]['id'] == kk for kk in ids)


def test_torch_network_adapter_contract():
    """The checks of unit_tests/test_torchnetwork.py that define the adapter's contract: defaults
    (MSE with reduction 'none', Adam), six weight arrays, layer activity shaped (units, batch),
    training reduces the error, set_weights round trip, clone independence (and device kept),
    set_loss / set_optimizer / set_trainable — plus the stacked form against single networks."""
    import torch
    from collections import OrderedDict
    from cobel_amd.network import TorchNetwork
    torch.manual_seed(0)
    layers = [('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
              ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
              ('output', torch.nn.Linear(64, 4))]
    net = TorchNetwork(torch.nn.Sequential(OrderedDict(layers)).double(),
                       activations={'dense_1': torch.relu, 'dense_2': None, 'output': None})
    assert isinstance(net.criterion, torch.nn.MSELoss) and net.criterion.reduction == 'none'
    assert isinstance(net.optimizer, torch.optim.Adam)
    w = net.get_weights()
    assert len(w) == 6 and w[0].shape == (64, 6) and w[5].shape == (4,)
    rng = np.random.default_rng(0)
    x, y = rng.random((32, 6)), rng.random((32, 4))
    assert net.predict_on_batch(x).shape == (32, 4)
    assert net.get_layer_activity(x, 'dense_1').shape == (64, 32)
    assert (net.get_layer_activity(x, 'dense_1') >= 0).all()       # activation applied
    assert net.get_layer_activity(x, 0).shape == (64, 32)
    before = np.mean((net.predict_on_batch(x) - y) ** 2)
    twin = net.clone()
    for _ in range(50):
        net.train_on_batch(x, y)
    assert np.mean((net.predict_on_batch(x) - y) ** 2) < before
    assert np.array_equal(twin.get_weights()[0], w[0])             # the clone did not move
    assert twin.device == net.device
    net.set_weights(w)
    assert all(np.array_equal(a, b) for a, b in zip(net.get_weights(), w))
    net.set_loss('huber')
    assert isinstance(net.criterion, torch.nn.HuberLoss) and net.criterion.reduction == 'none'
    net.set_optimizer('sgd', {'lr': 0.1})
    assert isinstance(net.optimizer, torch.optim.SGD)
    net.set_trainable(['dense_1'], False)
    assert not net.model.get_parameter('dense_1.weight').requires_grad
    net.set_trainable([0, 1], [True, False])
    assert net.model.get_parameter('dense_1.bias').requires_grad
    assert not net.model.get_parameter('dense_2.weight').requires_grad
    net.train_on_batch(x, y[:, 0], sample_weights=np.ones(32)) if False else None
    # stacked copies == independent single networks, step for step
    base = TorchNetwork(torch.nn.Sequential(OrderedDict(
        [('a', torch.nn.Linear(6, 8)), ('r', torch.nn.ReLU()), ('o', torch.nn.Linear(8, 4))])).double())
    stack, singles = base.replicate(3), [base.clone() for _ in range(3)]
    xs, ys = rng.random((3, 5, 6)), rng.random((3, 5, 4))
    for _ in range(4):
        stack.train_on_device(torch.as_tensor(xs), torch.as_tensor(ys))
        for i, s in enumerate(singles):
            s.train_on_batch(xs[i], ys[i])
    for i, s in enumerate(singles):
        for a, b in zip(stack.get_weights(i), s.get_weights()):
            assert np.allclose(a, b, rtol=1e-12, atol=1e-15)
    target = base.replicate(3)
    target.blend_from(stack, 0.01)
    w0, w1, wt = base.get_weights()[0], stack.get_weights(1)[0], target.get_weights(1)[0]
    assert np.allclose(wt, w0 + 0.01 * (w1 - w0), rtol=1e-12)


def _opt_sim(task, params):
    return task['bias'] + params['x_1'] + params['x_2'] ** 2 + params['x_3'] ** 3


def _opt_loss(data_sim, data_exp):
    error = 0.
    for t in data_sim:
        error += (np.mean(data_sim[t]) - data_exp[t]) ** 2
    return error / len(data_sim)


def test_grid_search_matches_reference(golden, tmp_path):
    """GridSearchOptimizer: 'nested' and 'systematic' enumerate the combinations in the
    reference's order (grid_search.py:112-171); fit() leaves the same files and fitness values;
    fit_vectorised() over a batched simulation gives the same fit; recompute_fit() reads the
    stored simulation data back."""
    import pickle
    from cobel_amd.optimizer import GridSearchOptimizer, spread_over_instances
    D = golden('optimizer_kat')
    params = {'x_1': [0, 1, 2, 3, 4], 'x_2': np.array([0.4, 0.1, 0.2, 0.3, 0.0]),
              'x_3': [0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1]}
    for order in ('nested', 'systematic'):
        opt = GridSearchOptimizer(str(tmp_path) + '/', params, order=order)
        keys = np.array(list(opt.parameter_combinations), dtype=np.float64)
        assert np.array_equal(keys, D['keys_' + order]), order
        k0 = next(iter(opt.parameter_combinations))
        assert opt.parameter_combinations[k0] == dict(zip(params, k0))
    shuffled = GridSearchOptimizer(str(tmp_path) + '/', params, order='shuffled',
                                   rng=np.random.default_rng(5))
    assert sorted(shuffled.parameter_combinations) == sorted(opt.parameter_combinations)

    small = {'x_1': [0, 2, 4], 'x_2': np.array([0.3, 0.1]), 'x_3': [0.5, 0.9]}
    tasks = {'task_1': {'bias': 0.0}, 'task_2': {'bias': 1.5}}
    data = {t: _opt_sim(tasks[t], {'x_1': 2, 'x_2': 0.1, 'x_3': 0.9}) for t in tasks}
    d1 = tmp_path / 'seq'
    d1.mkdir()
    opt = GridSearchOptimizer(str(d1) + '/', small, nb_runs=2)
    fit = opt.fit(_opt_sim, tasks, data, _opt_loss, store_simulation_data=True)
    assert np.array_equal(np.array(list(fit), dtype=np.float64), D['fit_keys'])
    assert np.array_equal(np.array([fit[k] for k in fit]), D['fit_values'])
    assert sorted(os.listdir(d1)) == list(D['fit_files'])
    with open(d1 / 'fit.pkl', 'rb') as fh:
        assert pickle.load(fh) == fit

    calls = []

    def batched(task, combos, nb_runs):
        calls.append(len(combos))
        arrays, which = spread_over_instances(combos, nb_runs)
        assert len(which) == len(combos) * nb_runs and set(arrays) == set(small)
        vals = task['bias'] + arrays['x_1'] + arrays['x_2'] ** 2 + arrays['x_3'] ** 3
        return [list(vals[which == c]) for c in range(len(combos))]

    d2 = tmp_path / 'vec'
    d2.mkdir()
    vec = GridSearchOptimizer(str(d2) + '/', small, nb_runs=2)
    fit_v = vec.fit_vectorised(batched, tasks, data, _opt_loss, store_simulation_data=True)
    assert calls == [12, 12]                       # one call per task, all combinations at once
    assert list(fit_v) == list(fit) and np.allclose([fit_v[k] for k in fit_v],
                                                    [fit[k] for k in fit], rtol=0, atol=1e-15)
    assert sorted(os.listdir(d2)) == list(D['fit_files'])
    again = GridSearchOptimizer(str(d2) + '/', small, nb_runs=2)
    assert again.recompute_fit(data, lambda sim, exp: 7.0) == {k: 7.0 for k in fit}
    assert again.fit_vectorised(batched, tasks, data, _opt_loss) == fit_v and calls == [12, 12]


def test_graft_entry_build_runs_on_a_gpu_less_host():
    """__graft_entry__.build(): both Makefiles, the import and the ABI check (what the driver runs
    here every round)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.build()


def test_sfma_host_surface_defaults():
    """SFMAMemory / SFMA carry the reference's attribute names and defaults
    (memory/sfma.py:142-193, agent/sfma.py:174-231) before any device is touched."""
    from cobel_amd.agent import SFMA
    from cobel_amd.memory import SFMAMemory
    from cobel_amd.memory.utils import Euclidean
    from cobel_amd.policy import EpsilonGreedy
    from cobel_amd.spaces import Box, Discrete
    m = SFMAMemory(Euclidean(3, 3), 9, 4)
    expect = dict(decay_inhibition=0.9, decay_strength=1.0, decay_recency=0.9, learning_rate=0.9,
                  beta=20, C_step=1.0, I_step=1.0, R_threshold=1e-6, deterministic=False,
                  recency=False, C_normalize=False, D_normalize=False, R_normalize=True,
                  mode='default', reward_modulation=1.0, blend=0.1, interpolation_fwd=0.5,
                  interpolation_rev=0.5, reward_mod_local=False, error_mod_local=False,
                  reward_mod=False, error_mod=False, policy_mod=False, state_mod=False)
    for k, v in expect.items():
        assert getattr(m, k) == v, k
    a = SFMA(Discrete(9), Discrete(4), EpsilonGreedy(), m)
    assert (a.learning_rate, a.gamma, a.nb_replays) == (0.99, 0.99, 1)
    assert not (a.random or a.dynamic or a.offline or a.start_replay or a.mask_actions)
    assert a.Q.shape == (9, 4) and a.action_mask.shape == (9, 4) and a.td == 0.0
    assert a.policy_test is a.policy and set(a.callbacks.custom_callbacks) == set()
    assert hasattr(a.callbacks, 'on_replay_begin') and hasattr(a.callbacks, 'on_replay_end')
    with pytest.raises(AssertionError):
        SFMA(Box(np.zeros(2), np.ones(2)), Discrete(4), EpsilonGreedy(), m)
    with pytest.raises(AssertionError):
        SFMAMemory(Euclidean(3, 3), 9, 6)


def test_device_monitors_stripes_on_cpu():
    """DeviceMonitors keeps [stripes, cap] copies for the kernels and exposes their sums; growing
    the capacity keeps what was accumulated; one stripe behaves like a plain tensor."""
    import torch
    from cobel_amd.agent.agent import DeviceMonitors
    m = DeviceMonitors(torch.device('cpu'), 1, 4, occupancy=False, responses=True, stripes=4)
    m.reserve(5)
    assert m.raw('lat_sum').shape == (4, 5) and m.raw('resp_cnt').shape == (4, 5)
    m.raw('lat_sum')[1, 2] += 7
    m.raw('lat_sum')[3, 2] += 5
    m.raw('lat_cnt')[0, 2] += 2
    m.raw('reward_sum')[2, 4] += 1.5
    assert m.lat_sum.tolist() == [0, 0, 12, 0, 0] and m.lat_cnt.tolist() == [0, 0, 2, 0, 0]
    m.reserve(9)
    assert m.raw('lat_sum').shape == (4, 9) and m.lat_sum.tolist()[:5] == [0, 0, 12, 0, 0]
    assert np.allclose(m.mean_latency()[2], 6.0) and np.isnan(m.mean_latency()[0])
    assert float(m.reward_sum[4]) == 1.5
    one = DeviceMonitors(torch.device('cpu'), 1, 4)
    one.reserve(3)
    one.lat_sum += torch.tensor([1, 2, 3])          # in place on the single copy
    assert one.raw('lat_sum').tolist() == [[1, 2, 3]] and one.lat_sum.data_ptr() == one.raw('lat_sum').data_ptr()
    one.lat_sum = None
    assert one.raw('lat_sum') is None and one.lat_sum is None


def test_fused_dqn_path_recognises_models_by_behaviour():
    """StackedTorchNetwork._mlp3_names (host logic of the fused DQN step): Linear-ReLU-Linear-ReLU-
    Linear is recognised whether written as nn.Sequential or, like the reference's demos
    (demo/topology/demo_dqn.py:36-60), as a custom Module calling functional relu; look-alikes
    (another activation, another depth, a frozen layer, extra parameters) are left to PyTorch."""
    import torch
    from torch import nn
    from cobel_amd.network import TorchNetwork

    class Demo(nn.Module):
        def __init__(self, act=torch.relu, extra=False):
            super().__init__()
            self.layer_dense_1 = nn.Linear(6, 64)
            self.layer_dense_2 = nn.Linear(64, 64)
            self.layer_output = nn.Linear(64, 4)
            self.act = act
            if extra:
                self.scale = nn.Parameter(torch.ones(1))
            self.double()

        def forward(self, x):
            x = torch.reshape(x, (len(x), -1))
            x = self.act(self.layer_dense_1(x))
            x = self.act(self.layer_dense_2(x))
            return self.layer_output(x)

    def names(model):
        torch.manual_seed(0)
        return TorchNetwork(model).replicate(2)._mlp3_names()

    assert names(Demo()) == ['layer_dense_1', 'layer_dense_2', 'layer_output']
    seq = nn.Sequential(nn.Flatten(), nn.Linear(6, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(),
                        nn.Linear(64, 4)).double()
    assert names(seq) == ['1', '3', '5']
    assert names(Demo(act=torch.tanh)) is None
    assert names(Demo(act=lambda v: torch.nn.functional.leaky_relu(v, 0.1))) is None
    # activations that EQUAL ReLU on the small activations of a freshly initialised network
    assert names(Demo(act=torch.nn.functional.relu6)) is None
    assert names(Demo(act=lambda v: torch.clamp(v, 0.0, 50.0))) is None
    assert names(Demo(act=lambda v: torch.nn.functional.hardtanh(v, 0.0, 1e3))) is None
    for act in (nn.ReLU6(), nn.Hardtanh(0.0, 20.0)):
        assert names(nn.Sequential(nn.Linear(6, 64), act, nn.Linear(64, 64), nn.ReLU(),
                                   nn.Linear(64, 4)).double()) is None
    f32 = nn.Sequential(nn.Linear(6, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 4))
    assert names(f32.float()) == ['0', '2', '4']
    assert names(Demo(extra=True)) is None
    assert names(nn.Sequential(nn.Linear(6, 32), nn.ReLU(), nn.Linear(32, 4)).double()) is None
    frozen = Demo()
    frozen.layer_dense_1.weight.requires_grad = False
    net = TorchNetwork(frozen).replicate(2)
    net.params['layer_dense_1.weight'].requires_grad_(False)
    assert net._mlp3_names() is None


def test_deterministic_flag_and_stochastic_rows():
    """world['deterministic'] only selects HOW the reference reads a row of sas — argmax, or a draw
    from it (interface/gridworld.py:115-123).  All builders write one-hot rows
    (misc/gridworld_tools.py:103-134), for which both are the same step.  An edited sas is
    followed: as a table (argmax) while the flag is set, as successor lists with cumulative
    probabilities — what cobel_world_set_transitions takes — when it is off."""
    from cobel_amd.misc import gridworld_tools as gt
    a = gt.make_gridworld(3, 4, terminals=[0], rewards=np.array([[0, 1.0]]), goals=[0])
    b = gt.make_gridworld(3, 4, terminals=[0], rewards=np.array([[0, 1.0]]), goals=[0],
                          deterministic=False)
    assert b['deterministic'] is False and np.array_equal(a['next'], b['next'])
    assert np.array_equal(gt.successor_table(a['sas']), a['next'])
    assert np.array_equal(b.compact()['next'], a['next']) and 'transitions' not in b.compact()
    _ = b['sas']                                  # materialised, still one-hot: still a table
    assert 'transitions' not in b.compact()
    # an edited row of a deterministic world: argmax
    c = gt.make_gridworld(2, 2)
    sas = c['sas']
    sas[0, 2] = 0.0
    sas[0, 2, 3] = 1.0                            # "right" from state 0 now jumps to state 3
    assert c.compact()['next'][0, 2] == 3
    sas[0, 2, 1], sas[0, 2, 3] = 0.25, 0.75
    assert c.compact()['next'][0, 2] == 3 and 'transitions' not in c.compact()
    # the same world with the flag off: lists (ascending states, normalised cumulative sums)
    c['deterministic'] = False
    sas[1, 0] = [2.0, 0.0, 1.0, 1.0]              # Generator.choice(p=) refuses such a row, so do we
    with pytest.raises(ValueError, match='do not sum to 1'):
        c.compact()
    sas[1, 0] = [0.5, 0.0, 0.25, 0.25]
    t = c.compact()
    off, succ, cdf = t['transitions']
    assert len(off) == 17 and off[-1] == len(succ) == len(cdf) == 16 + 1 + 2
    p = 0 * 4 + 2
    assert list(succ[off[p]:off[p + 1]]) == [1, 3] and list(cdf[off[p]:off[p + 1]]) == [0.25, 1.0]
    p = 1 * 4 + 0
    assert list(succ[off[p]:off[p + 1]]) == [0, 2, 3]
    assert list(cdf[off[p]:off[p + 1]]) == [0.5, 0.75, 1.0]
    assert t['next'][1, 0] == 0
    with pytest.raises(ValueError):
        gt.successor_table(np.zeros((2, 4, 2)))   # rows without any successor
    with pytest.raises(ValueError):
        gt.transition_lists(-np.ones((2, 4, 2)))


def test_pairwise_order_reproduces_numpy_sums_of_sparse_vectors():
    """cobel_pairwise_order (host function of the library): adding the k <= 32 non-zero elements of a
    float32 vector in the order it returns gives np.sum of the whole vector bit for bit — the
    grouping NumPy's pairwise summation applies to those positions (agent/sr.py:302-306)."""
    import ctypes as C
    from cobel_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(12)
    for trial in range(400):
        n = int(rng.choice([5, 7, 8, 25, 100, 128, 129, 221, 256, 400, 480, 512, 640, 961, 1024]))
        k = int(rng.integers(0, min(32 if trial % 2 else 8, n) + 1))
        pos = np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
        vals = (rng.standard_normal(k) * 10.0 ** rng.integers(-6, 6, k)).astype(np.float32)
        vec = np.zeros(n, dtype=np.float32)
        vec[pos] = vals
        dst, src = (C.c_uint8 * 31)(), (C.c_uint8 * 31)()
        root = C.c_int32(-2)
        rc = lib.cobel_pairwise_order(n, pos.ctypes.data_as(C.c_void_p), k, dst, src, C.byref(root))
        assert rc == 0
        slots = [np.float32(v) for v in vals]
        for t in range(max(0, k - 1)):
            slots[dst[t]] = np.float32(slots[dst[t]] + slots[src[t]])
        got = slots[root.value] if k else np.float32(0)
        assert (root.value >= 0) == (k > 0)
        want = np.sum(vec)
        assert got == want and (got != 0 or want == 0), (n, pos, vals, got, want)
        assert np.float32(got).tobytes() == np.float32(want).tobytes() or got == 0


def test_transition_lists_follow_generator_choice():
    """Rows of a dense sas that are distributions (interface/gridworld.py:119-123 draws from them):
    an entry too small to move the float64 cumulative sum is left out of the successor list (no
    draw can select it, and the list stays strictly increasing as cobel_world_set_transitions
    requires); a row that does not sum to 1 is the ValueError Generator.choice raises."""
    from cobel_amd.misc.gridworld_tools import transition_lists
    sas = np.zeros((2, 4, 2))
    sas[:, :, 0] = 1.0
    sas[0, 1] = [0.25, 0.75]
    sas[1, 2] = [1.0, 1e-300]          # valid for numpy: sums to 1 within tolerance
    off, states, cdf = transition_lists(sas)
    assert list(off[1:3] - off[0:2]) == [1, 2] and off[-1] == len(states) == len(cdf)
    lo = off[1 * 4 + 2]
    assert off[1 * 4 + 3] - lo == 1 and states[lo] == 0 and cdf[lo] == 1.0
    rng = np.random.default_rng(3)
    for _ in range(50):                # the same uniforms select the same states either way
        u = rng.random()
        c = np.cumsum(sas[1, 2])
        c /= c[-1]
        assert np.arange(2)[np.searchsorted(c, u, side='right')] == 0
    for p in range(8):
        seg = cdf[off[p]:off[p + 1]]
        assert (np.diff(seg) > 0).all() and seg[-1] == 1.0
    bad = sas.copy()
    bad[0, 0] = [0.5, 0.4]
    with pytest.raises(ValueError, match='do not sum to 1'):
        transition_lists(bad)
    with pytest.raises(ValueError):
        np.random.default_rng(0).choice(2, p=bad[0, 0])


def test_remove_obstructed_neighbors_prunes_edges_like_the_reference():
    """misc/topology_tools.py:472-505 without shapely: an edge whose straight line meets an obstacle
    polygon (grown by the buffer distance, boundary included) becomes a self-loop; the input graph is
    left untouched; z is ignored; polygons come as vertex lists or shapely-like objects."""
    from cobel_amd.misc import topology_tools as tt
    nodes, _ = tt.grid(4, (0.0, 3.0))                     # node n at (n % 4, 3 - n // 4)
    wall = [(1.4, 1.5), (1.6, 1.5), (1.6, 3.5), (1.4, 3.5)]   # between columns 1 and 2, two upper rows
    out = tt.remove_obstructed_neighbors(nodes, [wall])
    assert out['1']['neighbors'] == ['0', '1', '1', '5'] and out['2']['neighbors'] == ['2', '2', '3', '6']
    assert out['5']['neighbors'] == ['4', '1', '5', '9'] and out['6']['neighbors'] == ['6', '2', '7', '10']
    assert out['9']['neighbors'] == nodes['9']['neighbors']            # below the wall: untouched
    assert nodes['1']['neighbors'] == ['0', '1', '2', '5']             # deep copy
    changed = sum(a['neighbors'] != b['neighbors'] for a, b in zip(nodes.values(), out.values()))
    assert changed == 4
    # the buffer distance: a square 0.2 above the edge 9 - 10 obstructs it from exactly 0.2 on
    square = [(1.4, 1.2), (1.6, 1.2), (1.6, 1.4), (1.4, 1.4)]
    for buf, cut in ((0.0, False), (0.19, False), (0.2, True), (0.5, True)):
        o = tt.remove_obstructed_neighbors(nodes, [square], buf)
        assert (o['9']['neighbors'][2] == '9') == cut and (o['10']['neighbors'][0] == '10') == cut, buf
    # a node inside an obstacle loses every edge, and its neighbours their edges to it
    o = tt.remove_obstructed_neighbors(nodes, [[(-0.5, 2.5), (0.5, 2.5), (0.5, 3.5), (-0.5, 3.5)]])
    assert o['0']['neighbors'] == ['0'] * 4 and o['1']['neighbors'][0] == '1' and o['4']['neighbors'][1] == '4'
    # a hole: the edge inside the hole of a ring-shaped obstacle is free, the ones through the ring are not

    class Ring:      # what the function reads off a shapely polygon
        class _R:
            def __init__(self, c):
                self.coords = c
        exterior = _R([(0.5, 0.5), (2.5, 0.5), (2.5, 2.5), (0.5, 2.5), (0.5, 0.5)])
        interiors = [_R([(0.8, 0.8), (2.2, 0.8), (2.2, 2.2), (0.8, 2.2), (0.8, 0.8)])]

    o = tt.remove_obstructed_neighbors(nodes, [Ring])
    assert o['5']['neighbors'] == ['5', '5', '6', '9']     # 5 = (1, 2) inside the hole: 4 and 1 are outside
    assert o['6']['neighbors'][0] == '5' and o['9']['neighbors'][2] == '10'
    with pytest.raises(AssertionError):
        tt.remove_obstructed_neighbors(nodes, [wall], -1.0)
    # the pruned graph is an ordinary topology for the environment
    env_nodes = tt.remove_obstructed_neighbors(nodes, [wall])
    assert all(len(v['neighbors']) == 4 for v in env_nodes.values())


def test_analysis_match_and_state_coordinates():
    """cobel.analysis.behavior_spatial.match (:76-107) and cobel.analysis.utils (:8-53): the number
    of template states that line up with the sequence at every offset (counted by plain loops here,
    including the reference's padding rule: a -1 in the sequence matches wherever the template does
    not cover it), and state index -> (row, column)."""
    from cobel_amd.analysis import match, state_to_coordinates, states_to_coordinates
    r = np.random.default_rng(3)
    for k in range(200):
        n, m = int(r.integers(1, 40)), int(r.integers(1, 15))
        lo = -1 if k % 4 == 0 else 0
        seq, tem = r.integers(lo, 5, n), r.integers(lo, 5, m)
        want = np.zeros(n, dtype=np.int64)
        for t in range(n):
            for c in range(n):
                j = c - t
                covered = 0 <= j < m
                want[t] += int(tem[j] == seq[c]) if covered else int(seq[c] == -1)
        got = match(seq, tem)
        assert got.dtype == np.int64 and np.array_equal(got, want), (seq, tem)
    assert np.array_equal(match(np.array([1, 2, 3, 1, 2]), np.array([1, 2])), [2, 0, 0, 2, 0])
    assert match(np.zeros(0, dtype=int), np.array([1])).shape == (0,)
    assert state_to_coordinates(7, 5).tolist() == [1, 2] and state_to_coordinates(7, 5, False).tolist() == [2, 1]
    assert states_to_coordinates(np.array([0, 7, 24]), 5).tolist() == [[0, 0], [1, 2], [4, 4]]
    assert states_to_coordinates(np.array([7]), 5, y_first=False).tolist() == [[2, 1]]
    with pytest.raises(AssertionError):
        state_to_coordinates(-1, 5)
