"""CPU tests of the host logic and of the C-ABI surface (no compute calls: there is no GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


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
    assert lib.cobel_abi_version() == 1000
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r'\bT (cobel_\w+)', out))
    assert exported == declared


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the run structs have the size the C compiler gives them."""
    from cobel_amd import _lib
    src = ('#include "cobel_hip.h"\n#include <stdio.h>\n'
           'int main(){printf("%zu %zu\\n", sizeof(cobel_tab_run_t), sizeof(cobel_sr_run_t));}')
    exe = '/tmp/cobel_sizeof_%d' % os.getpid()
    subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe],
                   input=src.encode(), check=True)
    a, b = subprocess.check_output([exe]).split()
    os.remove(exe)
    assert int(a) == C.sizeof(_lib.TabRun) and int(b) == C.sizeof(_lib.SRRun)


def test_argument_errors_map_to_reference_exceptions():
    """Bad arguments are rejected before any HIP call, with the exception the reference raises."""
    from cobel_amd import _lib
    lib = _lib.lib()
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_world_create(None, None, None, None, None, 25, 1, 0, None))
    with pytest.raises(AssertionError):
        _lib.check(lib.cobel_tab_run(None, None, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_tab_query(1000000, 1, 10, None, None))
    with pytest.raises(NotImplementedError):
        _lib.check(lib.cobel_tab_query(25, 1, 63, None, None))
    lds = C.c_int32()
    _lib.check(lib.cobel_tab_query(1024, 1, 50, C.byref(lds), None))
    assert lds.value == 1024 * (16 + 8 + 4) + 2048   # Q, compact model, visit counters, hash
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
    from cobel_amd.monitor import EscapeLatencyMonitor
    k = golden('monitor_kat')
    mon = EscapeLatencyMonitor(len(k['steps']), int(k['max_steps']))
    for t in k['order']:
        mon.update({'trial': int(t), 'steps': int(k['steps'][t])})
    assert np.array_equal(mon.latency_trace, k['latency'], equal_nan=True)
    assert np.array_equal(mon.latency_trace_avg, k['latency_avg'], equal_nan=True)
    assert mon.get_trace() is mon.latency_trace


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
from cobel_amd.monitor import EscapeLatencyMonitor
rank = int(os.environ['RANK'])
dist.init_process_group('gloo')
# each rank owns half of 8 instances: per-trial latencies are rank-local partial sums
lat = np.array([[3, 5, 7], [4, 6, 8], [1, 1, 1], [9, 9, 9], [2, 4, 6], [0, 0, 0], [5, 5, 5], [7, 8, 9]])
mine = lat[rank * 4:(rank + 1) * 4]
mon = DeviceMonitors(torch.device('cpu'), 1, 4, occupancy=True)
mon.reserve(3)
mon.lat_sum += torch.as_tensor(mine.sum(axis=0)); mon.lat_cnt += 4
mon.reward_sum += float(rank + 1); mon.occupancy += rank + 1; mon.steps_done += 100 * (rank + 1)
el = EscapeLatencyMonitor(3, 10)
el.update_from_device(mon)            # all-reduces, then fills the trace with per-trial means
assert np.allclose(el.latency_trace, lat.mean(axis=0)), el.latency_trace
assert int(mon.steps_done.item()) == 300 and int(mon.occupancy.sum().item()) == 12
assert np.allclose(mon.reward_sum.numpy(), 3.0)
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_monitor_allreduce_two_ranks_gloo(tmp_path):
    """The N > 1 path's only collective (monitor reduction), world_size 2 on the gloo backend."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('ok' in o for o in outs)


def test_topology_builders_match_reference(golden):
    """cobel_amd.misc.topology_tools == the reference's linear_track / t_maze / grid as tables."""
    from cobel_amd.misc import topology_tools as tt
    k = golden('topology_kat')
    built = {'linear_10x2': tt.linear_track(10, 2, 1., 20., 'right'),
             'linear_5x1_left': tt.linear_track(5, 1, 0.5, 2., 'left'),
             't_maze_4_3_1': tt.t_maze(4, 3, 1),
             't_maze_3_2_2_left': tt.t_maze(3, 2, 2, 2.0, 3.0, 'left'),
             'grid_4x3': tt.grid((4, 3)), 'grid_5': tt.grid(5, (0.0, 2.0), 7.0, '12')}
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
