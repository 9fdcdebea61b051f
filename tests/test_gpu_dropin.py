"""Drop-in check: with ``cobel_amd.install_as_cobel()`` the import lines and call sequences of the
reference's own smoke tests and gridworld demos for this path run unchanged on the HIP library
(unit_tests/test_gridworld.py, test_dyna_q.py, test_q.py [Gridworld case], test_sr.py, test_sfma.py;
demo/gridworld/demo_dyna_q.py, demo_sfma.py without the Qt widget).  The reference's scripts
assert nothing beyond "runs"; here every simulation additionally has to leave finite tables and
the documented attribute surface."""
from itertools import product

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cobel():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    import cobel_amd
    cobel_amd.install_as_cobel()
    import cobel
    return cobel


def test_dyna_q_simulations(cobel):
    from cobel.agent import DynaQ
    from cobel.interface import Gridworld
    from cobel.misc.gridworld_tools import make_open_field
    from cobel.policy import EpsilonGreedy
    for use_test_policy, mask_actions in product([True, False], [True, False]):
        env = Gridworld(make_open_field(5, 5, 0, 1))
        policy_test = EpsilonGreedy(0.) if use_test_policy else None
        agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(), policy_test)
        agent.mask_actions = mask_actions
        agent.train(env, 5, 20, 32)
        agent.train(env, 5, 20, 32, True)
        agent.test(env, 5, 20)
        assert agent.current_trial == 15 and agent.Q.shape == (25, 4) and np.isfinite(agent.Q).all()
        assert agent.M.states.shape == (25, 4)


def test_q_and_sr_simulations(cobel):
    from cobel.agent import SR, QAgent
    from cobel.interface import Gridworld
    from cobel.misc.gridworld_tools import make_open_field
    from cobel.policy import EpsilonGreedy
    for use_test_policy in (True, False):
        env = Gridworld(make_open_field(5, 5, 0, 1))
        agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(),
                       EpsilonGreedy(0.) if use_test_policy else None)
        agent.train(env, 5, 20, 32)
        agent.test(env, 5, 20)
        assert agent.current_trial == 10 and np.isfinite(agent.Q).all()
    for use_test_policy, mask_actions in product([True, False], [True, False]):
        env = Gridworld(make_open_field(5, 5, 0, 1))
        agent = SR(env.observation_space, env.action_space, EpsilonGreedy(),
                   EpsilonGreedy(0.) if use_test_policy else None)
        agent.mask_actions = mask_actions
        agent.train(env, 5, 20)
        agent.test(env, 5, 20)
        assert agent.current_trial == 10
        assert np.isfinite(agent.predict_on_batch(np.arange(25))).all()


def test_sfma_simulations(cobel):
    """The 120 combinations of unit_tests/test_sfma.py:98-107."""
    from cobel.agent import SFMA
    from cobel.interface import Gridworld
    from cobel.memory import SFMAMemory
    from cobel.memory.utils import DR, SR, Euclidean
    from cobel.misc.gridworld_tools import make_gridworld
    from cobel.policy import EpsilonGreedy
    walls = [(3, 4), (4, 3), (8, 9), (9, 8), (13, 14), (14, 13)]
    metrics = ['DR', 'SR', 'Euclidean']
    modes = ['default', 'reverse', 'forward', 'dynamic', 'sweeping']
    for metric, mode, use_test_policy, mask_actions, random_replay in product(
            metrics, modes, [True, False], [True, False], [True, False]):
        gridworld = make_gridworld(5, 5, terminals=[4], rewards=np.array([[4, 10]]), goals=[4],
                                   invalid_transitions=walls)
        gridworld['starting_states'] = np.array([12])
        env = Gridworld(gridworld)
        if metric == 'DR':
            sim = DR(env.world['width'], env.world['height'], env.world['sas'], 0.9,
                     env.world['invalid_transitions'])
        elif metric == 'SR':
            sim = SR(env.world['sas'], 0.9)
        else:
            sim = Euclidean(env.world['width'], env.world['height'])
        memory = SFMAMemory(sim, env.world['states'], 4)
        agent = SFMA(env.observation_space, env.action_space, EpsilonGreedy(), memory,
                     EpsilonGreedy(0.) if use_test_policy else None)
        agent.M.mode = mode
        agent.mask_actions = mask_actions
        agent.random = random_replay
        agent.train(env, 5, 50, 32)
        agent.train(env, 5, 50, 32, True)
        agent.test(env, 5, 50)
        assert agent.current_trial == 15 and agent.M.mode == mode
        assert np.isfinite(agent.Q).all() and np.isfinite(agent.M.C).all()
        assert agent.M.C.sum() == agent.M.state[0, 0].item()   # one unit of strength per store


def test_demo_dyna_q_and_sfma_flows(cobel):
    """demo/gridworld/demo_dyna_q.py:36-56 and demo_sfma.py:30-75 (widget=None): monitors through
    on_trial_end, per-step hooks, Q readable afterwards."""
    from cobel.agent import SFMA, DynaQ
    from cobel.interface import Gridworld
    from cobel.memory import SFMAMemory
    from cobel.memory.utils import DR
    from cobel.misc.gridworld_tools import make_gridworld, make_open_field
    from cobel.monitor import EscapeLatencyMonitor
    from cobel.policy import EpsilonGreedy
    env = Gridworld(make_open_field(5, 5, 0, 1))
    el = EscapeLatencyMonitor(80, 50)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(), EpsilonGreedy(0.0),
                  custom_callbacks={'on_trial_end': [el.update]})
    agent.train(env, 50, 50, 32)
    agent.test(env, 30, 50)
    trace = np.asarray(el.get_trace())
    assert np.isfinite(trace[:80]).all() and trace[50:80].mean() < trace[:10].mean()

    walls = [(3, 4), (4, 3), (8, 9), (9, 8), (13, 14), (14, 13), (18, 19), (19, 18)]
    gridworld = make_gridworld(5, 5, terminals=[4], rewards=np.array([[4, 10]]), goals=[4],
                               invalid_transitions=walls)
    gridworld['starting_states'] = np.array([12])
    env = Gridworld(gridworld, seed=20260101)   # (seeded: the goal lies behind a five-cell corridor)
    steps_seen = []
    el = EscapeLatencyMonitor(60, 50)
    metric = DR(env.world['width'], env.world['height'], env.world['sas'], 0.9,
                env.world['invalid_transitions'])
    agent = SFMA(env.observation_space, env.action_space, EpsilonGreedy(),
                 SFMAMemory(metric, env.world['states'], 4),
                 custom_callbacks={'on_step_end': [lambda logs: steps_seen.append(logs['state'])],
                                   'on_trial_end': [el.update]})
    agent.M.mode = 'reverse'
    agent.mask_actions = True
    agent.train(env, 60, 50, 32)
    reached = bool((np.asarray(el.get_trace()) < 49).any())     # some trial ended at the goal
    assert len(steps_seen) > 60 and (agent.Q.max() > 0) == reached
    assert np.isfinite(agent.predict_on_batch(np.arange(25))).all()


@pytest.mark.parametrize('name', ['track_b0_f32', 'track_b8_f32', 'track5_b4_f32', 'hex5_b0_f32',
                                  'hex5_b8_f32', 'hex4_b70_f32'])
def test_qagent_on_topology_matches_reference(cobel, golden, name):
    """QAgent on pose observations (unit_tests/test_q.py "Topology", demo/topology/demo.py): the
    reference keys Q by tuple(pose); trajectory in node indices, TD errors, Q rows and
    predict_on_batch on poses match its float32 run bit for bit."""
    from conftest import SEED
    from cobel.agent import QAgent
    from cobel.interface import Topology
    from cobel.misc.topology_tools import linear_track
    from cobel.policy import EpsilonGreedy
    Z = golden('qagent_topology_traces')
    g = lambda k: Z['%s/%s' % (name, k)]      # noqa: E731
    inst, f32, trials, steps, B = [int(x) for x in g('cfg')]
    a, b, sp, rw = g('track')
    if str(g('side')) == 'hex':     # six actions per node: the general kernels
        from cobel.misc.topology_tools import hexagonal
        nodes, starts = {'hex5': hexagonal(5, (0.0, 2.0), 3.0, '7'), 'hex4': hexagonal(4)}[name[:4]]
    else:
        nodes, starts = linear_track(int(a), int(b), float(sp), float(rw), str(g('side')))
    env = Topology(nodes, starts, seed=SEED, instance_base=inst)
    assert int(env.action_space.n) == len(nodes[starts[0]]['neighbors']) == g('Q').shape[1]
    sarsn, tds, steps_log = [], [], []
    cbs = {'on_step_end': [lambda l: (sarsn.append((l['state'], l['action'], l['reward'],
                                                    l['next_state'], l['terminal'])),
                                      tds.append(l['td']))],
           'on_trial_end': [lambda l: steps_log.append(l['steps'])]}
    agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                   custom_callbacks=cbs)
    agent.train(env, trials, steps, B)
    arr = np.array(sarsn, dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(arr[:, col], g(key)), key
    if B == 0:
        # with replay the reference's logs['td'] is whatever update_q wrote LAST into the experience
        # dict, which also sits in the replay memory and may be re-sampled in the same step
        # (q.py:213-216, :353-354); the kernel reports the online TD error
        assert np.array_equal(np.array(tds, dtype=np.float64), g('td'))
    assert np.array_equal(steps_log, g('steps'))
    assert np.array_equal(np.asarray(agent.Q, dtype=np.float64), g('Q'))
    assert np.array_equal(agent.predict_on_batch(g('probe')).astype(np.float64), g('probe_q'))
    assert len(agent.M) == int(g('log_len'))     # (also at B = 0: q.py:213 appends regardless)
    qd = agent.Q_dict
    assert len(qd) == len(nodes) and all(len(k) == 6 for k in qd)
    with pytest.raises(KeyError):
        agent.predict_on_batch(np.full((1, 6), 123.0))
    # vectorised: instance `inst` of a batch behaves the same
    env2 = Topology(nodes, starts, n_envs=inst + 2, seed=SEED)
    ag2 = QAgent(env2.observation_space, env2.action_space, EpsilonGreedy(0.1))
    ag2.train(env2, trials, steps, B)
    assert np.array_equal(ag2.Q[inst].cpu().numpy().astype(np.float64), g('Q'))


class _Model:
    """The MLP of the reference's network demos (demo_dyna_dqn.py:29-46, demo_dqn.py:44-66)."""

    def __new__(cls, input_size, output_size):
        from torch import reshape
        from torch.nn import Linear, Module
        from torch.nn.functional import relu

        class Model(Module):
            def __init__(self):
                super().__init__()
                units = input_size if type(input_size) is int else int(np.prod(input_size))
                self.layer_dense_1 = Linear(in_features=units, out_features=64)
                self.layer_dense_2 = Linear(in_features=64, out_features=64)
                self.layer_output = Linear(in_features=64, out_features=output_size)
                self.double()

            def forward(self, layer_input):
                x = reshape(layer_input, (len(layer_input), -1))
                x = relu(self.layer_dense_1(x))
                x = relu(self.layer_dense_2(x))
                return self.layer_output(x)

        return Model()


def test_topology_and_sr_demo_flows(cobel):
    """demo/topology/demo.py (QAgent without replay on linear / grid / t-maze graphs, escape
    latency monitor) and demo/gridworld/demo_sr.py."""
    from cobel.agent import SR, QAgent
    from cobel.interface import Gridworld, Topology
    from cobel.misc.gridworld_tools import make_open_field
    from cobel.misc.topology_tools import grid, linear_track, t_maze
    from cobel.monitor import EscapeLatencyMonitor
    from cobel.policy import EpsilonGreedy
    from cobel.typing import CallbackDict, Node, NodeID  # noqa: F401
    for nodes, starting_nodes in (linear_track(10, 2, 1.0, 20, 'right'), grid(5, (0.0, 1.0)),
                                  t_maze(6, 3, 2, 1.0)):
        interface = Topology(nodes, starting_nodes, None, None)
        el_monitor = EscapeLatencyMonitor(120, 50, None)
        callbacks = {'on_trial_end': [el_monitor.update]}
        agent = QAgent(interface.observation_space, interface.action_space, EpsilonGreedy(0.1),
                       custom_callbacks=callbacks)
        agent.train(interface, 120, 50, 0)
        trace = np.asarray(el_monitor.get_trace())
        assert np.isfinite(trace).all() and trace[-20:].mean() <= trace[:20].mean()
    env = Gridworld(make_open_field(5, 5, 0, 1))
    el_monitor = EscapeLatencyMonitor(60, 50)
    agent = SR(env.observation_space, env.action_space, EpsilonGreedy(), EpsilonGreedy(0.0),
               custom_callbacks={'on_trial_end': [el_monitor.update]})
    agent.train(env, 40, 50)
    agent.test(env, 20, 50)
    assert np.isfinite(el_monitor.get_trace()).all()
    assert agent.predict_on_batch(np.arange(25)).shape == (25, 4)


def test_network_demo_flows(cobel):
    """demo/gridworld/demo_dyna_dqn.py, demo_dyna_dsr.py and demo/topology/demo_dqn.py with short
    schedules (widget None): constructors, keyword arguments, callbacks and the returned
    Q-function shapes as in the scripts."""
    from cobel.agent import DQN, DynaDQN, DynaDSR
    from cobel.interface import Gridworld, Topology
    from cobel.misc.gridworld_tools import make_open_field
    from cobel.misc.topology_tools import linear_track
    from cobel.monitor import EscapeLatencyMonitor
    from cobel.network import FlexibleTorchNetwork, TorchNetwork
    from cobel.policy import EpsilonGreedy
    trials_train, trials_test, steps = 6, 3, 20
    env = Gridworld(make_open_field(5, 5, 0, 1), widget=None)
    el_monitor = EscapeLatencyMonitor(trials_train + trials_test, steps, None)
    custom_callbacks = {'on_trial_end': [el_monitor.update],
                        'on_step_end': [env.update_visualization]}
    agent = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(),
                    TorchNetwork(_Model(25, 4)), gamma=0.8, policy_test=EpsilonGreedy(0.0),
                    custom_callbacks=custom_callbacks)
    agent.train(env, trials_train, steps, 32)
    agent.test(env, trials_test, steps)
    q = agent.predict_on_batch(np.arange(25))
    assert q.shape == (25, 4) and np.isfinite(q).all()
    assert np.isfinite(el_monitor.get_trace()).all()

    env = Gridworld(make_open_field(5, 5, 0, 1), widget=None)
    el_monitor = EscapeLatencyMonitor(trials_train + trials_test, steps, None)
    agent = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(),
                    TorchNetwork(_Model(25, 25)), TorchNetwork(_Model(25, 1)), gamma=0.8,
                    policy_test=EpsilonGreedy(0.0),
                    custom_callbacks={'on_trial_end': [el_monitor.update],
                                      'on_step_end': [env.update_visualization]})
    agent.train(env, trials_train, steps, 32)
    agent.test(env, trials_test, steps)
    q = agent.predict_on_batch(np.arange(25))
    assert q.shape == (25, 4) and np.isfinite(q).all()

    nodes, starting_nodes = linear_track(10, 2, 1.0, 20, 'right')
    interface = Topology(nodes, starting_nodes, None, None)
    el_monitor = EscapeLatencyMonitor(9, 30, None)
    assert type(interface.observation_space.shape) is tuple
    model = FlexibleTorchNetwork(_Model(interface.observation_space.shape, 4))
    agent = DQN(interface.observation_space, interface.action_space, EpsilonGreedy(0.3), model,
                policy_test=EpsilonGreedy(0.0), custom_callbacks={'on_trial_end': [el_monitor.update]})
    agent.train(interface, 6, 30)
    agent.test(interface, 3, 30)
    assert np.isfinite(el_monitor.get_trace()).all()


def test_dynaq_memory_store_and_retrieve_batch_match_reference(cobel):
    """DynaQMemory.store / retrieve / retrieve_batch as host calls (memory/dyna_q.py:77-157): the
    reference's own sequence of stores and batch draws (float32 tables, the generator fed from the
    memory stream of instance 3) reproduced through the device table — and interleaved with a
    kernel launch, which continues the same stream and sees the stored records."""
    import os
    import torch
    from conftest import SEED
    from cobel.agent import DynaQ
    from cobel.interface import Gridworld
    from cobel.memory.dyna_q import DynaQMemory
    from cobel.misc.gridworld_tools import make_open_field
    from cobel.policy import EpsilonGreedy
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                             'dynaq_memory_kat.npz'))
    inst = int(k['f32/instance'])
    M = DynaQMemory(25, 4)
    M._bind(1, None, SEED, inst)          # the stream the fixture was generated with
    stores, batches, row = iter(k['f32/stores']), k['f32/batches'], 0
    for op in k['f32/ops']:
        if op == 0:
            s, a, r, ns, nt = next(stores)
            M.store({'state': int(s), 'action': int(a), 'reward': float(r), 'next_state': int(ns),
                     'terminal': int(nt)})
        else:
            batch = M.retrieve_batch(int(op))
            assert isinstance(batch, list) and len(batch) == op
            got = np.array([[e['state'], e['action'], e['reward'], e['next_state'], e['terminal']]
                            for e in batch], dtype=np.float64)
            assert np.array_equal(got, batches[row: row + op])
            row += int(op)
    assert row == len(batches)
    assert np.array_equal(M.rewards.astype(np.float64), k['f32/rewards'])
    assert np.array_equal(M.states, k['f32/states'])
    assert np.array_equal(M.terminals, k['f32/terminals'])
    last = k['f32/stores'][-1]
    one = M.retrieve(int(last[0]), int(last[1]))
    assert [float(one['reward']), one['next_state'], one['terminal']] == list(k['f32/retrieve_last'])
    # the digest the planning kernel reads stayed in step with the table
    fresh = torch.empty_like(M.index)
    from cobel_amd import _lib
    _lib.check(_lib.lib().cobel_model_index_build(_lib.ptr(M.table), _lib.ptr(fresh), 1, 25, None))
    assert torch.equal(fresh, M.index)

    # host calls between launches: the agent's planning continues the stream after the host draw
    def run(host_draw):
        env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=1, seed=SEED)
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        ag.train(env, 2, 20, 8)
        if host_draw:
            b = ag.M.retrieve_batch(8)
            assert len(b) == 8 and all(0 <= e['state'] < 25 for e in b)
        ag.train(env, 2, 20, 8)
        return ag
    a, b = run(False), run(True)
    assert int(b.M.counter[0].item()) == int(a.M.counter[0].item()) + 1
    assert not np.array_equal(a.Q, b.Q)       # the host draw consumed one batch of the stream


def test_dqn_single_instance_honours_hooks_and_stop(cobel, golden):
    """agent/dqn.py:170-215 with one environment: step hooks fire around every step with the
    reference's logs keys, trial hooks after every trial, `agent.stop` ends the session at the next
    trial boundary — and neither mode changes the outcome (same transitions and weights as the
    hook-free run, which takes the two fused kernels)."""
    import torch
    from conftest import SEED
    from cobel.agent import DQN
    from cobel.interface import Topology
    from cobel.misc.topology_tools import linear_track
    from cobel.network import TorchNetwork
    from cobel.policy import EpsilonGreedy
    nodes, starts = linear_track(10, 2, 1., 20., 'right')

    def run(callbacks):
        torch.manual_seed(11)
        env = Topology(nodes, starts, seed=SEED, instance_base=4)
        ag = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                 TorchNetwork(_Model(6, 4)), gamma=0.8, custom_callbacks=callbacks)
        ag.train(env, 4, 20, 32)
        torch.cuda.synchronize()
        return ag

    steps, begins, trials = [], [], []
    plain = run(None)
    hooked = run({'on_step_begin': [lambda l: begins.append(l['step'])],
                  'on_step_end': [lambda l: steps.append(dict(l))],
                  'on_trial_end': [lambda l: trials.append((l['trial'], l['steps'], l['trial_reward']))]})
    per_trial = run({'on_trial_end': [lambda l: None]})
    assert plain.fused_steps > 0 and hooked.fused_steps == 0 and per_trial.fused_steps > 0
    n = int(plain.M.size[0].item())
    for other in (hooked, per_trial):
        assert int(other.M.size[0].item()) == n and other.current_trial == 4
        assert torch.equal(other.M.actions[0, :n], plain.M.actions[0, :n])
        assert torch.equal(other.M.next_states[0, :n], plain.M.next_states[0, :n])
        assert torch.equal(other.monitors.lat_sum[:4], plain.monitors.lat_sum[:4])
        for a, b in zip(other._online.get_weights(0), plain._online.get_weights(0)):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    assert len(steps) == n == len(begins) and [t[0] for t in trials] == [0, 1, 2, 3]
    for key in ('trial', 'trial_session', 'step', 'state', 'action', 'reward', 'next_state',
                'terminal', 'trial_reward', 'replay', 'agent'):
        assert key in steps[0], key
    assert steps[0]['state'].shape == (6,) and isinstance(steps[0]['action'], int)
    assert [t[1] for t in trials] == plain.monitors.lat_sum[:4].cpu().numpy().tolist()
    # agent.stop set by a trial hook ends the session after that trial
    seen = []

    def stop_after_two(logs):
        seen.append(logs['trial'])
        if logs['trial'] == 1:
            logs['agent'].stop = True
    stopped = run({'on_trial_end': [stop_after_two]})
    assert seen == [0, 1] and stopped.current_trial == 2
    stopped2 = run({'on_trial_end': [stop_after_two], 'on_step_end': [lambda l: None]})
    assert stopped2.current_trial == 2


def _offline_obs(kind, nodes):
    """unit_tests/test_topology.py:14-57 / test_q.py:24-41: every pose observed as two copies."""
    from cobel.spaces import Box, Dict, Tuple
    if kind == 'List':
        obs = {node['pose']: [np.array(node['pose']), np.array(node['pose'])] for node in nodes.values()}
        return obs, Tuple([Box(-np.inf, np.inf, (6,)), Box(-np.inf, np.inf, (6,))])
    obs = {node['pose']: {'1': np.array(node['pose']), '2': np.array(node['pose'])}
           for node in nodes.values()}
    return obs, Dict({'1': Box(-np.inf, np.inf, (6,)), '2': Box(-np.inf, np.inf, (6,))})


def test_topology_with_prerendered_observations(cobel):
    """unit_tests/test_topology.py:60-110 in all three observation modes: pose, list and dictionary
    observations of an OfflineSimulator (interface/simulator/offline.py:51-72), the same known
    trajectory in each; vectorised, the gathered components are device tensors."""
    from cobel.interface import OfflineSimulator, Topology
    from cobel.misc.topology_tools import t_maze
    from cobel.spaces import Box, Dict, Discrete, Tuple
    nodes, nodes_starting = t_maze(4, 3, 1)
    for kind in (None, 'List', 'Dict'):
        simulator = None if kind is None else OfflineSimulator(*_offline_obs(kind, nodes))
        env = Topology(nodes, nodes_starting, simulator)
        assert isinstance(env.action_space, Discrete) and env.action_space.n == 4
        if kind is None:
            assert isinstance(env.observation_space, Box) and env.observation_space.shape == (6,)
        elif kind == 'List':
            assert isinstance(env.observation_space, Tuple)
            assert [s.shape for s in env.observation_space.spaces] == [(6,), (6,)]
        else:
            assert isinstance(env.observation_space, Dict)
            assert env.observation_space.spaces['1'].shape == (6,)
        obs, _ = env.reset()
        assert env.current_node == '10'
        pose = np.array(nodes['10']['pose'])
        if kind == 'List':
            assert isinstance(obs, list) and all(np.array_equal(o, pose) for o in obs)
        elif kind == 'Dict':
            assert sorted(obs) == ['1', '2'] and all(np.array_equal(o, pose) for o in obs.values())
        states, rewards, terminals = [], [], []
        for action in [1, 1, 1, 1, 1, 2, 2, 2]:
            obs, reward, terminal, _, _ = env.step(action)
            states.append(env.current_node)
            rewards.append(reward)
            terminals.append(terminal)
            here = np.array(nodes[env.current_node]['pose'])
            got = obs if kind is None else (obs[0] if kind == 'List' else obs['2'])
            assert np.array_equal(got, here)
        assert states == ['9', '8', '7', '3', '3', '4', '5', '6']
        assert rewards == [0.] * 7 + [1.] and terminals == [False] * 7 + [True]
    # a pose without an observation is a KeyError (the reference raises it at the first visit)
    obs, space = _offline_obs('Dict', nodes)
    del obs[nodes['5']['pose']]
    with pytest.raises(KeyError):
        Topology(nodes, nodes_starting, OfflineSimulator(obs, space))
    # vectorised: rows gathered on the device
    env = Topology(nodes, nodes_starting, OfflineSimulator(*_offline_obs('Dict', nodes)), n_envs=7, seed=3)
    obs, _ = env.reset()
    assert obs['1'].shape == (7, 6) and obs['1'].is_cuda
    want = np.array([nodes[env.ids[int(s)]]['pose'] for s in env.state.cpu().numpy()])
    assert np.array_equal(obs['2'].cpu().numpy(), want)


def test_q_simulations_on_dictionary_observations(cobel):
    """unit_tests/test_q.py:46-87, the "Topology-Dict" case with and without a test policy."""
    from cobel.agent import QAgent
    from cobel.interface import OfflineSimulator, Topology
    from cobel.misc.topology_tools import linear_track
    from cobel.policy import EpsilonGreedy
    for use_test_policy in (True, False):
        nodes, starting_nodes = linear_track(10, 2, 1, 20)
        simulator = OfflineSimulator(*_offline_obs('Dict', nodes))
        env = Topology(nodes, starting_nodes, simulator)
        agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(),
                       EpsilonGreedy(0.) if use_test_policy else None)
        agent.train(env, 5, 20, 32)
        agent.test(env, 5, 20)
        assert agent.current_trial == 10 and np.isfinite(agent.Q).all()
        assert all(len(k) == 12 for k in agent.Q_dict)


@pytest.mark.parametrize('name', ['trackdict_b8_f32', 'trackdict_b0_f32'])
def test_qagent_on_dictionary_observations_matches_reference(cobel, golden, name):
    """QAgent on an OfflineSimulator-backed Topology: the reference keys Q by the concatenated
    observation components (agent/q.py:156-158); trajectory, TD errors and Q rows of its float32
    run, bit for bit."""
    from conftest import SEED
    from cobel.agent import QAgent
    from cobel.interface import OfflineSimulator, Topology
    from cobel.misc.topology_tools import linear_track
    from cobel.policy import EpsilonGreedy
    Z = golden('qagent_topology_traces')
    g = lambda k: Z['%s/%s' % (name, k)]      # noqa: E731
    inst, f32, trials, steps, B = [int(x) for x in g('cfg')]
    a, b, sp, rw = g('track')
    nodes, starts = linear_track(int(a), int(b), float(sp), float(rw), str(g('side')))
    env = Topology(nodes, starts, OfflineSimulator(*_offline_obs('Dict', nodes)), seed=SEED,
                   instance_base=inst)
    sarsn, tds, steps_log = [], [], []
    cbs = {'on_step_end': [lambda l: (sarsn.append((l['state'], l['action'], l['reward'],
                                                    l['next_state'], l['terminal'])),
                                      tds.append(l['td']))],
           'on_trial_end': [lambda l: steps_log.append(l['steps'])]}
    agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1), custom_callbacks=cbs)
    agent.train(env, trials, steps, B)
    arr = np.array(sarsn, dtype=np.float64)
    for col, key in enumerate(('state', 'action', 'reward', 'next_state', 'nonterminal')):
        assert np.array_equal(arr[:, col], g(key)), key
    if B == 0:
        assert np.array_equal(np.array(tds, dtype=np.float64), g('td'))
    assert np.array_equal(steps_log, g('steps'))
    assert np.array_equal(np.asarray(agent.Q, dtype=np.float64), g('Q')) and g('Q').max() > 0
    probe = [{'1': p, '2': p} for p in g('probe')]
    assert np.array_equal(agent.predict_on_batch(probe).astype(np.float64), g('probe_q'))
    assert len(agent.M) == int(g('log_len'))
    assert all(len(k) == 12 for k in agent.Q_dict)


def test_prerendered_observations_keep_their_dtype_and_aliases_are_refused(cobel):
    """interface/topology.py:174-193 hands out the stored observation objects themselves: uint8
    images come back as uint8, float32 vectors as float32 (one env: NumPy, several: fresh device
    tensors).  Two nodes with the SAME observation would share a table row in the reference
    (agent/q.py:150-158 keys Q by the observation): refused, not silently learned per node."""
    import torch
    from cobel.agent import QAgent
    from cobel.interface import OfflineSimulator, Topology
    from cobel.misc.topology_tools import linear_track
    from cobel.policy import EpsilonGreedy
    from cobel.spaces import Box, Dict
    nodes, starts = linear_track(4, 2, 1., 5., 'right')
    obs = {n['pose']: {'image': np.full((3, 3), k, dtype=np.uint8),
                       'vec': np.array(n['pose'], dtype=np.float32)}
           for k, n in enumerate(nodes.values())}
    space = Dict({'image': Box(0, 255, (3, 3), np.uint8), 'vec': Box(-np.inf, np.inf, (6,), np.float32)})
    env = Topology(nodes, starts, OfflineSimulator(obs, space), seed=5)
    o, _ = env.reset()
    assert o['image'].dtype == np.uint8 and o['vec'].dtype == np.float32 and o['image'].shape == (3, 3)
    many = Topology(nodes, starts, OfflineSimulator(obs, space), n_envs=5, seed=5)
    first, _ = many.reset()
    assert first['image'].dtype == torch.uint8 and first['vec'].dtype == torch.float32
    kept = first['vec'].clone()
    many.step(torch.zeros(5, dtype=torch.int64, device=first['vec'].device))
    assert torch.equal(first['vec'], kept)            # not a view of the gather buffer
    QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1)).train(env, 2, 5, 0)
    # node 1 renders like node 0
    poses = [n['pose'] for n in nodes.values()]
    obs[poses[1]] = obs[poses[0]]
    aliased = Topology(nodes, starts, OfflineSimulator(obs, space), seed=5)
    with pytest.raises(NotImplementedError, match='share their observation'):
        QAgent(aliased.observation_space, aliased.action_space, EpsilonGreedy(0.1)).train(aliased, 1, 5, 0)


def test_pickled_worlddict_steps_and_learns_like_the_oracle(cobel):
    """A WorldDict as the reference's gridworld editor pickles it (misc/gridworld_gui.py:203-239),
    loaded with load_world: cobel_env_step walks its transition table (every (state, action) pair
    against the pickled dense `sas`), and a vectorised Dyna-Q run on it leaves the Q tables and escape
    latencies of the NumPy oracle driven by the same streams."""
    import os
    import pickle
    import torch
    from conftest import GOLDEN, SEED
    from cobel.agent import DynaQ
    from cobel.interface import Gridworld
    from cobel.misc.gridworld_tools import load_world
    from cobel.policy import EpsilonGreedy
    from oracle import philox, ref_loop
    path = os.path.join(GOLDEN, 'double_t_maze_2_1.pkl')
    world = load_world(path)
    with open(path, 'rb') as fh:
        raw = pickle.load(fh)
    S = int(raw['states'])
    sas = np.asarray(raw['sas'])
    # env.step over every pair
    env = Gridworld(world, n_envs=S * 4, seed=3)
    pairs = np.stack(np.meshgrid(np.arange(S), np.arange(4), indexing='ij'), -1).reshape(-1, 2)
    env.state.copy_(torch.as_tensor(pairs[:, 0], device=env.device).to(torch.int32))
    ns, r, done, _, _ = env.step(torch.as_tensor(pairs[:, 1], device=env.device))
    want = sas[pairs[:, 0], pairs[:, 1]].argmax(axis=1)
    assert np.array_equal(env.state.cpu().numpy(), want)
    assert np.array_equal(r.cpu().numpy(), np.asarray(raw['rewards'])[want].astype(np.float32))
    assert np.array_equal(done.cpu().numpy(), np.asarray(raw['terminals'])[want].astype(bool))
    # Dyna-Q on it against the oracle
    n, trials, steps, batch = 48, 5, 40, 16
    env = Gridworld(world, n_envs=n, seed=SEED)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.2))
    agent.track_instances = True
    agent.train(env, trials, steps, batch)
    torch.cuda.synchronize()
    q, lat = agent.Q.cpu().numpy(), agent.monitors.lat_trace.cpu().numpy()
    tabs = world.compact()
    for i in (0, 11, 47):
        renv = ref_loop.RefGridworld(tabs, philox.TapeRNG(SEED, i, philox.STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(0.2, philox.TapeRNG(SEED, i, philox.STREAM_POLICY))
        ref = ref_loop.RefDynaQ(S, 4, pol, philox.TapeRNG(SEED, i, philox.STREAM_MEMORY), dtype=np.float32)
        tr = ref_loop.new_trace()
        ref.train(renv, trials, steps, batch, trace=tr)
        assert np.array_equal(lat[i, :trials], np.array(tr['steps'])), i
        assert np.array_equal(q[i], ref.Q), i


def test_agent_entry_points_check_the_agent_kind():
    """SURVEY.md section 8b's `cobel_dynaq_run` / `cobel_q_run`: cobel_tab_run with run->agent
    checked (the Python classes call them: every Dyna-Q / QAgent test goes through them) — the wrong
    kind is COBEL_E_ARG and launches nothing."""
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=8, seed=3)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    agent.train(env, 2, 10, 8)
    steps = agent.env_steps()
    assert steps > 0
    seen = {}
    real = _lib.lib().cobel_dynaq_run

    def spy(world, run, stream):        # the very struct the class hands over, to the OTHER entry
        seen['rc'] = _lib.lib().cobel_q_run(world, run, stream)
        seen['msg'] = _lib.lib().cobel_last_error().decode()
        return real(world, run, stream)

    lib = _lib.lib()
    try:
        lib.cobel_dynaq_run = spy
        agent.train(env, 1, 10, 8)
    finally:
        lib.cobel_dynaq_run = real
    assert seen['rc'] == _lib.E_ARG and 'COBEL_AGENT_Q' in seen['msg']
    assert agent.env_steps() > steps
