#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REAL reference.

Runs only in the build container (needs /root/reference); the .npz files it
writes are committed and are the only thing that travels to the GPU box.

The reference targets Python >= 3.11 with gymnasium/PyQt6/pyqtgraph installed;
this container has Python 3.10 and none of them, so a small import shim
 provides the typing names and inert stand-ins for the
GUI / space modules.  No reference arithmetic is replaced: worlds, env, policy,
memory and agents below are the reference's own classes, driven by
``oracle.philox.TapeRNG`` so that they consume the build's Philox streams.

    python tests/golden/gen_golden.py            # rewrite every fixture
"""
from __future__ import annotations

import importlib.metadata as md
import os
import sys
import types
import typing

import numpy as np
import typing_extensions as te

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
# Where the fixtures are written: this directory, or a scratch directory for the test that
# re-generates them and compares with the committed files (tests/test_golden_regen.py).
OUT = os.environ.get('COBEL_GOLDEN_OUT', HERE)


def _out(name: str) -> str:
    return os.path.join(OUT, name)


def load_reference():
    typing.NotRequired, typing.Self = te.NotRequired, te.Self
    real = md.distribution
    md.distribution = (
        lambda n: types.SimpleNamespace(version='3.0.1') if n == 'cobel' else real(n)
    )
    gym, sp = types.ModuleType('gymnasium'), types.ModuleType('gymnasium.spaces')

    class Space:
        pass

    class Discrete(Space):
        def __init__(self, n):
            self.n = np.int64(n)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float64):
            self.low, self.high = np.asarray(low), np.asarray(high)
            self.shape = tuple(shape) if shape is not None else self.low.shape
            self.dtype = dtype

    class Dict(Space, dict):
        def __init__(self, d=None):
            dict.__init__(self, d or {})
            self.spaces = self

    class Tuple(Space):
        def __init__(self, s):
            self.spaces = list(s)

    class Env:
        pass

    for k, v in dict(Space=Space, Discrete=Discrete, Box=Box, Dict=Dict, Tuple=Tuple).items():
        setattr(sp, k, v)
    gym.spaces, gym.Space, gym.Env = sp, Space, Env
    sys.modules.update({'gymnasium': gym, 'gymnasium.spaces': sp})

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return type(k, (), {})

    for name in ['pyqtgraph', 'pyqtgraph.Qt', 'PyQt6', 'PyQt6.QtGui', 'PyQt6.QtCore',
                 'PyQt6.QtWidgets', 'shapely', 'shapely.ops', 'shapely.affinity', 'cv2']:
        m = _Any(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.path.insert(0, '/root/reference/src')
    import cobel  # noqa: F401
    return cobel


load_reference()

from cobel.agent import DynaQ  # noqa: E402
from cobel.agent.q import QAgent  # noqa: E402
from cobel.agent.sr import SR  # noqa: E402
from cobel.analysis.behavior_spatial import get_occupancy_map  # noqa: E402
from cobel.interface import Gridworld  # noqa: E402
from cobel.misc import gridworld_tools as gt  # noqa: E402
from cobel.monitor import EscapeLatencyMonitor  # noqa: E402
from cobel.policy import EpsilonGreedy  # noqa: E402

from oracle.philox import (STREAM_ENV, STREAM_MEMORY, STREAM_POLICY,  # noqa: E402
                           TapeRNG)

SEED = 0xC0BE1


def compact(world) -> dict:
    """Compact tables of a reference WorldDict (argmax over the one-hot sas)."""
    return dict(
        height=np.int64(world['height']), width=np.int64(world['width']),
        next=np.argmax(world['sas'], axis=2).astype(np.uint16),
        reward=np.asarray(world['rewards'], dtype=np.float64),
        terminal=np.asarray(world['terminals']).astype(np.uint8),
        starts=np.asarray(world['starting_states']).astype(np.uint16),
        coordinates=np.asarray(world['coordinates'], dtype=np.float64),
        invalid_states=np.asarray(sorted(world['invalid_states']), dtype=np.int64),
    )


def walls_8x8():
    inv_states = [10, 11, 12, 20, 28, 36, 44, 45, 50]
    inv_trans = [(0, 1), (1, 0), (62, 63), (63, 62), (33, 34)]
    return gt.make_gridworld(
        8, 8, terminals=[7, 56], rewards=np.array([[7, 1.0], [56, -0.5], [30, 0.25]]),
        goals=[7], invalid_states=inv_states, invalid_transitions=inv_trans,
        starting_states=[63, 32, 3, 27],
    )


def maze_cells(seed: int, h: int = 32, w: int = 32, p: float = 0.20):
    """SURVEY.md §8d C3 obstacle recipe: wall cells ~ Bernoulli(p), goal 0,
    redrawn until the goal is reachable from >= 50 % of the free cells."""
    rng = np.random.default_rng(seed)
    while True:
        wall = rng.random(h * w) < p
        wall[0] = False
        free = ~wall
        seen = np.zeros(h * w, dtype=bool)
        seen[0] = True
        stack = [0]
        while stack:
            s = stack.pop()
            y, x = divmod(s, w)
            for ny, nx in ((y, x - 1), (y - 1, x), (y, x + 1), (y + 1, x)):
                if 0 <= ny < h and 0 <= nx < w:
                    t = ny * w + nx
                    if free[t] and not seen[t]:
                        seen[t] = True
                        stack.append(t)
        if seen.sum() >= 0.5 * free.sum():
            return [int(i) for i in np.flatnonzero(wall)]


def gen_worlds():
    out = {}
    worlds = {
        'open_5x5': gt.make_open_field(5, 5, 0, 1),
        'kat_5x5': gt.make_gridworld(5, 5, [0], np.array([[0, 10.]]), starting_states=[24]),
        'open_4x7_goal9': gt.make_open_field(4, 7, 9, 2.5),
        'empty_3x3': gt.make_empty_field(3, 3),
        'walls_8x8': walls_8x8(),
        'windy_7x10_up': gt.make_windy_gridworld(
            7, 10, np.array([0, 0, 0, 1, 1, 1, 2, 2, 1, 0]), 37, 1.0, 'up'),
        'windy_5x6_down': gt.make_windy_gridworld(
            5, 6, np.array([0, 1, 2, 1, 0, 3]), 3, 1.0, 'down'),
        'open_32x32': gt.make_open_field(32, 32, 0, 1),
        'open_32x32_near': dict(gt.make_open_field(32, 32, 0, 1),
                                starting_states=np.array([1, 2, 32, 33, 34, 64, 65, 66])),
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
    }
    for seed in (1234, 1235):
        walls = maze_cells(seed)
        free_nt = [s for s in range(1024) if s not in set(walls) and s != 0]
        worlds['maze_32x32_%d' % seed] = gt.make_gridworld(
            32, 32, terminals=[0], rewards=np.array([[0, 1.0]]), goals=[0],
            invalid_states=walls, starting_states=free_nt)
    for name, w in worlds.items():
        for k, v in compact(w).items():
            out['%s/%s' % (name, k)] = v
    np.savez_compressed(_out('worlds.npz'), **out)
    # a WorldDict as the reference's gridworld editor pickles it (misc/gridworld_gui.py:225)
    import pickle
    with open(_out('double_t_maze_2_1.pkl'), 'wb') as fh:
        pickle.dump(dict(gt.make_double_t_maze(2, 1)), fh, protocol=4)
    for k, v in compact(gt.make_double_t_maze(2, 1)).items():
        out['double_t_maze_2_1/%s' % k] = v
    np.savez_compressed(_out('worlds.npz'), **out)
    return worlds


def gen_gridworld_kat():
    """unit_tests/test_gridworld.py:16-41 replayed through the reference."""
    world = gt.make_gridworld(5, 5, [0], np.array([[0, 10.]]), starting_states=[24])
    env = Gridworld(world)
    state, _ = env.reset()
    actions = [0, 0, 0, 0, 0, 1, 1, 1, 1]
    states, rewards, terms = [], [], []
    for a in actions:
        s, r, t, _, _ = env.step(a)
        states.append(s), rewards.append(r), terms.append(t)
    assert states == [23, 22, 21, 20, 20, 15, 10, 5, 0]
    assert rewards == [0] * 8 + [10.] and terms == [False] * 8 + [True]
    np.savez_compressed(
        _out('gridworld_kat.npz'), start=np.int64(state),
        actions=np.array(actions), states=np.array(states),
        rewards=np.array(rewards, dtype=np.float64), terminals=np.array(terms),
        n_obs=np.int64(env.observation_space.n), n_act=np.int64(env.action_space.n))


def gen_eps_greedy():
    """policy/greedy.py:40-88: probabilities and injected-u selections."""
    rows = []
    vs = [
        [0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [0.5, 0.5, 0, 0], [0.1, 0.7, 0.7, 0.2],
        [-1, -1, -2, -1], [3, 3, 3, 2], [1e-8, 0, 1e-8, 0], [0.25, 0.5, 0.75, 1.0],
        [-0.5, -0.25, -0.125, -1.0],
    ]
    masks = [None, [1, 1, 1, 1], [1, 0, 1, 0], [0, 1, 1, 1], [0, 0, 0, 1], [1, 1, 0, 0]]
    us = [0.0, 0.024999, 0.025, 0.0250001, 0.05, 0.1, 0.25, 0.3, 0.5, 0.74999, 0.75,
          0.9, 0.925, 0.95, 0.975, 0.999999999, 1.0 - 2.0 ** -53]
    for dt in (np.float64, np.float32):
        for eps in (0.0, 0.1, 0.3, 1.0):
            pol = EpsilonGreedy(eps, rng=TapeRNG(SEED, 0, STREAM_POLICY))
            for v in vs:
                v = np.array(v, dtype=dt)
                for m in masks:
                    mm = None if m is None else np.array(m, dtype=bool)
                    p = pol.get_action_probs(v, mm)
                    for u in us:
                        pol.rng.random = lambda size=None, u=u: u
                        a = int(pol.select_action(v, mm))
                        rows.append((dt == np.float32, eps, *[float(x) for x in v],
                                     0xF if m is None else sum(b << i for i, b in enumerate(m)),
                                     u, a, *p))
    rows = np.array(rows, dtype=np.float64)
    # six values per row (hexagonal Topology: action space = 6 neighbours, topology.py:110-112)
    rows6 = []
    vs6 = [[0] * 6, [0, 0, 1, 0, 0, 0], [2, 1, 2, 0, 2, -1], [0.5, 0.25, 0.5, 0.5, 0.125, 0.5],
           [-1, -2, -3, -1, -1, -4], [1e-8, 0, 0, 1e-8, 0, 0], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]]
    masks6 = [None, [1] * 6, [1, 0, 1, 0, 1, 0], [0, 0, 0, 0, 0, 1], [0, 1, 1, 1, 1, 0]]
    us6 = [0.0, 0.016, 1 / 60., 0.0167, 0.05, 1 / 6., 0.1667, 0.25, 1 / 3., 0.5, 0.65, 2 / 3.,
           0.8, 5 / 6., 0.834, 0.95, 0.999999999, 1.0 - 2.0 ** -53]
    for eps in (0.0, 0.1, 0.3, 1.0):
        pol = EpsilonGreedy(eps, rng=TapeRNG(SEED, 0, STREAM_POLICY))
        for v in vs6:
            v = np.array(v, dtype=np.float32)
            for m in masks6:
                mm = None if m is None else np.array(m, dtype=bool)
                p = pol.get_action_probs(v, mm)
                for u in us6:
                    pol.rng.random = lambda size=None, u=u: u
                    a = int(pol.select_action(v, mm))
                    rows6.append((eps, *[float(x) for x in v],
                                  0x3F if m is None else sum(b << i for i, b in enumerate(m)),
                                  u, a, *p))
    np.savez_compressed(_out('eps_greedy_kat.npz'), rows=rows, columns=np.array(
        ['is_f32', 'eps', 'v0', 'v1', 'v2', 'v3', 'mask', 'u', 'action', 'p0', 'p1', 'p2', 'p3']),
        rows6=np.array(rows6, dtype=np.float64), columns6=np.array(
        ['eps'] + ['v%d' % i for i in range(6)] + ['mask', 'u', 'action'] + ['p%d' % i for i in range(6)]))


class Tracer:
    """Collects the per-step experience the reference hands to its callbacks."""

    def __init__(self, agent=None):
        self.agent = agent
        self.sarsn, self.td, self.steps, self.trial_reward, self.Q_trial = [], [], [], [], []
        self.q = []

    def on_step_end(self, logs):
        first = lambda x: x[0] if isinstance(x, tuple) else x  # QAgent keys states as tuples
        self.sarsn.append((first(logs['state']), logs['action'], logs['reward'],
                           first(logs['next_state']), logs['terminal']))
        td = logs.get('td', 0.0)  # test() logs carry no TD error
        self.td.append(float(td) if np.ndim(td) == 0 else 0.0)

    def on_trial_end(self, logs):
        self.steps.append(logs['steps'])
        self.trial_reward.append(float(logs['trial_reward']))
        if self.agent is not None and hasattr(self.agent, 'Q') and isinstance(self.agent.Q, np.ndarray):
            self.Q_trial.append(np.array(self.agent.Q, dtype=np.float64))

    def callbacks(self):
        return {'on_step_end': [self.on_step_end], 'on_trial_end': [self.on_trial_end]}

    def pack(self) -> dict:
        a = np.array(self.sarsn, dtype=np.float64).reshape(-1, 5)
        d = dict(
            state=a[:, 0].astype(np.int16), action=a[:, 1].astype(np.int8), reward=a[:, 2],
            next_state=a[:, 3].astype(np.int16), nonterminal=a[:, 4].astype(np.int8),
            td=np.array(self.td), steps=np.array(self.steps, dtype=np.int32),
            trial_reward=np.array(self.trial_reward))
        if self.Q_trial:
            d['Q_trial'] = np.array(self.Q_trial)
        return d


def _env(world, inst):
    return Gridworld(world, rng=TapeRNG(SEED, inst, STREAM_ENV))


def bump_mask(world):
    """Action mask that forbids moves which leave the state unchanged."""
    nxt = np.argmax(world['sas'], axis=2)
    m = nxt != np.arange(world['states'])[:, None]
    m[~m.any(axis=1)] = True
    return m


def gen_dynaq(worlds):
    cases = {
        # name: (world, instance, f32, trials, steps, B, kwargs)
        'open5_b32_f64': ('open_5x5', 0, False, 30, 50, 32, {}),
        'open5_b32_f32': ('open_5x5', 0, True, 30, 50, 32, {}),
        'open5_b50_f32_i7': ('open_5x5', 7, True, 12, 50, 50, {}),
        'open5_noreplay_f32': ('open_5x5', 1, True, 40, 50, 32, {'no_replay': True}),
        'open5_episodic_f32': ('open_5x5', 2, True, 25, 50, 16, {'episodic': True}),
        'walls8_b8_f64': ('walls_8x8', 3, False, 25, 60, 8, {}),
        'walls8_b8_f32': ('walls_8x8', 3, True, 25, 60, 8, {}),
        'walls8_mask_f32': ('walls_8x8', 4, True, 25, 60, 8, {'mask': True}),
        'walls8_traintest_f32': ('walls_8x8', 5, True, 20, 40, 12, {'test_trials': 10}),
        'maze32_b50_f32': ('maze_32x32_1234', 6, True, 3, 200, 50, {}),
        # more updates per step than one wavefront takes in a pass (round 6: 62 + 62 + 6)
        'walls8_b130_f32': ('walls_8x8', 7, True, 10, 40, 130, {}),
    }
    out = {}
    for name, (wname, inst, f32, trials, steps, B, kw) in cases.items():
        world = worlds[wname]
        env = _env(world, inst)
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        ag = DynaQ(env.observation_space, env.action_space, pol)
        ag.M.rng = TapeRNG(SEED, inst, STREAM_MEMORY)
        if f32:
            ag.Q = ag.Q.astype(np.float32)
            ag.M.rewards = ag.M.rewards.astype(np.float32)
        if kw.get('mask'):
            ag.mask_actions = True
            ag.action_mask = bump_mask(world)
        ag.episodic_replay = bool(kw.get('episodic'))
        tr = Tracer(ag)
        ag.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            ag.callbacks.custom_callbacks.setdefault(k, [])
        ag.train(env, trials, steps, B, bool(kw.get('no_replay')))
        n_train_steps = len(tr.sarsn)
        if kw.get('test_trials'):
            ag.test(env, kw['test_trials'], steps)
        d = tr.pack()
        d.update(Q=np.array(ag.Q, dtype=np.float64), M_rewards=np.array(ag.M.rewards, dtype=np.float64),
                 M_states=ag.M.states.astype(np.int16), M_terminals=ag.M.terminals.astype(np.int8),
                 cfg=np.array([inst, f32, trials, steps, B, bool(kw.get('no_replay')),
                               bool(kw.get('episodic')), bool(kw.get('mask')),
                               kw.get('test_trials', 0), n_train_steps], dtype=np.int64),
                 alpha=np.float64(ag.learning_rate), gamma=np.float64(ag.gamma),
                 eps=np.float64(0.1), model_lr=np.float64(ag.M.learning_rate))
        if kw.get('mask'):
            d['action_mask'] = ag.action_mask
        if 'Q_trial' in d and name.startswith('maze32'):
            del d['Q_trial']
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
        out['%s/world' % name] = np.array(wname)
    out['cfg_columns'] = np.array(['instance', 'f32', 'trials', 'steps', 'B', 'no_replay',
                                   'episodic', 'mask', 'test_trials', 'n_train_steps'])
    np.savez_compressed(_out('dynaq_traces.npz'), **out)


def slippery(world, p_slip: float = 0.2):
    """A reference world made stochastic the way a user would: rows of the dense ``sas`` edited in
    place and ``deterministic`` switched off (interface/gridworld.py:115-123 then DRAWS the
    successor).  The intended move keeps 1 - p_slip, each perpendicular move gets p_slip / 2."""
    det = np.argmax(world['sas'], axis=2)
    sas = np.zeros_like(world['sas'])
    for s in range(world['states']):
        for a in range(4):
            sas[s, a, det[s, a]] += 1.0 - p_slip
            sas[s, a, det[s, (a + 1) % 4]] += p_slip / 2
            sas[s, a, det[s, (a + 3) % 4]] += p_slip / 2
    world['sas'] = sas
    world['deterministic'] = False
    return world


def gen_stochastic():
    """Worlds whose transition rows are distributions: the env's own known answers (successor
    for a given uniform) and Dyna-Q / Q-learning runs, the env stream now serving integer draws
    (trial starts) and doubles (one per step) from one counter."""
    out = {}
    worlds = {
        'slip_4x4': slippery(gt.make_gridworld(4, 4, terminals=[3], rewards=np.array([[3, 1.0]]),
                                               goals=[3], invalid_transitions=[(5, 6), (6, 5)])),
        'slip_5x6_wind': slippery(gt.make_gridworld(
            5, 6, terminals=[5, 24], rewards=np.array([[5, 2.0], [24, -1.0], [14, 0.25]]),
            goals=[5], invalid_states=[8, 9], wind=np.tile(np.array([[0, 0], [0, 1], [0, 0]]), (10, 1))),
            0.35),
    }
    for wname, world in worlds.items():
        for k, v in compact(world).items():
            out['%s/%s' % (wname, k)] = v
        out[wname + '/sas'] = np.asarray(world['sas'], dtype=np.float64)
        # env known answers: (state, action, u) -> successor, through the reference's step()
        rows = []
        rng = np.random.default_rng(5)
        for _ in range(400):
            s, a, u = int(rng.integers(world['states'])), int(rng.integers(4)), float(rng.random())
            if rng.random() < 0.1:       # uniforms on and next to the edges of the distribution
                u = float(rng.choice([0.0, 0.8, 0.9, 0.65, 0.825, np.nextafter(0.8, 0), np.nextafter(1.0, 0)]))
            env = Gridworld(world, rng=TapeRNG(SEED, 0, STREAM_ENV, double_sub=1))
            env.current_state = s
            env.rng.random = lambda size=None, u=u: u
            ns, r, end, _, _ = env.step(a)
            rows.append((s, a, u, ns, r, end))
        out[wname + '/step_kat'] = np.array(rows, dtype=np.float64)
    cases = {
        # name: (world, agent, instance, trials, steps, B)
        'slip4_dynaq_b8': ('slip_4x4', 'dynaq', 0, 30, 40, 8),
        'slip4_q_b4': ('slip_4x4', 'q', 2, 30, 40, 4),
        'slip4_q_b0': ('slip_4x4', 'q', 3, 20, 40, 0),
        'slip56_dynaq_b70': ('slip_5x6_wind', 'dynaq', 1, 20, 60, 70),
        # SR.train only ever calls interface.step (agent/sr.py:170-182): the successor is drawn
        'slip4_sr': ('slip_4x4', 'sr', 1, 25, 40, 0),
        'slip56_sr': ('slip_5x6_wind', 'sr', 2, 15, 50, 0),
    }
    for name, (wname, kind, inst, trials, steps, B) in cases.items():
        world = worlds[wname]
        env = Gridworld(world, rng=TapeRNG(SEED, inst, STREAM_ENV, double_sub=1))
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        if kind == 'sr':
            ag = SR(env.observation_space, env.action_space, pol)
            ag.SR = ag.SR.astype(np.float32)
            ag.rewards = ag.rewards.astype(np.float32)
        elif kind == 'dynaq':
            ag = DynaQ(env.observation_space, env.action_space, pol)
            ag.M.rng = TapeRNG(SEED, inst, STREAM_MEMORY)
            ag.Q = ag.Q.astype(np.float32)
            ag.M.rewards = ag.M.rewards.astype(np.float32)
        else:
            ag = QAgent(env.observation_space, env.action_space, pol,
                        rng=TapeRNG(SEED, inst, STREAM_MEMORY))
            for s_ in range(world['states']):     # float32 rows (created lazily as float64 otherwise)
                ag.Q[(s_,)] = np.zeros(4, dtype=np.float32)
        tr = Tracer(ag)
        ag.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            ag.callbacks.custom_callbacks.setdefault(k, [])
        if kind == 'sr':
            ag.train(env, trials, steps)
        else:
            ag.train(env, trials, steps, B)
        d = tr.pack()
        d.pop('Q_trial', None)
        if kind == 'sr':
            d.pop('td', None)
            d.update(SR=np.array(ag.SR, dtype=np.float64), rewards=np.array(ag.rewards, dtype=np.float64),
                     T=np.argmax(ag.transitions, axis=-1).astype(np.int16))
        elif kind == 'dynaq':
            d.update(Q=np.array(ag.Q, dtype=np.float64),
                     M_rewards=np.array(ag.M.rewards, dtype=np.float64),
                     M_states=ag.M.states.astype(np.int16), M_terminals=ag.M.terminals.astype(np.int8))
        else:
            q = np.zeros((world['states'], 4))
            for key, row in ag.Q.items():
                q[key[0]] = np.asarray(row, dtype=np.float32)
            d.update(Q=q, log_len=np.int64(len(ag.M)))
        d['cfg'] = np.array([inst, trials, steps, B], dtype=np.int64)
        d['env_draws'] = np.int64(env.rng.index)
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
        out[name + '/world'] = np.array(wname)
        out[name + '/agent'] = np.array(kind)
    np.savez_compressed(_out('stochastic_traces.npz'), **out)



def gen_dynaq_memory():
    """memory/dyna_q.py:62-157 on its own: stores interleaved with retrieve_batch, the memory's
    generator fed from the build's memory stream (one vector draw per batch)."""
    from cobel.memory.dyna_q import DynaQMemory
    out = {}
    for name, f32, inst in (('f32', True, 3), ('f64', False, 11)):
        M = DynaQMemory(25, 4, rng=TapeRNG(SEED, inst, STREAM_MEMORY))
        if f32:
            M.rewards = M.rewards.astype(np.float32)
        src = np.random.default_rng(77)
        ops, stores, batches = [], [], []
        for k in range(60):
            if k % 5 == 4:
                B = [1, 8, 32, 50][(k // 5) % 4]
                batch = M.retrieve_batch(B)
                ops.append(B)
                batches.append(np.array([[e['state'], e['action'], e['reward'], e['next_state'],
                                          e['terminal']] for e in batch], dtype=np.float64))
            else:
                exp = {'state': int(src.integers(25)), 'action': int(src.integers(4)),
                       'reward': float(src.choice([0.0, 1.0, -0.5, 0.3])),
                       'next_state': int(src.integers(25)), 'terminal': int(src.integers(2))}
                M.store(exp)
                ops.append(0)
                stores.append([exp['state'], exp['action'], exp['reward'], exp['next_state'],
                               exp['terminal']])
        one = M.retrieve(int(stores[-1][0]), int(stores[-1][1]))
        out.update({name + '/ops': np.array(ops), name + '/stores': np.array(stores),
                    name + '/batches': np.concatenate(batches), name + '/instance': np.int64(inst),
                    name + '/rewards': np.array(M.rewards, dtype=np.float64),
                    name + '/states': M.states.astype(np.int64),
                    name + '/terminals': M.terminals.astype(np.int64),
                    name + '/retrieve_last': np.array([one['reward'], one['next_state'],
                                                       one['terminal']], dtype=np.float64)})
    np.savez_compressed(_out('dynaq_memory_kat.npz'), **out)


def gen_qagent(worlds):
    cases = {
        'open5_b0_f32': ('open_5x5', 0, True, 40, 50, 0),
        'open5_b0_f64': ('open_5x5', 0, False, 40, 50, 0),
        'open5_b8_f32': ('open_5x5', 1, True, 30, 50, 8),
        'walls8_b16_f32': ('walls_8x8', 2, True, 20, 60, 16),
        'walls8_b16_f64': ('walls_8x8', 2, False, 20, 60, 16),
        # more replayed experiences per step than one wavefront takes in a pass (round 6: 62 + 38)
        'walls8_b100_f32': ('walls_8x8', 3, True, 10, 40, 100),
    }
    out = {}
    for name, (wname, inst, f32, trials, steps, B) in cases.items():
        world = worlds[wname]
        env = _env(world, inst)
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        ag = QAgent(env.observation_space, env.action_space, pol,
                    rng=TapeRNG(SEED, inst, STREAM_MEMORY))
        if f32:  # rows are otherwise created lazily as float64 (q.py:197-204)
            for s in range(world['states']):
                ag.Q[(s,)] = np.zeros(4, dtype=np.float32)
        tr = Tracer(None)
        ag.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            ag.callbacks.custom_callbacks.setdefault(k, [])
        ag.train(env, trials, steps, B)
        d = tr.pack()
        Q = np.zeros((world['states'], 4))
        for (s,), row in ag.Q.items():
            Q[s] = row
        d.update(Q=Q, cfg=np.array([inst, f32, trials, steps, B], dtype=np.int64),
                 alpha=np.float64(ag.learning_rate), gamma=np.float64(ag.gamma), eps=np.float64(0.1),
                 log_len=np.int64(len(ag.M)))
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
        out['%s/world' % name] = np.array(wname)
    np.savez_compressed(_out('qagent_traces.npz'), **out)


def gen_qagent_topology():
    """QAgent on a Topology (unit_tests/test_q.py:57-78 "Topology", demo/topology/demo.py):
    observations are 6-float poses, Q is keyed by tuple(pose) (agent/q.py:154-155)."""
    from cobel.interface import Topology
    from cobel.misc import topology_tools as tt
    cases = {
        # name: (builder args, instance, f32, trials, steps, B)
        'track_b0_f32': ((10, 2, 1., 20., 'right'), 0, True, 25, 100, 0),
        'track_b8_f32': ((10, 2, 1., 20., 'right'), 1, True, 20, 100, 8),
        'track_b8_f64': ((10, 2, 1., 20., 'right'), 1, False, 20, 100, 8),
        'track5_b4_f32': ((5, 1, 0.5, 2., 'left'), 2, True, 25, 12, 4),
        # six actions: hexagonal(5) (misc/topology_tools.py:175-272), goal at node 7
        'hex5_b0_f32': (('hex', 5, (0.0, 2.0), 3.0, '7'), 4, True, 30, 40, 0),
        'hex5_b8_f32': (('hex', 5, (0.0, 2.0), 3.0, '7'), 5, True, 30, 40, 8),
        'hex4_b70_f32': (('hex', 4, (0.0, 1.0), 1.0, None), 6, True, 12, 30, 70),
        # pre-rendered dictionary observations (unit_tests/test_q.py:24-41, 57-62 "Topology-Dict"):
        # OfflineSimulator, Q keyed by the concatenated components (agent/q.py:156-158)
        'trackdict_b8_f32': ((10, 2, 1., 20., 'right', 'dict'), 7, True, 25, 60, 8),
        'trackdict_b0_f32': ((10, 2, 1., 20., 'right', 'dict'), 8, True, 20, 80, 0),
    }
    out = {}
    for name, (args, inst, f32, trials, steps, B) in cases.items():
        dict_obs = len(args) == 6 and args[5] == 'dict'
        if dict_obs:
            args = args[:5]
        if args[0] == 'hex':
            nodes, starts = tt.hexagonal(*args[1:])
            args = (float(args[1]), 0.0, 0.0, 0.0, 'hex')
        else:
            nodes, starts = tt.linear_track(*args)
        n_act = len(nodes[starts[0]]['neighbors'])
        simulator = None
        if dict_obs:
            import gymnasium
            from cobel.interface import OfflineSimulator
            obs = {node['pose']: {'1': np.array(node['pose']), '2': np.array(node['pose'])}
                   for node in nodes.values()}
            space = gymnasium.spaces.Dict({'1': gymnasium.spaces.Box(0., 1., (6,)),
                                           '2': gymnasium.spaces.Box(0., 1., (6,))})
            simulator = OfflineSimulator(obs, space)
        env = Topology(nodes, starts, simulator, rng=TapeRNG(SEED, inst, STREAM_ENV))
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        ag = QAgent(env.observation_space, env.action_space, pol,
                    rng=TapeRNG(SEED, inst, STREAM_MEMORY))
        ids = list(nodes.keys())
        rep = 2 if dict_obs else 1
        key = {tuple(np.tile(np.array(nodes[k]['pose']).flatten(), rep)): i
               for i, k in enumerate(ids)}
        if f32:
            for k in key:
                ag.Q[k] = np.zeros(n_act, dtype=np.float32)
        sarsn, tds, steps_log, rewards = [], [], [], []

        def on_step_end(logs, sarsn=sarsn, tds=tds, key=key):
            sarsn.append((key[logs['state']], logs['action'], logs['reward'],
                          key[logs['next_state']], logs['terminal']))
            tds.append(float(logs['td']))

        def on_trial_end(logs, steps_log=steps_log, rewards=rewards):
            steps_log.append(logs['steps'])
            rewards.append(float(logs['trial_reward']))

        ag.callbacks.custom_callbacks = {'on_step_end': [on_step_end], 'on_trial_end': [on_trial_end],
                                         'on_trial_begin': [], 'on_step_begin': []}
        ag.train(env, trials, steps, B)
        a = np.array(sarsn, dtype=np.float64).reshape(-1, 5)
        Q = np.zeros((len(ids), n_act))
        for k, row in ag.Q.items():
            Q[key[k]] = row
        probe = np.array([nodes[k]['pose'] for k in ids[:5]], dtype=np.float64)
        if dict_obs:
            # (the reference's predict_on_batch on dictionary observations raises TypeError — it
            #  looks Q up with an ndarray, agent/q.py:340 — so the rows are read from Q directly)
            probe_q = np.array([ag.Q[tuple(np.tile(p, 2))] for p in probe], dtype=np.float64)
        else:
            probe_q = np.array(ag.predict_on_batch(probe), dtype=np.float64)
        d = dict(state=a[:, 0].astype(np.int16), action=a[:, 1].astype(np.int8), reward=a[:, 2],
                 next_state=a[:, 3].astype(np.int16), nonterminal=a[:, 4].astype(np.int8),
                 td=np.array(tds), steps=np.array(steps_log, dtype=np.int32),
                 trial_reward=np.array(rewards), Q=Q, log_len=np.int64(len(ag.M)),
                 cfg=np.array([inst, f32, trials, steps, B], dtype=np.int64),
                 track=np.array(args[:4], dtype=np.float64), side=np.array(args[4]),
                 probe=probe, probe_q=probe_q)
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
    np.savez_compressed(_out('qagent_topology_traces.npz'), **out)


def gen_sr(worlds):
    cases = {
        'open5_f64': ('open_5x5', 0, False, 30, 50, False),
        'open5_f32': ('open_5x5', 0, True, 30, 50, False),
        'walls8_f32': ('walls_8x8', 1, True, 20, 60, False),
        'walls8_f64': ('walls_8x8', 1, False, 20, 60, False),
        'walls8_mask_f32': ('walls_8x8', 2, True, 15, 60, True),
    }
    out = {}
    for name, (wname, inst, f32, trials, steps, mask) in cases.items():
        world = worlds[wname]
        env = _env(world, inst)
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        ag = SR(env.observation_space, env.action_space, pol)
        if f32:
            ag.SR = ag.SR.astype(np.float32)
            ag.rewards = ag.rewards.astype(np.float32)
        if mask:
            ag.mask_actions = True
            ag.action_mask = bump_mask(world)
        tr = Tracer(None)
        qs = []
        orig = ag.retrieve_q

        def spy(state, orig=orig, qs=qs):
            q = orig(state)
            qs.append(np.asarray(q, dtype=np.float64))
            return q

        ag.retrieve_q = spy
        ag.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            ag.callbacks.custom_callbacks.setdefault(k, [])
        ag.train(env, trials, steps)
        d = tr.pack()
        del d['td']
        d.update(SR=np.array(ag.SR, dtype=np.float64), rewards=np.array(ag.rewards, dtype=np.float64),
                 T=np.argmax(ag.transitions, axis=-1).astype(np.int16), q=np.array(qs),
                 cfg=np.array([inst, f32, trials, steps, mask], dtype=np.int64),
                 alpha=np.float64(ag.learning_rate), gamma=np.float64(ag.gamma), eps=np.float64(0.1))
        if mask:
            d['action_mask'] = ag.action_mask
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
        out['%s/world' % name] = np.array(wname)
    np.savez_compressed(_out('sr_traces.npz'), **out)


def gen_sr32(worlds=None):
    """SR at config C4's own size (32 x 32 = 1 024 states: `retrieve_q`'s row sums change their
    association at 128-element leaves there, agent/sr.py:288-308).  Three float32-coerced runs of
    the reference's `SR.train`: trials that start next to the goal (the reward estimate gets its
    one non-zero entry the way C4 does), and two runs whose reward estimates are pre-loaded — all
    1 024 entries non-zero / twenty of them — so that the sums have many terms.  Stored: the rows
    of SR that differ from the identity and their indices, `rewards`, argmax(`transitions`),
    `retrieve_q` of every step."""
    if worlds is None:
        worlds = {'open_32x32': gt.make_open_field(32, 32, 0, 1),
                  'open_32x32_near': dict(gt.make_open_field(32, 32, 0, 1),
                                          starting_states=np.array([1, 2, 32, 33, 34, 64, 65, 66]))}
    gen = np.random.default_rng(20261005)
    dense = gen.random(1024).astype(np.float32)
    sparse = np.zeros(1024, dtype=np.float32)
    sparse[gen.choice(1024, size=20, replace=False)] = (gen.random(20) * 2 - 0.5).astype(np.float32)
    cases = {
        'open32_near_f32': ('open_32x32_near', 0, 8, 60, None),
        'open32_dense_f32': ('open_32x32', 1, 2, 90, dense),
        'open32_r20_f32': ('open_32x32', 2, 2, 90, sparse),
    }
    out = {}
    for name, (wname, inst, trials, steps, rw0) in cases.items():
        world = worlds[wname]
        env = _env(world, inst)
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        ag = SR(env.observation_space, env.action_space, pol)
        ag.SR = ag.SR.astype(np.float32)
        ag.rewards = ag.rewards.astype(np.float32) if rw0 is None else rw0.copy()
        tr = Tracer(None)
        qs = []
        orig = ag.retrieve_q

        def spy(state, orig=orig, qs=qs):
            q = orig(state)
            qs.append(np.asarray(q, dtype=np.float64))
            return q

        ag.retrieve_q = spy
        ag.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            ag.callbacks.custom_callbacks.setdefault(k, [])
        ag.train(env, trials, steps)
        assert ag.SR.dtype == np.float32 and ag.rewards.dtype == np.float32
        d = tr.pack()
        del d['td']
        rows = np.flatnonzero((ag.SR != np.eye(1024, dtype=np.float32)).any(axis=1))
        d.update(SR_rows=rows.astype(np.int16), SR_values=ag.SR[rows], rewards=np.array(ag.rewards),
                 T=np.argmax(ag.transitions, axis=-1).astype(np.int16), q=np.array(qs),
                 cfg=np.array([inst, 1, trials, steps, 0], dtype=np.int64),
                 alpha=np.float64(ag.learning_rate), gamma=np.float64(ag.gamma), eps=np.float64(0.1))
        if rw0 is not None:
            d['rewards0'] = rw0
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
        out['%s/world' % name] = np.array(wname)
    np.savez_compressed(_out('sr32_traces.npz'), **out)


MODES = ['default', 'reverse', 'forward', 'blend_forward', 'blend_reverse', 'interpolate',
         'sweeping']


def sfma_world_5x5():
    """The world of unit_tests/test_sfma.py:44-60 and demo/gridworld/demo_sfma.py:36-55."""
    inv = [(3, 4), (4, 3), (8, 9), (9, 8), (13, 14), (14, 13)]
    w = gt.make_gridworld(5, 5, terminals=[4], rewards=np.array([[4, 10]]), goals=[4],
                          invalid_transitions=inv)
    w['starting_states'] = np.array([12])
    return w


def sfma_world_6x7():
    """Walls, two rewards of different sign, several starts; short trials time out (replay then
    starts from an experience drawn by strength, memory/sfma.py:262-270)."""
    inv_t = [(8, 9), (9, 8), (15, 16), (16, 15), (22, 23), (23, 22), (30, 37), (37, 30)]
    return gt.make_gridworld(6, 7, terminals=[6, 35], rewards=np.array([[6, 1.0], [35, -0.5], [17, 0.25]]),
                             goals=[6], invalid_states=[10, 24, 25], invalid_transitions=inv_t,
                             starting_states=[41, 21, 3, 29])


def gen_sfma():
    """agent/sfma.py:233-458 + memory/sfma.py:195-416 + memory/utils/metrics.py, driven by the
    build's streams: env / policy as everywhere, SFMAMemory.rng = memory stream with doubles on
    sub 1, SFMA.rng = agent stream."""
    from cobel.agent import SFMA
    from cobel.memory import SFMAMemory
    from cobel.memory.utils import DR, SR as SRMetric, Euclidean
    from oracle.philox import STREAM_AGENT
    worlds = {'sfma_5x5': sfma_world_5x5(), 'sfma_6x7': sfma_world_6x7()}
    out = {}
    for wname, w in worlds.items():
        for k, v in compact(w).items():
            out['world/%s/%s' % (wname, k)] = v
        out['world/%s/invalid_transitions' % wname] = np.array(
            w['invalid_transitions'], dtype=np.int64).reshape(-1, 2)
        out['metric/%s/DR' % wname] = DR(w['width'], w['height'], w['sas'], 0.9,
                                         w['invalid_transitions']).D
        out['metric/%s/SR' % wname] = SRMetric(w['sas'], 0.9).D
        out['metric/%s/Euclidean' % wname] = Euclidean(w['width'], w['height']).D
    # name: (world, instance, f32, metric, mode, trials, steps, B, options)
    cases = {
        'dr_default_f64': ('sfma_5x5', 0, False, 'DR', 'default', 30, 50, 32, {}),
        'dr_default_f32': ('sfma_5x5', 0, True, 'DR', 'default', 30, 50, 32, {}),
        'dr_reverse_f32': ('sfma_5x5', 1, True, 'DR', 'reverse', 10, 50, 32, {'mask': True}),
        'sr_forward_f32': ('sfma_5x5', 2, True, 'SR', 'forward', 8, 50, 32, {}),
        'eu_sweeping_f32': ('sfma_5x5', 3, True, 'Euclidean', 'sweeping', 8, 50, 32, {}),
        'dr_dynamic_f32': ('sfma_5x5', 4, True, 'DR', 'default', 30, 50, 32, {'dynamic': True}),
        'dr_dynamic_f64': ('sfma_5x5', 4, False, 'DR', 'default', 30, 50, 32, {'dynamic': True}),
        'sr_random_mask_f32': ('sfma_5x5', 5, True, 'SR', 'default', 20, 50, 32,
                               {'random': True, 'mask': True}),
        'dr_random_f32': ('sfma_5x5', 6, True, 'DR', 'default', 24, 50, 20, {'random': True}),
        'dr_traintest_f32': ('sfma_5x5', 7, True, 'DR', 'reverse', 14, 50, 32,
                             {'mask': True, 'noreplay_trials': 5, 'test_trials': 5}),
        'w67_dr_reverse_f32': ('sfma_6x7', 8, True, 'DR', 'reverse', 12, 14, 24, {}),
        'w67_dr_reverse_f64': ('sfma_6x7', 8, False, 'DR', 'reverse', 12, 14, 24, {}),
        'w67_sr_blendf_f32': ('sfma_6x7', 9, True, 'SR', 'blend_forward', 20, 14, 16, {}),
        'w67_sr_blendr_f32': ('sfma_6x7', 10, True, 'SR', 'blend_reverse', 10, 14, 16,
                              {'mask': True}),
        'w67_eu_interp_f32': ('sfma_6x7', 11, True, 'Euclidean', 'interpolate', 10, 14, 16, {}),
        'w67_dr_recency_f32': ('sfma_6x7', 12, True, 'DR', 'default', 20, 14, 16,
                               {'recency': True, 'C_normalize': True, 'D_normalize': True}),
        'w67_dr_determ_f32': ('sfma_6x7', 13, True, 'DR', 'reverse', 10, 14, 16,
                              {'deterministic': True, 'R_normalize': False}),
        'w67_dr_start_f32': ('sfma_6x7', 14, True, 'DR', 'default', 8, 14, 12,
                             {'start_replay': True, 'nb_replays': 2}),
        'w67_sr_mods_f32': ('sfma_6x7', 15, True, 'SR', 'forward', 10, 14, 16,
                            {'reward_mod_local': True, 'reward_mod': True, 'state_mod': True,
                             'decay_strength': 0.95, 'decay_inhibition': 0.8,
                             'reward_modulation': 0.5}),
        'w67_dr_dynamic_f32': ('sfma_6x7', 16, True, 'DR', 'default', 12, 14, 16,
                               {'dynamic': True, 'beta': 5.0}),
    }
    for name, (wname, inst, f32, mname, mode, trials, steps, B, kw) in cases.items():
        w = worlds[wname]
        env = _env(w, inst)
        if mname == 'DR':
            metric = DR(w['width'], w['height'], w['sas'], 0.9, w['invalid_transitions'])
        elif mname == 'SR':
            metric = SRMetric(w['sas'], 0.9)
        else:
            metric = Euclidean(w['width'], w['height'])
        mem = SFMAMemory(metric, w['states'], 4, rng=TapeRNG(SEED, inst, STREAM_MEMORY, double_sub=1))
        if 'decay_inhibition' in kw:
            mem.decay_inhibition = kw['decay_inhibition']
        if 'decay_strength' in kw:
            mem.decay_strength = kw['decay_strength']
        pol = EpsilonGreedy(0.1, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        pol_test = EpsilonGreedy(0.0, rng=TapeRNG(SEED, inst, 3)) if kw.get('test_trials') else None
        ag = SFMA(env.observation_space, env.action_space, pol, mem, pol_test,
                  rng=TapeRNG(SEED, inst, STREAM_AGENT))
        ag.M.mode = mode
        for k in ('recency', 'C_normalize', 'D_normalize', 'R_normalize', 'deterministic',
                  'reward_mod_local', 'reward_mod', 'state_mod', 'reward_modulation', 'beta'):
            if k in kw:
                setattr(ag.M, k, kw[k])
        for k in ('dynamic', 'random', 'start_replay', 'nb_replays'):
            if k in kw:
                setattr(ag, k, kw[k])
        if f32:
            ag.Q = ag.Q.astype(np.float32)
            ag.M.rewards = ag.M.rewards.astype(np.float32)
        if kw.get('mask'):
            ag.mask_actions = True
            ag.action_mask = bump_mask(w)
        tr = Tracer(ag)
        rp, modes, td_acc, started = [], [], [], []

        def on_replay_begin(logs, started=started):
            started.append(('steps' not in logs))   # start-of-trial replays come before 'steps'

        def on_replay_end(logs, rp=rp, ag=ag, started=started):
            kind = 1 if started[-1] else 0
            for e in logs['replay']:
                rp.append((ag.current_trial - (0 if kind else 1), kind, e['state'], e['action'],
                           float(e['reward']), e['next_state'], e['terminal'],
                           float(e.get('td', np.nan))))

        def on_trial_end(logs, modes=modes, td_acc=td_acc, ag=ag):
            modes.append(MODES.index(logs['replay_mode']))
            td_acc.append(float(ag.td))

        cbs = {k: list(v) for k, v in tr.callbacks().items()}
        cbs['on_trial_end'].append(on_trial_end)
        cbs['on_replay_begin'] = [on_replay_begin]
        cbs['on_replay_end'] = [on_replay_end]
        for k in ('on_trial_begin', 'on_step_begin'):
            cbs.setdefault(k, [])
        ag.callbacks.custom_callbacks = cbs
        ag.train(env, trials, steps, B)
        n_train = len(tr.sarsn)
        if kw.get('noreplay_trials'):
            ag.train(env, kw['noreplay_trials'], steps, B, True)
        n_train2 = len(tr.sarsn)
        if kw.get('test_trials'):
            ag.test(env, kw['test_trials'], steps)
        d = tr.pack()
        rpa = np.array(rp, dtype=np.float64).reshape(-1, 8)
        d.update(
            Q=np.array(ag.Q, dtype=np.float64), M_rewards=np.array(ag.M.rewards, dtype=np.float64),
            M_states=ag.M.states.astype(np.int16), M_terminals=ag.M.terminals.astype(np.int8),
            C=ag.M.C.copy(), T=ag.M.T.copy(), I=ag.M.I.copy(),
            rp_trial=rpa[:, 0].astype(np.int32), rp_kind=rpa[:, 1].astype(np.int8),
            rp_state=rpa[:, 2].astype(np.int16), rp_action=rpa[:, 3].astype(np.int8),
            rp_reward=rpa[:, 4], rp_next=rpa[:, 5].astype(np.int16),
            rp_nonterminal=rpa[:, 6].astype(np.int8), rp_td=rpa[:, 7],
            replay_mode=np.array(modes, dtype=np.int8), td_acc=np.array(td_acc),
            final_mode=np.int64(MODES.index(ag.M.mode)),
            n_train_steps=np.array([n_train, n_train2], dtype=np.int64),
            cfg=np.array([inst, f32, trials, steps, B], dtype=np.int64),
            alpha=np.float64(ag.learning_rate), gamma=np.float64(ag.gamma), eps=np.float64(0.1),
            ctr=np.array([env.rng.index, pol.rng.index, mem.rng.index, ag.rng.index], dtype=np.int64),
            opts=np.array(repr(dict(kw, metric=mname, mode=mode, world=wname))))
        if kw.get('mask'):
            d['action_mask'] = ag.action_mask
        for k, v in d.items():
            out['%s/%s' % (name, k)] = v
    np.savez_compressed(_out('sfma_traces.npz'), **out)


def gen_monitor(worlds):
    """monitor/behavior.py:73-97 and analysis/behavior_spatial.py:9-73."""
    rng = np.random.default_rng(7)
    trials, max_steps = 40, 50
    mon = EscapeLatencyMonitor(trials, max_steps)
    steps = rng.integers(0, max_steps, trials)
    order = [t for t in range(trials) if t not in (5, 6, 17)]  # leave NaN gaps
    for t in order:
        mon.update({'trial': t, 'steps': int(steps[t])})
    world = worlds['walls_8x8']
    traj_states = [rng.integers(0, 64, n) for n in (30, 1, 77)]
    trajs = [world['coordinates'][s] for s in traj_states]
    occ = {m: get_occupancy_map(trajs, 8, 8, 1.0, m) for m in ('expand', 'include', 'ignore')}
    occ2 = get_occupancy_map(trajs, 8, 8, 2.0, 'expand')
    # RewardMonitor / ResponseMonitor (behavior.py:174-199, :269-290) on the same trial order
    from cobel.monitor import ResponseMonitor, RewardMonitor
    rewards = np.round(rng.random(trials) * 2.0 - 0.5, 3)
    responses = rng.integers(0, 2, trials)
    rew_mon, resp_a, resp_b = RewardMonitor(trials, (-0.5, 1.5)), ResponseMonitor(trials), \
        ResponseMonitor(trials)
    for t in order:
        rew_mon.update({'trial': t, 'trial_reward': float(rewards[t])})
        resp_a.update({'trial': t, 'trial_reward': float(rewards[t])})
        resp_b.update({'trial': t, 'trial_reward': float(rewards[t]), 'response': int(responses[t])})
    np.savez_compressed(
        _out('monitor_kat.npz'), steps=steps, order=np.array(order),
        rewards=rewards, responses=responses, reward_trace=rew_mon.reward_trace,
        reward_avg=rew_mon.reward_trace_avg, resp_default=resp_a.responses, crc_default=resp_a.CRC,
        resp_given=resp_b.responses, crc_given=resp_b.CRC,
        latency=mon.latency_trace, latency_avg=mon.latency_trace_avg, max_steps=np.int64(max_steps),
        traj_states=np.concatenate(traj_states), traj_len=np.array([30, 1, 77]),
        occ_expand=occ['expand'], occ_include=occ['include'], occ_ignore=occ['ignore'],
        occ_bin2=occ2)


def gen_topology():
    """misc/topology_tools.py builders as index tables, unit_tests/test_topology.py:92-109 and a
    draw-injected random walk through interface/topology.py."""
    from cobel.interface import Topology
    from cobel.misc import topology_tools as tt
    out = {}
    built = {'linear_10x2': tt.linear_track(10, 2, 1., 20., 'right'),
             'linear_5x1_left': tt.linear_track(5, 1, 0.5, 2., 'left'),
             't_maze_4_3_1': tt.t_maze(4, 3, 1), 't_maze_3_2_2_left': tt.t_maze(3, 2, 2, 2.0, 3.0, 'left'),
             'grid_4x3': tt.grid((4, 3)), 'grid_5': tt.grid(5, (0.0, 2.0), 7.0, '12'),
             'cross_2_1_rot30': tt.cross(2, 1, 0.5, 30.0), 'cross_3_2': tt.cross(3, 2, 2.0),
             'hex_4': tt.hexagonal(4), 'hex_5_goal7': tt.hexagonal(5, (0.0, 2.0), 3.0, '7')}
    for name, (nodes, starts) in built.items():
        ids = list(nodes.keys())
        idx = {k: i for i, k in enumerate(ids)}
        out[name + '/nbr'] = np.array([[idx[m] for m in nodes[k]['neighbors']] for k in ids])
        out[name + '/pose'] = np.array([nodes[k]['pose'] for k in ids], dtype=np.float64)
        out[name + '/reward'] = np.array([nodes[k]['reward'] for k in ids], dtype=np.float64)
        out[name + '/terminal'] = np.array([bool(nodes[k]['terminal']) for k in ids])
        out[name + '/starts'] = np.array([idx[k] for k in starts])
    nodes, starts = built['t_maze_4_3_1']
    env = Topology(nodes, starts)
    env.reset()
    assert env.current_node == '10'
    actions = [1, 1, 1, 1, 1, 2, 2, 2]
    st, rw, tm = [], [], []
    for a in actions:
        _, r, t, _, _ = env.step(a)
        st.append(int(env.current_node)), rw.append(r), tm.append(t)
    assert st == [9, 8, 7, 3, 3, 4, 5, 6] and rw == [0.] * 7 + [1.] and tm == [False] * 7 + [True]
    out.update(kat_actions=np.array(actions), kat_states=np.array(st), kat_rewards=np.array(rw),
               kat_terminals=np.array(tm))
    nodes, starts = built['linear_10x2']
    env = Topology(nodes, starts, rng=TapeRNG(SEED, 3, STREAM_ENV))
    rng = np.random.default_rng(11)
    walk_a, walk_s, walk_obs, walk_r, walk_t = [], [], [], [], []
    obs, _ = env.reset()
    first = int(env.current_node)
    for _ in range(60):
        a = int(rng.integers(0, 4))
        obs, r, t, trunc, _ = env.step(a)
        assert trunc == t
        walk_a.append(a), walk_s.append(int(env.current_node)), walk_obs.append(obs)
        walk_r.append(r), walk_t.append(t)
        if t:
            env.reset()
            walk_s[-1] = int(env.current_node) + 1000   # mark: state after the reset
    out.update(walk_first=np.int64(first), walk_actions=np.array(walk_a), walk_states=np.array(walk_s),
               walk_obs=np.array(walk_obs), walk_rewards=np.array(walk_r),
               walk_terminals=np.array(walk_t))
    np.savez_compressed(_out('topology_kat.npz'), **out)


def gen_dqn():
    """agent/dqn.py + memory/dqn.py + network/network_torch.py on linear_track(10, 2): float64
    6-64-64-4 MLP, Adam, MSE, tau 0.01, draw-injected env / policy / memory streams."""
    import torch
    from collections import OrderedDict
    from cobel.agent import DQN
    from cobel.interface import Topology
    from cobel.memory import DQNMemory
    from cobel.misc.topology_tools import linear_track
    from cobel.network import TorchNetwork
    out = {}
    for name, inst, trials, steps, ddqn, sessions, cap in (
            ('dqn_i0', 0, 3, 25, False, (3,), 100000), ('dqn_i2', 2, 2, 30, False, (2,), 100000),
            # two train() calls on one agent: the memory keeps the first session's experiences
            ('dqn_two_sessions', 1, 4, 25, False, (2, 2), 100000),
            # ... and a memory smaller than the run: FIFO eviction (memory/dqn.py:113-119)
            ('dqn_two_sessions_cap40', 3, 4, 25, False, (2, 2), 40)):
        torch.manual_seed(1234 + inst)
        layers = [('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
                  ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
                  ('output', torch.nn.Linear(64, 4))]
        net = torch.nn.Sequential(OrderedDict(layers)).double()
        model = TorchNetwork(net)
        init = model.get_weights()
        nodes, starts = linear_track(10, 2, 1., 20., 'right')
        env = Topology(nodes, starts, rng=TapeRNG(SEED, inst, STREAM_ENV))
        pol = EpsilonGreedy(0.3, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        agent = DQN(env.observation_space, env.action_space, pol, model, gamma=0.8,
                    memory=DQNMemory(cap, rng=TapeRNG(SEED, inst, STREAM_MEMORY)))
        agent.DDQN = ddqn
        tr = Tracer(None)
        nodes_seen = []
        agent.callbacks.custom_callbacks = {
            'on_step_end': [lambda logs, e=env, ns=nodes_seen: ns.append(int(e.current_node)),
                            lambda logs, t=tr: t.td.append(0.0)],
            'on_trial_end': [tr.on_trial_end], 'on_trial_begin': [], 'on_step_begin': []}
        torch.set_num_threads(1)
        for part in sessions:
            agent.train(env, part, steps, 32)
        out[name + '/sessions'] = np.array(sessions)
        out[name + '/capacity'] = np.int64(cap)
        poses = np.array([nodes[k]['pose'] for k in nodes])
        for i, w in enumerate(init):
            out['%s/init_%d' % (name, i)] = w
        for i, w in enumerate(agent.model_online.get_weights()):
            out['%s/online_%d' % (name, i)] = w
        for i, w in enumerate(agent.model_target.get_weights()):
            out['%s/target_%d' % (name, i)] = w
        out[name + '/nodes'] = np.array(nodes_seen)
        out[name + '/actions'] = agent.M.actions.astype(np.int64)
        out[name + '/rewards'] = agent.M.rewards
        out[name + '/steps'] = np.array(tr.steps)
        out[name + '/q_all'] = agent.predict_on_batch(poses)
        out[name + '/cfg'] = np.array([inst, trials, steps, 32, ddqn])
    np.savez_compressed(_out('dqn_trace.npz'), **out)


def gen_dyna_dqn():
    """agent/dyna_q.py:333-708 (DynaDQN) on a 4x4 open field: one-hot observations, float64
    16-32-4 MLP, draw-injected streams."""
    import torch
    from collections import OrderedDict
    from cobel.agent.dyna_q import DynaDQN
    from cobel.network import TorchNetwork
    out = {}
    for name, inst, trials, steps, B in (('ddqn_i0', 0, 3, 20, 16), ('ddqn_i1', 1, 2, 25, 8),
                                         # the network shape and batch the fused DQN step covers
                                         ('ddqn_mlp64', 2, 3, 20, 32)):
        torch.manual_seed(99 + inst)
        if name == 'ddqn_mlp64':
            net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 64)), ('relu_1', torch.nn.ReLU()),
                ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
                ('output', torch.nn.Linear(64, 4))])).double()
        else:
            net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 32)), ('relu_1', torch.nn.ReLU()),
                ('output', torch.nn.Linear(32, 4))])).double()
        model = TorchNetwork(net)
        init = model.get_weights()
        world = gt.make_open_field(4, 4, 0, 1)
        env = Gridworld(world, rng=TapeRNG(SEED, inst, STREAM_ENV))
        pol = EpsilonGreedy(0.2, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        agent = DynaDQN(env.observation_space, env.action_space, pol, model, gamma=0.9)
        agent.M.rng = TapeRNG(SEED, inst, STREAM_MEMORY)
        tr = Tracer(None)
        agent.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            agent.callbacks.custom_callbacks.setdefault(k, [])
        tr.on_step_end = None
        torch.set_num_threads(1)
        agent.train(env, trials, steps, B)
        d = tr.pack()
        for i, w in enumerate(init):
            out['%s/init_%d' % (name, i)] = w
        for i, w in enumerate(agent.model_online.get_weights()):
            out['%s/online_%d' % (name, i)] = w
        for i, w in enumerate(agent.model_target.get_weights()):
            out['%s/target_%d' % (name, i)] = w
        out[name + '/state'], out[name + '/action'] = d['state'], d['action']
        out[name + '/steps'] = d['steps']
        out[name + '/M_rewards'] = agent.M.rewards
        out[name + '/M_states'] = agent.M.states
        out[name + '/M_terminals'] = agent.M.terminals
        out[name + '/q_all'] = agent.predict_on_batch(np.arange(16))
        out[name + '/cfg'] = np.array([inst, trials, steps, B])
    np.savez_compressed(_out('dyna_dqn_trace.npz'), **out)


def gen_dyna_dsr():
    """agent/dyna_q.py:711-1150 (DynaDSR) on a 4x4 open field: one-hot observations, float64
    16-24-16 successor networks (one per action), 16-12-1 reward network, draw-injected streams;
    one run with the default switches, one with use_DR / use_follow_up_state / terminality
    respected / periodic target copies."""
    import torch
    from collections import OrderedDict
    from cobel.agent.dyna_q import DynaDSR
    from cobel.network import TorchNetwork
    out = {}
    runs = (('ddsr_default', 0, 3, 15, 12, dict()),
            ('ddsr_switches', 1, 2, 18, 10, dict(use_DR=True, use_follow_up_state=True,
                                                 ignore_terminality=False, target_update=3)),
            # the networks of demo/gridworld/demo_dyna_dsr.py (64-64 ReLU) and its batch of 32:
            # the shape the fused MLP kernels cover
            ('ddsr_mlp64', 2, 3, 15, 32, dict()),
            ('ddsr_mlp64_switches', 3, 2, 18, 32, dict(use_DR=True, use_follow_up_state=True,
                                                       ignore_terminality=False,
                                                       target_update=0.05)))
    for name, inst, trials, steps, B, switches in runs:
        torch.manual_seed(7 + inst)
        if 'mlp64' in name:
            sr_net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 64)), ('relu_1', torch.nn.ReLU()),
                ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
                ('output', torch.nn.Linear(64, 16))])).double()
            rw_net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 64)), ('relu_1', torch.nn.ReLU()),
                ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
                ('output', torch.nn.Linear(64, 1))])).double()
        else:
            sr_net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 24)), ('relu_1', torch.nn.ReLU()),
                ('output', torch.nn.Linear(24, 16))])).double()
            rw_net = torch.nn.Sequential(OrderedDict([
                ('dense_1', torch.nn.Linear(16, 12)), ('relu_1', torch.nn.ReLU()),
                ('output', torch.nn.Linear(12, 1))])).double()
        model_sr, model_rw = TorchNetwork(sr_net), TorchNetwork(rw_net)
        init_sr, init_rw = model_sr.get_weights(), model_rw.get_weights()
        world = gt.make_open_field(4, 4, 0, 1)
        env = Gridworld(world, rng=TapeRNG(SEED, inst, STREAM_ENV))
        pol = EpsilonGreedy(0.25, rng=TapeRNG(SEED, inst, STREAM_POLICY))
        agent = DynaDSR(env.observation_space, env.action_space, pol, model_sr, model_rw, gamma=0.9)
        for k, v in switches.items():
            setattr(agent, k, v)
        agent.M.rng = TapeRNG(SEED, inst, STREAM_MEMORY)
        tr = Tracer(None)
        agent.callbacks.custom_callbacks = {k: list(v) for k, v in tr.callbacks().items()}
        for k in ('on_trial_begin', 'on_step_begin'):
            agent.callbacks.custom_callbacks.setdefault(k, [])
        tr.on_step_end = None
        torch.set_num_threads(1)
        agent.train(env, trials, steps, B)
        d = tr.pack()
        for i, w in enumerate(init_sr):
            out['%s/init_sr_%d' % (name, i)] = w
        for i, w in enumerate(init_rw):
            out['%s/init_rw_%d' % (name, i)] = w
        for a in range(4):
            for i, w in enumerate(agent.models_online[a].get_weights()):
                out['%s/online_%d_%d' % (name, a, i)] = w
            for i, w in enumerate(agent.models_target[a].get_weights()):
                out['%s/target_%d_%d' % (name, a, i)] = w
        for i, w in enumerate(agent.model_reward.get_weights()):
            out['%s/reward_%d' % (name, i)] = w
        out[name + '/state'], out[name + '/action'] = d['state'], d['action']
        out[name + '/steps'] = d['steps']
        out[name + '/q_all'] = agent.predict_on_batch(np.arange(16))
        out[name + '/cfg'] = np.array([inst, trials, steps, B])
    np.savez_compressed(_out('dyna_dsr_trace.npz'), **out)


def _opt_sim(task, params):
    return task['bias'] + params['x_1'] + params['x_2'] ** 2 + params['x_3'] ** 3


def _opt_loss(data_sim, data_exp):
    error = 0.
    for t in data_sim:
        error += (np.mean(data_sim[t]) - data_exp[t]) ** 2
    return error / len(data_sim)


def gen_optimizer():
    """GridSearchOptimizer: enumeration orders (grid_search.py:112-171) and a fit of the
    docstring example (:68-88) on two tasks."""
    import tempfile
    from cobel.optimizer import GridSearchOptimizer
    params = {'x_1': [0, 1, 2, 3, 4], 'x_2': np.array([0.4, 0.1, 0.2, 0.3, 0.0]),
              'x_3': [0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1]}
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for order in ('nested', 'systematic'):
            opt = GridSearchOptimizer(tmp + '/', params, order=order)
            out['keys_' + order] = np.array(list(opt.parameter_combinations), dtype=np.float64)
        small = {'x_1': [0, 2, 4], 'x_2': np.array([0.3, 0.1]), 'x_3': [0.5, 0.9]}
        tasks = {'task_1': {'bias': 0.0}, 'task_2': {'bias': 1.5}}
        data = {'task_1': _opt_sim(tasks['task_1'], {'x_1': 2, 'x_2': 0.1, 'x_3': 0.9}),
                'task_2': _opt_sim(tasks['task_2'], {'x_1': 2, 'x_2': 0.1, 'x_3': 0.9})}
        opt = GridSearchOptimizer(tmp + '/', small, nb_runs=2)
        fit = opt.fit(_opt_sim, tasks, data, _opt_loss, store_simulation_data=True)
        out['fit_keys'] = np.array(list(fit), dtype=np.float64)
        out['fit_values'] = np.array([fit[k] for k in fit], dtype=np.float64)
        out['fit_files'] = np.array(sorted(os.listdir(tmp)))
    np.savez_compressed(_out('optimizer_kat.npz'), **out)


def main():
    worlds = gen_worlds()
    gen_optimizer()
    gen_dyna_dsr()
    gen_dyna_dqn()
    gen_dqn()
    gen_topology()
    gen_gridworld_kat()
    gen_eps_greedy()
    gen_dynaq(worlds)
    gen_stochastic()
    gen_dynaq_memory()
    gen_qagent(worlds)
    gen_sr(worlds)
    gen_sr32(worlds)
    gen_monitor(worlds)
    gen_sfma()
    gen_qagent_topology()
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print('%-24s %8d B' % (f, os.path.getsize(_out(f))))


if __name__ == '__main__':
    if len(sys.argv) > 1:      # regenerate selected fixtures only: gen_golden.py gen_sfma ...
        for fn in sys.argv[1:]:
            globals()[fn]()
    else:
        main()
