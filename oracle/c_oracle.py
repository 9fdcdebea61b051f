"""TEST INFRASTRUCTURE — ctypes front end of oracle/cobel_oracle.c (the many-instance C
restatement of the reference's hot loop).  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libcobel_oracle.so')

F_LEARN, F_NO_REPLAY, F_EPISODIC, F_MASK, F_TEST_STREAM = 1, 2, 4, 8, 16
AG_Q, AG_DYNAQ = 0, 1


class World(C.Structure):
    _fields_ = [('n_states', C.c_int32), ('n_worlds', C.c_int32), ('next', C.c_void_p),
                ('reward', C.c_void_p), ('terminal', C.c_void_p), ('starts', C.c_void_p),
                ('start_off', C.c_void_p)]


class Inst(C.Structure):
    _fields_ = [('state', C.c_int32), ('step', C.c_int32), ('trial', C.c_int32),
                ('ctr_env', C.c_uint32), ('ctr_policy', C.c_uint32), ('ctr_memory', C.c_uint32),
                ('log_len', C.c_uint32), ('flags', C.c_uint32), ('trial_reward', C.c_double),
                ('steps', C.c_uint64)]


INST_DTYPE = np.dtype([('state', 'i4'), ('step', 'i4'), ('trial', 'i4'), ('ctr_env', 'u4'),
                       ('ctr_policy', 'u4'), ('ctr_memory', 'u4'), ('log_len', 'u4'),
                       ('flags', 'u4'), ('trial_reward', 'f8'), ('steps', 'u8')], align=True)
assert INST_DTYPE.itemsize == C.sizeof(Inst)


class Cfg(C.Structure):
    _fields_ = [('n', C.c_int32), ('agent', C.c_int32), ('f32', C.c_int32), ('batch', C.c_int32),
                ('trials_target', C.c_int32), ('steps_per_trial', C.c_int32),
                ('step_budget', C.c_int32), ('trial_cap', C.c_int32), ('log_cap', C.c_int32),
                ('instance_base', C.c_uint32), ('flags', C.c_uint32), ('alpha', C.c_double),
                ('gamma', C.c_double), ('epsilon', C.c_double), ('model_lr', C.c_double),
                ('seed', C.c_uint64)]


_lib = None


def lib(path: str | None = None):
    global _lib
    if _lib is None or path is not None:
        target = path or os.environ.get('COBEL_ORACLE_LIB') or LIB   # e.g. the `make asan` build
        if path is None and (not os.path.exists(LIB) or os.path.getmtime(LIB) <
                             os.path.getmtime(os.path.join(HERE, 'cobel_oracle.c'))):
            subprocess.check_call(['make', '-C', HERE], stdout=subprocess.DEVNULL)
        h = C.CDLL(target)
        h.orc_eps_greedy.restype = C.c_int
        h.orc_eps_greedy.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        h.orc_pairwise_dot.restype = C.c_double
        h.orc_pairwise_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        h.orc_philox.restype = None
        h.orc_tab_run.restype = C.c_int
        h.orc_sr_run.restype = C.c_int
        if path is not None:
            return h
        _lib = h
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleWorld:
    """Compact tables of 1..W worlds (same S) in the layout the C oracle takes."""

    def __init__(self, tabs: list) -> None:
        self.S = int(np.asarray(tabs[0]['next']).shape[0])
        self.W = len(tabs)
        self.next = np.ascontiguousarray(np.stack([t['next'] for t in tabs]), dtype=np.uint16)
        self.reward = np.ascontiguousarray(np.stack([t['reward'] for t in tabs]), dtype=np.float64)
        self.terminal = np.ascontiguousarray(np.stack([t['terminal'] for t in tabs]) != 0,
                                             dtype=np.uint8)
        self.starts = np.ascontiguousarray(np.concatenate([t['starts'] for t in tabs]),
                                           dtype=np.uint16)
        self.off = np.zeros(self.W + 1, dtype=np.int32)
        self.off[1:] = np.cumsum([len(t['starts']) for t in tabs])
        self.c = World(self.S, self.W, self.next.ctypes.data, self.reward.ctypes.data,
                       self.terminal.ctypes.data, self.starts.ctypes.data, self.off.ctypes.data)


class TabOracle:
    """N instances of the tabular agents (Q / Dyna-Q); state persists across ``run`` calls."""

    def __init__(self, world: OracleWorld, n: int, agent: int, seed: int, f32: bool = True,
                 instance_base: int = 0, alpha=0.99, gamma=0.99, epsilon=0.1, model_lr=0.9,
                 trial_cap: int = 0, log_cap: int = 0, action_mask=None, occupancy=False) -> None:
        S = world.S
        self.world, self.n, self.agent = world, n, agent
        self.cfg = Cfg(n=n, agent=agent, f32=int(f32), instance_base=instance_base, alpha=alpha,
                       gamma=gamma, epsilon=epsilon, model_lr=model_lr, seed=seed,
                       trial_cap=trial_cap, log_cap=log_cap)
        self.inst = np.zeros(n, dtype=INST_DTYPE)
        self.inst['ctr_env'] = 1   # Gridworld.__init__ already consumed draw 0 (gridworld.py:89)
        self.Q = np.zeros((n, S, 4))
        self.MR = self.MS = self.MT = None
        if agent == AG_DYNAQ:
            self.MR = np.zeros((n, S, 4))
            self.MS = np.ascontiguousarray(
                np.broadcast_to(np.arange(S, dtype=np.int32)[None, :, None], (n, S, 4)))
            self.MT = np.zeros((n, S, 4), dtype=np.int32)
        self.log = None
        if log_cap:
            self.log = [np.zeros((n, log_cap), dtype=np.int32), np.zeros((n, log_cap), np.int32),
                        np.zeros((n, log_cap)), np.zeros((n, log_cap), np.int32),
                        np.zeros((n, log_cap), np.int32)]
        self.mask = None
        if action_mask is not None:
            m = np.asarray(action_mask, dtype=bool).reshape(S, 4)
            self.mask = (m * np.array([1, 2, 4, 8])).sum(axis=1).astype(np.uint8)
        self.lat_trace = np.full((n, max(trial_cap, 1)), -1, dtype=np.int32)
        self.lat_sum = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.lat_cnt = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.reward_sum = np.zeros(max(trial_cap, 1))
        self.resp_cnt = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.occupancy = np.zeros((world.W, S), dtype=np.uint64) if occupancy else None

    def run(self, trials_target: int, steps_per_trial: int, batch: int = 0, flags: int = F_LEARN,
            step_budget: int = 0, trace_inst: int = -1, trace_cap: int = 0, epsilon=None):
        c = self.cfg
        c.trials_target, c.steps_per_trial, c.batch = trials_target, steps_per_trial, batch
        c.flags, c.step_budget = flags | (F_MASK if self.mask is not None else 0), step_budget
        if epsilon is not None:
            c.epsilon = epsilon
        trace = np.zeros((max(trace_cap, 1), 6))
        tlen = C.c_int64(0)
        lg = self.log or [None] * 5
        rc = lib().orc_tab_run(
            C.byref(self.world.c), C.byref(c), _p(self.inst), _p(self.Q), _p(self.MR),
            _p(self.MS), _p(self.MT), _p(lg[0]), _p(lg[1]), _p(lg[2]), _p(lg[3]), _p(lg[4]),
            _p(self.mask), _p(self.lat_trace), _p(self.lat_sum), _p(self.lat_cnt),
            _p(self.reward_sum), _p(self.resp_cnt), _p(self.occupancy), C.c_int32(trace_inst),
            _p(trace) if trace_cap else None, C.c_int64(trace_cap), C.byref(tlen))
        assert rc == 0
        return trace[: tlen.value]


class SROracle:
    def __init__(self, world: OracleWorld, n: int, seed: int, f32: bool = True,
                 instance_base: int = 0, alpha=0.1, gamma=0.99, epsilon=0.1, trial_cap: int = 0,
                 action_mask=None, occupancy=False) -> None:
        S = world.S
        self.world, self.n = world, n
        self.cfg = Cfg(n=n, f32=int(f32), instance_base=instance_base, alpha=alpha, gamma=gamma,
                       epsilon=epsilon, seed=seed, trial_cap=trial_cap)
        self.inst = np.zeros(n, dtype=INST_DTYPE)
        self.inst['ctr_env'] = 1   # Gridworld.__init__ already consumed draw 0 (gridworld.py:89)
        self.SR = np.ascontiguousarray(np.broadcast_to(np.eye(S)[None], (n, S, S)))
        self.T = np.ascontiguousarray(
            np.broadcast_to(np.arange(S, dtype=np.int32)[None, :, None], (n, S, 4)))
        self.RW = np.zeros((n, S))
        self.mask = None
        if action_mask is not None:
            m = np.asarray(action_mask, dtype=bool).reshape(S, 4)
            self.mask = (m * np.array([1, 2, 4, 8])).sum(axis=1).astype(np.uint8)
        self.lat_trace = np.full((n, max(trial_cap, 1)), -1, dtype=np.int32)
        self.lat_sum = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.lat_cnt = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.reward_sum = np.zeros(max(trial_cap, 1))
        self.resp_cnt = np.zeros(max(trial_cap, 1), dtype=np.uint64)
        self.occupancy = np.zeros((world.W, S), dtype=np.uint64) if occupancy else None

    def run(self, trials_target: int, steps_per_trial: int, flags: int = F_LEARN,
            step_budget: int = 0, trace_inst: int = -1, trace_cap: int = 0):
        c = self.cfg
        c.trials_target, c.steps_per_trial = trials_target, steps_per_trial
        c.flags, c.step_budget = flags | (F_MASK if self.mask is not None else 0), step_budget
        trace = np.zeros((max(trace_cap, 1), 6))
        qtrace = np.zeros((max(trace_cap, 1), 4))
        tlen = C.c_int64(0)
        rc = lib().orc_sr_run(
            C.byref(self.world.c), C.byref(c), _p(self.inst), _p(self.SR), _p(self.T),
            _p(self.RW), _p(self.mask), _p(self.lat_trace), _p(self.lat_sum), _p(self.lat_cnt),
            _p(self.reward_sum), _p(self.resp_cnt), _p(self.occupancy), C.c_int32(trace_inst),
            _p(trace) if trace_cap else None, _p(qtrace) if trace_cap else None,
            C.c_int64(trace_cap), C.byref(tlen))
        assert rc == 0
        return trace[: tlen.value], qtrace[: tlen.value]


def eps_greedy(v, mask_bits: int, eps: float, u: float):
    v = np.ascontiguousarray(v, dtype=np.float64)
    p = np.zeros(4)
    a = lib().orc_eps_greedy(_p(v), mask_bits, eps, u, _p(p))
    return a, p


def pairwise_dot(a, b, f32: bool) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().orc_pairwise_dot(_p(a), _p(b), len(a), int(f32))
