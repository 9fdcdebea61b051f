"""Tabular successor-representation agent — ``cobel.agent.sr.SR`` (agent/sr.py:22-324).

Same constructor (note ``learning_rate`` defaults to 0.1 as in the code, not the 0.99 of the
reference's docstring), ``train`` / ``test`` / ``predict_on_batch`` / ``retrieve_q`` and the
attributes ``SR``, ``rewards``, ``transitions``.  On device: ``SR`` float32 ``[N, S, S]``, the
agent's transition table as indices ``T[N, S, 4]`` (uint16) instead of the one-hot
``[S, A, S]`` tensor, ``rewards`` float32 ``[N, S]``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..spaces import Discrete
from .agent import FusedAgent


class SR(FusedAgent):
    def __init__(self, observation_space, action_space, policy, policy_test=None,
                 learning_rate: float = 0.1, gamma: float = 0.99, custom_callbacks=None) -> None:
        assert type(observation_space) is Discrete, 'SR requires a discrete observation space!'
        assert type(action_space) is Discrete, 'SR requires a discrete action space!'
        super().__init__(observation_space, action_space, policy, policy_test, custom_callbacks)
        self.learning_rate = learning_rate
        self.gamma = gamma
        self._sr = self._T = self._rw = None
        self.stream_rows = False   # True: always the row-streaming kernel (A/B measurements)
        self.traffic = None        # [4] device counters of the sparse-reward kernel (cobel_hip.h)

    def _alloc_tables(self) -> None:
        S, N = self.n_states, self.n_envs
        self._sr = torch.empty((N, S, S), dtype=torch.float32, device=self.device)
        self._T = torch.empty((N, S, 4), dtype=torch.int16, device=self.device)
        self._rw = torch.empty((N, S), dtype=torch.float32, device=self.device)
        self.traffic = torch.zeros(4, dtype=torch.int64, device=self.device)
        _lib.check(_lib.lib().cobel_sr_init(_lib.ptr(self._sr), _lib.ptr(self._T),
                                            _lib.ptr(self._rw), N, S,
                                            _lib.current_stream(self.device)))

    def _view(self, t):
        return t[0].cpu().numpy() if self.n_envs == 1 else t

    @property
    def SR(self):
        return np.eye(self.n_states, dtype=np.float32) if self._sr is None else self._view(self._sr)

    @property
    def rewards(self):
        return np.zeros(self.n_states, dtype=np.float32) if self._rw is None else self._view(self._rw)

    @property
    def T(self):
        """Learned successor of each (state, action) as an index table ``[S, 4]``."""
        if self._T is None:
            return np.repeat(np.arange(self.n_states), 4).reshape(-1, 4)
        return self._view(self._T)

    @property
    def transitions(self):
        """One-hot ``[S, A, S]`` form of ``T`` for instance 0 (sr.py:131-135)."""
        T = np.asarray(self.T if self.n_envs in (None, 1) else self._T[0].cpu().numpy()).astype(int)
        out = np.zeros((self.n_states, 4, self.n_states))
        out[np.arange(self.n_states)[:, None], np.arange(4)[None, :], T] = 1.0
        return out

    def _launch(self, interface, pol, flags, trials_target, steps, budget, batch) -> None:
        mon = self.monitors
        run = _lib.SRRun()
        run.sr, run.trans, run.rewards = _lib.ptr(self._sr), _lib.ptr(self._T), _lib.ptr(self._rw)
        run.inst = _lib.ptr(self.inst)
        self._mask_dev = self._mask_bits() if (flags & _lib.F_MASK_ACTIONS) else None
        run.action_mask = _lib.ptr(self._mask_dev)
        run.lat_sum, run.lat_cnt = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt'))
        run.reward_sum, run.lat_trace = _lib.ptr(mon.raw('reward_sum')), _lib.ptr(mon.lat_trace)
        run.resp_cnt = _lib.ptr(mon.raw('resp_cnt'))
        run.mon_stripes = mon.stripes
        run.occupancy, run.steps_done = _lib.ptr(mon.occupancy), _lib.ptr(mon.steps_done)
        run.last_exp = _lib.ptr(self._last_exp) if budget == 1 else None
        run.n, run.trial_cap = self.n_envs, mon.cap
        run.instance_base = interface.instance_base
        run.flags = flags | (_lib.F_SR_STREAM_ROWS if self.stream_rows else 0)
        run.trials_target, run.steps_per_trial, run.step_budget = trials_target, steps, budget
        run.seed = interface.seed
        run.traffic = _lib.ptr(self.traffic)
        self._hyper(run, self.learning_rate, self.gamma, pol.epsilon)
        _lib.check(_lib.lib().cobel_sr_run(interface.handle.ptr, C.byref(run),
                                           _lib.current_stream(self.device)))

    def train(self, interface, trials: int, steps: int) -> None:
        self._session(interface, trials, steps, 0, True)

    def test(self, interface, trials: int, steps: int) -> None:
        self._session(interface, trials, steps, 0, False)

    def retrieve_q(self, state):
        """Q-values ``V[T[s, a]]`` for a state (int, one instance) or per-instance states."""
        assert self._sr is not None, 'retrieve_q needs a bound agent (train or test first)'
        states = torch.as_tensor(np.atleast_1d(state) if not torch.is_tensor(state) else state,
                                 device=self.device).to(torch.int32)
        if states.numel() == 1 and self.n_envs > 1:
            states = states.expand(self.n_envs)
        states = states.contiguous()
        q = torch.empty((self.n_envs, 4), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().cobel_sr_retrieve_q(
            _lib.ptr(self._sr), _lib.ptr(self._T), _lib.ptr(self._rw), _lib.ptr(states),
            _lib.ptr(q), self.n_envs, self.n_states, _lib.current_stream(self.device)))
        return q[0].cpu().numpy() if self.n_envs == 1 else q

    def predict_on_batch(self, batch):
        out = [self.retrieve_q(int(s)) for s in np.array(batch).astype(int)]
        return np.array(out) if self.n_envs == 1 else torch.stack(out, dim=1)
