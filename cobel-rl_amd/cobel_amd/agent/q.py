"""Tabular Q-learning with experience replay — ``cobel.agent.QAgent`` (agent/q.py:26-354).

Discrete observations (gridworlds) and the pose observations of a ``Topology`` (Box; the
reference keys Q by ``tuple(pose)``, here the key is the node index and ``predict_on_batch`` /
``Q_dict`` translate): the reference's lazily created dict rows are a dense zero table.  The replay memory is the per-instance log of experienced transitions
(q.py:143,213), kept on device; its capacity must be announced with ``reserve_replay`` (or is
sized by each ``train`` call).  ``batch_size=0`` disables replay (demo/topology/demo.py:76); the
experiences are logged all the same, as in the reference (``log_experiences = False`` skips that).
Any action count up to 8 (a hexagonal ``Topology`` has six) and any ``batch_size``: runs outside
what the wavefront kernels cover take the general kernel of ``cobel_tab_run``, same results.
"""
from __future__ import annotations

import torch

from .. import _lib
from ..spaces import Box, Dict, Discrete
from .tabular import TabularAgent


class QAgent(TabularAgent):
    agent_kind = _lib.AGENT_Q

    def __init__(self, observation_space, action_space, policy, policy_test=None,
                 learning_rate: float = 0.9, gamma: float = 0.8, custom_callbacks=None,
                 rng=None) -> None:
        # (Dict: the pre-rendered observations of an OfflineSimulator-backed Topology, which the
        #  reference concatenates in key order, agent/q.py:156-158)
        assert type(observation_space) in (Discrete, Box, Dict), \
            'Discrete observations, Topology poses (Box) and dictionary observations are accelerated'
        assert type(action_space) is Discrete, 'Wrong action space!'
        super().__init__(observation_space, action_space, policy, policy_test, learning_rate,
                         gamma, custom_callbacks)
        self.rng = rng
        self.nb_actions = self.n_actions
        self._log = None
        self._log_cap = 0
        # The reference appends every experience to ``M`` whether or not it replays (q.py:213), so
        # a later session with batch_size > 0 samples from all of them.  On device that is 8 B per
        # instance and step — of HBM for the log ([N, trials * steps] int64, reserved up front) and
        # of write traffic in the kernel that otherwise only touches LDS.  None (default): sessions
        # with batch_size 0 log while the log stays within ``log_budget_bytes`` and otherwise run
        # without one, with a warning (65 536 instances x 100 trials x 200 steps would reserve
        # 10 TB); True: always log (the reservation may fail); False: never log at batch_size 0.
        self.log_experiences = None
        self.log_budget_bytes = 4 << 30
        self._log_now = True

    def _log_words(self) -> int:
        """``COBEL_LOG_WORDS``: 64-bit words per logged experience (two where states and action
        do not fit beside the reward in one)."""
        return 2 if ((self.n_actions > 8 and self.n_states > 8192) or self.n_states > 16384) else 1

    def reserve_replay(self, entries: int) -> None:
        """Make room for ``entries`` logged experiences per instance (8 B each; 16 B in worlds
        beyond 16 384 states, or beyond 8 192 with more than eight actions)."""
        if entries <= self._log_cap:
            return
        w = self._log_words()
        new = torch.zeros((self.n_envs, entries * w), dtype=torch.int64, device=self.device)
        if self._log is not None:
            new[:, : self._log_cap * w] = self._log
        self._log, self._log_cap = new, entries

    @property
    def M(self):
        """Logged experiences of instance 0 as ``(state, action, reward, next_state, terminal)``."""
        if self._log is None:
            return []
        n = int(self.inst[0, _lib.I_LOG_LEN].item())
        if self._log_words() == 2:   # {reward, action | nonterminal << 8}, {state, next state}
            raw = self._log[0, :2 * n].cpu().numpy().reshape(n, 2)
            rew = (raw[:, 0] & 0xFFFFFFFF).astype('uint32').view('float32')
            meta = (raw[:, 0] >> 32) & 0xFFFFFFFF
            return [{'state': (int(w & 0xFFFFFFFF),), 'action': int(m & 0xFF), 'reward': float(r),
                     'next_state': (int((w >> 32) & 0xFFFFFFFF),), 'terminal': int((m >> 8) & 1)}
                    for r, m, w in zip(rew, meta, raw[:, 1])]
        raw = self._log[0, :n].cpu().numpy()
        lo = (raw & 0xFFFFFFFF).astype('uint32').view('float32')
        hi = (raw >> 32) & 0xFFFFFFFF
        # (more than four actions: three action bits, the flag moves up; more than eight: five
        #  action bits and 13-bit states — cobel_hip.h)
        if self.n_actions > 8:
            return [{'state': (int(h & 0x1FFF),), 'action': int((h >> 26) & 31), 'reward': float(r),
                     'next_state': (int((h >> 13) & 0x1FFF),), 'terminal': int((h >> 31) & 1)}
                    for r, h in zip(lo, hi)]
        a_mask, t_shift = (3, 30) if self.n_actions <= 4 else (7, 31)
        return [{'state': (int(h & 0x3FFF),), 'action': int((h >> 28) & a_mask), 'reward': float(r),
                 'next_state': (int((h >> 14) & 0x3FFF),), 'terminal': int((h >> t_shift) & 1)}
                for r, h in zip(lo, hi)]

    @property
    def Q_dict(self) -> dict:
        """The reference's view of Q for instance 0: ``{observation tuple: float32[4]}``."""
        q = self._q[0].cpu().numpy() if self._q is not None else self._q_host
        if self._poses is not None:
            return {tuple(p.flatten()): q[i] for i, p in enumerate(self._poses)}
        return {(i,): q[i] for i in range(q.shape[0])}

    def _extra(self, run) -> None:
        keep = run.batch > 0 or self._log_now
        run.replay_log = _lib.ptr(self._log) if keep else None
        run.log_cap = self._log_cap if keep else 0

    def train(self, interface, trials: int, steps: int = 32, batch_size: int = 32) -> None:
        assert batch_size >= 0     # (above _lib.MAX_BATCH: the general kernel, any size)
        self._bind(interface)
        used = int(self.inst[:, _lib.I_LOG_LEN].max().item())
        need = (used + trials * steps) * 8 * self._log_words() * self.n_envs
        self._log_now = bool(self.log_experiences) or (
            self.log_experiences is None and (need <= self.log_budget_bytes
                                              or (used + trials * steps) <= self._log_cap))
        if batch_size == 0 and self.log_experiences is None and not self._log_now:
            import warnings
            warnings.warn('QAgent.train(batch_size=0): logging every experience of this session '
                          'would reserve %.1f GB (%d instances x %d steps x 8 B, log_budget_bytes = '
                          '%.1f GB); the session runs without the log.  Set log_experiences = True '
                          'to force it, False to silence this.'
                          % (need / 1e9, self.n_envs, trials * steps, self.log_budget_bytes / 1e9))
        if batch_size > 0 or self._log_now:
            self.reserve_replay(used + trials * steps)
        self._session(interface, trials, steps, batch_size, True)

    def test(self, interface, trials: int, steps: int = 32) -> None:
        self._session(interface, trials, steps, 0, False)
