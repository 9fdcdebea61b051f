"""Memory module of the SFMA agent — ``cobel.memory.SFMAMemory`` (memory/sfma.py:19-416).

Same constructor and attributes as the reference.  ``store`` / ``replay`` /
``retrieve_random_batch`` are folded into the fused kernel (``cobel_sfma_run``); this class owns
the parameters and the device tables:

  ``table``     packed model records [N, S, 4] (float32 reward estimate, next state, nonterminal),
                decoded by ``rewards`` / ``states`` / ``terminals``
  ``strength``  experience strengths ``C`` float64 [N, 4S], experience index a * S + s
  ``stamp``     store clock of each experience; ``T`` is derived from it (the reference rescales the
                whole recency vector on every store: T[j] = decay_recency ** age, by repeated
                multiplication, 0 after the end-of-trial reset)
  ``state``     per-instance words: clock, epoch, replay mode, |TD| sum, agent-stream counter
``I`` (inhibition) only lives inside a replay and is not kept.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


class SFMAMemory:
    def __init__(self, metric, nb_states: int, nb_actions: int, decay_inhibition: float = 0.9,
                 decay_strength: float = 1.0, learning_rate: float = 0.9, rng=None) -> None:
        assert nb_actions == 4, 'the model record layout covers 4-action worlds'
        self.rng = rng
        self.nb_states, self.nb_actions = nb_states, nb_actions
        self.decay_inhibition, self.decay_strength = decay_inhibition, decay_strength
        self.decay_recency = 0.9
        self.learning_rate = learning_rate
        self.beta = 20
        self.rlAgent = None
        self.reward_mod_local = self.error_mod_local = False
        self.reward_mod = self.error_mod = False
        self.policy_mod = self.state_mod = False
        self.metric = metric
        self.C_step = self.I_step = 1.0
        self.R_threshold = 10.0 ** -6
        self.deterministic = self.recency = False
        self.C_normalize = self.D_normalize = False
        self.R_normalize = True
        self.mode = 'default'
        self.reward_modulation = 1.0
        self.blend = 0.1
        self.interpolation_fwd, self.interpolation_rev = 0.5, 0.5
        self.table = self.strength = self.stamp = self.state = self.counter = None
        self._metric_dev = self._metric_src = None
        self._recency = None

    # -- device state ---------------------------------------------------------------------------
    def _bind(self, n_envs: int, device) -> None:
        if self.table is not None:
            return
        S = self.nb_states
        self.table = torch.empty((n_envs, S, 4), dtype=torch.int64, device=device)
        _lib.check(_lib.lib().cobel_model_init(_lib.ptr(self.table), n_envs, S,
                                               _lib.current_stream(device)))
        self.strength = torch.zeros((n_envs, 4 * S), dtype=torch.float64, device=device)
        self.stamp = torch.zeros((n_envs, 4 * S), dtype=torch.int32, device=device)
        self.state = torch.zeros((n_envs, _lib.SI_WORDS), dtype=torch.int32, device=device)
        self.state[:, _lib.SI_FLAGS] = 1       # agent.td starts as a weak Python float
        self.state[:, _lib.SI_MODE] = self._mode_id()
        self._mode_seen = self.mode
        self.counter = torch.zeros(n_envs, dtype=torch.int32, device=device)

    def _metric_on(self, device, n_worlds: int):
        """metric.D on the device, [n_worlds, S, S]: one matrix shared by all worlds, or a stack."""
        D = np.asarray(self.metric.D if hasattr(self.metric, 'D') else self.metric,
                       dtype=np.float64)
        if self._metric_dev is None or self._metric_src is not D:
            S = self.nb_states
            if D.ndim == 2:
                D3 = np.broadcast_to(D, (n_worlds, S, S))
            else:
                D3 = D
            assert D3.shape == (n_worlds, S, S), 'metric.D must be [S, S] or [n_worlds, S, S]'
            self._metric_dev = torch.as_tensor(np.array(D3, dtype=np.float64, order='C'), device=device)
            self._metric_src = D
        return self._metric_dev

    def _recency_table(self, device):
        """1, d, fl(d d), ...: what ``T *= decay_recency`` leaves after k stores."""
        if self._recency is None or self._recency[0] != self.decay_recency:
            vals, v = [1.0], 1.0
            while len(vals) < 16384:
                nv = v * self.decay_recency
                if nv == v:
                    break
                vals.append(nv)
                v = nv
            self._recency = (self.decay_recency,
                             torch.as_tensor(np.array(vals, dtype=np.float64), device=device))
        return self._recency[1]

    def _mode_id(self) -> int:
        # memory/sfma.py:289-307 is an if/elif chain over the known names: any other string
        # (unit_tests/test_sfma.py assigns 'dynamic') leaves the similarity as in 'default'
        return _lib.SFMA_MODES.index(self.mode) if self.mode in _lib.SFMA_MODES else 0

    def _sync_mode(self) -> None:
        if self.mode != self._mode_seen:
            self.state[:, _lib.SI_MODE] = self._mode_id()
            self._mode_seen = self.mode

    def _read_mode(self) -> None:
        if self.state.shape[0] == 1:
            self.mode = self._mode_seen = _lib.SFMA_MODES[int(self.state[0, _lib.SI_MODE])]

    # -- the reference's tables -----------------------------------------------------------------
    def _squeeze(self, a):
        return a[0] if a.shape[0] == 1 else a

    def _decode(self):
        raw = self.table.cpu().numpy()
        lo = (raw & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        hi = (raw >> 32) & 0xFFFFFFFF
        return lo, (hi & 0xFFFF).astype(np.int64), ((hi >> 16) & 1).astype(np.int64)

    @property
    def rewards(self):
        return self._squeeze(self._decode()[0])

    @property
    def states(self):
        return self._squeeze(self._decode()[1])

    @property
    def terminals(self):
        return self._squeeze(self._decode()[2])

    @property
    def C(self):
        return self._squeeze(self.strength.cpu().numpy())

    @C.setter
    def C(self, value) -> None:
        self.strength.copy_(torch.as_tensor(np.asarray(value, dtype=np.float64),
                                            device=self.strength.device).expand_as(self.strength))

    @property
    def T(self):
        tab = self._recency_table(self.stamp.device).cpu().numpy()
        st = self.stamp.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        w = self.state.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        clock, epoch = w[:, _lib.SI_CLOCK:_lib.SI_CLOCK + 1], w[:, _lib.SI_EPOCH:_lib.SI_EPOCH + 1]
        age = np.minimum(clock - st, len(tab) - 1)
        return self._squeeze(np.where(st > epoch, tab[np.maximum(age, 0)], 0.0))

    @property
    def I(self):  # noqa: E743
        return np.zeros(self.nb_states)

    @property
    def modes(self):
        """Replay mode of every instance (they differ in dynamic mode)."""
        return [_lib.SFMA_MODES[int(m)] for m in self.state[:, _lib.SI_MODE].cpu().numpy()]

    def retrieve(self, state: int, action: int) -> dict:
        r, s, t = self._decode()
        return {'state': state, 'action': action, 'reward': r[0, state, action],
                'next_state': s[0, state, action], 'terminal': t[0, state, action]}
