"""Tabular world model of Dyna-Q — ``cobel.memory.DynaQMemory`` (memory/dyna_q.py:17-157).

On device the model is one packed 8-byte record per (state, action):
``{float32 reward estimate, uint16 next state, uint8 nonterminal flag}``, ``[N, S, 4]``.
``rewards`` / ``states`` / ``terminals`` decode it into the reference's three tables.  ``store``
and ``retrieve_batch`` are folded into the fused agent kernel (cobel_tab_run); the methods here
serve single host-side calls on instance 0 and go through the same table.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


class DynaQMemory:
    def __init__(self, states: int, actions: int, learning_rate: float = 0.9, rng=None) -> None:
        assert actions == 4, 'the model record layout covers 4-action worlds'
        self.rng = rng
        self.number_of_states = states
        self.number_of_actions = actions
        self.learning_rate = learning_rate
        self.table = None      # torch int64 [N, S, 4] packed records
        self.counter = None    # torch int32 [N] replay-batch counters (COBEL_STREAM_MEMORY)
        self.index = None      # torch int16 [N, S, 4] digest of `table` for the planning kernel

    def _bind(self, n_envs: int, device) -> None:
        if self.table is not None:
            return
        self.table = torch.empty((n_envs, self.number_of_states, 4), dtype=torch.int64,
                                 device=device)
        _lib.check(_lib.lib().cobel_model_init(_lib.ptr(self.table), n_envs,
                                               self.number_of_states,
                                               _lib.current_stream(device)))
        self.counter = torch.zeros(n_envs, dtype=torch.int32, device=device)
        # 16-bit digest of the table read by the planning kernel (cobel_model_index_build)
        self.index = torch.empty((n_envs, self.number_of_states, 4), dtype=torch.int16,
                                 device=device)
        self.rebuild_index()

    def rebuild_index(self) -> None:
        """Re-derive the digest after editing ``table`` by hand."""
        _lib.check(_lib.lib().cobel_model_index_build(
            _lib.ptr(self.table), _lib.ptr(self.index), self.table.shape[0],
            self.number_of_states, _lib.current_stream(self.table.device)))

    def _decode(self):
        raw = self.table.cpu().numpy()
        lo = (raw & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        hi = (raw >> 32) & 0xFFFFFFFF
        return lo, (hi & 0xFFFF).astype(np.int64), ((hi >> 16) & 1).astype(np.int64)

    def _squeeze(self, a):
        return a[0] if a.shape[0] == 1 else a

    @property
    def rewards(self):
        return self._squeeze(self._decode()[0])

    @property
    def states(self):
        return self._squeeze(self._decode()[1])

    @property
    def terminals(self):
        return self._squeeze(self._decode()[2])

    def retrieve(self, state: int, action: int) -> dict:
        r, s, t = self._decode()
        return {'state': state, 'action': action, 'reward': r[0, state, action],
                'next_state': s[0, state, action], 'terminal': t[0, state, action]}
