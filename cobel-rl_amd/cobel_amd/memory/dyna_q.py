"""Tabular world model of Dyna-Q — ``cobel.memory.DynaQMemory`` (memory/dyna_q.py:17-157).

On device the model is one packed 8-byte record per (state, action):
``{float32 reward estimate, uint16 next state, uint8 nonterminal flag}``, ``[N, S, 4]``.
``rewards`` / ``states`` / ``terminals`` decode it into the reference's three tables.  During
``agent.train`` ``store`` and ``retrieve_batch`` are folded into the fused agent kernel
(cobel_tab_run); the methods here serve single host-side calls (memory/dyna_q.py:77-157) on one
instance and go through the same device table, the same float32 arithmetic and the same memory
stream (counter ``counter[instance]``, one vector draw per batch) as the kernel, so host calls and
launches can be interleaved freely.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


class DynaQMemory:
    def __init__(self, states: int, actions: int, learning_rate: float = 0.9, rng=None) -> None:
        self.rng = rng
        self.number_of_states = states
        self.number_of_actions = actions
        self.learning_rate = learning_rate
        self.table = None      # torch int64 [N, S, 4] packed records
        self.counter = None    # torch int32 [N] replay-batch counters (COBEL_STREAM_MEMORY)
        self.index = None      # torch int16 [N, S, 4] digest of `table` for the planning kernel

    def _bind(self, n_envs: int = 1, device=None, seed: int | None = None, base: int = 0) -> None:
        if self.table is not None:
            return
        if device is None:      # stand-alone use: one instance on the current GPU
            device = torch.device('cuda', torch.cuda.current_device())
        if seed is None:
            from ..interface.gridworld import _as_seed
            seed = _as_seed(self.rng)
        self.seed, self.base = int(seed), int(base)
        assert self.number_of_actions == 4, 'the model records are laid out for 4-action worlds'
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

    def _record(self, instance: int, state: int, action: int):
        """(float32 reward estimate, next state, nonterminal flag) of one pair."""
        import ctypes as C
        rec = int(self.table[instance, state, action].item()) & 0xFFFFFFFFFFFFFFFF
        r, ns, nt = C.c_float(), C.c_uint16(), C.c_uint8()
        _lib.lib().cobel_unpack_model(rec, C.byref(r), C.byref(ns), C.byref(nt))
        return np.float32(r.value), int(ns.value), int(nt.value)

    def store(self, experience: dict, instance: int = 0) -> None:
        """memory/dyna_q.py:77-96: ``rewards[s, a] += lr * (r - rewards[s, a])`` (float32, as the
        kernel computes it: d = r - R; R + lr * d), ``states[s, a] = next_state``,
        ``terminals[s, a] = terminal``."""
        self._bind()
        s, a = int(experience['state']), int(experience['action'])
        old, _, _ = self._record(instance, s, a)
        d = np.float32(experience['reward']) - old
        new = np.float32(old + np.float32(self.learning_rate) * d)
        rec = int(_lib.lib().cobel_pack_model(float(new), int(experience['next_state']),
                                              int(bool(experience['terminal']))))
        self.table[instance, s, a] = rec - (1 << 64) if rec >= (1 << 63) else rec
        digest = (int(experience['next_state']) & 0x3FFF) | (int(bool(experience['terminal'])) << 14) \
            | (0x8000 if np.float32(new).view(np.uint32) != 0 else 0)
        self.index[instance, s, a] = digest - (1 << 16) if digest >= (1 << 15) else digest

    def retrieve(self, state: int, action: int, instance: int = 0) -> dict:
        self._bind()
        r, s, t = self._record(instance, int(state), int(action))
        return {'state': state, 'action': action, 'reward': r, 'next_state': s, 'terminal': t}

    def retrieve_batch(self, batch_size: int = 32, instance: int = 0) -> list:
        """memory/dyna_q.py:122-157: ``batch_size`` pairs drawn uniformly over ALL S x A pairs in
        one vector draw of the memory stream (draw number ``counter[instance]``, element j from
        sub-stream j — what the planning kernel consumes for one replay), as experience dicts."""
        self._bind()
        assert batch_size > 0
        pairs = self.number_of_states * self.number_of_actions
        ctr = self.counter[instance: instance + 1]
        idx = torch.empty((1, batch_size), dtype=torch.int32, device=self.table.device)
        _lib.check(_lib.lib().cobel_rng_bounded(
            _lib.ptr(ctr), self.seed, _lib.STREAM_MEMORY, self.base + instance, pairs,
            _lib.ptr(idx), 1, batch_size, 1, _lib.current_stream(self.table.device)))
        idx = idx[0].cpu().numpy().astype(np.int64)
        recs = self.table[instance].reshape(-1)[torch.as_tensor(idx, device=self.table.device)]
        raw = recs.cpu().numpy()
        lo = (raw & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        hi = (raw >> 32) & 0xFFFFFFFF
        return [{'state': int(k // self.number_of_actions), 'action': int(k % self.number_of_actions),
                 'reward': lo[j], 'next_state': int(hi[j] & 0xFFFF),
                 'terminal': int((hi[j] >> 16) & 1)} for j, k in enumerate(idx)]
