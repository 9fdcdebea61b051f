"""Replay memory of the DQN agent — ``cobel.memory.DQNMemory`` (memory/dqn.py:32-219) as a
device-resident ring buffer, one ring per instance.

Semantics kept: FIFO with ``capacity`` (the reference drops the oldest entry once over capacity,
memory/dqn.py:113-119), uniform sampling with replacement over the stored entries
(``rng.integers(size, size=batch_size)``, :137) — here drawn from COBEL_STREAM_MEMORY with one
counter per instance — and ``retrieve`` returning ``(states, actions, rewards, next_states,
terminals)`` where ``terminals`` is the non-terminal flag ``1 - end_trial`` (agent/dqn.py:191).
The reference grows NumPy arrays with ``np.append`` on every store (O(n) per step); the ring
preallocates ``min(capacity, needed)`` slots on first use.
"""
from __future__ import annotations

import torch

from .. import _lib


class DQNMemory:
    def __init__(self, capacity: int = 100000, rng=None) -> None:
        assert capacity > 0, 'Memory capacity must be greater than zero!'
        self.capacity = capacity
        self.rng = rng
        self.n = None
        self.counter = None     # [N] int32 sample-batch counters (COBEL_STREAM_MEMORY)

    def _bind(self, n: int, obs_shape, dtype, device, slots: int, seed: int, base: int) -> None:
        slots = max(1, min(self.capacity, slots))
        if self.n is not None and slots <= self.slots:
            return
        old = None if self.n is None else (self.states, self.next_states, self.actions,
                                           self.rewards, self.terminals, self.slots)
        self.n, self.slots, self.device, self.seed, self.base = n, slots, device, seed, base
        self.states = torch.zeros((n, slots) + tuple(obs_shape), dtype=dtype, device=device)
        self.next_states = torch.zeros_like(self.states)
        self.actions = torch.zeros((n, slots), dtype=torch.int64, device=device)
        self.rewards = torch.zeros((n, slots), dtype=dtype, device=device)
        self.terminals = torch.zeros((n, slots), dtype=dtype, device=device)
        if old is None:
            self.size = torch.zeros(n, dtype=torch.int32, device=device)    # stored entries
            self.head = torch.zeros(n, dtype=torch.int64, device=device)    # slot of the oldest
            self.counter = torch.zeros(n, dtype=torch.int32, device=device)
            self._rows = torch.arange(n, device=device)
        else:   # grow: unroll the old rings so that logical order is preserved
            s, ns, a, r, t, old_slots = old
            order = (self.head[:, None] + torch.arange(old_slots, device=device)[None]) % old_slots
            take = lambda x: torch.gather(  # noqa: E731
                x, 1, order.reshape(order.shape + (1,) * (x.dim() - 2)).expand(
                    order.shape + x.shape[2:]))
            self.states[:, :old_slots] = take(s)
            self.next_states[:, :old_slots] = take(ns)
            self.actions[:, :old_slots], self.rewards[:, :old_slots] = take(a), take(r)
            self.terminals[:, :old_slots] = take(t)
            self.head.zero_()

    def store_batch(self, state, action, reward, next_state, nonterminal, active=None) -> None:
        """Append one experience per (active) instance."""
        rows = self._rows if active is None else self._rows[active]
        full = self.size[rows] >= self.slots
        slot = (self.head[rows] + self.size[rows].to(torch.int64)) % self.slots
        sel = (lambda x: x) if active is None else (lambda x: x[active])
        self.states[rows, slot] = sel(state)
        self.next_states[rows, slot] = sel(next_state)
        self.actions[rows, slot] = sel(action).to(torch.int64)
        self.rewards[rows, slot] = sel(reward).to(self.rewards.dtype)
        self.terminals[rows, slot] = sel(nonterminal).to(self.terminals.dtype)
        self.head[rows] = torch.where(full, (self.head[rows] + 1) % self.slots, self.head[rows])
        self.size[rows] = torch.where(full, self.size[rows], self.size[rows] + 1)

    def retrieve(self, batch_size: int = 32):
        """``(states, actions, rewards, next_states, terminals)``, each ``[N, batch, ...]``."""
        assert batch_size > 0
        idx = torch.empty((self.n, batch_size), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().cobel_rng_bounded_each(
            _lib.ptr(self.counter), self.seed, _lib.STREAM_MEMORY, self.base,
            _lib.ptr(self.size), _lib.ptr(idx), self.n, batch_size, 1,
            _lib.current_stream(self.device)))
        slot = (self.head[:, None] + idx.to(torch.int64)) % self.slots
        rows = self._rows[:, None]
        return (self.states[rows, slot], self.actions[rows, slot], self.rewards[rows, slot],
                self.next_states[rows, slot], self.terminals[rows, slot])
