"""Epsilon-greedy policy — ``cobel.policy.greedy.EpsilonGreedy`` (policy/greedy.py:11-88).

``get_action_probs`` / ``select_action`` evaluate on the GPU through ``cobel_eps_greedy``: the
probabilities are float64, ties are exact equality on the float32 values, and the action is
``searchsorted(cumsum(p) / cumsum(p)[-1], u, 'right')`` exactly like ``Generator.choice``.
Inputs may be one row ``v[A]`` (returns an int / ``p[A]``, as the reference) or a batch
``v[N, A]``; A = 4 takes ``cobel_eps_greedy``, any other count up to 8 ``cobel_eps_greedy_n``.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .policy import Policy


def _mask_bits(mask, n: int, device, actions: int = 4):
    if mask is None:
        return None
    m = torch.as_tensor(np.asarray(mask), device=device).reshape(n, actions).to(torch.int64)
    assert bool((m.sum(dim=1) > 0).all()), 'The action mask masks all actions!'
    w = (1 << torch.arange(actions, dtype=torch.int64, device=device))
    bits = (m * w).sum(dim=1)
    # (one byte per row up to eight actions, one 32-bit word beyond: cobel_hip.h)
    if actions <= 8:
        return bits.to(torch.uint8).contiguous()
    # (the low words of the little-endian int64 sums)
    return bits.contiguous().view(torch.int32)[::2].contiguous()


class EpsilonGreedy(Policy):
    def __init__(self, epsilon: float = 0.1, rng=None) -> None:
        super().__init__(rng)
        assert np.all(np.asarray(epsilon) >= 0.0) and np.all(np.asarray(epsilon) <= 1.0)
        self.epsilon = epsilon

    # -- helpers ----------------------------------------------------------------------------
    def _bind(self, n: int, device) -> None:
        from ..interface.gridworld import _as_seed
        if self.seed is None:
            self.seed = _as_seed(self.rng)
        if self.stream is None:
            self.stream = _lib.STREAM_POLICY
        if self.counter is None or self.counter.numel() != n or self.counter.device != device:
            self.counter = torch.zeros(n, dtype=torch.int32, device=device)

    def _run(self, v, mask, u):
        device = torch.device('cuda', torch.cuda.current_device())
        vals = torch.as_tensor(np.asarray(v, dtype=np.float32) if not torch.is_tensor(v) else v)
        single = vals.dim() == 1
        A = int(vals.shape[-1])
        assert 1 <= A <= _lib.MAX_ACTIONS, 'up to %d actions' % _lib.MAX_ACTIONS
        vals = vals.reshape(-1, A).to(device=device, dtype=torch.float32).contiguous()
        n = vals.shape[0]
        bits = _mask_bits(mask, n, device, A)
        if u is None:
            self._bind(n, device)
            u = torch.empty(n, dtype=torch.float64, device=device)
            _lib.check(_lib.lib().cobel_rng_uniform(
                _lib.ptr(self.counter), self.seed, self.stream, 0, _lib.ptr(u), n, 1,
                _lib.current_stream(device)))
        else:
            u = torch.as_tensor(u, dtype=torch.float64, device=device).reshape(n).contiguous()
        act = torch.empty(n, dtype=torch.uint8, device=device)
        probs = torch.empty((n, A), dtype=torch.float64, device=device)
        if A == 4:
            _lib.check(_lib.lib().cobel_eps_greedy(
                _lib.ptr(vals), _lib.ptr(bits), _lib.ptr(u), float(self.epsilon), _lib.ptr(act),
                _lib.ptr(probs), n, _lib.current_stream(device)))
        else:
            _lib.check(_lib.lib().cobel_eps_greedy_n(
                _lib.ptr(vals), _lib.ptr(bits), _lib.ptr(u), float(self.epsilon), _lib.ptr(act),
                _lib.ptr(probs), n, A, _lib.current_stream(device)))
        return single, act, probs

    # -- reference surface ------------------------------------------------------------------
    def select_action(self, v, mask=None, u=None):
        """Action(s) for Q-value row(s) ``v``; ``u`` injects the uniform draw(s) (tests)."""
        single, act, _ = self._run(v, mask, u)
        return int(act[0].item()) if single else act

    def get_action_probs(self, v, mask=None):
        single, _, probs = self._run(v, mask, 0.0 if np.ndim(v) == 1 else np.zeros(len(v)))
        p = probs.cpu().numpy()
        return p[0] if single else p
