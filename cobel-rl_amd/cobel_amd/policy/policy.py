"""Policy base class — contract of ``cobel.policy.policy.Policy`` (policy/policy.py:11-67):
``select_action(v, mask)`` and ``get_action_probs(v, mask)``.

A policy here does not own a ``numpy.random.Generator``: its uniform draws come from one of the
library's counter-based streams, identified by ``(seed, stream)`` and a per-instance draw
counter.  Agents adopt the policy's ``epsilon`` / ``stream`` for their fused kernels and keep
the counter in their per-instance state, so single calls of ``select_action`` and fused runs
continue one and the same sequence.
"""
from __future__ import annotations

import abc


class Policy(abc.ABC):
    def __init__(self, rng=None) -> None:
        self.rng = rng          # int seed / numpy Generator (seed donor) / None
        self.seed = None        # bound by the agent (or lazily) to a 64-bit seed
        self.stream = None      # library stream id; agents assign POLICY or POLICY_TEST
        self.counter = None     # torch int32 [N] draw counters, created on first use

    @abc.abstractmethod
    def select_action(self, v, mask=None):
        ...

    @abc.abstractmethod
    def get_action_probs(self, v, mask=None):
        ...
