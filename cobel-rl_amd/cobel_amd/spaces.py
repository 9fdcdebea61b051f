"""Observation / action space descriptors.

The reference takes these from ``gymnasium.spaces`` (interface/gridworld.py:4,85-86) and its
agents only test ``type(space) is Discrete`` and read ``.n`` / ``.shape`` (agent/dyna_q.py:117-122).
When gymnasium is installed its classes are used, so objects are interchangeable with the
reference's; otherwise minimal stand-ins with the same attributes are defined here.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Discrete  # type: ignore
except Exception:  # gymnasium absent

    class Discrete:  # noqa: D101
        def __init__(self, n: int) -> None:
            self.n = np.int64(n)
            self.shape = ()
            self.dtype = np.int64

        def __repr__(self) -> str:
            return 'Discrete(%d)' % int(self.n)

        def __eq__(self, other) -> bool:
            return isinstance(other, Discrete) and int(other.n) == int(self.n)

    class Box:  # noqa: D101
        def __init__(self, low, high, shape=None, dtype=np.float64) -> None:
            self.low, self.high = np.asarray(low), np.asarray(high)
            self.shape = tuple(shape) if shape is not None else self.low.shape
            self.dtype = dtype


try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Dict, Tuple  # type: ignore
except Exception:  # gymnasium absent

    class Dict(dict):  # noqa: D101
        """``gymnasium.spaces.Dict``: named sub-spaces (``.spaces`` is the mapping itself)."""

        def __init__(self, spaces=None) -> None:
            dict.__init__(self, spaces or {})
            self.spaces = self

    class Tuple:  # noqa: D101
        """``gymnasium.spaces.Tuple``: a sequence of sub-spaces in ``.spaces``."""

        def __init__(self, spaces) -> None:
            self.spaces = list(spaces)

        def __len__(self) -> int:
            return len(self.spaces)

        def __getitem__(self, k):
            return self.spaces[k]
