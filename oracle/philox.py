"""TEST INFRASTRUCTURE — not part of the product path.

NumPy restatement of the build's counter-based random streams (Philox-4x32-10,
Salmon et al., SC'11 "Parallel random numbers: as easy as 1, 2, 3") and the
``TapeRNG`` object that feeds those streams INTO the reference's classes.

Why: the reference draws from three unseeded ``numpy.random.Generator`` objects
(env ``/root/reference/src/cobel/interface/gridworld.py:83``, policy
``src/cobel/policy/policy.py:30``, memory ``src/cobel/memory/dyna_q.py:69``;
QAgent replay ``src/cobel/agent/q.py:138``).  Every consumer only calls
``rng.random()``, ``rng.integers(lo, hi, size)`` and ``rng.choice(a, size, p)``
(gridworld.py:120,142; greedy.py:58; memory/dyna_q.py:137; q.py:353;
topology.py:109,170), so a duck-typed object can replace the Generator and make
the reference consume exactly the draws the HIP kernels generate on device.

Stream layout (shared with ``cobel-rl_amd/csrc/cobel_rng.h`` — keep in sync):

    key = (seed & 0xffffffff, seed >> 32)
    ctr = (block, sub, instance, stream)
    a stream is consumed through a draw counter c:
      bounded integer c : word (c & 3) of block (c >> 2)              -> mulhi32(word, n)
      uniform double  c : words 2(c & 1), 2(c & 1) + 1 of block (c >> 1)
    STREAM_ENV    = 0  c = number of resets so far
    STREAM_POLICY = 1  c = number of select_action()s
    STREAM_MEMORY = 2  c = number of replay batches, sub = position in the batch
    STREAM_AGENT  = 4  c = draws of the agent's own generator (SFMA dynamic mode, sfma.py:312)

A generator that mixes integer and double draws (SFMAMemory.rng: ``integers`` at the start of a
replay, ``choice(p=...)`` per reactivation, memory/sfma.py:258-333) takes its doubles from
``sub = double_sub + j`` (j = position in a vector draw) with ``double_sub = 1``: an integer draw
and a double draw then never share a Philox block, whatever their counters.

Bounded integers are ``(word * n) >> 32`` (Lemire multiply-shift without the
rejection step; bias <= n / 2**32).  Uniform doubles follow NumPy's recipe
``(a >> 5, b >> 6) -> (a * 2**26 + b) / 2**53``.
"""
from __future__ import annotations

import numpy as np

STREAM_ENV, STREAM_POLICY, STREAM_MEMORY, STREAM_AUX, STREAM_AGENT = 0, 1, 2, 3, 4

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32(ctr, key, rounds: int = 10):
    """Philox-4x32-R on arrays: ``ctr`` (..., 4) and ``key`` (..., 2) uint32.

    Returns the (..., 4) uint32 output block.
    """
    ctr = np.asarray(ctr, dtype=np.uint64)
    key = np.asarray(key, dtype=np.uint64)
    c0, c1, c2, c3 = (ctr[..., i] & _MASK for i in range(4))
    k0, k1 = key[..., 0] & _MASK, key[..., 1] & _MASK
    for r in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _block(seed: int, instance, index, sub, stream: int):
    instance, index, sub = np.broadcast_arrays(
        np.asarray(instance, dtype=np.uint64),
        np.asarray(index, dtype=np.uint64),
        np.asarray(sub, dtype=np.uint64),
    )
    ctr = np.stack(
        [index, sub, instance, np.full(index.shape, stream, dtype=np.uint64)], axis=-1
    )
    key = np.empty(index.shape + (2,), dtype=np.uint64)
    key[..., 0] = seed & 0xFFFFFFFF
    key[..., 1] = (seed >> 32) & 0xFFFFFFFF
    return philox4x32(ctr, key)


def draw_bounded(seed: int, instance, index, sub, stream: int, n):
    """Integer draw number ``index`` (array-like allowed) in [0, n); ``n`` may be an array."""
    index = np.asarray(index, dtype=np.uint64)
    b = _block(seed, instance, index >> np.uint64(2), sub, stream)
    w = np.take_along_axis(b, np.broadcast_to(index & np.uint64(3), b.shape[:-1])[..., None]
                           .astype(np.int64), axis=-1)[..., 0].astype(np.uint64)
    return ((w * np.asarray(n, dtype=np.uint64)) >> _S32).astype(np.int64)


def draw_double(seed: int, instance, index, sub, stream: int):
    """Double draw number ``index``: 53-bit uniform in [0, 1)."""
    index = np.asarray(index, dtype=np.uint64)
    b = _block(seed, instance, index >> np.uint64(1), sub, stream)
    k = (np.broadcast_to(index & np.uint64(1), b.shape[:-1]) * np.uint64(2)).astype(np.int64)
    hi = np.take_along_axis(b, k[..., None], axis=-1)[..., 0]
    lo = np.take_along_axis(b, k[..., None] + 1, axis=-1)[..., 0]
    a = (hi >> np.uint32(5)).astype(np.float64)
    c = (lo >> np.uint32(6)).astype(np.float64)
    return (a * 67108864.0 + c) / 9007199254740992.0


class TapeRNG:
    """Duck-typed stand-in for ``numpy.random.Generator`` driven by one stream.

    ``choice(a)`` == ``a[integers(0, len(a))]`` and ``choice(a, p=p)`` ==
    ``a[searchsorted(cumsum(p) / cumsum(p)[-1], random(), 'right')]`` reproduce
    NumPy's ``Generator.choice`` bit for bit (re-checked in
    tests/test_oracle_rng.py against a recording Generator).
    """

    def __init__(self, seed: int, instance: int, stream: int, start: int = 0,
                 double_sub: int = 0) -> None:
        self.seed, self.instance, self.stream = int(seed), int(instance), int(stream)
        self.index = int(start)
        self.double_sub = int(double_sub)
        self.log: list = []  # every value handed out, in order (for fixtures)

    def random(self, size=None):
        if size is None:
            u = float(draw_double(self.seed, self.instance, self.index, self.double_sub,
                                  self.stream))
            self.index += 1
            self.log.append(u)
            return u
        u = draw_double(self.seed, self.instance, self.index,
                        self.double_sub + np.arange(int(size)), self.stream)
        self.index += 1
        self.log.append(u.copy())
        return u

    def integers(self, low, high=None, size=None):
        if high is None:
            low, high = 0, low
        n = int(high) - int(low)
        assert 0 < n <= 2**32
        if size is None:
            k = int(draw_bounded(self.seed, self.instance, self.index, 0, self.stream, n))
            self.index += 1
            self.log.append(k + int(low))
            return np.int64(k + int(low))
        size = int(size)
        k = draw_bounded(
            self.seed, self.instance, self.index, np.arange(size), self.stream, n
        ) + int(low)
        self.index += 1
        self.log.append(k.copy())
        return k

    def choice(self, a, size=None, p=None):
        a = np.arange(a) if np.isscalar(a) else np.asarray(a)
        if p is None:
            return a[self.integers(0, len(a), size)]
        # Generator.choice converts p to float64 before the cumulative sum — and refuses NaN
        # probabilities (numpy/random/_generator.pyx: "probabilities contain NaN")
        p = np.asarray(p, dtype=np.float64)
        if np.isnan(p).any():
            raise ValueError('probabilities contain NaN')
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        return a[cdf.searchsorted(self.random(size), side='right')]
