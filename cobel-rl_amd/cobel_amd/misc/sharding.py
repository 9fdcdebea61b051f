"""How a batch of independent agent-env instances is split over the GPUs of a node.

Instances never interact, so rank g of G simply owns the contiguous range
of GLOBAL instance ids ``[base, base + count)``; random streams and the world of an instance are
functions of the global id (``Gridworld(..., instance_base=base)``), so a run's results do not
depend on G.  The reference has no counterpart (single process, one agent).
"""
from __future__ import annotations


def shard_instances(total: int, world_size: int, rank: int) -> tuple[int, int]:
    """(base, count) of rank ``rank``: contiguous, disjoint, complete, sizes differing by at
    most one (the first ``total % world_size`` ranks take the extra instance)."""
    assert world_size >= 1 and 0 <= rank < world_size and total >= 0
    q, r = divmod(int(total), int(world_size))
    base = rank * q + min(rank, r)
    return base, q + (1 if rank < r else 0)
