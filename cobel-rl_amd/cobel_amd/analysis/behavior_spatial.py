"""Spatial occupancy maps — ``cobel.analysis.behavior_spatial.get_occupancy_map``
(analysis/behavior_spatial.py:9-73) — and ``match`` (:76-107: how many states of a template
sequence line up with a state sequence at every offset).

Two entry points:
  * ``get_occupancy_map(trajectories, width, height, bin_size, margins)`` keeps the reference's
    signature and result (offline analysis of coordinate lists on the host);
  * ``occupancy_from_counts(counts, coordinates, ...)`` builds the same map from the per-state
    visit counts the agent kernels accumulate on device (``agent.monitors.occupancy``, one
    increment per env step at the state entered — what a ``TrajectoryMonitor`` registered under
    ``on_step_end`` would have recorded, monitor/behavior.py:371-374), summed over instances and
    ranks.
Quirk kept from the reference: the x coordinate is binned along the ``height`` bin count and y
along ``width`` (behavior_spatial.py:50,63-68); it is invisible for square arenas.
"""
from __future__ import annotations

from typing import Literal

import numpy as np


def _bins(width: float, height: float, bin_size: float, margins: str):
    assert width > 0 and height > 0, 'Invalid environment dimensions! Dimensions must be positive!'
    assert bin_size > 0 and bin_size <= min(width, height), (
        'Invalid bin size! Bin size must be positive and less than environmental dimensions!')
    assert margins in ['expand', 'include', 'ignore'], (
        "Invalid handling mode for margins! Must be 'expand', 'include' or 'ignore'!")
    bins = np.array([int(height / bin_size), int(width / bin_size)])
    if margins == 'expand':
        bins += (np.array([height, width]) - bins * bin_size) > 0.0
    return bins


def _bin_index(v: np.ndarray, n_bins: int, bin_size: float) -> np.ndarray:
    """Index of the uniform bin holding v over [0, n_bins * bin_size]; -1 outside.  The last bin
    is closed on the right, as in numpy.histogram."""
    edges = np.linspace(0.0, n_bins * bin_size, n_bins + 1)
    idx = np.searchsorted(edges, v, side='right') - 1
    idx[v == edges[-1]] = n_bins - 1
    idx[(v < edges[0]) | (v > edges[-1])] = -1
    return idx


def _accumulate(occ, xs, ys, weights, bins, bin_size, margins):
    if margins == 'include':
        xs = np.clip(xs, 0, bins[0] * bin_size)
        ys = np.clip(ys, 0, bins[1] * bin_size)
    ix, iy = _bin_index(xs, bins[0], bin_size), _bin_index(ys, bins[1], bin_size)
    ok = (ix >= 0) & (iy >= 0)
    np.add.at(occ, (ix[ok], iy[ok]), weights[ok] if weights is not None else 1.0)


def get_occupancy_map(trajectories: list, width: float, height: float, bin_size: float,
                      margins: Literal['expand', 'include', 'ignore'] = 'expand') -> np.ndarray:
    bins = _bins(width, height, bin_size, margins)
    occ = np.zeros(tuple(bins))
    for traj in trajectories:
        traj = np.asarray(traj)
        assert len(traj.shape) == 2 and traj.shape[1] == 2, 'Trajectories must be 2-dimensional!'
        assert np.amin(traj) >= 0.0, 'Invalid coordinates! Coordinates must be non-negative!'
        _accumulate(occ, traj[:, 0].astype(float), traj[:, 1].astype(float), None, bins,
                    bin_size, margins)
    return occ


def occupancy_from_counts(counts, coordinates, width: float, height: float, bin_size: float,
                          margins: Literal['expand', 'include', 'ignore'] = 'expand') -> np.ndarray:
    """Occupancy map from per-state visit counts ``counts[S]`` and ``world['coordinates']``."""
    counts = np.asarray(counts.cpu() if hasattr(counts, 'cpu') else counts, dtype=float)
    coordinates = np.asarray(coordinates, dtype=float)
    bins = _bins(width, height, bin_size, margins)
    occ = np.zeros(tuple(bins))
    _accumulate(occ, coordinates[:, 0], coordinates[:, 1], counts, bins, bin_size, margins)
    return occ


def match(sequence: np.ndarray, template: np.ndarray) -> np.ndarray:
    """Number of matching states between a sequence of state indices and a template laid over it
    at every offset: ``out[t] = #{j < len(template) : t + j < len(sequence) and template[j] ==
    sequence[t + j]}`` for t = 0 .. len(sequence) - 1 — the reference's ``match``
    (analysis/behavior_spatial.py:76-107), which builds an (n + 2m) x (n + 2m) band matrix padded
    with -1 and compares it with the padded sequence.  Here: one comparison of the (n, m) matrix of
    sliding windows, O(n m) instead of O((n + 2m)^2) memory.  (Faithful to the padding too: a -1
    IN the sequence matches the band matrix's -1 wherever the template does not cover it.)"""
    sequence, template = np.asarray(sequence), np.asarray(template)
    assert sequence.ndim == 1 and template.ndim == 1 and template.shape[0] >= 1
    n, m = sequence.shape[0], template.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    tail = np.concatenate((sequence.astype(np.float64), np.full(m - 1, np.nan)))   # NaN: past the end
    windows = np.lib.stride_tricks.sliding_window_view(tail, m)                    # [t, j] = sequence[t + j]
    out = (windows == template.astype(np.float64)[None, :]).sum(axis=1).astype(np.int64)
    minus = sequence == -1
    if minus.any():
        csum = np.concatenate(([0], np.cumsum(minus)))
        covered = csum[np.minimum(np.arange(n) + m, n)] - csum[np.arange(n)]
        out += int(minus.sum()) - covered
    return out
