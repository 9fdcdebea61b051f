"""Gridworld state index <-> grid coordinates — ``cobel.analysis.utils`` (analysis/utils.py:8-53):
state s of a gridworld of width w sits in row s // w, column s % w."""
from __future__ import annotations

import numpy as np


def state_to_coordinates(state: int, width: int, y_first: bool = True) -> np.ndarray:
    """Coordinates of one state: ``[y, x]`` (row first) or ``[x, y]``."""
    assert state >= 0 and width > 0
    y, x = divmod(state, width)
    return np.array([y, x]) if y_first else np.array([x, y])


def states_to_coordinates(states: np.ndarray, width: int, y_first: bool = True) -> np.ndarray:
    """Coordinates of a vector of states as an (n, 2) array, ``[y, x]`` rows or ``[x, y]`` rows —
    what a trajectory log of the tabular kernels (``agent.logs`` / ``TrajectoryMonitor``) is turned
    into before ``get_occupancy_map``."""
    states = np.asarray(states)
    assert np.amin(states) >= 0 and width > 0
    y, x = np.divmod(states.reshape((states.shape[0], 1)), width)
    return np.hstack((y, x)) if y_first else np.hstack((x, y))
