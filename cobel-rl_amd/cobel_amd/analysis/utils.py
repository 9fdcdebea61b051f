"""Gridworld state index <-> grid coordinates (the call surface of ``cobel.analysis.utils``,
analysis/utils.py:8-53): state s of a gridworld of width w sits in row s // w, column s % w."""
from __future__ import annotations

import numpy as np


def state_to_coordinates(state: int, width: int, y_first: bool = True) -> np.ndarray:
    """Coordinates of one state: ``[row, column]`` (``y_first``) or ``[column, row]``."""
    assert state >= 0 and width > 0
    row, col = int(state) // int(width), int(state) % int(width)
    return np.asarray((row, col) if y_first else (col, row))


def states_to_coordinates(states: np.ndarray, width: int, y_first: bool = True) -> np.ndarray:
    """Coordinates of a vector of states as an (n, 2) integer array, one ``[row, column]`` (or
    ``[column, row]``) pair per state — what a trajectory log of the tabular kernels (``agent.logs``,
    ``TrajectoryMonitor``) is turned into before ``get_occupancy_map``."""
    flat = np.asarray(states).reshape(-1)
    assert flat.size > 0 and int(flat.min()) >= 0 and width > 0
    rows, cols = flat // width, flat % width
    return np.stack((rows, cols) if y_first else (cols, rows), axis=1)
