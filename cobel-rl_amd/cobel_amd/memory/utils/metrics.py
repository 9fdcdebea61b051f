"""State-state similarity metrics for SFMA — ``cobel.memory.utils.metrics``
(memory/utils/metrics.py:9-268 of the reference): ``Euclidean``, ``SR`` (successor
representation under the uniform policy) and ``DR`` (default representation with a low-rank
correction for walls).  Same constructors and attributes (``D``, ``update_transitions``).

A metric is a one-off S x S float64 matrix per world, computed on the host with the same LAPACK
inverse the reference calls; the SFMA kernel reads it from device memory (one copy per world,
shared by every instance).  ``sas`` may be the dense one-hot tensor ``world['sas']`` as in the
reference, or the compact successor table ``world['next']`` ([S, 4] integers), which avoids
expanding the tensor.
"""
from __future__ import annotations

import abc

import numpy as np


def _uniform_policy_matrix(sas) -> np.ndarray:
    """``np.sum(sas, axis=1) / sas.shape[1]`` for either form of the transition table."""
    sas = np.asarray(sas)
    if sas.ndim == 3:
        return np.sum(sas, axis=1) / sas.shape[1]
    assert sas.ndim == 2, 'expected sas[S, A, S] or next[S, A]'
    n, acts = sas.shape
    T = np.zeros((n, n))
    rows = np.arange(n)
    for a in range(acts):
        np.add.at(T, (rows, sas[:, a].astype(np.int64)), 1.0)
    return T / acts


class Metric(abc.ABC):
    def __init__(self) -> None:
        self.D: np.ndarray

    @abc.abstractmethod
    def update_transitions(self) -> None:
        """Recompute the metric after the environment changed."""


class Euclidean(Metric):
    """D[s1, s2] = exp(-euclidean distance of the grid cells) (metrics.py:28-61)."""

    def __init__(self, width: int, height: int) -> None:
        super().__init__()
        rows, cols = np.divmod(np.arange(width * height), width)
        d2 = (rows[:, None] - rows[None, :]) ** 2 + (cols[:, None] - cols[None, :]) ** 2
        self.D = np.exp(-np.sqrt(d2))

    def update_transitions(self) -> None:
        pass


class SR(Metric):
    """(I - gamma T)^-1 with T the uniform-policy transition matrix (metrics.py:63-105)."""

    def __init__(self, sas, gamma: float) -> None:
        super().__init__()
        self.sas = sas
        self.gamma = gamma
        self.update_transitions()

    def update_transitions(self) -> None:
        T = _uniform_policy_matrix(self.sas)
        self.D = np.linalg.inv(np.eye(T.shape[0]) - self.gamma * T)


class DR(Metric):
    """Default representation (metrics.py:107-268): the SR of the wall-free grid, ``D0``,
    corrected for the rows of the states that invalid transitions start from by the Woodbury
    identity, ``D = D0 - D0[:, J] (I + delta D0[:, J])^-1 delta D0``."""

    def __init__(self, width: int, height: int, sas, gamma: float, invalid_transitions,
                 T_default=None) -> None:
        super().__init__()
        self.width, self.height = width, height
        self.nb_states = width * height
        self.sas = sas
        self.gamma = gamma
        self.invalid_transitions = invalid_transitions
        if T_default is None:
            self.build_default_transition_matrix()
        else:
            self.T_default = T_default
        self.D0 = np.linalg.inv(np.eye(self.nb_states) - self.gamma * self.T_default)
        self.update_transitions()

    def update_transitions(self) -> None:
        self.T_new = _uniform_policy_matrix(self.sas)
        self.B = np.zeros(self.T_new.shape)
        if len(self.invalid_transitions) > 0:
            self.states = np.unique(np.array(self.invalid_transitions)[:, 0])
            eye = np.eye(self.nb_states)
            delta = (eye - self.gamma * self.T_new)[self.states] \
                - (eye - self.gamma * self.T_default)[self.states]
            cols = self.D0[:, self.states]
            alpha = np.linalg.inv(np.eye(self.states.shape[0]) + np.matmul(delta, cols))
            self.B = np.matmul(np.matmul(cols, alpha), np.matmul(delta, self.D0))
        self.D = self.D0 - self.B

    def build_default_transition_matrix(self) -> None:
        """Uniform-policy transitions of the open field (moves clamp at the border)."""
        n, w, h = self.nb_states, self.width, self.height
        rows, cols = np.divmod(np.arange(n), w)
        self.T_default = np.zeros((n, n))
        for r, c in ((rows, np.maximum(0, cols - 1)), (np.maximum(0, rows - 1), cols),
                     (rows, np.minimum(w - 1, cols + 1)), (np.minimum(h - 1, rows + 1), cols)):
            np.add.at(self.T_default, (np.arange(n), r * w + c), 0.25)
