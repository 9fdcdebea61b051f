"""Abstract environment interface, same contract as the reference's
``cobel.interface.interface.Interface`` (interface/interface.py:17-86): ``step(action) ->
(observation, reward, end_trial, truncated, logs)``, ``reset() -> (observation, logs)``,
``get_position()``.  Vectorised implementations return per-instance tensors when ``n_envs > 1``
and the reference's scalar tuple when ``n_envs == 1``.
"""
from __future__ import annotations

import abc


class Interface(abc.ABC):
    def __init__(self, widget=None) -> None:
        self.widget = widget  # kept for signature compatibility; visualisation is out of scope
        self.n_envs = 1

    @abc.abstractmethod
    def step(self, action):
        ...

    @abc.abstractmethod
    def reset(self):
        ...

    @abc.abstractmethod
    def get_position(self):
        ...
