"""Abstract optimizer (optimizer/optimizer.py:13-47 of the reference)."""
from __future__ import annotations

import abc
from collections.abc import Callable
from typing import Any

Fit = dict
Simulation = Callable[[dict, dict], Any]
FitLoss = Callable[[dict, dict], Any]


class Optimizer(abc.ABC):
    @abc.abstractmethod
    def __init__(self) -> None:
        pass

    @abc.abstractmethod
    def fit(self, simulation: Simulation, tasks: dict, data: dict, loss: FitLoss) -> Fit:
        """Fit a model to behavioural data; returns {parameter combination: fitness}."""
