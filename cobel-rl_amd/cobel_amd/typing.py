"""Type aliases under the reference's names (``cobel.typing``), so that annotated user scripts
import unchanged.  Only the names the accelerated path's demos and tests use."""
from __future__ import annotations

from typing import Any, Callable

import numpy as np

NodeID = str
Pose = tuple
Node = dict            # {'pose', 'reward', 'terminal', 'neighbors'} (interface/topology.py:21-26)
Observation = Any      # int (gridworld state), numpy array (pose) or a dict of arrays
Action = Any
Logs = dict
Callback = Callable[[dict], Any]
CallbackDict = dict    # {'on_trial_begin' | 'on_step_begin' | 'on_step_end' | 'on_trial_end' | ...: [callables]}
NDArray = np.ndarray
