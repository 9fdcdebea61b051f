"""cobel_amd — MI355X-native hot path of the CoBeL-RL gridworld / tabular-agent loop.

The package mirrors the reference's module layout for the accelerated path only:

    cobel_amd.misc.gridworld_tools   make_gridworld, make_open_field, ...
    cobel_amd.interface              Gridworld, Topology (vectorised: n_envs instances)
    cobel_amd.policy                 EpsilonGreedy
    cobel_amd.memory                 DynaQMemory, DQNMemory, SFMAMemory (+ memory.utils metrics)
    cobel_amd.agent                  DynaQ, QAgent, SR, SFMA, DQN, DynaDQN, DynaDSR
    cobel_amd.monitor                EscapeLatencyMonitor, RewardMonitor
    cobel_amd.analysis               get_occupancy_map

All compute goes through ``libcobel_hip.so`` (see ``include/cobel_hip.h``); there is no CPU
fallback.  ``install_as_cobel()`` registers the package under the name ``cobel`` so that the
reference's gridworld demos and unit tests (``from cobel.agent import DynaQ`` ...) import it
unchanged.
"""
from __future__ import annotations

import sys

__version__ = '0.1.0'


def install_as_cobel() -> None:
    """Alias this package as ``cobel`` (drop-in for demo/gridworld, demo/topology and unit_tests
    scripts): every module of the package is registered under the corresponding ``cobel.`` name."""
    import importlib
    import pkgutil

    me = sys.modules[__name__]
    sys.modules.setdefault('cobel', me)
    for info in pkgutil.walk_packages(me.__path__, prefix=__name__ + '.'):
        if info.name.rsplit('.', 1)[-1].startswith('_'):
            continue
        mod = importlib.import_module(info.name)
        sys.modules.setdefault('cobel' + info.name[len(__name__):], mod)
