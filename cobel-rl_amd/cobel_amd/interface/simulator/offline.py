"""Pre-rendered observations — ``cobel.interface.OfflineSimulator``
(interface/simulator/offline.py:16-100): a dictionary pose -> observation and the observation
space that goes with it.  A ``Topology`` built with one turns the dictionary into one table per
observation component (node index -> flattened array) on the device and returns gathered rows, so
stepping stays ``cobel_env_step`` + ``cobel_gather_rows``.  The reference's class also carries the
socket / rendering no-ops its abstract ``Simulator`` base demands; nothing on this path calls them,
so only the two things a ``Topology`` uses are here: the table and the space.
"""
from __future__ import annotations


class OfflineSimulator:
    def __init__(self, observations: dict, observation_space) -> None:
        self.observations = observations
        self.observation_space = observation_space
        self.agent_pose = next(iter(observations))     # (offline.py:51: the first pose)

    def get_observation(self, pose):
        """The pre-rendered observation of a pose; an unknown pose is a KeyError, as a dictionary
        lookup is in the reference (offline.py:56-71)."""
        return self.observations[pose]
