"""Pre-rendered observations — ``cobel.interface.OfflineSimulator``
(interface/simulator/offline.py:16-100): a dictionary pose -> observation and the observation
space that goes with it.  A ``Topology`` built with one turns the dictionary into one table per
observation component (node index -> flattened array) on the device and returns gathered rows, so
stepping stays ``cobel_env_step`` + ``cobel_gather_rows``.  The socket / rendering methods of the
reference's ``Simulator`` base class are the same no-ops as in the reference.
"""
from __future__ import annotations

import numpy as np


class OfflineSimulator:
    def __init__(self, observations: dict, observation_space) -> None:
        self.agent_pose = list(observations)[0]
        self.observations = observations
        self.observation_space = observation_space

    def get_observation(self, pose):
        """The observation at a given pose (offline.py:56-71)."""
        return self.observations[pose]

    def connect_socket(self, connection_socket, port: int) -> None:
        pass

    def receive(self, connection_socket, data_size: int) -> bytes:
        return b'dummy'

    def receive_in_chunks(self, socket, chunk_size: int) -> bytes:
        return b'dummy'

    def move_agent(self, x: float, y: float, yaw: float):
        return np.array(self.agent_pose), np.ones(1)

    def move_object(self, object_id: str, pose) -> None:
        pass

    def set_illumination(self, light_source: str, color) -> None:
        pass

    def stop(self) -> None:
        pass
