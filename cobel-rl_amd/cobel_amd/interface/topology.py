"""Vectorised topology-graph environment on the HIP library.

API of the reference's ``cobel.interface.Topology`` (interface/topology.py:28-193):
``Topology(nodes, starting_nodes=None, simulator=None, widget=None, rng=None)`` with ``step`` /
``reset`` / ``get_observation`` / ``get_position`` and the attributes ``nodes``,
``starting_nodes``, ``current_node``, ``observation_space`` (Box of the 6-float pose),
``action_space``.  Nodes are compiled once into the same compact tables a gridworld uses
(neighbour table = ``next[S, 4]``), so ``step`` / ``reset`` are ``cobel_env_step`` /
``cobel_env_reset`` and observations are pose rows gathered on device
(``cobel_gather_rows``).  Reference quirks kept: the constructor draws one start node
(topology.py:109); ``step`` returns ``truncated == end_trial`` (topology.py:157).
Simulators (Godot / Unity / offline observation dictionaries) are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, Discrete
from .gridworld import WorldHandle, _as_seed
from .interface import Interface


class Topology(Interface):
    def __init__(self, nodes: dict, starting_nodes=None, simulator=None, widget=None, rng=None,
                 n_envs: int = 1, seed: int | None = None, device=None,
                 instance_base: int = 0) -> None:
        super().__init__(widget)
        assert simulator is None, 'simulator-backed observations are outside the accelerated path'
        self.nodes = nodes
        self.ids = list(nodes.keys())
        index = {k: i for i, k in enumerate(self.ids)}
        if starting_nodes is None:
            starting_nodes = [k for k, nd in nodes.items() if not nd['terminal']]
        self.starting_nodes = starting_nodes
        # the action space is the neighbour count of the start node (topology.py:110-112): four
        # on track / grid / maze graphs, six on hexagonal ones
        n_act = len(nodes[starting_nodes[0]]['neighbors'])
        assert 1 <= n_act <= _lib.MAX_ACTIONS and \
            all(len(nd['neighbors']) == n_act for nd in nodes.values()), \
            'every node needs the same number of neighbours (at most %d)' % _lib.MAX_ACTIONS
        S = len(self.ids)
        self.pose = np.array([nodes[k]['pose'] for k in self.ids], dtype=np.float64).reshape(S, 6)
        world = dict(
            states=S, next=np.array([[index[m] for m in nodes[k]['neighbors']] for k in self.ids],
                                    dtype=np.uint16),
            rewards=np.array([nodes[k]['reward'] for k in self.ids], dtype=np.float64),
            terminals=np.array([bool(nodes[k]['terminal']) for k in self.ids]),
            starting_states=np.array([index[k] for k in starting_nodes]), deterministic=True)
        self.world = world
        self.n_envs = int(n_envs)
        self.rng = rng
        self.seed = _as_seed(rng) if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self.instance_base = int(instance_base)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.handle = WorldHandle([world], self.device)
        self.simulator = None
        self.observation_space = Box(low=np.array([-np.inf, -np.inf, -np.inf, 0.0, 0.0, 0.0]),
                                     high=np.array([np.inf, np.inf, np.inf, 360.0, 360.0, 360.0]),
                                     dtype=np.float64)
        self.action_space = Discrete(n_act)
        self.state = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self.env_ctr = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self._reward = torch.zeros(self.n_envs, dtype=torch.float32, device=self.device)
        self._done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self._pose_dev = torch.as_tensor(self.pose, device=self.device).contiguous()
        self._obs = torch.zeros((self.n_envs, 6), dtype=torch.float64, device=self.device)
        self._draw()   # the constructor's start-node draw
        self.observation = None

    def _stream(self):
        return _lib.current_stream(self.device)

    def _draw(self, mask=None) -> None:
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        _lib.check(_lib.lib().cobel_env_reset(
            self.handle.ptr, _lib.ptr(self.state), _lib.ptr(m), _lib.ptr(self.env_ctr), self.seed,
            self.n_envs, self.instance_base, self._stream()))

    @property
    def current_node(self):
        if self.n_envs == 1:
            return self.ids[int(self.state[0].item())]
        return self.state

    def observe(self):
        """Pose rows of the current nodes, ``[N, 6]`` float64 on device."""
        _lib.check(_lib.lib().cobel_gather_rows(
            _lib.ptr(self._pose_dev), _lib.ptr(self.state), _lib.ptr(self._obs), self.n_envs, 6,
            len(self.ids), self._stream()))
        return self._obs

    def get_observation(self, pose=None):
        if pose is not None:
            self.observation = np.array(pose)
            return np.array(pose)
        obs = self.observe()
        self.observation = obs[0].cpu().numpy() if self.n_envs == 1 else obs
        return self.observation.copy() if self.n_envs == 1 else obs

    def step(self, action):
        if self.n_envs == 1 and not torch.is_tensor(action):
            a = int(action)
            assert 0 <= a < int(self.action_space.n), 'Invalid action type!'
            act = torch.full((1,), a, dtype=torch.uint8, device=self.device)
        else:
            act = torch.as_tensor(action, device=self.device).to(torch.uint8).contiguous()
        _lib.check(_lib.lib().cobel_env_step(
            self.handle.ptr, _lib.ptr(self.state), _lib.ptr(act), _lib.ptr(self._reward),
            _lib.ptr(self._done), self.n_envs, self.instance_base, self._stream()))
        obs = self.get_observation()
        if self.n_envs == 1:
            end = bool(self._done[0].item())
            return obs, float(self._reward[0].item()), end, end, {}
        done = self._done.bool()
        return obs, self._reward, done, done, {}

    def reset(self, mask=None):
        self._draw(mask)
        return self.get_observation(), {}

    def get_position(self):
        obs = self.observe()   # the reference returns the whole pose (topology.py:203)
        return obs[0].cpu().numpy() if self.n_envs == 1 else obs.cpu().numpy()

    def init_visualization(self) -> None:
        pass

    def update_visualization(self, logs=None) -> None:
        pass
