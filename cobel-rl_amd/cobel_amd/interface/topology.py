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
An ``OfflineSimulator`` (interface/simulator/offline.py:51-72: pre-rendered observations per pose,
plain arrays, lists or dictionaries of arrays — unit_tests/test_topology.py, test_q.py
"Topology-Dict") becomes one device table per observation component, gathered by node index like
the poses; the rendering simulators (Godot / Unity) are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, Dict, Discrete, Tuple  # noqa: F401
from .gridworld import WorldHandle, _as_seed
from .interface import Interface


def _torch_dtype(dt):
    """torch counterpart of a NumPy dtype (float64 where torch has none)."""
    try:
        return torch.from_numpy(np.zeros(1, dtype=dt)).dtype
    except TypeError:
        return torch.float64


class Topology(Interface):
    def __init__(self, nodes: dict, starting_nodes=None, simulator=None, widget=None, rng=None,
                 n_envs: int = 1, seed: int | None = None, device=None,
                 instance_base: int = 0) -> None:
        super().__init__(widget)
        assert simulator is None or hasattr(simulator, 'observations'), \
            'only pre-rendered observations (OfflineSimulator) are on the accelerated path'
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
        self.simulator = simulator
        if simulator is None:
            self.observation_space = Box(low=np.array([-np.inf, -np.inf, -np.inf, 0.0, 0.0, 0.0]),
                                         high=np.array([np.inf, np.inf, np.inf, 360.0, 360.0, 360.0]),
                                         dtype=np.float64)
        else:
            self.observation_space = simulator.observation_space
        self.action_space = Discrete(n_act)
        self.state = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self.env_ctr = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self._reward = torch.zeros(self.n_envs, dtype=torch.float32, device=self.device)
        self._done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self._pose_dev = torch.as_tensor(self.pose, device=self.device).contiguous()
        self._obs = torch.zeros((self.n_envs, 6), dtype=torch.float64, device=self.device)
        # pre-rendered observations: per component a table [S, D] (rows = nodes) on the device
        self._sim_kind, self._sim_keys, self._sim_tabs, self._sim_shapes, self._sim_out = \
            None, None, None, None, None
        if simulator is not None:
            self._compile_observations(simulator.observations)
        self._draw()   # the constructor's start-node draw
        self.observation = None

    def _compile_observations(self, observations: dict) -> None:
        """interface/simulator/offline.py:51-72 + topology.py:174-193: ``observations[pose]`` is an
        array, a list / tuple of arrays or a dictionary of arrays; every node's pose must be a key."""
        first = observations[tuple(self.nodes[self.ids[0]]['pose'])]
        if isinstance(first, dict):
            self._sim_kind, self._sim_keys = 'dict', list(first.keys())
            parts = lambda o: [o[k] for k in self._sim_keys]      # noqa: E731
        elif isinstance(first, (list, tuple)):
            self._sim_kind, self._sim_keys = 'list', list(range(len(first)))
            parts = lambda o: list(o)                              # noqa: E731
        else:
            self._sim_kind, self._sim_keys = 'array', [0]
            parts = lambda o: [o]                                  # noqa: E731
        rows = [[] for _ in self._sim_keys]
        for k in self.ids:
            pose = tuple(self.nodes[k]['pose'])
            if pose not in observations:
                raise KeyError(pose)        # what the reference raises at the first visit
            for c, part in enumerate(parts(observations[pose])):
                rows[c].append(np.asarray(part, dtype=np.float64))
        self._sim_shapes = [r[0].shape for r in rows]
        # (the device tables are float64 — every integer and float32 value is exact in it; the
        #  observations go back out in the dtype they were stored with, as the reference returns
        #  the stored objects themselves: topology.py:174-193)
        self._sim_dtypes = [np.asarray(parts(first)[c]).dtype for c in range(len(rows))]
        self._sim_tabs = [torch.as_tensor(np.stack([x.reshape(-1) for x in r]), device=self.device)
                          .contiguous() for r in rows]
        self._sim_out = [torch.zeros((self.n_envs, t.shape[1]), dtype=torch.float64,
                                     device=self.device) for t in self._sim_tabs]

    def observation_key_table(self) -> np.ndarray:
        """``[S, D]``: per node the tuple an agent keys its table by (agent/q.py:150-158: Box
        observations flattened, dictionary observations concatenated in key order)."""
        if self.simulator is None:
            return self.pose
        return np.concatenate([t.cpu().numpy() for t in self._sim_tabs], axis=1)

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

    def _observe_sim(self):
        """Gathered rows of every observation component, in the container the simulator uses."""
        for tab, out in zip(self._sim_tabs, self._sim_out):
            _lib.check(_lib.lib().cobel_gather_rows(
                _lib.ptr(tab), _lib.ptr(self.state), _lib.ptr(out), self.n_envs, tab.shape[1],
                len(self.ids), self._stream()))
        if self.n_envs == 1:
            parts = [o[0].cpu().numpy().reshape(sh).astype(dt, copy=False)
                     for o, sh, dt in zip(self._sim_out, self._sim_shapes, self._sim_dtypes)]
        else:
            # (fresh tensors, not views of the gather buffers: the next step overwrites those)
            parts = [o.reshape((self.n_envs,) + tuple(sh)).to(_torch_dtype(dt), copy=True)
                     for o, sh, dt in zip(self._sim_out, self._sim_shapes, self._sim_dtypes)]
        if self._sim_kind == 'dict':
            return dict(zip(self._sim_keys, parts))
        return parts if self._sim_kind == 'list' else parts[0]

    def get_observation(self, pose=None):
        if self.simulator is not None:
            if pose is not None:     # topology.py:190-191
                import copy
                self.observation = self.simulator.get_observation(tuple(pose))
                return copy.deepcopy(self.observation)
            self.observation = self._observe_sim()
            return self.observation
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
