"""Vectorised gridworld environment on the HIP library.

API of the reference's ``cobel.interface.Gridworld`` (interface/gridworld.py:33-156):
``Gridworld(world, widget=None, rng=None)`` with ``step`` / ``reset`` / ``get_position`` and the
attributes ``world``, ``observation_space``, ``action_space``, ``current_state``,
``current_coordinates``, ``rng``.  Additions: ``n_envs`` independent instances advance in
lockstep on one GPU, ``world`` may be a list of same-sized worlds (instance g uses world
g % len(worlds)), and randomness comes from the counter-based streams of the library
(``seed``; a ``numpy.random.Generator`` passed as ``rng`` only donates a seed).

The transition itself is ``cobel_env_step`` / ``cobel_env_reset`` (include/cobel_hip.h); agents
from ``cobel_amd.agent`` bypass this object during ``train`` and run the fused kernels on the
same world handle and the same per-instance state.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from ..spaces import Discrete
from .interface import Interface


def _as_seed(rng) -> int:
    if rng is None:
        return int.from_bytes(os.urandom(8), 'little')
    if isinstance(rng, (int, np.integer)):
        return int(rng) & 0xFFFFFFFFFFFFFFFF
    if hasattr(rng, 'integers'):  # numpy Generator: donate 64 bits
        return int(rng.integers(0, 2**63 - 1)) & 0xFFFFFFFFFFFFFFFF
    raise AssertionError('rng must be None, an int seed or a numpy Generator')


class WorldHandle:
    """Owns one ``cobel_world_t`` (device copies of the compact tables of 1..W worlds)."""

    def __init__(self, worlds: list, device: torch.device) -> None:
        S = int(worlds[0]['states'])
        assert all(int(w['states']) == S for w in worlds), 'worlds must have equal state counts'
        tabs = [w.compact() if hasattr(w, 'compact') else _compact(w) for w in worlds]
        nxt = np.ascontiguousarray(np.stack([t['next'] for t in tabs]), dtype=np.uint16)
        rew = np.ascontiguousarray(np.stack([t['reward'] for t in tabs]), dtype=np.float32)
        term = np.ascontiguousarray(np.stack([t['terminal'] for t in tabs]), dtype=np.uint8)
        starts = np.ascontiguousarray(np.concatenate([t['starts'] for t in tabs]), dtype=np.uint16)
        off = np.zeros(len(tabs) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(t['starts']) for t in tabs])
        assert nxt.ndim == 3 and nxt.shape[:2] == (len(tabs), S)
        self.n_states, self.n_worlds, self.device = S, len(tabs), device
        self.n_actions = int(nxt.shape[2])      # 4: gridworlds and 4-neighbour graphs
        self.ptr = C.c_void_p()
        _lib.check(_lib.lib().cobel_world_create_n(
            nxt.ctypes.data, rew.ctypes.data, term.ctypes.data, starts.ctypes.data,
            off.ctypes.data, S, len(tabs), self.n_actions, device.index or 0, C.byref(self.ptr)))
        # worlds whose transition rows are distributions: list form for all of them (a table row
        # is a list of one successor with cumulative probability 1)
        self.stochastic = any('transitions' in t for t in tabs)
        if self.stochastic:
            offs, sts, cdfs, base = [np.zeros(1, dtype=np.uint32)], [], [], 0
            for t in tabs:
                if 'transitions' in t:
                    o, st, cd = t['transitions']
                else:
                    flat = np.asarray(t['next'], dtype=np.uint16).reshape(-1)
                    o = np.arange(len(flat) + 1, dtype=np.uint32)
                    st, cd = flat, np.ones(len(flat))
                offs.append(o[1:].astype(np.uint64) + base)
                sts.append(st)
                cdfs.append(cd)
                base += len(st)
            o = np.ascontiguousarray(np.concatenate(offs), dtype=np.uint32)
            st = np.ascontiguousarray(np.concatenate(sts), dtype=np.uint16)
            cd = np.ascontiguousarray(np.concatenate(cdfs), dtype=np.float64)
            _lib.check(_lib.lib().cobel_world_set_transitions(
                self.ptr, o.ctypes.data, st.ctypes.data, cd.ctypes.data, len(st)))

    def __del__(self) -> None:
        try:
            if self.ptr:
                _lib.lib().cobel_world_destroy(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def _compact(world: dict) -> dict:
    """Compact tables of a reference-style WorldDict that only carries the dense ``sas``."""
    from ..misc.gridworld_tools import is_one_hot, successor_table, transition_lists
    out = {}
    if 'sas' in world and not world.get('deterministic', True) and not is_one_hot(world['sas']):
        out['transitions'] = transition_lists(world['sas'])
    nxt = successor_table(world['sas']) if 'sas' in world else world['next']
    return dict(out, next=np.asarray(nxt, dtype=np.uint16),
                reward=np.asarray(world['rewards'], dtype=np.float32),
                terminal=(np.asarray(world['terminals']) != 0).astype(np.uint8),
                starts=np.asarray(world['starting_states'], dtype=np.uint16))


class Gridworld(Interface):
    def __init__(self, world, widget=None, rng=None, n_envs: int = 1, seed: int | None = None,
                 device=None, instance_base: int = 0) -> None:
        super().__init__(widget)
        worlds = list(world) if isinstance(world, (list, tuple)) else [world]
        # (world['deterministic'] = False makes the reference DRAW the successor from the row of
        #  sas instead of taking its argmax, gridworld.py:115-123.  With one-hot rows — all any
        #  builder produces — that is the same step; rows that are distributions travel to the
        #  library as successor lists, WorldHandle below, and the draw happens on the device)
        self.worlds = worlds
        self.world = worlds[0]
        self.n_envs = int(n_envs)
        assert self.n_envs >= 1
        self.rng = rng
        self.seed = _as_seed(rng) if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self.instance_base = int(instance_base)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        assert self.device.type == 'cuda', 'cobel_amd runs on a GPU; there is no CPU fallback'
        self.handle = WorldHandle(worlds, self.device)
        self.observation_space = Discrete(self.handle.n_states)
        self.action_space = Discrete(4)
        self.state = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self.env_ctr = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self._reward = torch.zeros(self.n_envs, dtype=torch.float32, device=self.device)
        self._done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self._coords = [np.asarray(w['coordinates'], dtype=float) for w in worlds]
        self.reset()  # the reference's constructor draws a start state too (gridworld.py:89)

    # -- reference surface ------------------------------------------------------------------
    @property
    def current_state(self):
        return int(self.state[0].item()) if self.n_envs == 1 else self.state

    @current_state.setter
    def current_state(self, value) -> None:
        v = torch.as_tensor(value, dtype=torch.int32)
        if v.numel() and (int(v.min()) < 0 or int(v.max()) >= self.handle.n_states):
            raise IndexError('state outside the world (%d states)' % self.handle.n_states)
        if self.n_envs == 1 and not torch.is_tensor(value):
            self.state.fill_(int(value))
        else:
            self.state.copy_(v.to(self.device))

    @property
    def current_coordinates(self):
        return self.get_position()

    def _stream(self):
        return _lib.current_stream(self.device)

    def step(self, action):
        """``(observation, reward, end_trial, truncated, logs)``; see gridworld.py:92-129."""
        if self.n_envs == 1 and not torch.is_tensor(action):
            a = int(action)
            assert 0 <= a < 4, 'invalid action'
            act = torch.full((1,), a, dtype=torch.uint8, device=self.device)
        else:
            act = torch.as_tensor(action, device=self.device).to(torch.uint8).contiguous()
            assert act.shape == (self.n_envs,)
        # (worlds whose rows are distributions draw the successor: one double of the env stream at
        #  the counter the trial starts share; a world of tables steps as cobel_env_step does)
        _lib.check(_lib.lib().cobel_env_step_draw(
            self.handle.ptr, _lib.ptr(self.state), _lib.ptr(act), _lib.ptr(self._reward),
            _lib.ptr(self._done), _lib.ptr(self.env_ctr), self.seed, self.n_envs,
            self.instance_base, self._stream()))
        if self.n_envs == 1:
            out = torch.stack([self.state.to(torch.float64), self._reward.to(torch.float64),
                               self._done.to(torch.float64)]).cpu().numpy()[:, 0]
            return int(out[0]), np.float64(out[1]), bool(out[2]), False, {}
        return self.state, self._reward, self._done.bool(), False, {}

    def reset(self, mask=None):
        """``(observation, logs)``; uniform draw over ``starting_states`` (gridworld.py:131-145)."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        _lib.check(_lib.lib().cobel_env_reset(
            self.handle.ptr, _lib.ptr(self.state), _lib.ptr(m), _lib.ptr(self.env_ctr), self.seed,
            self.n_envs, self.instance_base, self._stream()))
        return self.current_state, {}

    def get_position(self):
        states = self.state.cpu().numpy()
        if self.n_envs == 1:
            return np.copy(self._coords[self.instance_base % len(self._coords)][states[0]])
        w = (self.instance_base + np.arange(self.n_envs)) % len(self._coords)
        return np.stack([self._coords[wi][s] for wi, s in zip(w, states)])

    # visualisation hooks of the reference are accepted and ignored (widget is always None here)
    def init_visualization(self) -> None:
        pass

    def update_visualization(self, logs=None) -> None:
        pass
