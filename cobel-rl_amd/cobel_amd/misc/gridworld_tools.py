"""Gridworld builders producing the compact world tables the HIP kernels consume.

Mirrors the call surface of the reference's ``cobel.misc.gridworld_tools``
(``/root/reference/src/cobel/misc/gridworld_tools.py:10-234``): same function names, arguments
and ``WorldDict`` keys.  The one structural change is the transition tensor: the reference
materialises a dense one-hot ``sas[S, 4, S]`` float64 array (32 MiB at 32x32); here the primary
form is ``world['next'][S, 4]`` (uint16) and ``world['sas']`` is expanded from it only if somebody
asks for the key.

Semantics restated from gridworld_tools.py:103-133: state = h * width + w; actions 0 left,
1 up (h - 1), 2 right, 3 down (h + 1), clamped at the border; wind is added after the move and
clamped again (walls are ignored by wind); a move into an invalid state or along an invalid
transition leaves the agent where it was; by default every non-terminal state — invalid ones
included — is a starting state.
"""
from __future__ import annotations

from typing import Literal

import numpy as np


class World(dict):
    """``WorldDict`` (interface/gridworld.py:17-30) plus ``'next'``; ``'sas'`` is lazy."""

    def __missing__(self, key):
        if key == 'sas':
            n = int(self['states'])
            sas = np.zeros((n, 4, n))
            sas[np.arange(n)[:, None], np.arange(4)[None, :], self['next'].astype(np.int64)] = 1.0
            self['sas'] = sas
            return sas
        raise KeyError(key)

    def compact(self) -> dict:
        """Tables in the layout ``cobel_world_create`` takes."""
        return dict(
            next=np.ascontiguousarray(self['next'], dtype=np.uint16),
            reward=np.ascontiguousarray(self['rewards'], dtype=np.float32),
            terminal=np.ascontiguousarray(self['terminals'] != 0, dtype=np.uint8),
            starts=np.ascontiguousarray(self['starting_states'], dtype=np.uint16),
        )


def make_gridworld(
    height: int,
    width: int,
    terminals: None | list[int] = None,
    rewards: None | np.ndarray = None,
    goals: None | list[int] = None,
    starting_states: None | list[int] = None,
    invalid_states: None | list[int] = None,
    invalid_transitions: None | list[tuple[int, int]] = None,
    wind: None | np.ndarray = None,
    deterministic: bool = True,
) -> World:
    """Build a gridworld; arguments as in the reference's ``make_gridworld``."""
    n = int(height) * int(width)
    assert 0 < n <= 16384, 'the compact tables index states with 14 bits'
    world = World()
    world['height'], world['width'], world['states'] = height, width, n
    world['goals'] = [] if goals is None else goals
    world['terminals'] = np.zeros(n, dtype=int)
    if terminals is not None:
        world['terminals'][terminals] = 1
    world['rewards'] = np.zeros(n, dtype=float)
    if rewards is not None:
        rewards = np.asarray(rewards)
        world['rewards'][rewards[:, 0].astype(int)] = rewards[:, 1]
    if starting_states is not None and len(starting_states) > 0:
        world['starting_states'] = np.array(starting_states)
    else:
        world['starting_states'] = np.flatnonzero(world['terminals'] == 0)
    world['wind'] = np.zeros((n, 2), dtype=int)
    if wind is not None:
        wind = np.asarray(wind)
        world['wind'][wind[:, 0].astype(int)] = wind[:, 1:].astype(int)
    world['invalid_states'] = [] if invalid_states is None else invalid_states
    world['invalid_transitions'] = [] if invalid_transitions is None else invalid_transitions

    s = np.arange(n)
    h, w = s // width, s % width
    world['coordinates'] = np.stack([w, height - 1 - h], axis=1).astype(float)
    dh = np.array([0, -1, 0, 1])
    dw = np.array([-1, 0, 1, 0])
    nh = np.clip(h[:, None] + dh[None, :], 0, height - 1)
    nw = np.clip(w[:, None] + dw[None, :], 0, width - 1)
    nh = np.clip(nh + world['wind'][:, 0:1], 0, height - 1)
    nw = np.clip(nw + world['wind'][:, 1:2], 0, width - 1)
    nxt = nh * width + nw
    blocked = np.zeros(n, dtype=bool)
    if len(world['invalid_states']):
        blocked[np.asarray(world['invalid_states'], dtype=int)] = True
    stay = blocked[nxt]
    if len(world['invalid_transitions']):
        bad = {(int(a), int(b)) for a, b in world['invalid_transitions']}
        src = np.repeat(s, 4)
        hit = np.fromiter(((int(a), int(b)) in bad for a, b in zip(src, nxt.ravel())),
                          dtype=bool, count=4 * n)
        stay |= hit.reshape(n, 4)
    world['next'] = np.where(stay, s[:, None], nxt).astype(np.uint16)
    world['deterministic'] = deterministic
    assert deterministic, ('only deterministic worlds are supported: no builder of the reference '
                           'produces a non-one-hot sas (SURVEY.md §8a quirk 12)')
    return world


def make_open_field(height: int, width: int, goal_state: int = 0, reward: float = 1) -> World:
    """Open field with one terminal goal state (gridworld_tools.py:139-167)."""
    return make_gridworld(height, width, terminals=[goal_state],
                          rewards=np.array([[goal_state, reward]]), goals=[goal_state])


def make_empty_field(height: int, width: int) -> World:
    """Empty open field (gridworld_tools.py:170-186)."""
    return make_gridworld(height, width)


def make_windy_gridworld(height: int, width: int, columns: np.ndarray, goal_state: int = 0,
                         reward: float = 1, direction: Literal['up', 'down'] = 'up') -> World:
    """Windy gridworld (gridworld_tools.py:189-234): column-wise vertical wind."""
    sign = {'up': 1, 'down': -1}[direction]
    s = np.arange(height * width)
    wind = np.stack([s, np.asarray(columns)[s % width] * sign, np.zeros_like(s)], axis=1)
    return make_gridworld(height, width, terminals=[goal_state],
                          rewards=np.array([[goal_state, reward]]), goals=[goal_state], wind=wind)


def make_obstacle_maze(height: int, width: int, seed: int, density: float = 0.20,
                       goal_state: int = 0, reward: float = 1.0) -> World:
    """Random obstacle maze of the benchmark.

    Each non-goal cell is a wall with probability ``density`` (``default_rng(seed)``); the layout
    is redrawn until the goal is reachable from at least half of the free cells.  Starting
    states are the free non-terminal cells.
    """
    rng = np.random.default_rng(seed)
    n = height * width
    while True:
        wall = rng.random(n) < density
        wall[goal_state] = False
        seen = np.zeros(n, dtype=bool)
        seen[goal_state] = True
        frontier = [goal_state]
        while frontier:
            c = frontier.pop()
            y, x = divmod(c, width)
            for ny, nx in ((y, x - 1), (y - 1, x), (y, x + 1), (y + 1, x)):
                if 0 <= ny < height and 0 <= nx < width:
                    t = ny * width + nx
                    if not wall[t] and not seen[t]:
                        seen[t] = True
                        frontier.append(t)
        if seen.sum() >= 0.5 * (~wall).sum():
            break
    walls = [int(i) for i in np.flatnonzero(wall)]
    starts = [int(i) for i in np.flatnonzero(~wall) if i != goal_state]
    return make_gridworld(height, width, terminals=[goal_state],
                          rewards=np.array([[goal_state, reward]]), goals=[goal_state],
                          invalid_states=walls, starting_states=starts)


def corridor_gridworld(height: int, width: int, corridor, terminals, rewards, goals,
                       starting_states) -> World:
    """Gridworld whose walls are given by a boolean ``corridor[h, w]`` mask: every move between a
    corridor cell and a non-corridor cell is an invalid transition (both directions).  Cells
    outside the corridor remain ordinary, unreachable states — the structure the reference's
    maze templates build by listing border-crossing transitions explicitly."""
    inside = np.asarray(corridor, dtype=bool).reshape(height, width)
    s = np.arange(height * width).reshape(height, width)
    pairs = []
    for a, b, ia, ib in ((s[:, :-1], s[:, 1:], inside[:, :-1], inside[:, 1:]),
                         (s[:-1, :], s[1:, :], inside[:-1, :], inside[1:, :])):
        cross = ia != ib
        pairs += list(zip(a[cross].tolist(), b[cross].tolist()))
        pairs += list(zip(b[cross].tolist(), a[cross].tolist()))
    return make_gridworld(height, width, terminals, rewards, goals, starting_states,
                          invalid_transitions=pairs)


def make_t_maze(stem_length: int, arm_length: int, goal_arm: Literal['left', 'right'] = 'right',
                reward: float = 1) -> World:
    """T-maze (gridworld_tools.py:237-298): arms along the top row, stem below their centre; the
    agent starts at the foot of the stem, the end of the chosen arm is the terminal goal."""
    assert stem_length > 0 and arm_length > 0, 'Stem and arm length must be greater than zero!'
    height, width = stem_length + 1, arm_length * 2 + 1
    goal = 0 if goal_arm == 'left' else width - 1
    inside = np.zeros((height, width), dtype=bool)
    inside[0, :] = True
    inside[:, arm_length] = True
    return corridor_gridworld(height, width, inside, [goal], np.array([[goal, reward]]), [goal],
                              [height * width - arm_length - 1])


def load_world(path: str) -> World:
    """Load a WorldDict pickled by the reference's gridworld editor
    (``pickle.dump(self.world, ...)``, misc/gridworld_gui.py:225) or by user code, and derive the
    compact tables the kernels use.  The dense ``sas`` tensor is dropped after ``next`` has been
    taken from it (it is re-created on demand)."""
    import pickle
    with open(path, 'rb') as fh:
        raw = pickle.load(fh)
    assert raw.get('deterministic', True), 'only deterministic worlds are supported'
    world = World()
    for key, value in raw.items():
        if key != 'sas':
            world[key] = value
    world['next'] = (np.asarray(raw['next'], dtype=np.uint16) if 'next' in raw
                     else np.argmax(raw['sas'], axis=2).astype(np.uint16))
    world.setdefault('deterministic', True)
    return world
