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


def successor_table(sas) -> np.ndarray:
    """``next[S, A]`` of a dense ``sas[S, A, S]``: the argmax of every row — what the reference's
    ``Gridworld.step`` takes when ``world['deterministic']`` is set (interface/gridworld.py:115-117)
    and, for the one-hot rows every builder writes, the only possible successor."""
    sas = np.asarray(sas)
    if not (np.count_nonzero(sas, axis=2) >= 1).all():
        raise ValueError('world["sas"] has rows without any successor')
    return np.argmax(sas, axis=2).astype(np.uint16)


def is_one_hot(sas) -> bool:
    return bool((np.count_nonzero(np.asarray(sas), axis=2) == 1).all())


def transition_lists(sas):
    """List form of a dense ``sas[S, A, S]`` whose rows are distributions: offsets ``[S * A + 1]``,
    the possible successors of every (state, action) in ascending state order, and the normalised
    cumulative sum of their probabilities formed as ``Generator.choice`` forms it (float64 cumsum
    over the row, divided by its last entry; entries of probability zero add nothing to a cumsum,
    so leaving them out changes no value) — ``cobel_world_set_transitions`` (include/cobel_hip.h)."""
    sas = np.asarray(sas, dtype=np.float64)
    S, A, _ = sas.shape
    if (sas < 0).any() or not (np.count_nonzero(sas, axis=2) >= 1).all():
        raise ValueError('rows of world["sas"] must be probability vectors')
    # Generator.choice's own check (numpy/random/_generator.pyx: `abs(kahan_sum(p) - 1.) > atol`
    # with atol = sqrt(eps) of float64): the reference raises here too, at the first draw
    if (np.abs(sas.sum(axis=2) - 1.0) > np.sqrt(np.finfo(np.float64).eps)).any():
        raise ValueError('probabilities do not sum to 1')
    off = np.zeros(S * A + 1, dtype=np.uint32)
    states, cdf = [], []
    for p, row in enumerate(sas.reshape(S * A, S)):
        c = np.cumsum(row)
        c /= c[-1]
        # a successor is kept where the cumulative sum ADVANCES: an entry too small to move the
        # float64 cumsum can never be returned by searchsorted(side='right') — the same draws
        # select the same states with or without it — and the list stays strictly increasing
        keep = np.flatnonzero(np.diff(np.concatenate(([0.0], c))) > 0)
        states.append(keep.astype(np.uint16))
        cdf.append(c[keep])
        cdf[-1][-1] = 1.0
        off[p + 1] = off[p] + len(keep)
    return off, np.concatenate(states), np.concatenate(cdf)


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
        out = {}
        if dict.__contains__(self, 'sas'):
            # somebody materialised (and may have edited) the dense tensor: it is the reference's
            # source of truth.  The index table follows it (argmax, what the reference takes from a
            # row while world['deterministic'] is set); rows that are distributions of a world with
            # the flag off are DRAWN from (interface/gridworld.py:119-123) and travel as lists.
            sas = dict.__getitem__(self, 'sas')
            self['next'] = successor_table(sas)
            if not self.get('deterministic', True) and not is_one_hot(sas):
                out['transitions'] = transition_lists(sas)
        return dict(out, 
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
    # The flag only selects how Gridworld.step reads a row of sas (argmax, or a draw from it:
    # interface/gridworld.py:115-123); the rows this builder writes are one-hot either way
    # (gridworld_tools.py:103-134), and a draw from a one-hot row is its only outcome.
    world['deterministic'] = deterministic
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
    world = World()
    for key, value in raw.items():
        if key != 'sas':
            world[key] = value
    world['next'] = (np.asarray(raw['next'], dtype=np.uint16) if 'next' in raw
                     else successor_table(raw['sas']))
    world.setdefault('deterministic', True)
    return world


def make_double_t_maze(stem_length: int, arm_length: int,
                       goal_arm: Literal['left-left', 'left-right', 'right-left',
                                         'right-right'] = 'right-right',
                       reward: float = 1) -> World:
    """Double T-maze (gridworld_tools.py:301-432): a main stem leads up to a cross arm whose two
    ends carry a small T each; the four arm ends of the small Ts are the candidate goals."""
    assert stem_length > 0 and arm_length > 0, 'Stem and arm length must be greater than zero!'
    a = arm_length
    height, width = stem_length * 2 + 2, a * 4 + 3
    goal = {'left-left': 0, 'left-right': a * 2, 'right-left': a * 2 + 2,
            'right-right': a * 4 + 2}[goal_arm]
    inside = np.zeros((height, width), dtype=bool)
    inside[0, :2 * a + 1] = True                    # arms of the left small T
    inside[0, 2 * a + 2:] = True                    # arms of the right small T
    inside[1:stem_length + 1, a] = True             # stems of the small Ts
    inside[1:stem_length + 1, 3 * a + 2] = True
    inside[stem_length + 1, a:3 * a + 3] = True     # cross arm
    inside[stem_length + 1:, 2 * a + 1] = True      # main stem
    return corridor_gridworld(height, width, inside, [goal], np.array([[goal, reward]]), [goal],
                              [height * width - a * 2 - 2])


def make_8_maze(center_height: int, lap_width: int,
                goal_location: Literal['left', 'right'] = 'right', reward: float = 1) -> World:
    """Figure-8 maze (gridworld_tools.py:669-735): two laps sharing the centre corridor; the goal
    sits halfway down the outer side of one lap, the agent starts right of the centre on top."""
    assert center_height > 0 and lap_width > 0, \
        'Center height and lap width must be greater than zero!'
    height, width = center_height + 2, lap_width * 2 + 3
    goal = int((center_height + 2) / 2) * width + (width - 1 if goal_location == 'right' else 0)
    inside = np.zeros((height, width), dtype=bool)
    inside[[0, height - 1], :] = True
    inside[:, [0, lap_width + 1, width - 1]] = True
    return corridor_gridworld(height, width, inside, [goal], np.array([[goal, reward]]), [goal],
                              [lap_width + 2])


def make_cross_maze(arm_length: int, arm_width: int,
                    goal_arm: Literal['left', 'top', 'right', 'bottom'] = 'top',
                    reward: float = 1.0) -> World:
    """Cross maze (gridworld_tools.py:898-974): four arms of width ``arm_width`` around a square
    centre; every cell at the end of the goal arm is a rewarded terminal, the agent starts
    anywhere in the centre.  As in the reference (:963) the goal cells pay 1 whatever ``reward``
    says."""
    assert arm_length > 0 and arm_width > 0
    assert goal_arm in ('left', 'top', 'right', 'bottom'), 'Invalid goal arm!'
    size = arm_length * 2 + arm_width
    band = np.arange(size)
    mid = (band >= arm_length) & (band < arm_length + arm_width)
    inside = mid[:, None] | mid[None, :]
    k = np.arange(arm_width)
    terminals = {'top': k + arm_length, 'left': k * size + arm_length * size,
                 'right': k * size + arm_length * size + (size - 1),
                 'bottom': k + arm_length + size * (size - 1)}[goal_arm].tolist()
    rewards = np.stack([np.array(terminals, dtype=float), np.ones(arm_width)], 1)
    starts = [arm_length * size + arm_length + i * size + j for i in range(arm_width)
              for j in range(arm_width)]
    return corridor_gridworld(size, size, inside, terminals, rewards, terminals, starts)


def make_two_sided_t_maze(stem_length: int, arm_length: int,
                          goal_arm: Literal['left-left', 'left-right', 'right-left',
                                            'right-right'] = 'right-right',
                          reward: float = 1) -> World:
    """Two-sided T-maze (gridworld_tools.py:435-506): a horizontal stem with a vertical arm pair
    at either end (an "H" lying on its side); the agent starts in the middle of the stem."""
    assert stem_length > 0 and arm_length > 0, 'Stem and arm length must be greater than zero!'
    height, width = arm_length * 2 + 1, stem_length + 2
    goal = {'left-right': 0, 'right-left': width - 1, 'left-left': width * (height - 1),
            'right-right': width * height - 1}.get(goal_arm, 0)
    inside = np.zeros((height, width), dtype=bool)
    inside[:, [0, width - 1]] = True
    inside[arm_length, :] = True
    return corridor_gridworld(height, width, inside, [goal], np.array([[goal, reward]]), [goal],
                              [arm_length * width + int(stem_length / 2)])


def make_two_choice_t_maze(center_height: int, lap_width: int, arm_length: int,
                           chirality: Literal['left', 'right'] = 'right',
                           goal_location: Literal['left', 'right'] = 'right',
                           reward: float = 1) -> World:
    """Two-choice T-maze (gridworld_tools.py:509-666): a figure-8 frame whose centre corridor
    stops halfway down at the bar of an inner T shifted to one side (``chirality``); the T's stem
    continues to the bottom corridor.  As in the reference (:554) the goal pays 1 whatever
    ``reward`` says, and the start column does not depend on the chirality (:551)."""
    assert arm_length > 0, '!'
    assert center_height > 2, '!'
    assert lap_width >= arm_length * 2 + 1, '!'
    assert chirality in ['left', 'right'], 'Invalid chirality!'
    height, width = center_height + 2, lap_width * 2 + 3
    goal = int(height / 2) * width + (width - 1 if goal_location == 'right' else 0)
    bar_row, centre = int((center_height - 1) / 2) + 1, lap_width + 1
    side = 1 if chirality == 'right' else -1
    inside = np.zeros((height, width), dtype=bool)
    inside[[0, height - 1], :] = True
    inside[:, [0, width - 1]] = True
    inside[:bar_row + 1, centre] = True
    lo, hi = sorted((centre, centre + side * 2 * arm_length))
    inside[bar_row, lo:hi + 1] = True
    inside[bar_row:, centre + side * arm_length] = True
    return corridor_gridworld(height, width, inside, [goal], np.array([[goal, 1.0]]), [goal],
                              [width * (height - 1) + lap_width + arm_length])


def make_detour_maze(width_small: int, height_small: int, width_large: int, height_large: int,
                     reward: float = 1) -> World:
    """Detour maze (gridworld_tools.py:738-895): a straight corridor from the start (bottom) to
    the goal (top) with a small loop hanging off its lower left and a large loop off its upper
    right, both joining it at the same crossing."""
    assert width_small > 0 and height_small > 0, \
        'Width and height of the small side piece must be greater than zero!'
    assert width_large > width_small and height_large > height_small, \
        'Width and height of the large side piece must be greater than those of the small side piece!'
    width, height = width_small + width_large + 3, height_small + height_large + 5
    spine, crossing = width_small + 1, height_large + 2
    low = crossing + height_small + 1
    inside = np.zeros((height, width), dtype=bool)
    inside[:, spine] = True
    inside[crossing, :] = True
    inside[1, spine:] = True                        # large loop: out along row 1, down the right edge
    inside[1:crossing + 1, width - 1] = True
    inside[crossing:low + 1, 0] = True              # small loop: down the left edge, back along `low`
    inside[low, :spine + 1] = True
    return corridor_gridworld(height, width, inside, [spine], np.array([[spine, reward]]), [spine],
                              [spine + width * (height - 1)])
