"""Topology-graph builders for the environments the HIP env kernels serve.

Call surface of the reference's ``cobel.misc.topology_tools`` (misc/topology_tools.py:14-373) for
``linear_track``, ``grid``, ``t_maze`` and ``cross``: same arguments, same ``(nodes, starting_nodes)`` result
with ``nodes[id] = {'id', 'pose', 'terminal', 'reward', 'neighbors'}``; ids are ``str(n)`` in
construction order, neighbours are ordered [left, up, right, down] and point back at the node
itself where the graph ends (topology_tools.py:61,70-81).  ``hexagonal`` builds the six-neighbour
graph as the reference does: a ``Topology`` over it has six actions and runs through the general
entry points (``cobel_world_create_n``, ``cobel_eps_greedy_n``, the general tabular kernel).
``remove_obstructed_neighbors`` (topology_tools.py:472-505) prunes the edges that cross obstacle
polygons — host geometry in NumPy here (the reference calls shapely, which this package does not
require); the pruned graph is an ordinary neighbour table for the env kernels.
"""
from __future__ import annotations

import copy
from itertools import product
from typing import Literal

import numpy as np


def _nodes_from_coordinates(coords: np.ndarray, spacing: float) -> dict:
    """Unit-lattice coordinates -> nodes; neighbours by the reference's distance rule
    (0 < dx < 1.5 and |dy| < 1 is 'left', ...), evaluated for all pairs at once."""
    n = len(coords)
    dx = coords[:, None, 0] - coords[None, :, 0]
    dy = coords[:, None, 1] - coords[None, :, 1]
    off = ~np.eye(n, dtype=bool)
    rules = [(0 < dx) & (dx < 1.5) & (np.abs(dy) < 1),      # left
             (0 > dy) & (dy > -1.5) & (np.abs(dx) < 1),     # up
             (0 > dx) & (dx > -1.5) & (np.abs(dy) < 1),     # right
             (0 < dy) & (dy < 1.5) & (np.abs(dx) < 1)]      # down
    nodes = {}
    for i, (x, y) in enumerate(coords):
        neighbors = []
        for rule in rules:
            hit = np.flatnonzero(rule[i] & off[i])
            neighbors.append(str(hit[-1]) if len(hit) else str(i))   # the last match wins
        nodes[str(i)] = {'id': str(i), 'pose': (float(x) * spacing, float(y) * spacing, 0.0, 0.0,
                                                0.0, 0.0),
                         'terminal': False, 'reward': 0.0, 'neighbors': neighbors}
    return nodes


def linear_track(nb_nodes_track: int, nb_nodes_width: int, spacing: float = 1.0,
                 reward: float = 1.0, location: Literal['left', 'right'] = 'right'):
    assert nb_nodes_track > 1, 'Track has to be at least 2 states long!'
    assert nb_nodes_width > 0, 'Track has to be at least 1 state wide!'
    assert spacing > 0, 'Node spacing must be positive!'
    assert location in ['left', 'right'], 'Invalid reward location!'
    coords = np.array([[i, nb_nodes_width - j - 1] for j in range(nb_nodes_width)
                       for i in range(nb_nodes_track)], dtype=float)
    nodes = _nodes_from_coordinates(coords, spacing)
    starting = []
    for i in range(nb_nodes_width):
        goal = str(i * nb_nodes_track + (nb_nodes_track - 1) * (location == 'right'))
        nodes[goal].update({'terminal': True, 'reward': reward})
        starting.append(str(i * nb_nodes_track + (nb_nodes_track - 1) * (location == 'left')))
    return nodes, starting


def t_maze(nb_nodes_stem: int, nb_nodes_arm: int, nb_nodes_width: int, spacing: float = 1.0,
           reward: float = 1.0, location: Literal['left', 'right'] = 'right'):
    assert nb_nodes_stem > 0 and nb_nodes_arm > 0 and nb_nodes_width > 0
    assert spacing > 0, 'Node spacing must be positive!'
    assert location in ['left', 'right'], 'Invalid reward location!'
    span = nb_nodes_arm * 2 + nb_nodes_width
    coords = [[j, nb_nodes_stem + nb_nodes_width - 1 - i] for i in range(nb_nodes_width)
              for j in range(span)]
    coords += [[nb_nodes_arm + j, nb_nodes_stem - 1 - i] for i in range(nb_nodes_stem)
               for j in range(nb_nodes_width)]
    nodes = _nodes_from_coordinates(np.array(coords, dtype=float), spacing)
    for i in range(nb_nodes_width):
        nodes[str(span * i + (span - 1) * int(location == 'right'))].update(
            {'terminal': True, 'reward': reward})
    return nodes, list(nodes.keys())[-nb_nodes_width:]


def grid(nb_nodes, limits=(0.0, 1.0), reward: float = 1.0, location=None):
    nx = nb_nodes if isinstance(nb_nodes, int) else nb_nodes[0]
    ny = nb_nodes if isinstance(nb_nodes, int) else nb_nodes[1]
    assert (nx > 1 and ny >= 1) or (nx >= 1 and ny > 1), 'Invalid environment dimensions!'
    lim_x = limits if isinstance(limits[0], float) else limits[0]
    lim_y = limits if isinstance(limits[0], float) else limits[1]
    assert lim_x[1] > lim_x[0], 'Invalid x coordinate range!'
    assert lim_y[1] > lim_y[0], 'Invalid y coordinate range!'
    xs, ys = np.linspace(lim_x[0], lim_x[1], nx), np.linspace(lim_y[0], lim_y[1], ny)
    nodes = {}
    for n, (y, x) in enumerate(product(ys, xs)):
        j, i = divmod(n, nx)
        nodes[str(n)] = {
            'id': str(n), 'pose': (float(x), float(lim_y[1] - (y - lim_y[0])), 0.0, 0.0, 0.0, 0.0),
            'terminal': False, 'reward': 0.0,
            'neighbors': [str(j * nx + max(i - 1, 0)), str(max(j - 1, 0) * nx + i),
                          str(j * nx + min(i + 1, nx - 1)), str(min(j + 1, ny - 1) * nx + i)]}
    if location is None or location not in nodes:
        location = str(nx - 1)
    nodes[location].update({'terminal': True, 'reward': reward})
    starting = list(nodes.keys())
    starting.remove(location)
    return nodes, starting


def cross(nb_nodes_arm: int, nb_nodes_width: int, spacing: float = 1.0, rotation: float = 0.0):
    """Cross arena (topology_tools.py:376-469): four arms of ``nb_nodes_arm`` x ``nb_nodes_width``
    nodes around a square centre, no terminal node, every node a starting node; node ids run
    through the upper arm (top row first), the horizontal bar, then the lower arm.  Poses are laid
    out on the unit square, scaled to ``spacing``, centred and rotated by ``rotation`` degrees."""
    assert nb_nodes_arm > 0, 'The arm must be at least 1 node long!'
    assert nb_nodes_width > 0, 'The corridors must be at least 1 node wide!'
    assert spacing > 0, 'Node spacing must be positive!'
    a, w = nb_nodes_arm, nb_nodes_width
    d = 2 * a + w
    ticks = np.linspace(0, 1.0, d)
    unit = [(ticks[a + j], ticks[-(1 + i)]) for i in range(a) for j in range(w)]
    unit += [(ticks[i], ticks[-(a + 1 + j)]) for j in range(w) for i in range(d)]
    unit += [(ticks[a + j], ticks[-(i + 1 + a + w)]) for i in range(a) for j in range(w)]
    unit = np.array(unit)
    n = len(unit)
    dx = unit[:, None, 0] - unit[None, :, 0]
    dy = unit[:, None, 1] - unit[None, :, 1]
    t = 1.1 / (d - 1.0)
    off = ~np.eye(n, dtype=bool)
    rules = [(0 < dx) & (dx < t) & (np.abs(dy) < t / 2), (0 > dy) & (dy > -t) & (np.abs(dx) < t / 2),
             (0 > dx) & (dx > -t) & (np.abs(dy) < t / 2), (0 < dy) & (dy < t) & (np.abs(dx) < t / 2)]
    extent = (d - 1) * spacing
    offset = spacing * (d - 1) / 2
    theta = np.deg2rad(rotation)
    rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    nodes = {}
    for i in range(n):
        neighbors = []
        for rule in rules:
            hit = np.flatnonzero(rule[i] & off[i])
            neighbors.append(str(hit[-1]) if len(hit) else str(i))
        x, y = rot @ (np.array((unit[i, 0], unit[i, 1])) * extent - offset)
        nodes[str(i)] = {'id': str(i), 'pose': (float(x), float(y), 0.0, 0.0, 0.0, 0.0),
                         'terminal': False, 'reward': 0.0, 'neighbors': neighbors}
    return nodes, list(nodes.keys())


def hexagonal(nb_nodes: int, limits=(0.0, 1.0), reward: float = 1.0, location=None):
    """Hexagonal lattice (topology_tools.py:175-272): every other row is shifted by half a
    spacing and loses the node that leaves the range; neighbours are the nodes closer than 1.5
    spacings, sorted into six 60-degree sectors and listed clockwise starting at the left; a
    missing neighbour is the node itself.  Six actions (``Topology`` takes the general kernels)."""
    assert nb_nodes > 1, 'Invalid number of nodes!'
    assert limits[1] > limits[0], 'Invalid coordinate range!'
    spacing = (limits[1] - limits[0]) / (nb_nodes - 1)
    line = np.linspace(limits[0], limits[1], nb_nodes)
    coords = np.array([[x, y] for y, x in product(line, line)])
    odd_rows = (np.arange(nb_nodes * nb_nodes) // nb_nodes) % 2 == 1
    coords[odd_rows, 0] += spacing / 2
    coords = coords[coords[:, 0] <= limits[1]]
    pose_xy = np.stack([coords[:, 0], limits[1] - (coords[:, 1] - limits[0])], axis=1)
    n = len(pose_xy)
    delta = pose_xy[None, :, :] - pose_xy[:, None, :]          # [from, to]
    dist = np.sqrt(np.sum(delta ** 2, axis=2))
    sectors = np.array([(i * 60 - 120) % 360 for i in range(6)])
    nodes = {}
    for i in range(n):
        slots = {int(a): str(i) for a in sectors}
        # the reference walks [self] * 6 first (angle 0) and then the real neighbours in id order
        for j in [i] + [k for k in range(n) if k != i and dist[i, k] < spacing * 1.5]:
            angle = np.angle(complex(delta[i, j, 0], delta[i, j, 1]), deg=True) % 360
            slots[int(sectors[int(np.argmin(np.abs(sectors - angle)))])] = str(j)
        nodes[str(i)] = {'id': str(i),
                         'pose': (pose_xy[i, 0], pose_xy[i, 1], 0.0, 0.0, 0.0, 0.0),
                         'terminal': False, 'reward': 0.0,
                         'neighbors': list(slots.values())[::-1]}
    if location is None or location not in nodes:
        location = str(nb_nodes - 1)
    nodes[location].update({'terminal': True, 'reward': reward})
    starting_nodes = [k for k in nodes if k != location]
    return nodes, starting_nodes


# ---------------------------------------------------------------------------------------------
def _rings(polygon) -> list:
    """Exterior ring first, then the holes, each an (n, 2) array of vertices (closing vertex
    dropped).  Accepts a sequence of (x, y[, z]) vertices or an object with shapely's
    ``exterior.coords`` / ``interiors``."""
    if hasattr(polygon, 'exterior'):
        rings = [polygon.exterior.coords] + [r.coords for r in getattr(polygon, 'interiors', [])]
    else:
        rings = [polygon]
    out = []
    for ring in rings:
        v = np.asarray(list(ring), dtype=np.float64)[:, :2]
        if len(v) > 1 and np.array_equal(v[0], v[-1]):
            v = v[:-1]
        assert len(v) >= 3, 'a polygon needs at least three vertices'
        out.append(v)
    return out


def _inside(p: np.ndarray, ring: np.ndarray) -> bool:
    """Even-odd rule (points on the boundary are settled by the distance test, not here)."""
    x, y = ring[:, 0], ring[:, 1]
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    cross = (y > p[1]) != (yn > p[1])
    with np.errstate(divide='ignore', invalid='ignore'):
        xi = x + (p[1] - y) * (xn - x) / (yn - y)
    return bool(np.count_nonzero(cross & (p[0] < xi)) % 2)


def _segment_distance(a: np.ndarray, b: np.ndarray, c: np.ndarray, d: np.ndarray) -> float:
    """Distance between the segments ab and cd (0 when they cross or touch)."""
    def orient(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])

    def point_segment(p, q, r):
        qr = r - q
        den = float(qr @ qr)
        t = 0.0 if den == 0.0 else min(1.0, max(0.0, float((p - q) @ qr) / den))
        return float(np.hypot(*(p - (q + t * qr))))

    def within(p, q, r):     # r is collinear with pq: does it lie between them?
        return (min(p[0], q[0]) <= r[0] <= max(p[0], q[0])
                and min(p[1], q[1]) <= r[1] <= max(p[1], q[1]))

    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    if o1 * o2 < 0 and o3 * o4 < 0:   # a proper crossing
        return 0.0
    # touching: an end of one segment exactly ON the other (a zero orientation inside the other's
    # bounding box) — decided by the orientation signs, not by a distance that has to round to 0.0
    if ((o1 == 0 and within(a, b, c)) or (o2 == 0 and within(a, b, d))
            or (o3 == 0 and within(c, d, a)) or (o4 == 0 and within(c, d, b))):
        return 0.0
    return min(point_segment(a, c, d), point_segment(b, c, d), point_segment(c, a, b),
               point_segment(d, a, b))


def _edge_obstructed(p1: np.ndarray, p2: np.ndarray, polygons: list, buffer_distance: float) -> bool:
    """Does the segment p1-p2 meet a polygon (boundary included) grown by buffer_distance?"""
    for rings in polygons:
        dist = np.inf
        for ring in rings:
            nxt = np.roll(ring, -1, axis=0)
            for c, d in zip(ring, nxt):
                dist = min(dist, _segment_distance(p1, p2, c, d))
        if dist <= buffer_distance:
            return True
        # no boundary within reach: the segment lies wholly inside or wholly outside the area
        if _inside(p1, rings[0]) and not any(_inside(p1, hole) for hole in rings[1:]):
            return True
    return False


def remove_obstructed_neighbors(nodes: dict, obstacles: list, buffer_distance: float = 0.0) -> dict:
    """Replace every edge of a topology graph that is obstructed by one of the obstacle polygons with
    a self-loop (the reference's ``remove_obstructed_neighbors``, topology_tools.py:472-505: a deep
    copy of ``nodes`` in which ``neighbors[i] = own id`` wherever the straight line between the two
    nodes' (x, y) positions intersects the obstacles grown by ``buffer_distance``; z is ignored).

    ``obstacles``: polygons as sequences of (x, y) vertices, or shapely polygons (exterior +
    holes are read from them; shapely itself is not needed).  The obstacles are grown by the exact
    Euclidean distance; shapely's ``buffer`` rounds corners with eight chords per quarter circle,
    which is smaller by at most 0.5 % of the distance — edges that pass a CORNER within that margin
    are the only ones on which the two can differ."""
    assert buffer_distance >= 0, 'The buffer distance has to be non-negative!'
    nodes_updated = copy.deepcopy(nodes)
    polygons = [_rings(p) for p in obstacles]
    for n, node in nodes_updated.items():
        pos_1 = np.asarray(node['pose'], dtype=np.float64)[:2]
        for i, neighbor in enumerate(node['neighbors']):
            pos_2 = np.asarray(nodes_updated[neighbor]['pose'], dtype=np.float64)[:2]
            if _edge_obstructed(pos_1, pos_2, polygons, float(buffer_distance)):
                node['neighbors'][i] = n
    return nodes_updated
