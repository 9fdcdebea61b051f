"""Shared fixtures.  ``-m gpu`` tests need an MI355X and run the HIP library through its C ABI;
everything else runs on CPU (oracle vs golden vectors, host logic, symbol checks)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'cobel-rl_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SEED = 0xC0BE1


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # the library honours its COBEL_DEBUG_* experiment variables only under the master switch
    # COBEL_DEBUG=1; the suite runs WITHOUT it (the production configuration) and strips stray
    # variables of a developer's shell — the tests that pin a kernel set both (debug_env below)
    for k in [k for k in os.environ if k == 'COBEL_DEBUG' or k.startswith('COBEL_DEBUG_')]:
        del os.environ[k]


def debug_env(monkeypatch, **variables):
    """Set COBEL_DEBUG_<NAME> experiment variables for one test, with the master switch."""
    monkeypatch.setenv('COBEL_DEBUG', '1')
    for k, v in variables.items():
        monkeypatch.setenv('COBEL_DEBUG_' + k, str(v))


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def golden_worlds(golden):
    data = golden('worlds')

    def get(name):
        return {k: data['%s/%s' % (name, k)] for k in
                ('next', 'reward', 'terminal', 'starts', 'coordinates', 'height', 'width')}
    return get


def cases(npz):
    return sorted({k.split('/')[0] for k in npz.files if '/' in k})


def as_world(tab):
    """Golden compact tables -> the WorldDict shape the host classes take."""
    from cobel_amd.misc.gridworld_tools import World
    w = World()
    h, wd = int(tab['height']), int(tab['width'])
    w.update(height=h, width=wd, states=h * wd, next=tab['next'], rewards=tab['reward'],
             terminals=tab['terminal'].astype(int), starting_states=tab['starts'].astype(int),
             coordinates=tab['coordinates'], deterministic=True, invalid_states=[],
             invalid_transitions=[], goals=[], wind=np.zeros((h * wd, 2), dtype=int))
    return w


def coinciding_trials(D, a, b):
    """Number of leading trials in which two recorded runs visit exactly the same (s, a) pairs."""
    sa, sb = D[a + '/steps'], D[b + '/steps']
    pa = np.stack([D[a + '/state'], D[a + '/action']], axis=1)
    pb = np.stack([D[b + '/state'], D[b + '/action']], axis=1)
    same, oa, ob = 0, 0, 0
    for t in range(min(len(sa), len(sb))):
        na, nb = int(sa[t]) + 1, int(sb[t]) + 1
        if na != nb or not np.array_equal(pa[oa:oa + na], pb[ob:ob + nb]):
            break
        same, oa, ob = same + 1, oa + na, ob + nb
    return same


def free_port() -> int:
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return int(sock.getsockname()[1])
