"""The committed golden vectors are what the REAL reference produces — re-generated here.

Runs only where the reference lies (/root/reference, the build container); skipped on the GPU
box, where only the committed fixtures travel.  One command regenerates every fixture:

    python tests/golden/gen_golden.py

and this test runs exactly that into a scratch directory and compares array by array.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
REFERENCE = '/root/reference/src/cobel'

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE),
                                reason='the reference is only present in the build container')


def _same(x: np.ndarray, y: np.ndarray) -> bool:
    if x.dtype != y.dtype or x.shape != y.shape:
        return False
    if x.dtype.kind in 'fc':
        return bool(np.array_equal(x, y, equal_nan=True))
    return bool(np.array_equal(x, y))


def test_every_fixture_regenerates_bit_identically(tmp_path):
    env = dict(os.environ, COBEL_GOLDEN_OUT=str(tmp_path))
    proc = subprocess.run([sys.executable, os.path.join(GOLDEN, 'gen_golden.py')], env=env,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-4000:]
    committed = sorted(glob.glob(os.path.join(GOLDEN, '*.npz')))
    assert len(committed) >= 14
    n_arrays = 0
    for path in committed:
        fresh_path = os.path.join(str(tmp_path), os.path.basename(path))
        assert os.path.exists(fresh_path), 'generator no longer writes %s' % os.path.basename(path)
        old, new = np.load(path, allow_pickle=False), np.load(fresh_path, allow_pickle=False)
        assert set(old.files) == set(new.files), os.path.basename(path)
        for key in old.files:
            assert _same(old[key], new[key]), '%s: %s differs' % (os.path.basename(path), key)
            n_arrays += 1
    assert n_arrays > 1500
    # the pickled WorldDicts (what the reference's gridworld editor writes, misc/gridworld_gui.py:
    # 203-239): unpickled and compared entry by entry
    import pickle
    pickles = sorted(glob.glob(os.path.join(GOLDEN, '*.pkl')))
    assert pickles
    for path in pickles:
        fresh_path = os.path.join(str(tmp_path), os.path.basename(path))
        assert os.path.exists(fresh_path), 'generator no longer writes %s' % os.path.basename(path)
        with open(path, 'rb') as fh:
            old = pickle.load(fh)
        with open(fresh_path, 'rb') as fh:
            new = pickle.load(fh)
        assert set(old) == set(new), os.path.basename(path)
        for key in old:
            a, b = old[key], new[key]
            if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                assert _same(np.asarray(a), np.asarray(b)), '%s: %s differs' % (os.path.basename(path), key)
            else:
                assert a == b, '%s: %s differs' % (os.path.basename(path), key)
    # nothing may be written next to the committed files when an output directory is given
    assert not [f for f in os.listdir(str(tmp_path)) if f.endswith('.py')]
