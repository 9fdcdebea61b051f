"""The C oracle under AddressSanitizer + UBSan (sanitizers run on the CPU
build — GPU sanitizers are not available on this pool).  `make -C oracle asan` builds
_build/libcobel_oracle_asan.so; a child process preloads the sanitizer runtime and drives every
entry point of the oracle (tabular Q / Dyna-Q with replay, masks, batches above 62, SR, the
epsilon-greedy and pairwise-sum helpers) on golden-sized inputs.  Any out-of-bounds access, signed
overflow or misaligned load aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, 'oracle')

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import c_oracle
k = np.load(os.path.join(sys.argv[1], 'tests', 'golden', 'worlds.npz'))
tab = {f: k['walls_8x8/' + f] for f in ('next', 'reward', 'terminal', 'starts')}
w = c_oracle.OracleWorld([tab, tab])
mask = np.ones((64, 4), dtype=bool); mask[5, 0] = False
for agent, batch, flags in ((c_oracle.AG_DYNAQ, 32, c_oracle.F_LEARN),
                            (c_oracle.AG_DYNAQ, 100, c_oracle.F_LEARN | c_oracle.F_EPISODIC),
                            (c_oracle.AG_Q, 24, c_oracle.F_LEARN), (c_oracle.AG_Q, 0, 0)):
    o = c_oracle.TabOracle(w, 9, agent, 7, True, trial_cap=6, log_cap=200 if agent == c_oracle.AG_Q else 0,
                           action_mask=mask, occupancy=True)
    o.run(6, 30, batch, flags=flags, trace_inst=2, trace_cap=500)
    o.run(8, 30, batch, flags=flags, step_budget=17)
    assert np.isfinite(o.Q).all()
s = c_oracle.SROracle(w, 5, 7, True, trial_cap=4, action_mask=mask, occupancy=True)
s.run(4, 25, trace_inst=1, trace_cap=200)
s.run(6, 25, step_budget=11)
assert np.isfinite(s.SR).all()
for S in (5, 128, 129, 1000):
    a = np.random.default_rng(S).standard_normal(S); b = np.random.default_rng(S + 1).standard_normal(S)
    for f32 in (0, 1):
        c_oracle.pairwise_dot(a, b, f32)
for bits in (1, 6, 15):
    c_oracle.eps_greedy(np.array([0.5, 0.5, 0.1, 0.5]), bits, 0.1, 0.73)
print('asan-ok')
'''


def test_c_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(['make', '-C', ORACLE, 'asan'], stdout=subprocess.DEVNULL)
    lib = os.path.join(ORACLE, '_build', 'libcobel_oracle_asan.so')
    assert os.path.exists(lib)
    runtime = subprocess.check_output(['gcc', '-print-file-name=libasan.so']).decode().strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip('no AddressSanitizer runtime in this toolchain')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, LD_PRELOAD=runtime, COBEL_ORACLE_LIB=lib,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and 'asan-ok' in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
