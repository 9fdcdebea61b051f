"""Random ticket plans of the persistent-workgroup Dyna-Q kernel against the one-wave-per-instance
kernel: instance counts of 3 400 .. 12 000 (a full chip's worth and up), random slices of the step
budget, a random number of whole-instance tickets per queue (with and without the flag that sends
the global-memory waves on to slices), one or two launches — Q, model, digest, instance state and
monitors must be identical (tests/test_gpu_pwg.py holds the fixed cases).

    python scripts/fuzz_pwg_tickets.py [first seed] [cases]        (on a GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cobel-rl_amd'), os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402

import test_gpu_pwg as T  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from cobel_amd import _lib
    failing = []
    os.environ['COBEL_DEBUG'] = '1'
    for seed in range(first, first + cases):
        r = np.random.default_rng(seed)
        n = int(r.integers(3400, 12001))
        steps = int(r.choice([128, 200, 256, 512]))
        k = int(r.integers(1, 6))
        cuts = np.sort(r.choice(np.arange(1, steps), size=k - 1, replace=False)) if k > 1 else np.array([], int)
        parts = np.diff(np.concatenate([[0], cuts, [steps]]))
        whole = int(r.integers(0, n // 8 + 2))
        if r.random() < 0.4:
            whole |= 0x80000000
        args = dict(n=n, side=32, seeds=[int(s) for s in r.integers(1, 10 ** 6, int(r.integers(1, 5)))],
                    launches=int(r.integers(1, 3)), steps=steps, spt=int(r.integers(20, 120)))
        for key in ('COBEL_DEBUG_PWG_SLICES', 'COBEL_DEBUG_PWG_WHOLE'):
            os.environ.pop(key, None)
        plain = T._run(_lib.F_NO_PWG, **args)
        auto = T._run(0, **args)                      # the library's own plan
        os.environ['COBEL_DEBUG_PWG_SLICES'] = ','.join(str(int(p)) for p in parts)
        os.environ['COBEL_DEBUG_PWG_WHOLE'] = str(whole)
        forced = T._run(0, **args)
        try:
            T._same(plain, auto)
            T._same(plain, forced)
            assert auto['kinds'] == {_lib.TAB_KERNEL_PWG}
        except AssertionError as e:
            failing.append((seed, n, steps, list(map(int, parts)), hex(whole), str(e)))
        if (seed - first + 1) % 5 == 0:
            print('... %d cases, %d failing' % (seed - first + 1, len(failing)), flush=True)
    print('cases %d, failing %d: %s' % (cases, len(failing), failing))
    return 1 if failing else 0


if __name__ == '__main__':
    sys.exit(main())
