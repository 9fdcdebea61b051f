"""Shape sweep of the fused network kernels against their PyTorch paths: every input width 1..32
of `cobel_dqn_replay` (float64 / float32, DQN / DDQN targets) and random (inputs, outputs) pairs of
`cobel_mlp_forward` / `cobel_mlp_fit`, through the bodies of the GPU tests that pin them
(tests/test_gpu_parity.py, tests/test_gpu_mlp.py).  float64 to round-off (1e-10 / 1e-9 relative);
float32 within 1e-4 absolute: Adam divides by |g| + 1e-8, which turns the rounding of gradient
entries near 1e-8 into ~1e-3 of a step (scripts/experiments/exp_f32_adam.py measures both paths against float64).

    python scripts/fuzz_networks.py [pairs]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'cobel-rl_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def main() -> int:
    import torch
    import test_gpu_mlp
    import test_gpu_parity
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    failed = []
    for n_in in range(1, 33):
        for dtype_name in ('f64', 'f32'):
            ddqn = bool((n_in + (dtype_name == 'f32')) % 2)
            for kernel in ('lds', 'stream'):   # both forms of the step, whichever the library picks
                try:
                    test_gpu_parity.test_fused_dqn_replay_equals_torch_path(
                        torch, dtype_name, n_in, ddqn, kernel, f32_atol=1e-4)
                except AssertionError as e:
                    failed.append(('dqn_replay', kernel, n_in, dtype_name, ddqn, str(e)[:200]))
                    print('MISMATCH', failed[-1], flush=True)
    print('dqn_replay sweep done, failing %d' % len(failed), flush=True)
    r = np.random.default_rng(5)
    shapes = {(1, 1), (32, 32), (1, 32), (32, 1)}
    while len(shapes) < pairs:
        shapes.add((int(r.integers(1, 33)), int(r.integers(1, 33))))
    for k, (D, O) in enumerate(sorted(shapes)):
        dtype_name = 'f32' if k % 4 == 3 else 'f64'
        try:
            test_gpu_mlp.test_mlp_forward_and_fit_match_pytorch(torch, D, O, dtype_name, f32_atol=1e-4)
        except AssertionError as e:
            failed.append(('mlp', D, O, dtype_name, str(e)[:200]))
            print('MISMATCH', failed[-1], flush=True)
    print('cases %d, failing %d: %s' % (64 + len(shapes), len(failed), failed))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
