"""Run only the C5 leg of bench.py (DQN on PyTorch-ROCm) — for profiling:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o c5 --output-format csv -- \
        python3 scripts/run_c5.py f64 64
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == '__main__':
    dt = sys.argv[1] if len(sys.argv) > 1 else 'f64'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    print(json.dumps(bench.run_c5(torch.device('cuda', 0), dt, iters=iters)))
