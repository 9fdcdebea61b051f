"""Run only the "next rows" legs of bench.py (Dyna-DQN, Dyna-DSR, grid search)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == '__main__':
    print(json.dumps(bench.run_next_rows(torch.device('cuda', 0)), indent=1))
