"""The Dyna-DSR leg of bench.py alone (profiling runs): prints its result object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
leg = sys.argv[2] if len(sys.argv) > 2 else 'dsr'
fn = {'dsr': bench.run_dyna_dsr, 'dqn': bench.run_dyna_dqn}[leg]
print(json.dumps(fn(torch.device('cuda', 0), n)))
