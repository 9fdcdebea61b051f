"""The C5 legs of bench.py alone: python scripts/run_c5_leg.py [f64|f32] [instances]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else 'f64'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
r = bench.run_c5(torch.device('cuda', 0), dt, n=n)
r['roofline'].pop('traffic', None)
print(json.dumps({k: r[k] for k in ('value', 'ms_per_step', 'dtype')} | {'frac': r['roofline']['frac']}))
