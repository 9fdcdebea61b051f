"""Throughput of the general tabular kernel (csrc/general.hip) beside the wavefront kernels:
C3's mazes with Dyna-Q at B = 50 (both kernels) and B = 100 (general only), QAgent on a hexagonal
Topology.  `python scripts/experiments/exp_general.py`"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)


def run(batch, general, n=65536, steps=64, launches=3):
    cfg = dict(bench.CONFIGS['C3'], instances=n, env_steps_per_launch=steps, batch=batch)
    env, ag = bench.build_agent('C3', cfg, n, 0, dev)
    ag.force_general = general
    r = bench.Runner(cfg, env, ag)
    r.launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for k in range(launches):
        r.launch()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]
    print(json.dumps({'batch': batch, 'general': general, 'instances': n, 'ms': [round(x, 2) for x in ms],
                      'env_steps_per_s': n * steps / (np.mean(ms) * 1e-3),
                      'td_updates_per_s': n * steps * (batch + 1) / (np.mean(ms) * 1e-3)}), flush=True)


if __name__ == '__main__':
    run(50, False)
    run(50, True)
    run(100, True)
    run(62, False)
    run(63, False)
    run(100, False)
    run(124, False)
    run(63, True)
