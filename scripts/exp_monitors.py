"""Cost of the per-trial monitor atomics: bench configs with and without monitor buffers.
python scripts/exp_monitors.py C6|C3|C1"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C6'
for monitors in (True, False):
    cfg = dict(bench.CONFIGS[name])
    env, agent = bench.build_agent(name, cfg, cfg['instances'], 0, torch.device('cuda', 0))
    if os.environ.get('STRIPES'):
        agent.monitor_stripes = int(os.environ['STRIPES'])
    runner = bench.Runner(cfg, env, agent)
    if not monitors:
        m = agent.monitors
        m.lat_sum = m.lat_cnt = m.reward_sum = m.resp_cnt = None
    runner.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    times = []
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        t1 = time.perf_counter()
        runner.launch()
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t1) * 1e3, 2))
    print(name, 'monitors' if monitors else 'no monitors', times)
