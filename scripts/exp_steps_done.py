"""Cost of the one-address steps_done atomic at the end of every workgroup."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

for name in sys.argv[1:] or ['C3', 'C6']:
    for keep in (True, False):
        cfg = dict(bench.CONFIGS[name])
        env, agent = bench.build_agent(name, cfg, cfg['instances'], 0, torch.device('cuda', 0))
        runner = bench.Runner(cfg, env, agent)
        if not keep:
            agent.monitors.steps_done = None
            if hasattr(agent, 'replays_done'):
                agent.replays_done = None
        times = []
        for _ in range(4):
            t1 = time.perf_counter()
            runner.launch()
            torch.cuda.synchronize()
            times.append(round((time.perf_counter() - t1) * 1e3, 2))
        print(name, 'with steps_done' if keep else 'without', times)
