"""C6 launch time with and without the end-of-trial replays (how much of a launch is online
stepping)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd import _lib  # noqa: E402

for no_replay in (False, True):
    cfg = dict(bench.CONFIGS['C6'])
    env, agent = bench.build_agent('C6', cfg, cfg['instances'], 0, torch.device('cuda', 0))
    runner = bench.Runner(cfg, env, agent)
    if no_replay:
        runner.flags |= _lib.F_NO_REPLAY
    times = []
    for _ in range(5):
        t1 = time.perf_counter()
        runner.launch()
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t1) * 1e3, 2))
    print('no replay' if no_replay else 'replay', times, 'reactivations', int(agent.replays_done.item()))
