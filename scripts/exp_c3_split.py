"""C3 launch time with planning (B = 50) and without (no_replay): how much of a step is the scalar
select -> env.step -> model store -> online TD chain."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd import _lib  # noqa: E402

for mode in ('planning', 'no_replay', 'B=10'):
    cfg = dict(bench.CONFIGS['C3'])
    if mode == 'B=10':
        cfg['batch'] = 10
    env, agent = bench.build_agent('C3', cfg, cfg['instances'], 0, torch.device('cuda', 0))
    runner = bench.Runner(cfg, env, agent)
    if mode == 'no_replay':
        runner.flags |= _lib.F_NO_REPLAY
    times = []
    for _ in range(4):
        t1 = time.perf_counter()
        runner.launch()
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t1) * 1e3, 2))
    print(mode, times)
