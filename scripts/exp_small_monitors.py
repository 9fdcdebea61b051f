"""Per-trial monitor cost for Dyna-Q on many small worlds (wave-per-instance kernel)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd.agent import DynaQ  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_open_field  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

n = 65536
cfg = dict(steps_per_trial=50, env_steps_per_launch=200, batch=32)
for monitors in (True, False):
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=1)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    if os.environ.get('STRIPES'):
        agent.monitor_stripes = int(os.environ['STRIPES'])
    runner = bench.Runner(cfg, env, agent)
    if not monitors:
        m = agent.monitors
        m.lat_sum = m.lat_cnt = m.reward_sum = m.resp_cnt = None
    times = []
    for _ in range(6):
        t1 = time.perf_counter()
        runner.launch()
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t1) * 1e3, 2))
    print('monitors' if monitors else 'no monitors', times)
