"""Dyna-Q on many small worlds (the reference's demo_dyna_q.py configuration, vectorised):
python scripts/exp_small_dynaq.py [SIDE] [N] [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

from cobel_amd.agent import DynaQ  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_open_field  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
env = Gridworld(make_open_field(side, side, 0, 1), n_envs=n, seed=1)
agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
agent.train(env, 2, 50, B)
torch.cuda.synchronize()
s0 = agent.env_steps()
t0 = time.perf_counter()
agent.train(env, 20, 50, B)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%dx%d n %d B %d: %.3g env-steps/s (%.1f ms)' % (side, side, n, B, (agent.env_steps() - s0) / dt,
                                                        dt * 1e3))
