"""SFMA throughput at other world sizes (experiments): python scripts/exp_sfma.py SIDE N [mode]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cobel_amd.agent import SFMA  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.memory import SFMAMemory  # noqa: E402
from cobel_amd.memory.utils import DR  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_gridworld  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

side, n = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else 'reverse'
w = make_gridworld(side, side, terminals=[side - 1], rewards=np.array([[side - 1, 10]]),
                   goals=[side - 1])
w['starting_states'] = np.array([side * side // 2])
env = Gridworld(w, n_envs=n, seed=1)
metric = DR(side, side, w['next'], 0.9, [])
agent = SFMA(env.observation_space, env.action_space, EpsilonGreedy(0.1),
             SFMAMemory(metric, side * side, 4))
agent.M.mode = mode
agent.train(env, 1, 4 * side, 32)
torch.cuda.synchronize()
s0, r0 = agent.env_steps(), int(agent.replays_done.item())
t0 = time.perf_counter()
agent.train(env, 4, 4 * side, 32)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s1, r1 = agent.env_steps(), int(agent.replays_done.item())
print('side %d n %d: %.3g env-steps/s, %.3g reactivations/s (%.1f ms)' %
      (side, n, (s1 - s0) / dt, (r1 - r0) / dt, dt * 1e3))
