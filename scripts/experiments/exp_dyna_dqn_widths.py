"""Dyna-DQN step time over input widths and dtypes (the staging kernel's DI = 2 / 4 / 8
instantiations): python scripts/experiments/exp_dyna_dqn_widths.py [n]"""
import os
import sys
os.environ.setdefault('COBEL_DEBUG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd.agent import DynaDQN  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_open_field  # noqa: E402
from cobel_amd.network import TorchNetwork  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda', 0)
for (h, w) in ((2, 3), (3, 4), (5, 5)):
    for dt in ('f32', 'f64'):
        torch.manual_seed(0)
        env = Gridworld(make_open_field(h, w, 0, 1), n_envs=n, seed=1, device=dev)
        agent = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                        TorchNetwork(bench._mlp(h * w, 4, dt)), gamma=0.8)
        r = bench._timed_network_agent(agent, env, n, 65, dev)
        print('%d inputs %s: %.4f ms per step (fused %s)' % (h * w, dt, r['ms_per_step'], agent.fused_steps > 0), flush=True)
