"""demo/topology/demo_dqn.py --env hexagonal at 8 192 instances: the two-kernel loop (streaming
replay kernel, six actions) against the PyTorch-ROCm loop: python scripts/experiments/exp_dqn_hex.py"""
import os
import sys
os.environ.setdefault('COBEL_DEBUG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd.agent import DQN  # noqa: E402
from cobel_amd.interface import Topology  # noqa: E402
from cobel_amd.misc.topology_tools import hexagonal  # noqa: E402
from cobel_amd.network import TorchNetwork  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nodes, starts = hexagonal(10, (0.0, 1.0))
for dt in ('f64', 'f32'):
    for fused in (True, False):
        torch.manual_seed(0)
        env = Topology(nodes, starts, n_envs=n, seed=1, device=dev)
        agent = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                    TorchNetwork(bench._mlp(6, 6, dt)), gamma=0.8)
        if not fused:
            agent.fused_loop = False
        r = bench._timed_network_agent(agent, env, n, 65 if fused else 17, dev)
        print('hexagonal(10), %d instances, %s, %s: %.3f ms per step (fused steps %d)' % (
            n, dt, 'two-kernel loop' if fused else 'PyTorch-ROCm loop', r['ms_per_step'], agent.fused_steps), flush=True)
