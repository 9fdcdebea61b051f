import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch
import bench
from cobel_amd.agent import DynaDSR
from cobel_amd.interface import Gridworld
from cobel_amd.misc.gridworld_tools import make_open_field
from cobel_amd.network import TorchNetwork
from cobel_amd.policy import EpsilonGreedy
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=bench.SEED, device=dev)
ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.1),
             TorchNetwork(bench._mlp(25, 25)), TorchNetwork(bench._mlp(25, 1)), gamma=0.8)
ag._run(env, 4096, 50, 32, True, budget=8)
torch.cuda.synchronize()
ag._run(env, 4096, 50, 32, True, budget=40)
torch.cuda.synchronize()
print('fused', ag.fused_steps)
