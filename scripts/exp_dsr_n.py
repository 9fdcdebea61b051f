"""Dyna-DSR throughput (PyTorch-ROCm path, one step replayed from a HIP graph) against the number
of instances.  `python scripts/exp_dsr_n.py`"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from cobel_amd.agent import DynaDSR  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_open_field  # noqa: E402
from cobel_amd.network import TorchNetwork  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

dev = torch.device('cuda', 0)
for n, fused in ((2048, True), (8192, True), (8192, False)):
    torch.manual_seed(0)
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=bench.SEED, device=dev)
    ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                 TorchNetwork(bench._mlp(25, 25)), TorchNetwork(bench._mlp(25, 1)), gamma=0.8)
    ag.use_graph = None if fused else True
    ag._run(env, 4096, 50, 32, True, budget=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ag._run(env, 4096, 50, 32, True, budget=48)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'instances': n, 'fused_steps': ag.fused_steps, 'ms_per_step': dt / 48 * 1e3, 'env_steps_per_s': n * 48 / dt}), flush=True)
    del ag, env
    import gc
    gc.collect()
    torch.cuda.empty_cache()
