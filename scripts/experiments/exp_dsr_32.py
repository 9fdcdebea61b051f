"""Dyna-DSR at the size of BASELINE config 4's worlds: 32x32 open field, one-hot inputs of 1 024,
successor networks 1024-64-64-1024 and a reward network 1024-64-64-1 in float64 per instance.
Networks of that width are outside the fused MLP kernels (inputs / outputs <= 32): the agent runs
its PyTorch-ROCm loop — stacked parameters through batched GEMMs (the library GEMM is the right
tool for 1024-wide layers), the fused Adam kernel, one step replayed from a HIP graph.
    python scripts/experiments/exp_dsr_32.py [instances]"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for n in ([int(a) for a in sys.argv[1:]] or [64, 256, 1024]):
    torch.manual_seed(0)
    env = Gridworld(make_open_field(32, 32, 0, 1), n_envs=n, seed=bench.SEED, device=dev)
    ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                 TorchNetwork(bench._mlp(1024, 1024)), TorchNetwork(bench._mlp(1024, 1)), gamma=0.8)
    ag._run(env, 4096, 200, 32, True, budget=4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ag._run(env, 4096, 200, 32, True, budget=16)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    params = 9 * (1024 * 64 + 64 + 64 * 64 + 64) + 8 * (64 * 1024 + 1024) + (64 + 1)
    print(json.dumps({'instances': n, 'fused_steps': ag.fused_steps, 'graph_replays': ag.graph_replays,
                      'ms_per_step': dt / 16 * 1e3, 'env_steps_per_s': n * 16 / dt,
                      'parameters_per_instance': params,
                      'GB_per_step_at_8_accesses': n * params * 8 * 8 / 1e9,
                      'mem_GiB': torch.cuda.max_memory_allocated() / 2**30}), flush=True)
    del ag, env
    gc.collect()
    torch.cuda.empty_cache()
