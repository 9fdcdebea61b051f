"""SR agent throughput against the world size (open fields with one rewarded goal): which kernel
cobel_sr_run takes and what it delivers.  python scripts/experiments/exp_sr_sizes.py"""
import gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch
from cobel_amd.agent import SR
from cobel_amd.interface import Gridworld
from cobel_amd.misc.gridworld_tools import make_open_field
from cobel_amd.policy import EpsilonGreedy
SIZES = ((5, 65536), (6, 65536), (8, 65536), (10, 65536), (16, 65536), (20, 32768), (24, 16384), (28, 16384), (32, 16384), (17, 32768), (25, 16384), (27, 16384), (29, 16384), (31, 16384))
if len(sys.argv) > 1:      # python scripts/experiments/exp_sr_sizes.py 25 27 31 [stream]
    SIZES = tuple((int(a), 16384) for a in sys.argv[1:] if a.isdigit())
for side, n in SIZES:
    env = Gridworld(make_open_field(side, side, 0, 1), n_envs=n, seed=1)
    ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    ag.stream_rows = 'stream' in sys.argv
    ag.train(env, 1, 64)
    torch.cuda.synchronize()
    before = int(ag.monitors.steps_done.item())
    t0 = time.perf_counter()
    ag.train(env, 4, 128)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(ag.monitors.steps_done.item()) - before
    moved = ag.traffic.cpu().numpy().tolist()
    print(json.dumps({'side': side, 'S': side * side, 'instances': n, 'ms': dt * 1e3,
                      'env_steps_per_s': steps / dt,
                      'kernel': 'k_sr_wave' if moved[1] else 'k_sr'}), flush=True)
    del ag, env; gc.collect(); torch.cuda.empty_cache()
