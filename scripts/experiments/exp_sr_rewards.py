"""SR on 32x32 worlds with 1 ... 32 rewarded states — the sparse-reward wave kernel (KX form for
three to eight, the 32-slot form for nine to 32: round 4) against the row-streaming kernel.
`python scripts/experiments/exp_sr_rewards.py`"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cobel_amd import _lib  # noqa: E402
from cobel_amd.agent import SR  # noqa: E402
from cobel_amd.interface import Gridworld  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_gridworld  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

dev = torch.device('cuda', 0)
n, steps, spt = 16384, 128, 200
rng = np.random.default_rng(4)
pos = [0, 1023, 517, 130, 300, 777, 40, 900] + [int(x) for x in rng.permutation(np.arange(2, 1020))[:40]]
pos = list(dict.fromkeys(pos))
for k in (1, 2, 4, 8, 9, 16, 24, 32):
    rw = np.array([[p, 1.0 / (j + 1)] for j, p in enumerate(pos[:k])])
    world = make_gridworld(32, 32, terminals=[0], goals=[0], rewards=rw)
    for stream in (False, True):
        env = Gridworld(world, n_envs=n, seed=5, device=dev)
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1), learning_rate=0.1, gamma=0.99)
        ag.stream_rows = stream
        ag._bind(env)
        ag._env_in(env)
        flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False)
        ag.monitors.reserve(4096, n, False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(3):      # warm-up: the agents find their rewards
            ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
        for i in range(5):
            ev[i].record()
            ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
        ev[5].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
        nzr = float((ag._rw != 0).sum(dim=1).float().mean().item())
        print('rewarded states %d, %-14s: %s ms/launch -> %.3e env-steps/s  (mean non-zero estimates %.2f)' % (
            k, 'row streaming' if stream else 'wave kernel', ' '.join('%.2f' % m for m in ms),
            n * steps / (min(ms) * 1e-3), nzr), flush=True)
        del env, ag
        import gc
        gc.collect()
        torch.cuda.empty_cache()
