"""SR on small worlds (S not a multiple of 64: the ANY_S instantiations of k_sr_wave), 1 / 3 / 12
rewarded states, deterministic and slippery: python scripts/experiments/exp_sr_small.py [n] [10x10,31x31,...]"""
import os
import sys
os.environ.setdefault('COBEL_DEBUG', '1')
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


def slippery(world, p_slip=0.2):
    det = np.argmax(world['sas'], axis=2)
    sas = np.zeros_like(world['sas'])
    for s in range(sas.shape[0]):
        for a in range(4):
            sas[s, a, det[s, a]] += 1.0 - p_slip
            sas[s, a, det[s, (a + 1) % 4]] += p_slip / 2
            sas[s, a, det[s, (a + 3) % 4]] += p_slip / 2
    world['sas'] = sas
    world['deterministic'] = False
    return world


dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps, spt = 128, 100
pos = [99, 5, 50, 73, 18, 31, 44, 67, 80, 92, 9, 26]
sizes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [(10, 10), (14, 14)]
for (h, w) in sizes:
    for k in (1, 3, 12):
        for slip in (False, True):
            rw = np.array([[p % (h * w), 1.0 / (j + 1)] for j, p in enumerate(pos[:k])])
            world = make_gridworld(h, w, terminals=[0], goals=[0], rewards=rw)
            if slip:
                world = slippery(world)
            env = Gridworld(world, n_envs=n, seed=5, device=dev)
            ag = SR(env.observation_space, env.action_space, EpsilonGreedy(0.1), learning_rate=0.1, gamma=0.99)
            ag._bind(env)
            ag._env_in(env)
            flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False)
            ag.monitors.reserve(4096, n, False)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            for i in range(3):
                ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
            for i in range(5):
                ev[i].record()
                ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
            ev[5].record()
            torch.cuda.synchronize()
            ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
            print('%dx%d, %2d rewarded, %-13s: %.3f ms/launch -> %.3e env-steps/s  (kernel %d)' % (
                h, w, k, 'slippery' if slip else 'deterministic', min(ms), n * steps / (min(ms) * 1e-3),
                ag.last_kernel if hasattr(ag, 'last_kernel') else -1), flush=True)
            del env, ag
            import gc
            gc.collect()
            torch.cuda.empty_cache()
