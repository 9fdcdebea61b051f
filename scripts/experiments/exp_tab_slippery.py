"""Q-learning / Dyna-Q on deterministic and slippery gridworlds (the generic wavefront kernel draws
the successor in the step): python scripts/experiments/exp_tab_slippery.py [n] [10x10,16x16,...]"""
import os
import sys
os.environ.setdefault('COBEL_DEBUG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cobel_amd import _lib  # noqa: E402
from cobel_amd.agent import DynaQ, QAgent  # noqa: E402
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
sizes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [(10, 10), (32, 32)]
for (h, w) in sizes:
    for kind, batch in (('dynaq', 32), ('q', 0), ('q', 32)):
        for slip in (False, True):
            world = make_gridworld(h, w, terminals=[0], goals=[0], rewards=np.array([[0, 1.0]]))
            if slip:
                world = slippery(world)
            env = Gridworld(world, n_envs=n, seed=5, device=dev)
            cls = DynaQ if kind == 'dynaq' else QAgent
            ag = cls(env.observation_space, env.action_space, EpsilonGreedy(0.1), learning_rate=0.9, gamma=0.99)
            t = []
            for rep in range(3):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ag.train(env, 8, 64, batch)
                e1.record()
                torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1))
            steps = float(ag.steps_total) if hasattr(ag, 'steps_total') else n * 8 * 64 * 3
            print('%dx%d %-5s batch %2d %-13s: %.1f ms per train(8 trials x 64 steps) call' % (
                h, w, kind, batch, 'slippery' if slip else 'deterministic', min(t)), flush=True)
            del env, ag
            import gc
            gc.collect()
            torch.cuda.empty_cache()
