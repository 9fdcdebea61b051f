"""Round 3: the reference's hexagonal topology demo (demo/topology/demo.py: hexagonal(10), QAgent,
batch_size 0) vectorised over 65 536 instances — which kernel runs it and how fast — beside the
four-action grid(10) on the lane-per-instance LDS kernel.  `python scripts/exp_hex_b0.py`"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402
from cobel_amd import _lib  # noqa: E402
from cobel_amd.agent import QAgent  # noqa: E402
from cobel_amd.interface import Topology  # noqa: E402
from cobel_amd.misc.topology_tools import grid, hexagonal  # noqa: E402
from cobel_amd.policy import EpsilonGreedy  # noqa: E402

dev = torch.device('cuda', 0)
n, steps, spt = 65536, 256, 50
for name, (nodes, starts) in (('hexagonal(10)', hexagonal(10, (0.0, 1.0))), ('grid(10)', grid(10, (0.0, 1.0))),
                              ('hexagonal(5)', hexagonal(5, (0.0, 1.0)))):
    env = Topology(nodes, starts, n_envs=n, seed=3, device=dev)
    ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1))
    ag.log_experiences = False
    ag._bind(env)
    ag._env_in(env)
    flags = _lib.F_LEARN | ag._policy_in(ag.policy, env, False)
    ag.monitors.reserve(4096, n, False)
    kind = ag.describe_launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
    for i in range(4):
        ev[i].record()
        ag._launch(env, ag.policy, flags, 0x7fffffff, spt, steps, 0)
    ev[4].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    print('%-14s %3d nodes, %d actions: kernel %d, %s ms per %d steps -> %.3e env-steps/s' % (
        name, len(nodes), int(env.action_space.n), kind['kernel'], ' '.join('%.2f' % m for m in ms), steps,
        n * steps / (min(ms) * 1e-3)), flush=True)
