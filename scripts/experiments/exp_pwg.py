"""Scratch (round 3): launch time of the persistent-workgroup Dyna-Q kernel on C3 for one mix of
LDS / global-memory waves (COBEL_DEBUG_PWG="nl,ng", read once per process) on trained agents.
`python scripts/experiments/exp_pwg.py [pretrain launches] [nopwg]`"""
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from cobel_amd import _lib  # noqa: E402
if os.environ.get('COBEL_LIB'):      # A/B builds of the library
    _lib.LIB_PATH = os.path.join(ROOT, 'cobel-rl_amd', 'lib', os.environ['COBEL_LIB'])
if os.environ.get('PWG_SIDE'):       # (occupancy experiments: the same mazes on another grid)
    _side = int(os.environ['PWG_SIDE'])
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze as _mk
    _orig = bench.make_worlds
    bench.make_worlds = lambda name: ([_mk(_side, _side, 1234 + k) for k in range(64)] if name == 'C3' else _orig(name))
dev = torch.device('cuda', 0)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = dict(bench.CONFIGS['C3'])
if os.environ.get('PWG_N'):          # a shard of the batch (what each GPU of a split runs)
    cfg['instances'] = int(os.environ['PWG_N'])
env, agent = bench.build_agent('C3', cfg, cfg['instances'], 0, dev)
if 'nopwg' in sys.argv:
    agent.extra_flags = _lib.F_NO_PWG
r = bench.Runner(cfg, env, agent)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(pre + 5)]
for k in range(pre + 4):
    ev[k].record()
    r.launch()
ev[pre + 4].record()
torch.cuda.synchronize()
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(pre + 4)]
what = r.describe()
print('n=%d lib=%s PWG=%s %s kernel %d: launches 2-5 %s | last 4 %s ms  -> %.3e steps/s' % (
    cfg['instances'], os.environ.get('COBEL_LIB', '-'), os.environ.get('COBEL_DEBUG_PWG', '-'), 'nopwg' if 'nopwg' in sys.argv else '', what['kernel'],
    ' '.join('%.2f' % m for m in ms[1:5]), ' '.join('%.2f' % m for m in ms[-4:]),
    cfg['instances'] * cfg['env_steps_per_launch'] / (min(ms[-4:]) * 1e-3)), flush=True)
