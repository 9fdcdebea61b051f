"""Scratch (round 3): what MORE resident instances per CU would buy the Dyna-Q kernel on TRAINED
agents.  24x24 mazes need 10 240 B of LDS (8 blocks of 1 280 B: 16 instances per CU, the register
limit); COBEL_DEBUG_LDS_PAD pads them down to 14 / 12 / 11 / 10 / 9 per CU — same work, different
occupancy."""
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from cobel_amd.misc.gridworld_tools import make_obstacle_maze  # noqa: E402
dev = torch.device('cuda', 0)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bench.make_worlds = lambda name: [make_obstacle_maze(side, side, 1234 + k) for k in range(64)]
cfg = dict(bench.CONFIGS['C3'])
env, agent = bench.build_agent('C3', cfg, cfg['instances'], 0, dev)
r = bench.Runner(cfg, env, agent)
for _ in range(pre):
    r.launch()
torch.cuda.synchronize()
base = side * side * 16 + 1024
for pad in (0, 1280, 2560, 3840, 5120, 6400, 0):
    os.environ['COBEL_DEBUG_LDS_PAD'] = str(pad)
    b0 = int(agent.batches_done.item())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for k in range(3):
        r.launch()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
    frac = (int(agent.batches_done.item()) - b0) / (3 * cfg['instances'] * cfg['env_steps_per_launch'])
    blocks = (base + pad + 1279) // 1280
    print('side %d pad %4d: %5d B = %2d blocks -> %2d per CU  %s ms/launch  evaluated %.3f  %.3e steps/s' % (
        side, pad, base + pad, blocks, 128 // blocks, ' '.join('%.2f' % m for m in ms), frac,
        cfg['instances'] * cfg['env_steps_per_launch'] / (min(ms) * 1e-3)), flush=True)
