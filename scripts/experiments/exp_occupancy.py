"""Scratch: what one or two more resident instances per CU buy the Dyna-Q kernel (k_tab_wpi MIDX).
Worlds of 30x30 / 28x28 states need 16 / 14 KiB of LDS (10 / 11 instances per CU); COBEL_DEBUG_LDS_PAD
pads them back to the 17 KiB (9 per CU) of a 32x32 world — same work, different occupancy."""
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch, bench
from cobel_amd.misc.gridworld_tools import make_obstacle_maze
dev = torch.device('cuda', 0)
for side, pads in [(32, [0]), (30, [0, 1024, 2048]), (28, [0, 1024, 2048, 3072])]:
    bench.make_worlds = lambda name, side=side: [make_obstacle_maze(side, side, 1234 + k) for k in range(64)]
    for pad in pads:
        os.environ['COBEL_DEBUG_LDS_PAD'] = str(pad)
        cfg = dict(bench.CONFIGS['C3'])
        env, agent = bench.build_agent('C3', cfg, cfg['instances'], 0, dev)
        r = bench.Runner(cfg, env, agent)
        r.launch(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        lds = side * side * 16 + 1024 + pad
        kib = (lds + 1023) // 1024
        print('side %d pad %4d: LDS %5d B (%d KiB -> %d per CU)  %.3f ms/launch  %.3e steps/s' % (
            side, pad, lds, kib, 160 // kib, dt * 1e3, cfg['instances'] * cfg['env_steps_per_launch'] / dt), flush=True)
        del r, env, agent
