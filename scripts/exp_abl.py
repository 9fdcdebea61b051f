"""Scratch: time C3 with ablated builds of the library (COBEL_LIB selects the .so)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch
from cobel_amd import _lib
if os.environ.get('COBEL_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'cobel-rl_amd', 'lib', os.environ['COBEL_LIB'])
import bench
dev = torch.device('cuda', 0)
for n, B in [(65536, 50), (65536, 0), (1536, 50), (1536, 0)]:
    cfg = dict(bench.CONFIGS['C3'], instances=n, env_steps_per_launch=256, batch=B)
    env, agent = bench.build_agent('C3', cfg, n, 0, dev)
    r = bench.Runner(cfg, env, agent)
    r.launch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r.launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('%-22s n=%6d B=%2d: %.3f ms/launch  %.3e steps/s' % (os.environ.get('COBEL_LIB', 'default'), n, B, dt * 1e3, n * 256 / dt))
