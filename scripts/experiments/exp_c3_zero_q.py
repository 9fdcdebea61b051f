"""C3: share of instances whose Q table is still all zero (no reward seen yet) after k launches of
512 steps — what a 'nothing to plan yet' shortcut could skip."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch
import bench
dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS['C3'], instances=16384)
env, agent = bench.build_agent('C3', cfg, 16384, 0, dev)
r = bench.Runner(cfg, env, agent)
out = []
for k in range(1, 13):
    r.launch(); torch.cuda.synchronize()
    zero = float((agent._q.abs().amax(dim=(1, 2)) == 0).float().mean())
    nz = float((agent._q != 0).float().mean())
    out.append((k * cfg['env_steps_per_launch'], round(zero, 4), round(nz, 5)))
print(json.dumps({'steps, share of instances with Q == 0, share of nonzero Q cells': out}))
