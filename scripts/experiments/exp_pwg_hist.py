"""How many planning lanes of a batch change their cell (a -DCOBEL_PWG_HIST build of
tabular_pwg.hip counts the first round of every LDS-wave batch into the scratch area):
COBEL_LIB=<that build> python scripts/experiments/exp_pwg_hist.py [pretrain launches]"""
import os
os.environ.setdefault('COBEL_DEBUG', '1')
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
dev = torch.device('cuda', 0)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = dict(bench.CONFIGS['C3'])
env, agent = bench.build_agent('C3', cfg, cfg['instances'], 0, dev)
r = bench.Runner(cfg, env, agent)
for k in range(pre):
    r.launch()
torch.cuda.synchronize()
agent._scratch[256:400].zero_()
r.launch()
torch.cuda.synchronize()
h = agent._scratch[256:256 + 66].cpu().long()
tot = int(h[:65].sum())
print('batches (LDS waves) %d, later rounds %d (%.3f per batch)' % (tot, int(h[65]), float(h[65]) / max(tot, 1)))
cum = 0
for k in range(65):
    if int(h[k]):
        cum += int(h[k])
        print('changed lanes %2d: %6.3f %%   (cumulative %6.2f %%)' % (k, 100.0 * int(h[k]) / tot, 100.0 * cum / tot))
