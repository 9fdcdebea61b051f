"""C6 (SFMA) launch time and reactivations per launch over a long run: is the growth of the launch
time over the bench window work (more trials end per launch as the agents learn) or slowdown?"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS['C6'])
env, ag = bench.build_agent('C6', cfg, cfg['instances'], 0, dev)
r = bench.Runner(cfg, env, ag)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
reps = []
ev[0].record()
for k in range(n):
    r.launch()
    ev[k + 1].record()
    reps.append(ag.replays_done.clone())
torch.cuda.synchronize()
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
rp = [int(x.item()) for x in reps]
per = [rp[0]] + [rp[k] - rp[k - 1] for k in range(1, n)]
trials = ag.inst[:, 2].double().mean().item()
for k in range(0, n, 4):
    print(json.dumps({'launch': k, 'ms': round(ms[k], 3), 'reactivations': per[k],
                      'ns_per_reactivation_equiv': round(ms[k] * 1e6 / max(per[k], 1), 3)}))
print(json.dumps({'mean_trials_per_instance': trials}))
