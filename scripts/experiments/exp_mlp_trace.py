"""Per-phase wall-clock stamps of k_dqn_replay / k_mlp_fit (COBEL_DEBUG_MLP_TRACE): thread 0 of
every workgroup stamps wall_clock64() (100 MHz) at the phase boundaries of its step.
    python scripts/experiments/exp_mlp_trace.py c5|dsr [f64|f32] [stream|lds] [reward|successor]"""
import json
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'c5'
dt = sys.argv[2] if len(sys.argv) > 2 else 'f64'
n = 8192
rows = n * 4 if what == 'dsr' else n
trace = torch.zeros((rows, 16), dtype=torch.int64, device='cuda')
os.environ['COBEL_DEBUG_MLP_TRACE'] = hex(trace.data_ptr())
form = sys.argv[3] if len(sys.argv) > 3 else 'stream'
os.environ['COBEL_DEBUG_DQN_KERNEL'] = form
if what == 'c5':
    bench.run_c5(torch.device('cuda', 0), dt, n=n, iters=33, warm=17)
else:
    bench.run_dyna_dsr(torch.device('cuda', 0), n, iters=17)
torch.cuda.synchronize()
tr = trace.cpu().double()
if what == 'dsr' and len(sys.argv) > 4:
    # (the reward networks' launch — n workgroups, the later one — overwrites rows 0 .. n - 1 of the
    #  successor networks' 4 n: `reward` / `successor` picks one launch's rows)
    tr = tr[:n] if sys.argv[4] == 'reward' else tr[n:]
names = ['start', 'target pass', 'inputs in LDS', 'forward + loss', 'output layer', 'second layer',
         'first layer', 'stores drained + extra rows in', 'extra rows out']
if form == 'lds' and what == 'c5':
    names = ['start', 'slots in LDS', 'target parameters + rows in LDS', 'target pass', 'online pass(es)',
             'targets + delta3', 'output layer', 'delta2', 'second layer', 'delta1', 'first layer',
             'Q of the next observation']
last = len(names) - 1
valid = tr[:, 0] > 0
tr = tr[valid]
out = {'workgroups': int(valid.sum()), 'tick_ns': 10}
prev = tr[:, 0]
for k in range(1, last + 1):
    ok = tr[:, k] > 0
    if not bool(ok.any()):
        continue
    d = (tr[:, k] - prev)[ok] * 0.01
    out[names[k]] = {'mean_us': round(float(d.mean()), 2), 'p10': round(float(d.quantile(0.1)), 2),
                     'p90': round(float(d.quantile(0.9)), 2)}
    prev = torch.where(ok, tr[:, k], prev)
tot = (tr[:, last] - tr[:, 0]) * 0.01
out['whole workgroup'] = {'mean_us': round(float(tot.mean()), 2)}
out['launch_span_us'] = round(float(tr[:, last].max() - tr[:, 0].min()) * 0.01, 1)
print(json.dumps(out, indent=1))
