"""C3 launch times (young agents / after 30 launches) with alternative builds of tabular.hip:
python scripts/exp_flag_variants.py cobel-rl_amd/lib/variants/libv1.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, os, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "cobel-rl_amd"))
from cobel_amd import _lib
if sys.argv[1] != "default": _lib.LIB_PATH = sys.argv[1]
import torch, bench
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["C3"])
env, agent = bench.build_agent("C3", cfg, cfg["instances"], 0, dev)
r = bench.Runner(cfg, env, agent)
r.launch(); torch.cuda.synchronize()
ms = []
for k in range(34):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.launch(); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "early": round(sum(ms[:4]) / 4, 3), "late": round(sum(ms[-4:]) / 4, 3)}))
''' % (ROOT, ROOT)
for lib in ['default'] + sys.argv[1:]:
    out = subprocess.run([sys.executable, '-c', code, lib], capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
