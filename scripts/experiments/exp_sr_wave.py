"""C4 with the sparse-reward SR kernel: launch times for a sweep of resident wavefronts per CU
(LDS padding limits them), against the row-streaming kernel.  `python scripts/experiments/exp_sr_wave.py`"""
import gc
import json
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)


def run(stream_rows, steps_per_launch, pad, launches=4):
    if pad:
        os.environ['COBEL_DEBUG_LDS_PAD'] = str(pad)
    else:
        os.environ.pop('COBEL_DEBUG_LDS_PAD', None)
    cfg = dict(bench.CONFIGS['C4'], env_steps_per_launch=steps_per_launch)
    env, ag = bench.build_agent('C4', cfg, cfg['instances'], 0, dev)
    ag.stream_rows = stream_rows
    r = bench.Runner(cfg, env, ag)
    r.launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for k in range(launches):
        r.launch()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]
    n = cfg['instances']
    print(json.dumps({'stream_rows': stream_rows, 'steps_per_launch': steps_per_launch, 'lds_pad': pad,
                      'ms': [round(x, 3) for x in ms],
                      'steps_per_s': n * steps_per_launch / (np.mean(ms) * 1e-3),
                      'traffic': ag.traffic.cpu().tolist()}), flush=True)
    del env, ag, r
    gc.collect()
    torch.cuda.empty_cache()


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'sweep'
    if which == 'sweep':
        run(True, 128, 0)
        for wg_per_cu in (24, 12, 8, 4):
            pad = 0 if wg_per_cu == 24 else (160 * 1024 // wg_per_cu) // 1280 * 1280 - 384
            run(False, 128, pad)
        run(False, 512, 0)
    else:
        run(False, 128, 0)
