"""Where C2's launch time goes (k_tab_lpi, one wave per SIMD): the same 65 536 instances with trial
ends made rare (steps_per_trial huge or 200) and as the bench runs them.

    python scripts/experiments/exp_c2_parts.py        (on a GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cobel-rl_amd')]
import torch  # noqa: E402

import bench  # noqa: E402


def timed(run, k=12):
    ms = []
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run.launch()
        e1.record()
        torch.cuda.synchronize()
        ms.append(round(e0.elapsed_time(e1), 4))
    return ms


def main():
    dev = torch.device('cuda', 0)
    for name, spt in (('bench (50 steps per trial)', 50), ('no trial ends', 10 ** 7), ('200 steps per trial', 200)):
        cfg = dict(bench.CONFIGS['C2'])
        cfg['steps_per_trial'] = spt
        env, agent = bench.build_agent('C2', cfg, cfg['instances'], 0, dev)
        run = bench.Runner(cfg, env, agent)
        print(name, timed(run), flush=True)


if __name__ == '__main__':
    main()
