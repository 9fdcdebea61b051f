"""Scratch timing sweep of the tabular kernel (instances x batch) — not part of the product."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch, bench
dev = torch.device('cuda', 0)
for cfgname, n, B, budget in [('C3', 2560, 50, 256), ('C3', 65536, 50, 256), ('C3', 65536, 0, 256),
                              ('C3', 2560, 0, 256), ('C3', 65536, 8, 256), ('C3', 65536, 25, 256),
                              ('C2', 65536, 0, 1024), ('C2', 2304 * 4, 0, 1024)]:
    cfg = dict(bench.CONFIGS[cfgname], instances=n, env_steps_per_launch=budget, batch=B)
    env, agent = bench.build_agent(cfgname, cfg, n, 0, dev)
    r = bench.Runner(cfg, env, agent)
    r.launch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r.launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('%s n=%6d B=%2d: %.3f ms/launch  %.3e steps/s  %.1f ns per wave-step (assuming %d resident)' % (
        cfgname, n, B, dt * 1e3, n * budget / dt, dt / budget / max(1, n / 2560) * 1e9, 2560))
