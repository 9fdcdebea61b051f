"""Scratch: lane-per-instance kernel with and without the per-trial monitor atomics."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch, bench
dev = torch.device('cuda', 0)
for mon in ('all', 'none', 'trace'):
    for n in (65536, 262144):
        cfg = dict(bench.CONFIGS['C2'], instances=n, env_steps_per_launch=1024)
        env, agent = bench.build_agent('C2', cfg, n, 0, dev)
        r = bench.Runner(cfg, env, agent)
        m = agent.monitors
        if mon != 'all':
            m.lat_sum = m.lat_cnt = m.reward_sum = None
        if mon == 'trace':
            m.lat_trace = torch.full((n, m.cap), -1, dtype=torch.int32, device=dev)
        r.launch(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print('monitors=%-5s n=%7d: %.3f ms/launch  %.3e steps/s' % (mon, n, dt * 1e3, n * 1024 / dt))
