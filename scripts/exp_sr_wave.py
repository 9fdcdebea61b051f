import sys, os, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/cobel-rl_amd')
import torch, numpy as np
import bench
dev = torch.device('cuda', 0)
for stream_rows in (True, False):
    for steps_per_launch in (128, 512):
        cfg = dict(bench.CONFIGS['C4'], env_steps_per_launch=steps_per_launch)
        env, ag = bench.build_agent('C4', cfg, cfg['instances'], 0, dev)
        ag.stream_rows = stream_rows
        r = bench.Runner(cfg, env, ag)
        r.launch(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        for k in range(4):
            r.launch(); ev[k+1].record()
        torch.cuda.synchronize()
        ms = [ev[k].elapsed_time(ev[k+1]) for k in range(4)]
        n = cfg['instances']
        print(json.dumps({'stream_rows': stream_rows, 'steps_per_launch': steps_per_launch, 'ms': ms,
              'steps_per_s': n*steps_per_launch/(np.mean(ms)*1e-3), 'traffic': ag.traffic.cpu().tolist()}), flush=True)
        del env, ag, r
        torch.cuda.empty_cache()
