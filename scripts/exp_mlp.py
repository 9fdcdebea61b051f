"""Scratch: launch time of cobel_dqn_replay alone (C5 shapes, gathered batches)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch, bench
from cobel_amd.network import TorchNetwork
dev = torch.device('cuda', 0)
n, B = 8192, 32
for dt_name in ('f64', 'f32'):
    dt = torch.float64 if dt_name == 'f64' else torch.float32
    proto = TorchNetwork(bench._mlp(6, 4, dt_name), optimizer_params={'lr': 1e-3})
    proto.set_device(dev)
    net = proto.replicate(n); tgt = net.clone()
    s = torch.rand((n, B, 6), device=dev, dtype=dt); ns = torch.rand((n, B, 6), device=dev, dtype=dt)
    a = torch.randint(0, 4, (n, B), device=dev); r = torch.rand((n, B), device=dev, dtype=dt)
    nt = torch.ones((n, B), device=dev, dtype=dt)
    for _ in (0,):
        for _ in range(2):
            assert net.dqn_replay_fused(tgt, s, a, r, ns, nt, 0.8, False, 0.01, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.dqn_replay_fused(tgt, s, a, r, ns, nt, 0.8, False, 0.01, None)
        torch.cuda.synchronize()
        print('%s: %.1f us per launch' % (dt_name, (time.perf_counter() - t0) / 10 * 1e6), flush=True)
