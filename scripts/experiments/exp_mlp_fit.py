"""cobel_mlp_fit on 32 768 float64 25-64-64-25 networks (the successor networks of 8 192 Dyna-DSR
agents): launch time of the full step and of variants that leave parts out."""
import ctypes as C
import json
import os
os.environ.setdefault('COBEL_DEBUG', '1')   # (master switch of the library's COBEL_DEBUG_* experiment variables)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from cobel_amd import _lib  # noqa: E402
from cobel_amd.agent.dyna_dsr import DynaDSR  # noqa: E402
from test_gpu_mlp import _ptrs, _stack  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
D = O = int(sys.argv[2]) if len(sys.argv) > 2 else 25    # (6: 78 KB of LDS, two workgroups per CU)
dt = torch.float64
net, tgt = _stack(torch, n, D, O, dt, 1), _stack(torch, n, D, O, dt, 2)
names = net._mlp3_names()
table = torch.eye(D, dtype=torch.float64, device='cuda')
index = torch.randint(0, D, (n // 4, 32), dtype=torch.int32, device='cuda')
y = torch.randn((n // 4, 32, O), dtype=dt, device='cuda')
mask = (torch.rand((n, 32), device='cuda') < 0.25).to(torch.uint8)
train = torch.ones(n, dtype=torch.uint8, device='cuda')
ep_idx = torch.randint(0, D, (n // 4,), dtype=torch.int32, device='cuda')
ep_out = torch.zeros((n, 1, O), dtype=dt, device='cuda')
Pn, Pt = DynaDSR._mlp_ptrs(net, names), DynaDSR._mlp_ptrs(tgt, names)
DynaDSR._step_counts(net)


def make(tau=0.01, ep=True, train_t=train, stage=0):
    fit = _lib.MLPFit()
    fit.debug_stage = stage
    for dst, key in ((fit.w, 'w'), (fit.b, 'b'), (fit.m_w, 'mw'), (fit.m_b, 'mb'), (fit.v_w, 'vw'),
                     (fit.v_b, 'vb')):
        _ptrs(_lib, dst, Pn[key])
    _ptrs(_lib, fit.w_target, Pt['w']); _ptrs(_lib, fit.b_target, Pt['b'])      # noqa: E702
    fit.lr, fit.beta1, fit.beta2, fit.eps, fit.weight_decay, fit.tau = 1e-3, 0.9, 0.999, 1e-8, 0.0, tau
    fit.steps = _lib.ptr(net._steps)
    fit.in_table, fit.in_index, fit.in_div = _lib.ptr(table), _lib.ptr(index), 4
    fit.targets, fit.tgt_div, fit.sample_mask = _lib.ptr(y), 4, _lib.ptr(mask)
    fit.train = _lib.ptr(train_t)
    if ep:
        fit.ep_table, fit.ep_index, fit.ep_div = _lib.ptr(table), _lib.ptr(ep_idx), 4
        fit.ep_rows, fit.ep_out = 1, _lib.ptr(ep_out)
    else:
        fit.ep_div = 1
    fit.n, fit.n_inputs, fit.n_outputs, fit.is_float64, fit.act_div = n, D, O, 1, 1
    return fit


def timeit(fit, reps=5):
    _lib.check(_lib.lib().cobel_mlp_fit(C.byref(fit), None))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        _lib.check(_lib.lib().cobel_mlp_fit(C.byref(fit), None))
        ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))


none = torch.zeros(n, dtype=torch.uint8, device='cuda')
print(json.dumps({'full': timeit(make()), 'no_blend': timeit(make(tau=0.0)),
                  'no_epilogue': timeit(make(ep=False)),
                  'no_training (stage + blend + epilogue)': timeit(make(train_t=none)),
                  'no_training, no_blend (stage + epilogue)': timeit(make(tau=0.0, train_t=none)),
                  'until_forward': timeit(make(stage=1)), 'until_output_layer': timeit(make(stage=2)),
                  'until_second_layer_grad': timeit(make(stage=3)), 'until_delta1': timeit(make(stage=4)),
                  'bytes_full_GB': n * 8 * (64 * D + 64 + 4160 + 64 * O + O) * 8 / 1e9,
                  'inputs_outputs': D, 'lds_pad': os.environ.get('COBEL_DEBUG_LDS_PAD')}))
