"""cobel_mlp_forward / cobel_mlp_fit (csrc/mlp_fit.hip) — TorchNetwork.predict_on_batch /
train_on_batch for stacks of Linear(D, 64)-ReLU-Linear(64, 64)-ReLU-Linear(64, O) networks — against
PyTorch on the same stacked parameters (float64 to round-off, float32 to its own), and the
five-launch Dyna-DSR step built from them against the PyTorch loop it replaces."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch


def _stack(torch, n, D, O, dtype, seed):
    from collections import OrderedDict
    from cobel_amd.network import TorchNetwork
    torch.manual_seed(seed)
    net = torch.nn.Sequential(OrderedDict([
        ('dense_1', torch.nn.Linear(D, 64)), ('relu_1', torch.nn.ReLU()),
        ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
        ('output', torch.nn.Linear(64, O))])).to(dtype)
    proto = TorchNetwork(net, optimizer_params={'lr': 3e-3})
    proto.set_device(torch.device('cuda', 0))
    st = proto.replicate(n)
    with torch.no_grad():       # every instance its own network
        for p in st.params.values():
            p.add_(torch.randn_like(p) * 0.05)
    return st


def _ptrs(lib_mod, dst, tensors):
    for k in range(3):
        dst[k] = lib_mod.ptr(tensors[k])


@pytest.mark.parametrize('D,O,dtype_name', [(25, 25, 'f64'), (25, 1, 'f64'), (6, 4, 'f64'),
                                            (32, 32, 'f64'), (16, 16, 'f32'), (3, 7, 'f64')])
def test_mlp_forward_and_fit_match_pytorch(torch_cuda, D, O, dtype_name, f32_atol=2e-5):
    torch = torch_cuda
    from cobel_amd import _lib
    from cobel_amd.agent.dyna_dsr import DynaDSR
    dt = torch.float64 if dtype_name == 'f64' else torch.float32
    tol = dict(rtol=1e-9, atol=1e-12) if dtype_name == 'f64' else dict(rtol=2e-4, atol=f32_atol)
    n, rows = 12, 40
    gen = torch.Generator(device='cpu').manual_seed(D * 100 + O)
    table = torch.randn((rows, D), generator=gen, dtype=torch.float64).cuda()
    index = torch.randint(0, rows, (n // 4, 32), generator=gen, dtype=torch.int32).cuda()
    net, tgt = _stack(torch, n, D, O, dt, 1), _stack(torch, n, D, O, dt, 2)
    names = net._mlp3_names()
    assert names is not None
    x = table[index.to(torch.int64)].to(dt)                       # [n / 4, 32, D]
    x_all = x[:, None].expand(n // 4, 4, 32, D).reshape(n, 32, D).contiguous()

    # ---- forward: inputs by table rows shared by 4 instances, and as a dense block -----------
    P = DynaDSR._mlp_ptrs(net, names)
    out = torch.zeros((n, 32, O), dtype=dt, device='cuda')
    fwd = _lib.MLPForward()
    _ptrs(_lib, fwd.w, P['w']); _ptrs(_lib, fwd.b, P['b'])       # noqa: E702
    fwd.in_table, fwd.in_index, fwd.in_div = _lib.ptr(table), _lib.ptr(index), 4
    fwd.out, fwd.n, fwd.net_div, fwd.act_div = _lib.ptr(out), n, 1, 1
    fwd.n_inputs, fwd.n_outputs, fwd.is_float64 = D, O, int(dt == torch.float64)
    _lib.check(_lib.lib().cobel_mlp_forward(C.byref(fwd), None))
    ref = net.predict_on_device(x_all)
    assert torch.allclose(out, ref, **tol), float((out - ref).abs().max())
    dense = torch.zeros_like(out)
    fwd.in_table, fwd.in_index, fwd.in_dense, fwd.in_div = None, None, _lib.ptr(x_all), 1
    fwd.out = _lib.ptr(dense)
    _lib.check(_lib.lib().cobel_mlp_forward(C.byref(fwd), None))
    assert torch.equal(dense, out)
    # one network rating the rows of four instances (net_div)
    shared = torch.zeros_like(out)
    fwd.net_div, fwd.out = 4, _lib.ptr(shared)
    _lib.check(_lib.lib().cobel_mlp_forward(C.byref(fwd), None))
    for j in (0, 5, 11):
        inp = x_all.clone()
        inp[j // 4] = x_all[j]
        assert torch.allclose(shared[j], net.predict_on_device(inp)[j // 4], **tol)

    # ---- fit: three steps with masks, networks that sit a step out, blend, extra rows ---------
    steps = torch.zeros(n, dtype=torch.float64, device='cuda')
    y = torch.randn((n // 4, 32, O), generator=gen, dtype=torch.float64).cuda().to(dt).contiguous()
    ref_net, ref_tgt = net.clone(), tgt.clone()
    ref_net.fused_adam = True
    tau = 0.07
    ep_idx = torch.randint(0, rows, (n // 4,), generator=gen, dtype=torch.int32).cuda()
    ep_out = torch.zeros((n, 1, O), dtype=dt, device='cuda')
    Pn, Pt = DynaDSR._mlp_ptrs(net, names), DynaDSR._mlp_ptrs(tgt, names)
    DynaDSR._step_counts(net)
    fit = _lib.MLPFit()
    for dst, key in ((fit.w, 'w'), (fit.b, 'b'), (fit.m_w, 'mw'), (fit.m_b, 'mb'), (fit.v_w, 'vw'),
                     (fit.v_b, 'vb')):
        _ptrs(_lib, dst, Pn[key])
    _ptrs(_lib, fit.w_target, Pt['w']); _ptrs(_lib, fit.b_target, Pt['b'])     # noqa: E702
    g = net.optimizer.param_groups[0]
    fit.lr, fit.beta1, fit.beta2 = float(g['lr']), float(g['betas'][0]), float(g['betas'][1])
    fit.eps, fit.weight_decay, fit.tau = float(g['eps']), float(g['weight_decay']), tau
    fit.steps = _lib.ptr(net._steps)
    fit.in_table, fit.in_index, fit.in_div = _lib.ptr(table), _lib.ptr(index), 4
    fit.targets, fit.tgt_div = _lib.ptr(y), 4
    fit.ep_table, fit.ep_index, fit.ep_div = _lib.ptr(table), _lib.ptr(ep_idx), 4
    fit.ep_rows, fit.ep_out = 1, _lib.ptr(ep_out)
    fit.n, fit.n_inputs, fit.n_outputs, fit.is_float64 = n, D, O, int(dt == torch.float64)
    fit.act_div = 1
    for it in range(3):
        mask = torch.rand((n, 32), generator=gen) < 0.4
        mask[1] = False                    # a network without samples: blend only
        mask[2, :] = False
        mask[2, 7] = True                  # a single sample
        mask = mask.cuda()
        train = mask.any(dim=1)
        m8, t8 = mask.to(torch.uint8).contiguous(), train.to(torch.uint8).contiguous()
        fit.sample_mask, fit.train = _lib.ptr(m8), _lib.ptr(t8)
        _lib.check(_lib.lib().cobel_mlp_fit(C.byref(fit), None))
        yy = y[:, None].expand(n // 4, 4, 32, O).reshape(n, 32, O)
        ref_net.train_on_device(x_all, yy, train, mask)
        ref_tgt.blend_from(ref_net, tau, None)
        for k in net.params:
            assert torch.allclose(net.params[k], ref_net.params[k], **tol), (it, k)
            assert torch.allclose(tgt.params[k], ref_tgt.params[k], **tol), (it, k)
        ex = table[ep_idx.to(torch.int64)].to(dt)[:, None].expand(n // 4, 4, D).reshape(n, 1, D)
        assert torch.allclose(ep_out, ref_net.predict_on_device(ex.contiguous()), **tol)
    assert torch.equal(net._steps, ref_net._steps)
    assert float(net._steps[1]) == 0.0 and float(net._steps[2]) == 3.0
    # refused shapes
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.lib().cobel_mlp_query(33, 64, 64, 4, 32, 1, None))
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.lib().cobel_mlp_query(6, 64, 32, 4, 32, 1, None))


def test_dyna_dsr_fused_step_equals_torch_loop(torch_cuda):
    """DynaDSR.train through cobel_dqn_act + 2 x cobel_mlp_forward + 2 x cobel_mlp_fit against the
    PyTorch loop: identical model tables, stream counters, trial counts and monitors; the nine
    networks of every instance to float64 round-off (instances finish at different times; two
    train() calls; default switches and use_DR / follow-up / terminality respected)."""
    torch = torch_cuda
    import bench
    from cobel_amd.agent import DynaDSR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    world = make_gridworld(4, 5, terminals=[3], rewards=np.array([[3, 1.0]]), goals=[3],
                           invalid_transitions=[(6, 7), (7, 6)])
    for switches in (False, True):
        def run(fused):
            torch.manual_seed(5)
            env = Gridworld(world, n_envs=24, seed=77, instance_base=2)
            ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                         TorchNetwork(bench._mlp(20, 20)), TorchNetwork(bench._mlp(20, 1)), gamma=0.9)
            if switches:
                ag.use_DR, ag.use_follow_up_state, ag.ignore_terminality = True, True, False
                ag.target_update = 0.1
            ag.fused_loop = None if fused else False
            ag.use_graph = None if fused else False
            ag.train(env, 4, 12, 32)
            ag.train(env, 2, 12, 32)
            torch.cuda.synchronize()
            return ag, env
        (a, ea), (b, eb) = run(True), run(False)
        assert a.fused_steps > 0 and b.fused_steps == 0 and a.fused_graph_steps > 0
        assert torch.equal(a.M.rewards, b.M.rewards) and torch.equal(a.M.states, b.M.states)
        assert torch.equal(a.M.terminals, b.M.terminals) and torch.equal(a.M.counter, b.M.counter)
        assert torch.equal(a.policy.counter, b.policy.counter) and torch.equal(ea.env_ctr, eb.env_ctr)
        assert torch.equal(a.trial, b.trial) and int(a.trial.min()) == 6
        for k in ('lat_sum', 'lat_cnt', 'reward_sum'):
            assert torch.equal(getattr(a.monitors, k), getattr(b.monitors, k)), k
        for i in (0, 9, 23):
            for act in range(4):
                for tgt in (False, True):
                    for x, y in zip(a.get_weights(act, tgt, i), b.get_weights(act, tgt, i)):
                        assert np.allclose(x, y, rtol=1e-9, atol=1e-12), (i, act, tgt)
            for x, y in zip(a.get_reward_weights(i), b.get_reward_weights(i)):
                assert np.allclose(x, y, rtol=1e-9, atol=1e-12)
        assert torch.equal(a._online._steps, b._online._steps)
        # the user-facing single networks hold instance 0's trained weights afterwards
        for x, y in zip(a.models_online[2].get_weights(), a.get_weights(2, False, 0)):
            assert np.array_equal(x, y)


@pytest.mark.parametrize('dtype_name,switches', [('f64', False), ('f64', True), ('f32', False)])
def test_dsr_targets_kernel_equals_torch_expressions(torch_cuda, dtype_name, switches):
    """cobel_dsr_targets (the regression targets of DynaDSR.replay, agent/dyna_q.py:1079-1131, in
    one launch) against the elementwise torch kernels it replaces inside the fused loop: the same
    run with ``fused_targets`` on and off ends in the same networks BIT FOR BIT in float64 (same
    expressions, same operation order; float32: round-off), with default switches and with
    use_DR / follow-up state / terminality respected."""
    torch = torch_cuda
    import bench
    from cobel_amd.agent import DynaDSR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    world = make_gridworld(4, 5, terminals=[3], rewards=np.array([[3, 1.0]]), goals=[3],
                           invalid_transitions=[(6, 7), (7, 6)])

    def run(fused_targets):
        torch.manual_seed(11)
        env = Gridworld(world, n_envs=20, seed=5, instance_base=1)
        ag = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                     TorchNetwork(bench._mlp(20, 20, dtype_name)),
                     TorchNetwork(bench._mlp(20, 1, dtype_name)), gamma=0.9)
        if switches:
            ag.use_DR, ag.use_follow_up_state, ag.ignore_terminality = True, True, False
        ag.fused_targets = fused_targets
        ag.use_graph = False                     # (eager steps: both forms launch kernel by kernel)
        ag.train(env, 3, 10, 32)
        torch.cuda.synchronize()
        assert ag.fused_steps > 0
        return ag

    a, b = run(True), run(False)
    assert torch.equal(a.trial, b.trial) and torch.equal(a.M.counter, b.M.counter)
    for i in (0, 7, 19):
        for act in range(4):
            for tgt in (False, True):
                for x, y in zip(a.get_weights(act, tgt, i), b.get_weights(act, tgt, i)):
                    if dtype_name == 'f64' and not switches:
                        assert np.array_equal(x, y), (i, act, tgt)
                    else:      # (use_DR: torch's mean may add the four actions in another order)
                        assert np.allclose(x, y, rtol=1e-9 if dtype_name == 'f64' else 2e-4,
                                           atol=1e-12 if dtype_name == 'f64' else 1e-5), (i, act, tgt)
        for x, y in zip(a.get_reward_weights(i), b.get_reward_weights(i)):
            assert np.allclose(x, y, rtol=1e-9 if dtype_name == 'f64' else 2e-4,
                               atol=1e-12 if dtype_name == 'f64' else 1e-5)
