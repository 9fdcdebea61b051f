"""Scratch: per-phase cycle shares of k_tab_wpi from the COBEL_STAMPS diagnostic build."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
import torch, numpy as np
from cobel_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'cobel-rl_amd', 'lib', os.environ.get('COBEL_LIB', 'libcobel_hip_stamps.so'))
import bench
dev = torch.device('cuda', 0)
for n, B in [(65536, 50)]:
    cfg = dict(bench.CONFIGS['C3'], instances=n, env_steps_per_launch=256, batch=B)
    env, agent = bench.build_agent('C3', cfg, n, 0, dev)
    r = bench.Runner(cfg, env, agent)
    r.launch(); torch.cuda.synchronize()
    def patched(interface, pol, flags, tt, steps, budget, batch):
        mon = agent.monitors
        run = _lib.TabRun()
        run.q = _lib.ptr(agent._q); run.inst = _lib.ptr(agent.inst); run.model = _lib.ptr(agent.M.table); run.model_index = _lib.ptr(agent.M.index)
        run.lat_sum, run.lat_cnt, run.reward_sum = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt')), _lib.ptr(mon.raw('reward_sum')); run.mon_stripes = mon.stripes
        run.steps_done = _lib.ptr(mon.steps_done); run.last_exp = _lib.ptr(agent._last_exp)
        run.n, run.trial_cap, run.instance_base = agent.n_envs, mon.cap, interface.instance_base
        run.agent, run.flags, run.trials_target, run.steps_per_trial, run.step_budget, run.batch = 1, flags, tt, steps, budget, batch
        run.alpha, run.gamma, run.epsilon, run.model_lr, run.seed = 0.99, 0.99, 0.1, 0.9, interface.seed
        agent.inst[:, _lib.I_CTR_MEMORY] = agent.M.counter
        _lib.check(_lib.lib().cobel_tab_run(interface.handle.ptr, C.byref(run), None))
    agent._launch = patched
    r.launch(); torch.cuda.synchronize()
    st = agent._last_exp.cpu().numpy().astype(np.float64) * 16 / 256   # s_memtime ticks per step
    names = ['select after LDS batch', 'store+TD (after prefetch issue)', 'refresh_draws', 'readlane draws + LDS batch reads', 'env readlanes', 'prefetch issue'] if os.environ.get('FINE') else ['draws+select', 'env+store+TD', 'plan: M16+hash', 'plan: candidates', 'plan: rounds', 'bookkeeping+prefetch+loop']
    tot = st.sum(axis=1).mean()
    print('n=%d B=%d: %.0f ticks per step per wave' % (n, B, tot))
    for k in range(6):
        print('   %-28s %7.0f ticks  %5.1f %%' % (names[k], st[:, k].mean(), 100 * st[:, k].mean() / tot))
