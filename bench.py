#!/usr/bin/env python3
"""Headline benchmark: gridworld env-steps/s over tens of thousands of parallel agent-env instances.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C4] [--instances n]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = ONE launch of the fused agent kernel that advances every instance on the GPU
by `env_steps_per_launch` env steps (select -> env.step -> learn -> B planning updates, with
per-instance auto-reset).  W untimed warm-up launches — on C3 followed by untimed pre-training until
the kernel reports that >= 95 % of the planning batches it draws are evaluated, so that the timed
window is the full-work state of trained agents (`pretraining`; the rate of the first launches is
kept as `young_agents`) —, then exactly K launches timed between barrier + torch.cuda.synchronize()
pairs; the max over ranks is used and rank 0 prints one JSON line.  `python bench.py --gpus N` with
no launcher around it starts its N ranks itself (torch.distributed.run) and relays that line.  Instances are independent: with --gpus N the configuration's instances are SPLIT evenly over
the ranks (BASELINE config 3: "batch split 1 -> 8 MI355X"; contiguous ranges of global instance
ids, cobel_amd.misc.sharding) — strong scaling — or, with --weak, every rank runs the full
instance count.  Global instance ids key the random streams and the world of an instance, so the
results do not depend on N.  The only collective is the reduction of the monitor buffers after
the last launch (one all-gather, inside the timed region).

Workloads, all synthetic, tables zero-initialised as the reference does:
  C3 (default, the configuration the metric is quoted on): 65 536 instances over 64 distinct
     32x32 obstacle mazes, Dyna-Q alpha .99 gamma .99 eps .1 model-lr .9, B = 50 planning
     updates per step, 200 steps per trial.
  C2: 65 536 x 5x5 open field, Q-learning alpha .9 gamma .8 eps .1, no replay, 50 steps/trial.
  C4: 16 384 x 32x32 open field, SR alpha .1 gamma .99 eps .1, 200 steps/trial (64 GiB of SR).
  C6: 65 536 x the 5x5 walled world of demo/gridworld/demo_sfma.py, SFMA with the DR metric in
      reverse mode, 32 reactivations per trial (reported beside the headline).

`roofline.achieved` = algorithmic bytes per env step (C2 67 B; C3 by the work done:
78 B per env step + 31 B per planning update of the batches the kernel evaluated; C4: the
SR rows and value elements the sparse-reward kernel actually asks for, counted by the kernel —
DESIGN.md §4.2; §8d's eight-row figure of 32 817 B is reported beside it as `sec8d_*`) x env steps
per launch / mean launch duration, the latter measured with HIP events on the launch stream.
`roofline.traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes and
`roofline.frac_measured` = traffic / launch duration / peak — the PHYSICAL HBM utilisation;
`roofline.limiter` names what bounds the kernel.  `cpu_baseline` = the NumPy
restatement of the reference's single-instance loop (oracle/ref_loop.py), one core, on a bounded
sample of the same workload (rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'cobel-rl_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    'C2': dict(instances=65536, env_steps_per_launch=1024, steps_per_trial=50, batch=0,
               bytes_per_step=67, agent='q', limiter='issue',
               desc='65536 x 5x5 open gridworld, tabular Q-learning (alpha .9, gamma .8, eps .1, '
                    'no replay), 50 steps/trial'),
    # (512 steps per launch: loading and storing the 16 KiB Q table of an instance once per launch
    #  is 2 GB of traffic for 65 536 instances; at 256 steps per launch C3 runs 3 % slower)
    'C3': dict(instances=65536, env_steps_per_launch=512, steps_per_trial=200, batch=50,
               bytes_per_step=1628, agent='dynaq', limiter='issue', train_until=0.95,
               desc='65536 instances over 64 32x32 obstacle mazes (p_wall .20, seeds 1234..1297), '
                    'Dyna-Q (alpha .99, gamma .99, eps .1, model lr .9), 50 planning updates/step, '
                    '200 steps/trial'),
    'C4': dict(instances=16384, env_steps_per_launch=128, steps_per_trial=200, batch=0,
               bytes_per_step=32817, agent='sr', limiter='hbm',
               desc='16384 x 32x32 open gridworld, successor representation (alpha .1, gamma .99, '
                    'eps .1), 200 steps/trial'),
    # SURVEY.md §8f rank 2 (not a BASELINE config): the reference's SFMA demo, vectorised
    'C6': dict(instances=65536, env_steps_per_launch=200, steps_per_trial=50, batch=32,
               bytes_per_step=98, bytes_per_reactivation=20 * 100 + 16 * 25 + 32, agent='sfma',
               limiter='issue', min_warmup=40,
               desc='65536 x 5x5 walled gridworld of demo/gridworld/demo_sfma.py, SFMA (alpha .99, '
                    'gamma .99, eps .1, DR metric gamma .9, reverse mode, action mask on), 32 '
                    'reactivations per trial, 50 steps/trial'),
}
SEED = 0xC0BE1


def run_c5(device, dtype_name, n=8192, iters=256, warm=8, graph=None):
    """C5: 8192 linear_track(10, 2) Topology envs, DQN 6-64-64-4 (gamma .8, eps .3, Adam 1e-3, MSE,
    tau .01, batch 32, 100 steps/trial), one network and one replay ring per instance.  graph =
    None: the two-kernel loop (cobel_dqn_act + cobel_dqn_replay); True: the PyTorch loop with one
    step captured as a HIP graph (forward passes by vmap, the replay step still cobel_dqn_replay)."""
    from collections import OrderedDict
    from cobel_amd.agent import DQN
    from cobel_amd.interface import Topology
    from cobel_amd.memory import DQNMemory
    from cobel_amd.misc.topology_tools import linear_track
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    torch.manual_seed(0)
    net = torch.nn.Sequential(OrderedDict([
        ('dense_1', torch.nn.Linear(6, 64)), ('relu_1', torch.nn.ReLU()),
        ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
        ('output', torch.nn.Linear(64, 4))]))
    net = net.double() if dtype_name == 'f64' else net.float()
    nodes, starts = linear_track(10, 2, 1., 20., 'right')
    env = Topology(nodes, starts, n_envs=n, seed=SEED, device=device)
    agent = DQN(env.observation_space, env.action_space, EpsilonGreedy(0.3),
                TorchNetwork(net, optimizer_params={'lr': 1e-3}), gamma=0.8,
                memory=DQNMemory(capacity=256))
    agent.use_graph = graph
    # warm-up: a short run, then one of the timed length (rings at their final size, and — after
    # the host-only CPU-baseline legs before this one — the GPU back at its working clocks: the
    # first 0.3 s after an idle phase run 1.3-1.6x slower)
    agent._run(env, 4096, 100, 32, True, budget=warm)
    agent._run(env, 4096, 100, 32, True, budget=iters)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    agent._run(env, 4096, 100, 32, True, budget=iters)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    how = ('two launches per lockstep step (cobel_dqn_act + cobel_dqn_replay), 16 steps per HIP graph'
           if agent.fused_steps
           else 'PyTorch-ROCm loop' + (', one step captured as a HIP graph and replayed' if graph
                                       else ''))
    # cobel_dqn_replay moves 8 streams over an instance's parameters (online, two Adam moments,
    # target: read + write each) = the algorithmic HBM bytes of a step
    n_params = sum(p.numel() for p in agent.model_online.model.parameters())
    bytes_per_step = 8 * n_params * (8 if dtype_name == 'f64' else 4)
    return {'value': n * iters / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / iters * 1e3,
            'dtype': dtype_name,
            'config': {'workload': 'C5: %d x linear_track(10,2) Topology, DQN 6-64-64-4 %s, one '
                                   'network + replay ring per instance, batch 32, %s'
                                   % (n, dtype_name, how), 'instances_per_gpu': n,
                       'lockstep_iterations': iters},
            'roofline': {'bound': 'hbm', 'achieved': bytes_per_step * n * iters / dt / 1e9,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': bytes_per_step * n * iters / dt / 1e9 / HBM_PEAK_GBS,
                         'traffic': _c5_traffic(dtype_name, n), 'kernel': 'k_dqn_replay_lds',
                         'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/)',
                         'algorithmic_bytes_per_launch': bytes_per_step * n,
                         'algorithmic_bytes_per_env_step': bytes_per_step}}


def _c5_traffic(dtype_name, n):
    """HBM bytes per launch of k_dqn_replay_lds from the committed PMC passes (scripts/pmc_c5.py),
    for the instance count they were collected with."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files or n != 8192:
        return None
    with open(files[-1]) as fh:
        entry = json.load(fh).get('C5_' + dtype_name)
    return entry['hbm_bytes_per_launch'] if entry else None


def _mlp(n_in, n_out, dtype_name='f64'):
    from collections import OrderedDict
    net = torch.nn.Sequential(OrderedDict([
        ('flatten', torch.nn.Flatten()),
        ('dense_1', torch.nn.Linear(n_in, 64)), ('relu_1', torch.nn.ReLU()),
        ('dense_2', torch.nn.Linear(64, 64)), ('relu_2', torch.nn.ReLU()),
        ('output', torch.nn.Linear(64, n_out))]))
    return net.double() if dtype_name == 'f64' else net.float()


def _timed_network_agent(agent, env, n, iters, device, warm=4):
    """Steady-state rate of a network agent's lockstep loop.  A run launches its first step directly,
    records ONE HIP graph of a chunk of steps (16 for the DQN loop, 8 for Dyna-DSR's 29 kernels per
    step) and replays it; `iters` = 1 + a whole number of chunks.  Recording the graph is a one-off
    of the run (~0.1 s for Dyna-DSR, with the GPU idle) that a training run of thousands of steps
    does not notice but a benchmark of 200 does: two runs, of `iters` and 2 * iters - 1 steps, are
    timed and the rate is taken from their DIFFERENCE (the marginal cost of a step); the whole-run
    figure of the longer one and the one-off are reported next to it."""
    def run(k):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        agent._run(env, 4096, 50, 32, True, budget=k)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    agent._run(env, 4096, 50, 32, True, budget=warm)
    run(iters)                       # (untimed: working clocks, rings at their final size)
    long_iters = 2 * iters - 1
    t_short, t_long = run(iters), run(long_iters)
    per_step = (t_long - t_short) / (long_iters - iters)
    return {'value': n / per_step, 'unit': 'env-steps/s', 'ms_per_step': per_step * 1e3,
            'dtype': 'f64', 'timing': {
                'method': 'marginal step: (t[%d steps] - t[%d steps]) / %d'
                          % (long_iters, iters, long_iters - iters),
                'whole_run_ms_per_step': t_long / long_iters * 1e3,
                'one_off_ms_per_run': max(0.0, (t_short - per_step * iters) * 1e3)}}


def _hbm_roofline(bytes_per_step, value, kernel, limiter, note=None):
    gbs = bytes_per_step * value / 1e9
    r = {'bound': 'hbm', 'limiter': limiter, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': gbs / HBM_PEAK_GBS, 'traffic': None, 'kernel': kernel,
         'algorithmic_bytes_per_env_step': bytes_per_step}
    if note:
        r['note'] = note
    return r


def run_dyna_dqn(device, n=8192, iters=129):
    """SURVEY.md §8f rank 1 (demo/gridworld/demo_dyna_dqn.py: 5x5 open field, one-hot inputs, a
    25-64-64-4 float64 network per instance, gamma .8, model-sampled batches of 32)."""
    from cobel_amd.agent import DynaDQN
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    torch.manual_seed(0)
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=SEED, device=device)
    agent = DynaDQN(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                    TorchNetwork(_mlp(25, 4)), gamma=0.8)
    r = _timed_network_agent(agent, env, n, iters, device)
    fused = agent.fused_steps > 0
    r['config'] = {'workload': 'Dyna-DQN: %d x 5x5 open field, MLP 25-64-64-4 f64 per instance, '
                               'model-sampled batches of 32, %s' % (
                                   n, 'two launches per lockstep step: cobel_dqn_act (world-model '
                                   'mode) + cobel_dqn_replay' if fused else 'PyTorch-ROCm loop'),
                   'instances_per_gpu': n, 'lockstep_iterations': [iters, 2 * iters - 1]}
    # cobel_dqn_replay: 8 streams over the instance's parameters (online, two Adam moments, target:
    # read + write) + the batch: 32 sampled model records (8 B) and their one-hot rows are indices
    n_params = sum(p.numel() for p in agent.model_online.model.parameters())
    bytes_per_step = 8 * n_params * 8 + 32 * 8 + 78
    r['roofline'] = _hbm_roofline(bytes_per_step, r['value'], 'k_dqn_replay', 'latency',
                                  '8 parameter streams of 8 B x %d parameters + 32 model records + '
                                  'the online step (78 B); the streaming form of the step (weight '
                                  'operands from memory, 52 KB of LDS: two eight-wave workgroups per '
                                  'CU, DESIGN.md section 4.4)' % n_params)
    return r


def run_dyna_dsr(device, n=8192, iters=97):
    """SURVEY.md §8f rank 1 (demo/gridworld/demo_dyna_dsr.py): four online + four target successor
    networks 25-64-64-25 and one reward network per instance, float64."""
    from cobel_amd.agent import DynaDSR
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.network import TorchNetwork
    from cobel_amd.policy import EpsilonGreedy
    torch.manual_seed(0)
    env = Gridworld(make_open_field(5, 5, 0, 1), n_envs=n, seed=SEED, device=device)
    agent = DynaDSR(env.observation_space, env.action_space, EpsilonGreedy(0.1),
                    TorchNetwork(_mlp(25, 25)), TorchNetwork(_mlp(25, 1)), gamma=0.8)
    r = _timed_network_agent(agent, env, n, iters, device, warm=8)
    fused = agent.fused_steps > 0
    r['config'] = {'workload': 'Dyna-DSR: %d x 5x5 open field, four online + four target successor '
                               'networks 25-64-64-25 and one reward network f64 per instance, batches '
                               'of 32, %s' % (n, 'six launches per lockstep step (cobel_dqn_act, 2 x '
                                              'cobel_mlp_forward, cobel_dsr_targets, 2 x '
                                              'cobel_mlp_fit), 8 steps per HIP graph' if fused
                                              else 'PyTorch-ROCm loop, one step per HIP graph'),
                   'instances_per_gpu': n, 'lockstep_iterations': [iters, 2 * iters - 1]}
    # per instance and step: the four online successor networks move 8 streams over their
    # parameters (p, m, v, target: read + write), the reward network 6 (no target), and the
    # forward passes read the four target networks and the reward network once more
    p_sr = sum(p.numel() for p in agent.models_online[0].model.parameters())
    p_rw = sum(p.numel() for p in agent.model_reward.model.parameters())
    bytes_per_step = (4 * 8 * p_sr + 6 * p_rw + 4 * p_sr + p_rw) * 8
    r['roofline'] = _hbm_roofline(bytes_per_step, r['value'], 'k_mlp_fit' if fused else 'torch',
                                  'latency', 'DESIGN.md section 4.4b')
    if fused and n == 8192:   # the successor-network fit launch alone (scripts/pmc_fit.py, profiles/)
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
        if files:
            with open(files[-1]) as fh:
                r['roofline']['fit_launch_pmc'] = json.load(fh).get('dyna_dsr_fit')
    return r


def run_grid_search(device, runs=16):
    """learning_rate x gamma x epsilon, every combination x run one instance
    of ONE Dyna-Q launch (optimizer/grid_search.py:173-262 simulates them one at a time)."""
    import tempfile
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.optimizer import GridSearchOptimizer, spread_over_instances
    from cobel_amd.policy import EpsilonGreedy
    grid = {'learning_rate': list(np.linspace(0.1, 0.99, 16)), 'gamma': list(np.linspace(0.5, 0.99, 16)),
            'epsilon': [0.05, 0.1, 0.2, 0.3]}
    trials, steps, batch = 50, 50, 32
    world = make_open_field(5, 5, 0, 1)
    stats = {'env_steps': 0, 'batches': 0, 'kernel_ms': 0.0}

    def simulation_batch(task, combinations, nb_runs):
        arrays, which = spread_over_instances(combinations, nb_runs)
        env = Gridworld(world, n_envs=len(which), seed=SEED, device=device)
        ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(arrays['epsilon']),
                   learning_rate=arrays['learning_rate'], gamma=arrays['gamma'])
        ag.track_instances = True
        # HIP events right around the library call (train() also fills and uploads 16 384
        # parameter sets and reserves the monitors: host work during which the GPU idles)
        ag.launch_events = []          # (one pair of events per launch the call issues)
        ag.train(env, trials, steps, batch)
        lat = ag.monitors.lat_trace[:, :trials].double().mean(dim=1).cpu().numpy()
        stats['env_steps'] += ag.env_steps()
        stats['batches'] += int(ag.batches_done.item())
        stats['kernel_ms'] += sum(e0.elapsed_time(e1) for e0, e1 in ag.launch_events)
        return [list(lat[which == c]) for c in range(len(combinations))]

    # (the parameter-set instantiation of the kernel is used by this leg only: its code object is
    #  loaded by a small untimed launch, not inside the HIP events of the timed one)
    warm_env = Gridworld(world, n_envs=64, seed=SEED, device=device)
    warm = DynaQ(warm_env.observation_space, warm_env.action_space,
                 EpsilonGreedy(np.full(64, 0.1)), learning_rate=np.full(64, 0.5), gamma=np.full(64, 0.9))
    warm.train(warm_env, 2, 10, batch)
    torch.cuda.synchronize(device)
    del warm, warm_env
    with tempfile.TemporaryDirectory() as tmp:
        opt = GridSearchOptimizer(tmp + '/', grid, nb_runs=runs)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        opt.fit_vectorised(simulation_batch, {'demo': {}}, {'demo': [5.0] * runs},
                           lambda sim, data: float(np.mean(sim['demo'])))
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
    combos = 16 * 16 * 4
    # SURVEY 8d by work done: 78 B per env step + 31 B per planning update of evaluated batches
    alg = 78 * stats['env_steps'] + 31 * batch * stats['batches']
    roof = _hbm_roofline(alg / max(1, stats['env_steps']), stats['env_steps'] / (stats['kernel_ms'] * 1e-3),
                         'k_tab_wpi<DYNAQ, PSETS>', 'issue',
                         'over the launch itself (HIP events around DynaQ.train: %.1f ms of the %.2f s '
                         'the whole fit takes, the rest is the host enumerating combinations and '
                         'writing the result files as the reference does); 5x5 tables are LDS '
                         'resident' % (stats['kernel_ms'], dt))
    iss = issue_roofline('grid_search', stats['env_steps'] / (stats['kernel_ms'] * 1e-3),
                         stats['kernel_ms'], device, runs == 16)
    if iss is not None:
        roof['issue'] = iss
    roof = to_issue_bound(roof)
    return {'value': combos * runs / dt, 'unit': 'simulations/s', 'dtype': 'f32',
            'env_steps': stats['env_steps'],
            'env_steps_per_s': stats['env_steps'] / dt, 'seconds': dt,
            'env_steps_per_s_kernel': stats['env_steps'] / (stats['kernel_ms'] * 1e-3),
            'config': {'workload': 'GridSearchOptimizer.fit_vectorised: %d combinations x %d runs of '
                                   'Dyna-Q (5x5, 32 planning updates, %d trials x <= %d steps) as %d '
                                   'instances of one launch, files written as the reference does'
                                   % (combos, runs, trials, steps, combos * runs)},
            'roofline': roof}


def run_c1(device, with_cpu=True):
    """C1, the reference's own CPU-runnable case (demo/gridworld/demo_dyna_q.py:36-56): ONE Dyna-Q
    agent on the 5x5 open field, 500 training + 300 test trials of <= 50 steps, batch 32 — through
    the drop-in classes (whole launches, no per-step hooks), with the reference loop's port timed
    on the same shape beside it."""
    from cobel_amd.agent import DynaQ
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_open_field
    from cobel_amd.policy import EpsilonGreedy
    world = make_open_field(5, 5, 0, 1)
    world['starting_states'] = np.arange(1, 25)
    env = Gridworld(world, seed=SEED, device=device)
    agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1), learning_rate=0.99)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    agent.train(env, 500, 50, 32)
    agent.test(env, 300, 50)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    steps = agent.env_steps()
    r = {'value': steps / dt, 'unit': 'env-steps/s', 'seconds': dt, 'env_steps': steps, 'dtype': 'f32',
         'config': {'workload': 'C1: one Dyna-Q agent, 5x5 open field, 500 train + 300 test trials x <= '
                                '50 steps, batch 32 (demo/gridworld/demo_dyna_q.py) through the '
                                'drop-in classes', 'instances_per_gpu': 1},
         'roofline': {'bound': 'latency', 'achieved': None, 'peak': None, 'frac': None, 'unit': None,
                      'traffic': None,
                      'note': 'one instance is one wavefront: the reference plumbing case, not a '
                              'throughput configuration'}}
    if with_cpu:
        cfg = dict(instances=1, steps_per_trial=50, batch=32, agent='dynaq')
        r['cpu_baseline'] = cpu_baseline('C1', cfg, 4.0)
    return r


def run_general(device, which, n=65536, env_steps=64, launches=4):
    """What leaving the four-action / <= 62-update wavefront kernels costs:
    `hex_q` = QAgent with a replay batch of 32 on a six-action hexagonal Topology
    (misc/topology_tools.py:175-272) — k_tab_wqn since round 4, one wavefront per instance with the
    tables in LDS (k_tab_general, one lane per instance, before); `dynaq_b100` = Dyna-Q
    on C3's mazes with 100 planning updates per step (agent/dyna_q.py:319-330 has no limit) — the
    generic k_tab_wpi in two passes of <= 62 lanes; `wide_q` = QAgent with a replay batch of 32 and
    an action mask on a random graph of 256 nodes with twelve neighbours each (interface/topology.py:
    110-112 takes any count) — k_tab_wqn on rows of 16 since round 5 (the selection's float64 CDF
    worked out by the wave); `wide_q_lane` = the same run forced onto k_tab_general, one lane per
    instance with every table in L2: the functional path, timed so that its price is on record."""
    from cobel_amd import _lib
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld, Topology
    from cobel_amd.misc.topology_tools import hexagonal
    from cobel_amd.policy import EpsilonGreedy
    if which in ('hex_q', 'wide_q', 'wide_q_lane'):
        if which == 'hex_q':
            nodes, starts = hexagonal(16)
        else:
            rg = np.random.default_rng(12)
            nbr = rg.integers(0, 256, (256, 12))
            nodes = {str(i): {'id': str(i), 'pose': np.array([float(i % 16), float(i // 16), 0., 0., 0., 0.]),
                              'neighbors': [str(int(j)) for j in nbr[i]],
                              'reward': 1.0 if i == 255 else 0.0, 'terminal': i == 255}
                     for i in range(256)}
            starts = None
        env = Topology(nodes, starts, n_envs=n, seed=SEED, device=device)
        agent = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        batch, spt, A = 32, 100, int(env.action_space.n)
        if which != 'hex_q':
            agent.force_general = which == 'wide_q_lane'
            agent.mask_actions = True
            agent.action_mask = np.ones((256, 12), dtype=bool)
            agent.action_mask[:, 11] = False            # (one neighbour of every node closed)
        agent._bind(env)
        agent.reserve_replay(env_steps * (launches + 2))
        desc = ('QAgent (alpha .9, gamma .8, eps .1, replay batch 32 from the experience log%s) on a '
                '%s of %d nodes, %d actions' % (', action mask' if which != 'hex_q' else '',
                                                'hexagonal Topology' if which == 'hex_q' else 'random graph',
                                                len(nodes), A))
        # online step: state r/w 8 + Q[s,:] 4A + next 2 + reward 4 + terminal 1 + Q[ns,:] 4A +
        # Q[s,a] write 4 + counters 16, log append 8; one replayed update: record 8 + Q[ns,:] 4A +
        # Q[s,a] RMW 8
        b_step, b_upd = 8 + 4 * A + 2 + 4 + 1 + 4 * A + 4 + 16 + 8, 8 + 4 * A + 8
    else:
        env = Gridworld(make_worlds('C3'), n_envs=n, seed=SEED, device=device)
        agent = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1))
        batch, spt = 100, 200
        desc = 'Dyna-Q on the 64 32x32 mazes of C3 with 100 planning updates per step'
        b_step, b_upd = 78, 31
    cfg = dict(env_steps_per_launch=env_steps, steps_per_trial=spt, batch=batch)
    runner = Runner(cfg, env, agent)
    if which == 'dynaq_b100':
        # trained agents (nearly every planning batch is evaluated), as the headline: untimed
        # pre-training with the headline's own launches, B = 50
        pre = Runner(dict(CONFIGS['C3']), env, agent)
        for _ in range(48):
            pre.launch()
    for _ in range(2):
        runner.launch()
    torch.cuda.synchronize(device)
    b0 = int(agent.batches_done.item())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for k in range(launches):
        runner.launch()
        ev[k + 1].record()
    torch.cuda.synchronize(device)
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]
    batches = int(agent.batches_done.item()) - b0
    steps = n * env_steps * launches
    sec = sum(ms) * 1e-3
    what = runner.describe()
    passes = -(-batch // _lib.MAX_BATCH)
    names = {_lib.TAB_KERNEL_GENERAL: 'k_tab_general',
             _lib.TAB_KERNEL_WPI: 'k_tab_wpi (generic, %d passes)' % passes,
             _lib.TAB_KERNEL_WPI_INDEX: 'k_tab_wpi (digest in HBM, %d passes)' % passes,
             _lib.TAB_KERNEL_WQN: 'k_tab_wqn'}
    kernel = names.get(what['kernel'], 'kernel %d' % what['kernel'])
    alg = b_step * steps + b_upd * batch * batches
    traffic = None
    if which in ('hex_q', 'wide_q', 'wide_q_lane'):
        traffic = pmc_traffic_leg('general_' + which, kernel) if n == 65536 else None
    roof = _hbm_roofline(alg / steps, steps / sec, kernel,
                         'latency' if what['kernel'] == _lib.TAB_KERNEL_GENERAL else 'issue',
                         '%d B per env step + %d B per replayed / planned update of the batches the '
                         'kernel evaluated (%d of %d drawn)' % (b_step, b_upd, batches, steps))
    if traffic is not None:
        roof['traffic'] = traffic
        roof['traffic_unit'] = 'HBM bytes per launch (rocprofv3 PMC, profiles/)'
        roof['frac_measured'] = traffic / (sec / launches) / 1e9 / HBM_PEAK_GBS
        roof['measured_over_algorithmic'] = traffic / (alg / launches)
    if which in ('hex_q', 'wide_q') and what['kernel'] == _lib.TAB_KERNEL_WQN:
        # (tables in LDS for the whole launch: held against instruction issue, as C2 / C6)
        iss = issue_roofline('general_' + which, steps / sec, float(np.mean(ms)), device, n == 65536)
        if iss is not None:
            roof['issue'] = iss
            roof = to_issue_bound(roof)
    if which == 'dynaq_b100':
        iss = issue_roofline('general_dynaq_b100', steps / sec, float(np.mean(ms)), device,
                             n == 65536)
        if iss is not None:
            roof['issue'] = iss
            roof = to_issue_bound(roof)
    return {'value': steps / sec, 'unit': 'env-steps/s', 'ms_per_step': float(np.mean(ms)),
            'td_updates_per_s': (steps + batch * batches) / sec, 'dtype': 'f32',
            'config': {'workload': desc, 'instances_per_gpu': n, 'env_steps_per_launch': env_steps},
            'roofline': roof}


def make_worlds(cfg_name):
    from cobel_amd.misc.gridworld_tools import make_obstacle_maze, make_open_field
    if cfg_name == 'C1':
        w = make_open_field(5, 5, 0, 1)
        w['starting_states'] = np.arange(1, 25)
        return [w]
    if cfg_name == 'C2':
        return [make_open_field(5, 5, 0, 1)]
    if cfg_name == 'C3':
        return [make_obstacle_maze(32, 32, 1234 + k) for k in range(64)]
    if cfg_name == 'C6':
        from cobel_amd.misc.gridworld_tools import make_gridworld
        walls = [(3, 4), (4, 3), (8, 9), (9, 8), (13, 14), (14, 13), (18, 19), (19, 18)]
        w = make_gridworld(5, 5, terminals=[4], rewards=np.array([[4, 10]]), goals=[4],
                           invalid_transitions=walls)
        w['starting_states'] = np.array([12])
        return [w]
    return [make_open_field(32, 32, 0, 1)]


def build_agent(cfg_name, cfg, n, base, device):
    """Environment + agent for `n` instances whose global ids start at `base`."""
    from cobel_amd.agent import SR, DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.policy import EpsilonGreedy
    env = Gridworld(make_worlds(cfg_name), n_envs=n, seed=SEED, device=device,
                    instance_base=base)
    pol = EpsilonGreedy(0.1)
    if cfg['agent'] == 'sfma':
        from cobel_amd.agent import SFMA
        from cobel_amd.memory import SFMAMemory
        from cobel_amd.memory.utils import DR
        w = env.world
        metric = DR(w['width'], w['height'], w['next'], 0.9, w['invalid_transitions'])
        agent = SFMA(env.observation_space, env.action_space, pol,
                     SFMAMemory(metric, env.observation_space.n, 4))
        agent.M.mode = 'reverse'
        agent.mask_actions = True
        return env, agent
    if cfg['agent'] == 'q':
        agent = QAgent(env.observation_space, env.action_space, pol)
    elif cfg['agent'] == 'dynaq':
        agent = DynaQ(env.observation_space, env.action_space, pol)
    else:
        agent = SR(env.observation_space, env.action_space, pol)
    return env, agent


class Runner:
    """Drives the fused kernel in fixed-size launches (the façade's train() runs whole trials)."""

    def __init__(self, cfg, env, agent):
        from cobel_amd import _lib
        self.cfg, self.env, self.agent, self._lib = cfg, env, agent, _lib
        agent._bind(env)
        agent._env_in(env)
        self.flags = _lib.F_LEARN | agent._policy_in(agent.policy, env, False)
        if getattr(agent, 'mask_actions', False):
            self.flags |= _lib.F_MASK_ACTIONS
        agent.monitors.reserve(4096, agent.n_envs, False)

    def launch(self):
        c = self.cfg
        self.agent._launch(self.env, self.agent.policy, self.flags, 0x7fffffff,
                           c['steps_per_trial'], c['env_steps_per_launch'], c['batch'])

    def describe(self):
        """Kernel variant / LDS footprint of the launches (tabular agents: cobel_tab_describe)."""
        if getattr(self.agent, 'describe_launch', None) is None:
            return None
        c = self.cfg
        return self.agent.describe_launch(self.env, self.agent.policy, self.flags, 0x7fffffff,
                                          c['steps_per_trial'], c['env_steps_per_launch'],
                                          c['batch'])


# rate of the port / rate of the real reference, one core, alternating 2-second slices in the build
# container where both exist (scripts/calibrate_cpu_baseline.py): the RANGE
# of the sessions so far (round 2, the round-5 review's run, round 6 on 2026-10-05: C1 1.11 / 1.11 /
# 1.07, C3 1.13 / 1.18 / 1.09, C4 1.35 / 1.07 / 1.15, C6 1.08 / 1.04 / 1.08) — sandbox noise of +-10 %
# around "the port is a little faster, never slower"; no derived reference figure is quoted.
PORT_OVER_REFERENCE = {'C1': (1.07, 1.11), 'C2': (1.07, 1.11), 'C3': (1.09, 1.18), 'C4': (1.07, 1.35),
                       'C6': (1.04, 1.08)}


def cpu_baseline(cfg_name, cfg, seconds=12.0):
    """NumPy restatement of the reference loop, 1 core, bounded sample of the same workload."""
    r = _cpu_baseline(cfg_name, cfg, seconds)
    rng = PORT_OVER_REFERENCE.get(cfg_name)
    r['port_over_reference'] = list(rng) if rng else None
    r['calibration'] = ('the port ran at %.2f-%.2f x the rate of the real reference on this leg in the '
                        'sessions where both exist (BASELINE.md section 4): never slower'
                        % rng if rng else None)
    return r


def _cpu_baseline(cfg_name, cfg, seconds=12.0):
    from oracle import philox, ref_loop
    world = make_worlds(cfg_name)[0]
    tabs = world.compact()
    S = int(world['states'])
    if cfg['agent'] == 'sfma':
        from oracle import sfma_loop
        D = sfma_loop.metric_dr(world['width'], world['height'], tabs['next'].astype(np.int64),
                                0.9, world['invalid_transitions'])
        ag, env = sfma_loop.run_case(tabs, D, SEED, 0, True, 'reverse',
                                     {'mask': np.ones((S, 4), dtype=bool)}, 0,
                                     cfg['steps_per_trial'], cfg['batch'])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ag.train(env, 1, cfg['steps_per_trial'], cfg['batch'])
        dt = time.perf_counter() - t0
        return {'value': len(ag.sarsn) / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
                'reactivations_per_s': len(ag.replayed) / dt,
                'sample': 'oracle/sfma_loop.py (NumPy restatement of the reference SFMA loop, '
                          'float32 tables), instance 0 of the same workload, %d env steps + %d '
                          'reactivations in %.1f s on 1 of %d host cores'
                          % (len(ag.sarsn), len(ag.replayed), dt, os.cpu_count() or 1)}
    env = ref_loop.RefGridworld(tabs, philox.TapeRNG(SEED, 0, philox.STREAM_ENV))
    pol = ref_loop.RefEpsilonGreedy(0.1, philox.TapeRNG(SEED, 0, philox.STREAM_POLICY))
    mem = philox.TapeRNG(SEED, 0, philox.STREAM_MEMORY)
    if cfg['agent'] == 'q':
        ag = ref_loop.RefQAgent(S, 4, pol, mem, dtype=np.float32)
        run = lambda: ag.train(env, 4, cfg['steps_per_trial'], 0, trace=tr)  # noqa: E731
    elif cfg['agent'] == 'dynaq':
        ag = ref_loop.RefDynaQ(S, 4, pol, mem, dtype=np.float32)
        run = lambda: ag.train(env, 1, cfg['steps_per_trial'], cfg['batch'], trace=tr)  # noqa: E731
    else:
        ag = ref_loop.RefSR(S, 4, pol, dtype=np.float32)
        run = lambda: ag.train(env, 1, cfg['steps_per_trial'], trace=tr)  # noqa: E731
    tr = ref_loop.new_trace()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run()
    dt = time.perf_counter() - t0
    steps = len(tr['sarsn'])
    return {'value': steps / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/ref_loop.py (NumPy restatement of the reference single-instance '
                      'loop, float32 tables), instance 0 of the same workload, %d env steps in '
                      '%.1f s on 1 of %d host cores' % (steps, dt, os.cpu_count() or 1)}


def pmc_traffic(cfg_name, cfg, kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this
    script, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only reported for
    the launch geometry the counters were collected with."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None
    ref = CONFIGS[cfg_name]
    if any(cfg[k] != ref[k] for k in ('instances', 'env_steps_per_launch', 'batch')):
        return None
    entry = json.load(open(files[-1])).get(cfg_name)
    if entry is None or (kernel is not None and not entry['kernel'].startswith(kernel.split('<')[0])):
        return None       # counters of another kernel than the one that ran
    return entry['hbm_bytes_per_launch']


VALU_PEAK_SIMDS = 1024          # 256 CUs x 4 SIMDs; a wave64 vector instruction occupies a SIMD for 4 cycles
VALU_CLOCK_GHZ = 2.4            # MI355X peak engine clock (MI355X_MICROARCH.md)


def issue_roofline(key, env_steps_per_s, launch_ms, device, same_geometry):
    """Instruction-issue view of a leg whose tables never leave the chip (LDS resident): vector
    instructions per env step and how busy the vector ALUs are, from the committed SQ counter
    passes (profiles/rNN_pmc_sq.json, scripts/pmc_sq.py: SQ_INSTS_* and SQ_ACTIVE_INST_VALU of the
    leg's kernel under `rocprofv3 --pmc`, the timed launches).  `floor_ms` = what the same
    instructions would take with the vector ALUs issuing every cycle."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_sq.json')))
    if not files:
        return None
    entry = json.load(open(files[-1])).get(key)
    if entry is None:
        return None
    clock_ghz = VALU_CLOCK_GHZ
    peak = VALU_PEAK_SIMDS * clock_ghz / 4.0            # G wave-instructions / s
    achieved = entry['valu_per_env_step'] * env_steps_per_s / 1e9
    r = {'valu_per_step': entry['valu_per_env_step'], 'salu_per_step': entry['salu_per_env_step'],
         'lds_per_step': entry.get('lds_per_env_step'),
         'valu_instr_per_s': achieved * 1e9, 'valu_peak_instr_per_s': peak * 1e9,
         'source': os.path.basename(files[-1]) + ' (' + entry['kernel'] + ')',
         'note': 'wave-level instructions per env step (a lane-per-instance kernel serves 64 env '
                 'steps with one).  The counter pass fixes how long the vector ALUs are busy per env '
                 'step (SQ_ACTIVE_INST_VALU x 4 cycles / clock / env steps: float64 and '
                 'transcendental instructions hold a SIMD longer than 4 cycles, so it exceeds '
                 'valu_per_step x 4 cycles where they matter); valu_busy_frac = that time x THIS '
                 "run's env steps per second / 1024 SIMDs, floor_ms = that time x the env steps of "
                 'a launch / 1024 SIMDs; profiled_valu_busy_frac is the fraction the counter pass '
                 'itself saw'}
    if same_geometry:   # the busy time per env step holds for the launch geometry of the counter pass
        # SIMD-seconds of vector-ALU work per env step, a property of the code, not of this run
        busy_s = entry['valu_busy_frac'] * VALU_PEAK_SIMDS * entry['seconds'] / entry['env_steps']
        r['profiled_valu_busy_frac'] = entry['valu_busy_frac']
        r['valu_busy_simd_ns_per_step'] = busy_s * 1e9
        r['valu_busy_frac'] = busy_s * env_steps_per_s / VALU_PEAK_SIMDS
        r['floor_ms'] = busy_s * env_steps_per_s * launch_ms / VALU_PEAK_SIMDS
    return r


def to_issue_bound(roof):
    """Roofline object of a leg bound by instruction issue: `frac` = how busy the vector ALUs are
    (`issue.valu_busy_frac` at the counter pass's own geometry, else vector instructions per second
    against one wave64 instruction per SIMD and four cycles), the byte accounting of SURVEY 8d
    moved to `hbm_accounting`."""
    iss = roof.get('issue')
    acct = {k: roof.pop(k) for k in ('achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_measured',
                                     'traffic_unit', 'algorithmic_bytes_per_launch',
                                     'algorithmic_bytes_per_env_step',
                                     'algorithmic_bytes_per_reactivation') if k in roof}
    acct['note'] = ('SURVEY 8d bytes per env step x env steps / launch time against the 8 TB/s HBM '
                    'peak; these tables are LDS resident, so the bytes are accounting, not traffic '
                    '(`traffic` = the L2 <-> fabric bytes of the PMC passes)')
    roof['hbm_accounting'] = acct
    roof['bound'] = 'issue'
    roof['unit'] = 'G vector instructions/s'
    if iss is None:
        roof.update({'achieved': None, 'peak': None, 'frac': None,
                     'note': 'no SQ counter pass for this kernel under profiles/'})
        return roof
    roof['achieved'] = iss['valu_instr_per_s'] / 1e9
    roof['peak'] = iss['valu_peak_instr_per_s'] / 1e9
    roof['frac'] = iss.get('valu_busy_frac', roof['achieved'] / roof['peak'])
    roof['traffic'] = acct.get('traffic')
    return roof


def pmc_traffic_leg(key, kernel):
    """HBM bytes per launch of a leg's kernel from the newest committed PMC passes, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None
    entry = json.load(open(files[-1])).get(key)
    if entry is None or not entry['kernel'].startswith(kernel.split(' ')[0]):
        return None
    return entry['hbm_bytes_per_launch']


def run_config(cfg_name, args, rank, world_size, device, dist, repeat_for=0.0):
    from cobel_amd.misc.sharding import shard_instances
    cfg = dict(CONFIGS[cfg_name])
    if args.instances:
        cfg['instances'] = args.instances
    if args.env_steps:
        cfg['env_steps_per_launch'] = args.env_steps
    if args.weak:      # every rank runs the whole configuration
        base, n, n_global = rank * cfg['instances'], cfg['instances'], cfg['instances'] * world_size
    else:              # the configuration's instances split over the ranks
        (base, n), n_global = shard_instances(cfg['instances'], world_size, rank), cfg['instances']
    env, agent = build_agent(cfg_name, cfg, n, base, device)
    runner = Runner(cfg, env, agent)
    # C6: the work per launch grows while the agents learn (trials get shorter, so more of them
    # end — each with its 32 reactivations — inside a launch of 200 env steps): 9.8e6 reactivations
    # and 12 ms in the first launch, 4.66e7 and 28.7 ms from the ~30th on
    # (scripts/experiments/exp_c6_trend.py).  The timed window starts in that steady state.
    warm = max(args.warmup, cfg.get('min_warmup', 0))
    dynaq = cfg['agent'] == 'dynaq'
    per_launch = n * cfg['env_steps_per_launch']          # env steps = planning batches drawn
    first_ms = None
    warm_ev, warm_batches = [], []      # per warm-up launch: HIP events, evaluated-batch counter

    def warm_launch():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        runner.launch()
        e1.record()
        warm_ev.append((e0, e1))
        if dynaq:
            warm_batches.append(agent.batches_done.clone())

    def evaluated_fraction(k):
        """planning batches evaluated / drawn in warm-up launch k (this rank; all ranks' minimum
        when there are several, so that every rank leaves the warm-up after the same launch)"""
        prev = int(warm_batches[k - 1].item()) if k else 0
        f = (int(warm_batches[k].item()) - prev) / max(1, per_launch)
        if dist is not None:
            t = torch.tensor([f], dtype=torch.float64,
                             device='cpu' if dist.get_backend() == 'gloo' else device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            f = float(t.item())
        return f

    for k in range(warm):
        warm_launch()
    torch.cuda.synchronize(device)
    # First use of the monitor reduction and of the barrier (buffers, communicator set-up: 10+ ms on
    # the host) belongs to the warm-up, not in front of the timed window: a GPU that has been idle
    # for a few milliseconds runs its next launch ~10 % slower (clocks), measured on C3 — 13.8 ms
    # against 12.5 for the launches that follow.
    agent.monitors.all_reduce()
    if dist is not None:
        dist.barrier()
    # C3: a Dyna-Q kernel does not evaluate a planning batch that cannot change a table (an
    # instance whose Q and reward estimates are still all +0.0f), so young
    # agents are cheaper per env step than trained ones.  The timed window is the FULL-WORK state:
    # untimed pre-training until the kernel itself reports that >= train_until of the batches it
    # draws are evaluated (checked every 8 launches, at most --max-pretrain launches).
    frac_at_start = None
    if dynaq and cfg.get('train_until') and not args.no_pretrain:
        while warm > 0 and len(warm_ev) < args.max_pretrain:
            frac_at_start = evaluated_fraction(len(warm_ev) - 1)
            if frac_at_start >= cfg['train_until']:
                break
            for _ in range(8):
                warm_launch()
            torch.cuda.synchronize(device)
        warm = len(warm_ev)
    if warm_ev:
        first_ms = warm_ev[0][0].elapsed_time(warm_ev[0][1])
    before = agent.monitors.all_reduce().steps_done       # (global; untimed)
    sfma = cfg['agent'] == 'sfma'
    replays_before = int(agent.replays_done.item()) if sfma else 0
    sr_before = agent.traffic.clone() if cfg['agent'] == 'sr' else None

    def window():
        """exactly args.steps launches between barrier + synchronize brackets; max over ranks"""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(args.steps):
            runner.launch()
            ev[k + 1].record()
        sums = agent.monitors.all_reduce()   # the path's only collective: monitor buffers, one gather
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        launch_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device='cpu' if dist.get_backend() == 'gloo' else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, launch_ms, sums

    batches_before = int(agent.batches_done.item()) if dynaq else 0
    elapsed, launch_ms, sums = window()
    batches_timed = (int(agent.batches_done.item()) - batches_before) if dynaq else 0
    if sfma:
        replays = int(agent.replays_done.item()) - replays_before
    if cfg['agent'] == 'sr':
        sr_moved = (agent.traffic - sr_before).cpu().numpy().astype(np.int64)
    # The reported window is the one above.  With --min-seconds the same window is repeated
    # (untimed for `value`) so that an outside observer sampling GPU utilisation every few seconds
    # sees the job; the spread of the repeats is reported next to the headline.
    repeats, spent = [], elapsed
    while repeat_for > 0 and spent < repeat_for and len(repeats) < 4096:
        e, _, _ = window()           # (elapsed is the max over ranks: every rank stops together)
        repeats.append(e / args.steps * 1e3)
        spent += e
    total_steps = sums.steps_done - before
    expect = n_global * cfg['env_steps_per_launch'] * args.steps
    assert total_steps == expect, 'kernel executed %d env steps, expected %d' % (total_steps, expect)
    mean_launch_s = float(np.mean(launch_ms)) * 1e-3
    steps_per_launch = n * cfg['env_steps_per_launch']           # this rank's launches
    sec8d_bytes_per_launch = cfg['bytes_per_step'] * steps_per_launch
    alg_bytes_per_launch, extra = sec8d_bytes_per_launch, {}
    if dynaq:
        # work DONE, not work drawn: SURVEY 8d's b0 + bm = 78 B per env step and br = 31 B per
        # planning update of the batches the kernel evaluated (counted by the kernel)
        alg_bytes_per_launch = (78 * steps_per_launch
                                + 31 * cfg['batch'] * batches_timed // args.steps)
    if sfma:   # reactivations per launch depend on the trial lengths: count them (this rank)
        alg_bytes_per_launch += cfg['bytes_per_reactivation'] * replays // args.steps
    kernel = {'q': 'k_tab_lpi', 'dynaq': 'k_tab_wpi<DYNAQ>', 'sr': 'k_sr', 'sfma': 'k_sfma'}[cfg['agent']]
    what = runner.describe()
    if what is not None and what['kernel'] == runner._lib.TAB_KERNEL_PWG:
        kernel = 'k_tab_pwg'      # one persistent workgroup per CU (csrc/tabular_pwg.hip)
    limiter = cfg['limiter']
    if cfg['agent'] == 'sr':
        moved = sr_moved
        if moved[1] > 0:     # the sparse-reward kernel ran and counted what it asked for
            S = int(env.observation_space.n)
            asked = int((moved[0] + moved[1]) * 4 * S + moved[2] * 4) // args.steps + 49 * steps_per_launch
            kernel = 'k_sr_wave'
            extra = {'sec8d_bytes_per_env_step': cfg['bytes_per_step'],
                     'sec8d_achieved': sec8d_bytes_per_launch / mean_launch_s / 1e9,
                     'rows_read_per_env_step': float(moved[0]) / (steps_per_launch * args.steps),
                     'rows_written_per_env_step': float(moved[1]) / (steps_per_launch * args.steps),
                     'value_gathers_per_env_step': float(moved[2]) / (steps_per_launch * args.steps),
                     'note': 'algorithmic bytes = SR rows read + written and 4-byte value elements '
                             'requested by k_sr_wave (counted in the kernel) + 49 B of scalars per '
                             'step; SURVEY 8d counts eight row streams per step (sec8d_*), of which '
                             'the sparse-reward form needs two'}
            alg_bytes_per_launch = asked
        else:
            limiter = 'latency'
    achieved = alg_bytes_per_launch / mean_launch_s / 1e9
    traffic = pmc_traffic(cfg_name, cfg, kernel) if (n == cfg['instances'] and args.scale == 1.0) else None
    res = {
        'metric': 'gridworld env-steps/sec (whole job)',
        'value': total_steps / elapsed,
        'unit': 'env-steps/s',
        'n_gpus': world_size, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': cfg_name + ': ' + cfg['desc'], 'instances_total': n_global,
                   'instances_per_gpu': n,
                   'env_steps_per_launch': cfg['env_steps_per_launch'],
                   'parallelism': ('%d instances per GPU x%d (weak)' % (n, world_size) if args.weak
                                   else '%d instances split x%d (strong)' % (n_global, world_size))
                   + ', contiguous global instance ids, no data-path collective, one all-gather '
                     'of the monitor buffers'},
        'roofline': {'bound': 'hbm', 'limiter': limiter, 'achieved': achieved,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'frac_measured': (None if traffic is None
                                       else traffic / mean_launch_s / 1e9 / HBM_PEAK_GBS),
                     'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/)',
                     'algorithmic_bytes_per_launch': alg_bytes_per_launch,
                     'kernel': kernel,
                     'algorithmic_bytes_per_env_step': alg_bytes_per_launch / steps_per_launch,
                     'launch_ms_mean': mean_launch_s * 1e3,
                     'launch_ms_all': [round(x, 4) for x in launch_ms]},
    }
    res['roofline'].update(extra)
    # C2 / C6 keep their tables in LDS for a whole launch (measured L2 <-> fabric traffic: 1 % / 0.3 %
    # of the HBM peak): what bounds them is instruction issue, and that is the roofline they are held
    # against; SURVEY 8d's byte accounting stays beside it as `hbm_accounting`.  C3 keeps the
    # north-star's HBM accounting and carries the issue view as well.
    iss = issue_roofline(cfg_name, steps_per_launch / mean_launch_s, mean_launch_s * 1e3, device,
                         n == CONFIGS[cfg_name]['instances'] and args.scale == 1.0
                         and not args.env_steps and world_size == 1)
    if iss is not None:
        res['roofline']['issue'] = iss
    if cfg_name in ('C2', 'C6'):
        res['roofline'] = to_issue_bound(res['roofline'])
    elif cfg_name == 'C3':
        res['roofline']['note'] = (
            'frac = SURVEY 8d bytes of the work done / launch time / 8 TB/s: the north-star\'s '
            'accounting.  Since round 6 the kernel keeps every Q table in LDS for the whole launch '
            '(ten per CU): `traffic` — the L2 <-> fabric bytes of the counter passes, staging, '
            'write-back and digest gathers — is a tenth of the accounted bytes, and what bounds the '
            'kernel is the chain of one wave\'s step (`limiter`, `issue`; DESIGN.md section 4.1d)')
    if dynaq:
        # (this rank's instances; counted by the kernel: cobel_tab_run_t.batches_done)
        drawn = steps_per_launch * args.steps
        res['roofline']['planning_batches_drawn'] = drawn
        res['roofline']['planning_batches_evaluated'] = batches_timed
        res['roofline']['evaluated_fraction'] = batches_timed / max(1, drawn)
        res['roofline']['sec8d_bytes_per_env_step'] = cfg['bytes_per_step']
        res['roofline']['note'] = (
            'algorithmic bytes = SURVEY 8d by work done: 78 B per env step (b0 + bm) + 31 B per '
            'planning update of the batches the kernel EVALUATED (cobel_tab_run_t.batches_done); a '
            'batch drawn for an instance whose Q and reward estimates are still all +0.0f cannot '
            'change a table and is skipped, bit-identically (DESIGN.md section 4.1).  The timed '
            'window starts after untimed pre-training (`pretraining`) so that nearly every batch '
            'is evaluated: value / ms_per_step / frac are full-work numbers; `young_agents` is the '
            'rate of the first launches after one warm-up launch')
        pre = {'launches': warm - args.warmup, 'untimed_launches_in_all': warm,
               'env_steps_per_instance': warm * cfg['env_steps_per_launch'],
               'evaluated_fraction_at_start': frac_at_start,
               'train_until': None if args.no_pretrain else cfg.get('train_until'),
               'note': 'untimed launches after the `warmup` ones; the agents learn exactly as in '
                       'the timed window'}
        res['pretraining'] = pre
        k_young = [k for k in range(1, min(5, len(warm_ev)))]
        if k_young:
            ms = [warm_ev[k][0].elapsed_time(warm_ev[k][1]) for k in k_young]
            ev = int(warm_batches[k_young[-1]].item()) - int(warm_batches[0].item())
            b = 78 * steps_per_launch * len(k_young) + 31 * cfg['batch'] * ev
            res['young_agents'] = {
                'launches': '2..%d' % (k_young[-1] + 1), 'launch_ms_mean': float(np.mean(ms)),
                'value_this_rank': steps_per_launch * len(k_young) / (sum(ms) * 1e-3),
                'evaluated_fraction': ev / (steps_per_launch * len(k_young)),
                'frac': b / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'note': 'agents 1..%d x %d steps old: most instances have not met a reward yet and '
                        'skip their planning batches' % (k_young[-1] + 1, cfg['env_steps_per_launch'])}
    # global monitor sums (after the one collective): identical for any split of the instances
    res['monitors'] = {'trials_finished': int(sums.lat_cnt.sum()),
                       'escape_latency_sum': int(sums.lat_sum.sum()),
                       'trial_reward_sum': float(sums.reward_sum.sum()),
                       'collectives_in_timed_region': 1 if dist is not None else 0}
    if repeats:
        res['repeat_windows'] = {'count': len(repeats), 'seconds': spent,
                                 'ms_per_step_median': float(np.median(repeats)),
                                 'ms_per_step_min': min(repeats), 'ms_per_step_max': max(repeats),
                                 'ms_per_step_last': repeats[-1],
                                 'value_last': total_steps / (repeats[-1] * 1e-3 * args.steps),
                                 'note': 'the timed window of exactly `steps` launches run again '
                                         'until --min-seconds of GPU work; `value` is the first '
                                         'window.  The instances keep learning through the repeats: '
                                         'on C3 a launch takes longer once Q has filled and most '
                                         'planning updates move their cell (DESIGN.md section 4.1), '
                                         'so the last window is the rate of well-trained agents'}
    if what is not None:
        res['roofline']['lds_bytes_per_workgroup'] = what['lds_bytes']
        res['roofline']['workgroups_per_cu_by_lds'] = what['workgroups_per_cu']
        res['roofline']['instances_per_workgroup'] = what['instances_per_workgroup']
    if sfma:
        res['reactivations_per_s'] = replays * world_size / elapsed
        res['roofline']['algorithmic_bytes_per_reactivation'] = cfg['bytes_per_reactivation']
        res['roofline']['reactivations_per_launch'] = replays // args.steps
        res['transient'] = {'first_launch_ms': first_ms, 'untimed_launches': warm,
                            'note': 'launch time grows with the reactivations per launch while the '
                                    'agents learn; the timed window is the steady state'}
    return res, cfg


def _sig(x, digits=6):
    """floats to `digits` significant digits (the compact line is held under 6 000 bytes)"""
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _short(text, n=110):
    return text if text is None or len(text) <= n else text[:n - 3] + '...'


COMPACT_LIMIT = 6000      # bytes; the driver keeps an 8 KB tail of stdout and parses its last line


def compact(res, full_path=None):
    """The ONE line bench.py prints: every key of the contract, the headline's roofline and CPU
    baseline with numbers only (no prose notes), the other legs reduced to {value, unit,
    ms_per_step, dtype, bound, frac, frac_measured, cpu_baseline}.  The full objects — notes,
    per-launch times, issue-counter provenance, every leg's configuration — go to `full_path`
    (bench_full.json) and to stderr."""
    out = _pick(res, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                      'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'))
    cfg = res['config']
    out['config'] = dict(_pick(cfg, ('instances_total', 'instances_per_gpu', 'env_steps_per_launch')),
                         workload=_short(cfg['workload']), parallelism=_short(cfg['parallelism'], 64))
    roof = res['roofline']
    r = _pick(roof, ('bound', 'limiter', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_measured',
                     'kernel', 'algorithmic_bytes_per_env_step', 'algorithmic_bytes_per_launch',
                     'launch_ms_mean', 'evaluated_fraction', 'sec8d_bytes_per_env_step',
                     'sec8d_achieved'))
    if roof.get('issue'):
        r['issue'] = _pick(roof['issue'], ('valu_per_step', 'salu_per_step', 'valu_busy_frac',
                                           'profiled_valu_busy_frac', 'floor_ms'))
    if roof.get('hbm_accounting'):
        r['hbm_accounting'] = _pick(roof['hbm_accounting'], ('achieved', 'frac', 'traffic'))
    out['roofline'] = r
    if 'cpu_baseline' in res:
        c = res['cpu_baseline']
        out['cpu_baseline'] = dict(_pick(c, ('value', 'unit', 'cores', 'kind', 'port_over_reference')),
                                   sample=_short(c.get('sample')))
    if 'pretraining' in res:
        out['pretraining'] = _pick(res['pretraining'], ('launches', 'untimed_launches_in_all',
                                                        'env_steps_per_instance',
                                                        'evaluated_fraction_at_start', 'train_until'))
    if 'young_agents' in res:
        out['young_agents'] = _pick(res['young_agents'], ('launches', 'launch_ms_mean', 'value_this_rank',
                                                          'evaluated_fraction', 'frac'))
    if 'monitors' in res:
        out['monitors'] = res['monitors']
    if 'repeat_windows' in res:
        out['repeat_windows'] = _pick(res['repeat_windows'], ('count', 'seconds', 'ms_per_step_median',
                                                              'ms_per_step_min', 'ms_per_step_max'))
    legs = {}
    for name, leg in res.get('other_configs', {}).items():
        if 'error' in leg:
            legs[name] = {'error': _short(leg['error'], 100)}
            continue
        lr = leg.get('roofline') or {}
        e = dict(_pick(leg, ('value', 'unit', 'ms_per_step', 'dtype')),
                 **_pick(lr, ('bound', 'frac', 'frac_measured')))
        if 'cpu_baseline' in leg:
            e['cpu_baseline'] = leg['cpu_baseline']['value']
        legs[name] = e
    if legs:
        out['other_configs'] = legs
    if full_path:
        out['full'] = os.path.basename(full_path)
    out = _sig(out)
    line = json.dumps(out, separators=(',', ':'))
    # (never longer than the limit, never an exception in front of the one line that counts: what is
    #  not part of the contract goes first)
    for key in ('repeat_windows', 'monitors', 'young_agents', 'pretraining', 'other_configs'):
        if len(line) < COMPACT_LIMIT:
            break
        out.pop(key, None)
        line = json.dumps(out, separators=(',', ':'))
    return line


def spawn_ranks(argv, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH processes through
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.
    Called before anything in this process has touched the GPU; this process never does."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = int(sock.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    for ln in proc.stdout.splitlines():
        if not ln.startswith('{'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    return proc.returncode if (proc.returncode or lines) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 4 launches x env_steps_per_launch = the T steps per instance of SURVEY.md §8d for
    # C2 (4 096) and C4 (512), twice that for C3 (2 048)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--min-seconds', type=float, default=3.0,
                    help='repeat the headline window until this much GPU time has been spent '
                         '(0: one window only); the reported value is always the first window')
    ap.add_argument('--config', default='C3', choices=sorted(CONFIGS))
    ap.add_argument('--also', default='C2,C4,C6', help='extra configs reported under "other_configs"')
    ap.add_argument('--weak', action='store_true',
                    help='every rank runs the full instance count (default: the instances of the '
                         'configuration are split over the ranks, BASELINE config 3)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo: rehearse several ranks on one GPU (ranks share the cards round robin)')
    ap.add_argument('--instances', type=int, default=0)
    ap.add_argument('--env-steps', type=int, default=0)
    ap.add_argument('--scale', type=float, default=1.0,
                    help='multiply the instance count of EVERY leg (tests: all legs in seconds)')
    ap.add_argument('--no-pretrain', action='store_true',
                    help='C3: time the window right after --warmup launches (young agents)')
    ap.add_argument('--max-pretrain', type=int, default=400,
                    help='C3: upper bound of the untimed pre-training, in launches')
    ap.add_argument('--dist-single', action='store_true',
                    help='with one rank: initialise the process group all the same, so that the '
                         'barriers, the MIN all-reduce of the pre-training and the monitor all-gather '
                         'go through the backend (RCCL on a one-GPU box)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--full', action='store_true',
                    help='print the full result object as the one stdout line (profiling scripts) '
                         'instead of the compact one')
    ap.add_argument('--full-out', default='bench_full.json',
                    help='where rank 0 writes the full result object ("" = nowhere)')
    ap.add_argument('--no-c5', action='store_true', help='skip the network legs (profiling runs)')
    ap.add_argument('--legs', default=None,
                    help='comma-separated subset of the legs behind --also (C5_f64, C5_f32, dyna_dqn, '
                         'dyna_dsr, C1, grid_search, general_hex_q, general_wide_q, general_wide_q_lane, '
                         'general_dynaq_b100); default: all')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(sys.argv[1:], args.gpus))   # (nothing here has touched the GPU yet)

    # stdout carries exactly ONE line, rank 0's JSON: everything else that lands on file descriptor
    # 1 — RCCL prints a five-line version banner there when its first communicator is created — goes
    # to stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    assert args.gpus == world_size, '--gpus must equal the number of launched ranks'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    n_dev = torch.cuda.device_count()
    assert args.backend == 'gloo' or local_rank < n_dev, \
        'rank %d has no GPU of its own (%d visible): RCCL needs one GPU per rank' % (local_rank, n_dev)
    device = torch.device('cuda', local_rank % n_dev)
    torch.cuda.set_device(device)
    dist = None
    if world_size > 1 or args.dist_single:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world_size == 1:      # --dist-single without a launcher: a group of one
            os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)   # "nccl" is RCCL on ROCm
        else:   # rehearsal of the N > 1 path with several ranks on ONE GPU (tests)
            dist.init_process_group(args.backend)

    def scaled(count):
        return max(64, int(round(count * args.scale))) if args.scale != 1.0 else count

    for c in CONFIGS.values():
        c['instances'] = scaled(c['instances'])
    res, cfg = run_config(args.config, args, rank, world_size, device, dist, args.min_seconds)
    others = {}
    if world_size == 1 and not args.instances and (args.also or args.legs):
        for name in [c for c in args.also.split(',') if c and c != args.config]:
            try:
                gc.collect()               # (the previous leg's tables: C4 alone holds 64 GiB)
                torch.cuda.empty_cache()
                r, _ = run_config(name, args, rank, world_size, device, dist)
                others[name] = {'value': r['value'], 'unit': r['unit'], 'dtype': r['dtype'],
                                'ms_per_step': r['ms_per_step'], 'config': r['config'],
                                'warmup': r['warmup'], 'roofline': r['roofline']}
                if 'transient' in r:
                    others[name]['transient'] = r['transient']
                if 'reactivations_per_s' in r:
                    others[name]['reactivations_per_s'] = r['reactivations_per_s']
                # the reference's single-process loop on the same world, timed beside
                # every leg that has one (C2 / C4 / C6; >= 2 000 env steps each)
                if rank == 0 and not args.no_cpu_baseline:
                    others[name]['cpu_baseline'] = cpu_baseline(name, CONFIGS[name], 8.0)
            except Exception as e:  # e.g. not enough HBM for C4 on a shared device
                others[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        gc.collect()
        torch.cuda.empty_cache()
        legs = [] if (args.no_c5 and args.legs is None) else [
            ('C5_f64', lambda: run_c5(device, 'f64', n=scaled(8192))),
            ('C5_f32', lambda: run_c5(device, 'f32', n=scaled(8192))),
            ('dyna_dqn', lambda: run_dyna_dqn(device, scaled(8192))),
            ('dyna_dsr', lambda: run_dyna_dsr(device, scaled(8192))),
            ('C1', lambda: run_c1(device, not args.no_cpu_baseline)),
            ('grid_search', lambda: run_grid_search(device, 16 if args.scale == 1.0 else 4)),
            ('general_hex_q', lambda: run_general(device, 'hex_q', scaled(65536))),
            ('general_wide_q', lambda: run_general(device, 'wide_q', scaled(65536))),
            ('general_wide_q_lane', lambda: run_general(device, 'wide_q_lane', scaled(65536))),
            ('general_dynaq_b100', lambda: run_general(device, 'dynaq_b100', scaled(65536))),
        ]
        if args.legs is not None:
            legs = [(name, leg) for name, leg in legs if name in args.legs.split(',')]
        for name, leg in legs:
            try:
                others[name] = leg()
            except Exception as e:
                import traceback
                traceback.print_exc()      # (stderr: the JSON line on stdout stays one line)
                others[name] = {'error': '%s: %s' % (type(e).__name__, e)}
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0:
        if world_size == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args.config, cfg)
        if others:
            res['other_configs'] = others
        sys.stdout.flush()
        full = json.dumps(res)
        if args.full_out:
            try:
                with open(args.full_out, 'w') as fh:
                    fh.write(full + '\n')
            except OSError as e:      # (a read-only cwd: the compact line is what counts)
                print('bench.py: could not write %s: %s' % (args.full_out, e), file=sys.stderr)
                args.full_out = ''
        print('bench.py full result: ' + full, file=sys.stderr)
        sys.stderr.flush()
        line = full if args.full else compact(res, args.full_out or None)
        os.write(json_fd, (line + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
