"""Randomised parity sweep for QAgent on Topology graphs with 1..8 actions (seeds from 10^6 on: 9..32;
from 2 * 10^6 on: 1..32 with a random action mask in half of the cases — bytes up to eight actions,
32-bit words beyond) — random directed graphs
(every node the same neighbour count, as `Topology` requires), random rewards / terminals / start
nodes, replay batches 0..130 — against the NumPy restatement of the reference's loop
(oracle/ref_loop.py, fed with the build's streams through TapeRNG).

    python scripts/fuzz_topology.py [first_seed] [count]

Four actions take the wavefront kernels, every other count the general kernel (csrc/general.hip);
three instances per case are re-run by the restatement: escape latencies, Q (float32, bit for
bit), length of the replay memory.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))

SEED = 0xC0BE1


def draw_case(seed: int) -> dict:
    r = np.random.default_rng(9_000_011 * seed + 3)
    S = int(r.integers(2, 61))
    A = int(r.choice([1, 2, 3, 4, 4, 5, 6, 6, 7, 8]))
    if seed >= 1_000_000:   # (seeds from 10^6 on: nine to 32 neighbours — the wide general kernels)
        A = int(r.choice([9, 10, 12, 13, 16, 17, 20, 24, 31, 32]))
    if seed >= 2_000_000:   # (round 5: any count, masks)
        A = int(r.choice([1, 2, 4, 6, 8, 9, 12, 16, 17, 31, 32]))
    nbr = r.integers(0, S, (S, A))
    stay = r.random((S, A)) < 0.2             # missing neighbours are the node itself
    nbr = np.where(stay, np.arange(S)[:, None], nbr)
    terminal = r.random(S) < 0.1
    if terminal.all():
        terminal[0] = False
    reward = np.where(r.random(S) < 0.15, r.choice([1.0, -1.0, 0.5, 2.0, 1e-3], S), 0.0)
    reward[terminal] = r.choice([1.0, 3.0, -0.5], int(terminal.sum()))
    free = np.flatnonzero(~terminal)
    starts = None
    if r.random() < 0.5:
        starts = sorted(set(int(x) for x in r.choice(free, int(r.integers(1, 5)))))
    mask = None
    if seed >= 2_000_000 and r.random() < 0.5:
        mask = r.random((S, A)) < 0.6
        mask[np.arange(S), r.integers(0, A, S)] = True     # (never all actions masked)
    return dict(seed=seed, S=S, A=A, nbr=nbr, terminal=terminal, reward=reward, starts=starts, mask=mask,
                n=int(r.choice([1, 2, 64, 130])), base=int(r.choice([0, 3, 1 << 16])),
                trials=int(r.integers(1, 6)), steps=int(r.integers(1, 41)),
                batch=int(r.choice([0, 0, 1, 8, 24, 62, 63, 70, 130])),
                alpha=float(r.choice([0.9, 0.5, 1.0, 0.1])), gamma=float(r.choice([0.8, 0.99, 0.0])),
                eps=float(r.choice([0.1, 0.3, 0.0, 1.0])), second=bool(r.random() < 0.3))


def describe(c: dict) -> str:
    return ('seed %(seed)d S=%(S)d A=%(A)d n=%(n)d base=%(base)d trials=%(trials)d steps=%(steps)d '
            'B=%(batch)d a=%(alpha)g g=%(gamma)g e=%(eps)g second=%(second)d' % c
            + (' masked' if c.get('mask') is not None else ''))


def run_case(c: dict):
    import torch
    from cobel_amd.agent import QAgent
    from cobel_amd.interface import Topology
    from cobel_amd.policy import EpsilonGreedy
    from oracle import ref_loop
    from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG
    S, A = c['S'], c['A']
    nodes = {}
    for i in range(S):
        nodes[str(i)] = {'id': str(i), 'pose': np.array([float(i % 7), float(i // 7), 0.0, 0.0, 0.0, 0.0]),
                         'neighbors': [str(int(j)) for j in c['nbr'][i]],
                         'reward': float(c['reward'][i]), 'terminal': bool(c['terminal'][i])}
    starts = None if c['starts'] is None else [str(s) for s in c['starts']]
    env = Topology(nodes, starts, n_envs=c['n'], seed=SEED, instance_base=c['base'])
    assert int(env.action_space.n) == A
    ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(c['eps']),
                learning_rate=c['alpha'], gamma=c['gamma'])
    ag.track_instances = True
    if c.get('mask') is not None:
        ag.mask_actions = True
        ag.action_mask = c['mask'].copy()
    ag.train(env, c['trials'], c['steps'], c['batch'])
    if c['second']:
        ag.train(env, c['trials'], c['steps'], c['batch'])
    torch.cuda.synchronize()
    total = c['trials'] * (2 if c['second'] else 1)
    w = env.world
    tab = dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'], starts=w['starting_states'])
    Q = ag._q.cpu().numpy()
    lat = ag.monitors.lat_trace.cpu().numpy()
    loglen = ag.inst[:, 6].cpu().numpy()
    bad = []
    for i in sorted({0, c['n'] // 2, c['n'] - 1}):
        g = c['base'] + i
        renv = ref_loop.RefGridworld(tab, TapeRNG(SEED, g, STREAM_ENV))
        pol = ref_loop.RefEpsilonGreedy(c['eps'], TapeRNG(SEED, g, STREAM_POLICY))
        ref = ref_loop.RefQAgent(S, A, pol, TapeRNG(SEED, g, STREAM_MEMORY), c['alpha'], c['gamma'],
                                 dtype=np.float32)
        if c.get('mask') is not None:
            ref.mask_actions, ref.action_mask = True, c['mask']
        tr = ref_loop.new_trace()
        ref.train(renv, c['trials'], c['steps'], c['batch'], trace=tr)
        if c['second']:
            ref.train(renv, c['trials'], c['steps'], c['batch'], trace=tr)
        if not np.array_equal(lat[i, :total], tr['steps']):
            bad.append('inst %d steps %s vs %s' % (i, lat[i, :total].tolist(), tr['steps']))
        if not np.array_equal(Q[i].reshape(S, A), ref.Q):
            d = np.argwhere(Q[i].reshape(S, A) != ref.Q)
            bad.append('inst %d Q: %d differ, first %s' % (i, len(d), d[0].tolist()))
        if int(loglen[i]) != len(ref.M):
            bad.append('inst %d len(M) %d vs %d' % (i, int(loglen[i]), len(ref.M)))
    return bad


def main() -> int:
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    failed, t0 = [], time.time()
    for seed in range(first, first + count):
        c = draw_case(seed)
        try:
            bad = run_case(c)
        except Exception as e:
            bad = ['%s: %s' % (type(e).__name__, str(e)[:300])]
        if bad:
            failed.append(seed)
            print('MISMATCH', bad[:4], describe(c), flush=True)
        if (seed - first) % 20 == 19:
            print('... %d cases, %d failing, %.0f s' % (seed - first + 1, len(failed), time.time() - t0),
                  flush=True)
    print('cases %d, failing %d: %s' % (count, len(failed), failed))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
