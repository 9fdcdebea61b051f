"""Randomised parity sweep for worlds whose transition rows are distributions (the reference's
Gridworld.step draws the successor, interface/gridworld.py:119-123): random gridworlds whose dense
sas is edited at random — 1 to 4 possible successors anywhere in the world, random weights
(normalised to 1 within float64 round-off: anything else is a ValueError in Generator.choice and here),
some rows left one-hot — Dyna-Q and Q-learning through cobel_tab_run's general kernel against the
NumPy restatement (oracle/ref_loop.py) fed with the build's streams.

    python scripts/fuzz_stochastic.py [first_seed] [count]
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
    r = np.random.default_rng(13_000_001 * seed + 7)
    h, w = int(r.integers(1, 8)), int(r.integers(2, 8))
    S = h * w
    terminals = sorted(set(int(x) for x in r.integers(0, S, int(r.integers(0, 3)))))
    if len(terminals) == S:
        terminals = terminals[:-1]
    rew = {int(x): float(r.choice([1.0, -0.5, 2.0, 0.25])) for x in r.integers(0, S, int(r.integers(1, 4)))}
    edits = []
    for s in range(S):
        for a in range(4):
            if r.random() < 0.6:
                k = int(r.integers(1, 5))
                edits.append((s, a, r.integers(0, S, k).tolist(),
                              r.choice([0.1, 0.25, 0.5, 1.0, 2.0, 1e-3], k).tolist()))
    return dict(seed=seed, h=h, w=w, terminals=terminals, rew=rew, edits=edits,
                kind=str(r.choice(['dynaq', 'q'])), n=int(r.choice([1, 3, 64, 130])),
                base=int(r.choice([0, 11, 1 << 16])), trials=int(r.integers(1, 6)),
                steps=int(r.integers(1, 41)), batch=int(r.choice([0, 1, 8, 32, 62, 63, 100])),
                alpha=float(r.choice([0.9, 0.5, 0.99])), gamma=float(r.choice([0.8, 0.99, 0.0])),
                eps=float(r.choice([0.1, 0.3, 1.0, 0.0])), second=bool(r.random() < 0.3))


def describe(c: dict) -> str:
    return ('seed %(seed)d %(kind)s %(h)dx%(w)d n=%(n)d base=%(base)d trials=%(trials)d '
            'steps=%(steps)d B=%(batch)d a=%(alpha)g g=%(gamma)g e=%(eps)g second=%(second)d' % c
            + ' edited rows=%d' % len(c['edits']))


def run_case(c: dict):
    import torch
    from cobel_amd.agent import DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    from oracle import ref_loop
    from oracle.philox import STREAM_ENV, STREAM_MEMORY, STREAM_POLICY, TapeRNG
    S = c['h'] * c['w']
    rewards = np.array([[s, v] for s, v in c['rew'].items()], dtype=np.float64)
    world = make_gridworld(c['h'], c['w'], terminals=c['terminals'], rewards=rewards,
                           goals=c['terminals'], deterministic=False)
    if len(world['starting_states']) == 0:
        return []
    sas = world['sas']
    for s, a, succ, wts in c['edits']:
        sas[s, a] = 0.0
        for t, p in zip(succ, wts):
            sas[s, a, t] += p
        sas[s, a] /= sas[s, a].sum()
    env = Gridworld(world, n_envs=c['n'], seed=SEED, instance_base=c['base'])
    cls = DynaQ if c['kind'] == 'dynaq' else QAgent
    ag = cls(env.observation_space, env.action_space, EpsilonGreedy(c['eps']),
             learning_rate=c['alpha'], gamma=c['gamma'])
    ag.track_instances = True
    ag.train(env, c['trials'], c['steps'], c['batch'])
    if c['second']:
        ag.train(env, c['trials'], c['steps'], c['batch'])
    torch.cuda.synchronize()
    total = c['trials'] * (2 if c['second'] else 1)
    tab = dict(next=world['next'], reward=world['rewards'], terminal=world['terminals'],
               starts=world['starting_states'], sas=np.array(sas))
    if not env.handle.stochastic:
        # every edited row came out one-hot: a world of tables.  The reference would still consume
        # one double per step (its flag is off) without looking at it; the build draws nothing for
        # a table, so its trial starts sit at other counters of the env stream — compare with the
        # restatement stepping the table
        del tab['sas']
    Q = ag._q.cpu().numpy()
    lat = ag.monitors.lat_trace.cpu().numpy()
    ctr = env.env_ctr.cpu().numpy()
    bad = []
    for i in sorted({0, c['n'] // 2, c['n'] - 1}):
        g = c['base'] + i
        erng = TapeRNG(SEED, g, STREAM_ENV, double_sub=1)
        renv = ref_loop.RefGridworld(tab, erng)
        pol = ref_loop.RefEpsilonGreedy(c['eps'], TapeRNG(SEED, g, STREAM_POLICY))
        if c['kind'] == 'dynaq':
            ref = ref_loop.RefDynaQ(S, 4, pol, TapeRNG(SEED, g, STREAM_MEMORY), c['alpha'], c['gamma'],
                                    dtype=np.float32)
        else:
            ref = ref_loop.RefQAgent(S, 4, pol, TapeRNG(SEED, g, STREAM_MEMORY), c['alpha'], c['gamma'],
                                     dtype=np.float32)
        tr = ref_loop.new_trace()
        for _ in range(2 if c['second'] else 1):
            ref.train(renv, c['trials'], c['steps'], c['batch'], trace=tr)
        if not np.array_equal(lat[i, :total], tr['steps']):
            bad.append('inst %d steps %s vs %s' % (i, lat[i, :total].tolist(), tr['steps']))
        if not np.array_equal(Q[i].reshape(S, 4), ref.Q):
            bad.append('inst %d Q' % i)
        if int(ctr[i]) != erng.index:
            bad.append('inst %d env draws %d vs %d' % (i, int(ctr[i]), erng.index))
        if c['kind'] == 'dynaq':
            M = ag.M
            sq = (lambda a: np.asarray(a) if c['n'] == 1 else np.asarray(a)[i])
            if not (np.array_equal(sq(M.states), ref.M.states)
                    and np.array_equal(sq(M.terminals), ref.M.terminals)
                    and np.array_equal(np.asarray(sq(M.rewards), dtype=np.float32), ref.M.rewards)):
                bad.append('inst %d model' % i)
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
