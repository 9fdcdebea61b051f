"""Randomised parity sweep for SFMA: the HIP path (csrc/sfma.hip through SFMA / SFMAMemory) against
the NumPy restatement (oracle/sfma_loop.py) on random worlds, metrics, replay modes and switches.

    python scripts/fuzz_sfma.py [first_seed] [count]

Per seed: one world, one configuration, n_envs instances in one launch sequence; three of the
instances are re-run by the restatement and compared — escape latencies, every replayed experience
with its TD error, Q, the memory's tables (rewards, states, strengths C, inhibition-free T) and
the last TD error.  tests/test_gpu_fuzz.py runs a fixed slice.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))



def draw_case(seed: int) -> dict:
    from cobel_amd import _lib
    r = np.random.default_rng(7_000_003 * seed + 5)
    if r.random() < 0.8:
        h, w = int(r.integers(2, 9)), int(r.integers(2, 9))
    else:
        h, w = int(r.integers(9, 17)), int(r.integers(9, 17))
    S = h * w
    goal = int(r.integers(0, S))
    terminals = [goal]
    if r.random() < 0.25:
        terminals.append(int(r.integers(0, S)))
    terminals = sorted(set(terminals))
    rew = {goal: float(r.choice([1.0, 1.0, 2.0, 0.5, -1.0]))}
    for _ in range(int(r.choice([0, 0, 1, 2]))):
        rew[int(r.integers(0, S))] = float(r.choice([0.5, -0.5, 1.0, 0.25]))
    walls = []
    for _ in range(int(r.integers(0, 5))):
        a = int(r.integers(0, S))
        b = a + int(r.choice([-1, 1, -w, w]))
        if 0 <= b < S and not (abs(b - a) == 1 and a // w != b // w):
            walls += [(a, b), (b, a)]
    opts = {'mode': _lib.SFMA_MODES[int(r.integers(0, len(_lib.SFMA_MODES)))]}
    for k in ('recency', 'C_normalize', 'D_normalize', 'deterministic', 'reward_mod_local',
              'reward_mod', 'state_mod', 'dynamic', 'start_replay'):
        if r.random() < 0.25:
            opts[k] = True
    if r.random() < 0.3:
        opts['R_normalize'] = False
    if r.random() < 0.3:
        opts['random'] = bool(r.random() < 0.5)
    if r.random() < 0.3:
        opts['beta'] = float(r.choice([1.0, 5.0, 9.0, 40.0]))
    if r.random() < 0.3:
        opts['decay_inhibition'] = float(r.choice([0.5, 0.8, 0.99, 1.0]))
    if r.random() < 0.3:
        opts['decay_strength'] = float(r.choice([0.9, 0.97, 0.5]))
    if r.random() < 0.2:
        opts['reward_modulation'] = float(r.choice([0.5, 2.0]))
    if r.random() < 0.2:
        opts['nb_replays'] = int(r.choice([1, 2, 3]))
    if r.random() < 0.2:
        opts['noreplay_trials'] = int(r.integers(1, 3))
    if r.random() < 0.2:
        opts['test_trials'] = int(r.integers(1, 3))
    mask = None
    if r.random() < 0.15:
        m = r.random((S, 4)) < 0.8
        m[np.arange(S), r.integers(0, 4, S)] = True
        mask = m
    opts['mask'] = mask
    steps = int(r.integers(3, 50 if S <= 64 else 70))
    B = int(r.choice([1, 4, 8, 16, 24, 32, 40]))
    trials = int(r.integers(1, 6))
    return dict(seed=seed, h=h, w=w, terminals=terminals, rew=rew, walls=walls, opts=opts,
                metric=str(r.choice(['DR', 'SR', 'Euclidean'])), steps=steps, B=B, trials=trials,
                n=int(r.choice([1, 3, 40])), base=int(r.choice([0, 1000, 1 << 18])),
                eps=float(r.choice([0.1, 0.1, 0.3, 0.0, 1.0])),
                general=bool(r.random() < 0.2))


def describe(c: dict) -> str:
    o = {k: v for k, v in c['opts'].items() if k != 'mask'}
    return ('seed %(seed)d %(h)dx%(w)d %(metric)s n=%(n)d base=%(base)d trials=%(trials)d '
            'steps=%(steps)d B=%(B)d eps=%(eps)g general=%(general)d' % c
            + ' terminals=%s rew=%s walls=%d mask=%s opts=%s'
            % (c['terminals'], c['rew'], len(c['walls']), c['opts']['mask'] is not None, o))


SKIPPED = []


def run_case(c: dict):
    import torch
    import test_gpu_sfma as T
    from cobel_amd.memory.utils import DR, SR, Euclidean
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from oracle import sfma_loop
    h, w = c['h'], c['w']
    rewards = np.array([[s, v] for s, v in c['rew'].items()], dtype=np.float64)
    world = make_gridworld(h, w, terminals=c['terminals'], rewards=rewards, goals=c['terminals'],
                           invalid_transitions=list(c['walls']))
    if len(world['starting_states']) == 0:
        return []
    if c['metric'] == 'DR':
        D = DR(w, h, world['next'], 0.9, world['invalid_transitions']).D
    elif c['metric'] == 'SR':
        D = SR(world['next'], 0.9).D
    else:
        D = Euclidean(w, h).D
    opts = c['opts']
    tab = dict(world.compact(), height=h, width=w, coordinates=world['coordinates'])
    SEED = T.SEED
    env, agent = T.build(tab, D, opts, c['n'], c['base'], eps=c['eps'])
    agent.force_general_kernel = c['general']
    T.run_schedule(env, agent, opts, c['trials'], c['steps'], c['B'])
    torch.cuda.synchronize()
    ow = T._oracle_world(world)
    total = c['trials'] + opts.get('noreplay_trials', 0) + opts.get('test_trials', 0)
    bad = []
    for i in sorted({0, c['n'] // 2, c['n'] - 1}):
        try:
            with np.errstate(all='ignore'):
                ag, _ = sfma_loop.run_case(ow, D, SEED, c['base'] + i, True, opts['mode'], opts,
                                           c['trials'], c['steps'], c['B'], eps=c['eps'])
        except ValueError as e:
            # NaN activation probabilities (0 / 0 strengths, overflowing exp without
            # R_normalize): the reference's rng.choice raises — nothing to compare with
            assert 'NaN' in str(e), e
            SKIPPED.append((c['seed'], i, 'reference raises'))
            continue
        if ag.M.nan_ratings:
            # deterministic mode walks on with NaN ratings (argmax = experience 0); the build
            # ends such a replay
            SKIPPED.append((c['seed'], i, 'NaN ratings'))
            continue

        def cmp(name, got, want, **kw):
            got, want = np.asarray(got), np.asarray(want)
            if got.shape != want.shape or not np.array_equal(got, want, **kw):
                bad.append('inst %d %s' % (i, name))

        def of(x):      # (one instance: the host classes hand out the reference's unbatched views)
            x = x.cpu().numpy() if hasattr(x, 'cpu') else np.asarray(x)
            return x if c['n'] == 1 else x[i]

        cmp('steps', agent.monitors.lat_trace[i].cpu().numpy()[:total], ag.steps)
        rp = np.array(ag.replayed, dtype=np.float64).reshape(-1, 8)
        ev = T.events_of(agent, i)
        try:
            T.check_events(ev, rp)
        except AssertionError:
            sa = ev['sa'].astype(np.int64)
            mine = np.stack([ev['trial'], (sa >> 25) & 1, sa & 0xFFFF, (sa >> 16) & 0xFF,
                             ev['reward'].astype(np.float64), ev['next'], (sa >> 24) & 1,
                             ev['td']], 1).astype(np.float64) if len(ev) else np.zeros((0, 8))
            m = min(len(mine), len(rp))
            diff = np.argwhere(~((mine[:m] == rp[:m]) | (np.isnan(mine[:m]) & np.isnan(rp[:m]))))
            k = int(diff[0, 0]) if len(diff) else m
            bad.append('inst %d replay events: %d vs %d, first at %d: %s vs %s'
                       % (i, len(mine), len(rp), k, mine[k].tolist() if k < len(mine) else None,
                          rp[k].tolist() if k < len(rp) else None))
        cmp('Q', of(agent.Q), ag.Q)
        cmp('M.rewards', of(agent.M.rewards), ag.M.rewards)
        cmp('M.states', of(agent.M.states), ag.M.states)
        cmp('M.C', of(agent.M.C), ag.M.C)
        cmp('M.T', of(agent.M.T), ag.M.T)
        td = float(np.asarray(agent.td).reshape(-1)[i])
        if not (td == float(ag.td) or (np.isnan(td) and np.isnan(float(ag.td)))):
            bad.append('inst %d td' % i)
    return bad


def main() -> int:
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    failed, t0 = [], time.time()
    for seed in range(first, first + count):
        c = draw_case(seed)
        try:
            bad = run_case(c)
        except Exception as e:
            bad = ['%s: %s' % (type(e).__name__, str(e)[:300])]
        if bad:
            failed.append(seed)
            print('MISMATCH', bad[:6], describe(c), flush=True)
        if (seed - first) % 10 == 9:
            print('... %d cases, %d failing, %.0f s' % (seed - first + 1, len(failed), time.time() - t0),
                  flush=True)
    print('cases %d, failing %d: %s; instances skipped because the reference has NaN ratings: %d'
          % (count, len(failed), failed, len(SKIPPED)))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
