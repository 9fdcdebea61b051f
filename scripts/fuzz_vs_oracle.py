"""Randomised parity sweep: the HIP tabular / SR paths against the C oracle on random worlds,
hyper-parameters, batch sizes, instance counts and launch shapes.

    python scripts/fuzz_vs_oracle.py [first_seed] [count]

Each seed draws one case (see `draw_case`), runs it through the product path (Gridworld + DynaQ /
QAgent / SR on cuda:0) and through oracle/cobel_oracle.c, and compares every table, counter and
per-trial monitor bit for bit.  Prints one line per failing seed and a summary; exit code 1 if any
case differs.  tests/test_gpu_fuzz.py runs a fixed slice of the same generator.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cobel-rl_amd'))


def draw_case(seed: int) -> dict:
    r = np.random.default_rng(1_000_003 * seed + 17)
    kind = ['dynaq', 'dynaq', 'q', 'sr'][int(r.integers(0, 4))]
    shape = r.random()
    if kind == 'sr' and shape < 0.45:      # the sizes the wave-per-instance SR kernel covers
        H, W = [(16, 16), (16, 32), (32, 32)][int(r.integers(0, 3))]
    elif shape < 0.8:
        H, W = int(r.integers(1, 11)), int(r.integers(2, 11))
    elif shape < 0.95:
        H, W = int(r.integers(8, 25)), int(r.integers(8, 25))
    else:
        H, W = int(r.integers(30, 45)), int(r.integers(30, 45))   # past the LDS-resident sizes
    S = H * W
    n_term = int(r.integers(0, 4))
    terminals = sorted(set(int(x) for x in r.integers(0, S, n_term)))
    n_rew = int(r.choice([0, 1, 1, 2, 2, 3, 6]))
    rew_states = sorted(set(int(x) for x in r.integers(0, S, n_rew)))
    for t in terminals:                      # most goals are rewarded terminals
        if r.random() < 0.7 and t not in rew_states:
            rew_states.append(t)
    vals = r.choice([1.0, 1.0, 0.5, -1.0, 2.5, 0.0, -0.25, 1e-3], len(rew_states))
    rewards = np.array([[s, v] for s, v in zip(rew_states, vals)], dtype=np.float64).reshape(-1, 2)
    n_inv = int(r.integers(0, max(1, S // 6)))
    invalid = sorted(set(int(x) for x in r.integers(0, S, n_inv)) - set(terminals))
    trans = []
    for _ in range(int(r.integers(0, 6))):
        a = int(r.integers(0, S))
        b = a + int(r.choice([-1, 1, -W, W]))
        if 0 <= b < S:
            trans.append((a, b))
    starts = None
    if r.random() < 0.4:
        starts = sorted(set(int(x) for x in r.integers(0, S, int(r.integers(1, 6)))) - set(terminals))
        starts = starts or None
    if starts is None and len(terminals) == S:
        terminals = terminals[:-1]
    wind = None
    if r.random() < 0.15:
        wind = r.integers(-1, 2, (S, 2))
    budget = 60_000 if S <= 100 else 25_000
    if kind == 'sr':
        budget = max(64, 3_000_000 // S)
    n = int(r.choice([1, 2, 3, 63, 64, 65, 130, 257]))
    trials = int(r.integers(1, 7))
    steps = int(r.integers(1, 61))
    batch = 0
    if kind == 'dynaq':
        batch = int(r.choice([0, 1, 5, 20, 32, 50, 62, 63, 64, 100, 130]))
    elif kind == 'q':
        batch = int(r.choice([0, 0, 1, 8, 24, 62, 63, 70]))
    while n * trials * steps * (batch + 1) > budget * 40 and n > 1:
        n = max(1, n // 2)
    case = dict(seed=seed, kind=kind, H=H, W=W, terminals=terminals, rewards=rewards,
                invalid=invalid, trans=trans, starts=starts, wind=wind, n=n, trials=trials,
                steps=steps, batch=batch,
                alpha=float(r.choice([0.99, 0.9, 0.5, 0.1, 1.0])),
                gamma=float(r.choice([0.99, 0.9, 0.5, 0.0, 1.0])),
                eps=float(r.choice([0.0, 0.1, 0.1, 0.3, 1.0])),
                model_lr=float(r.choice([0.9, 0.9, 1.0, 0.25])),
                base=int(r.choice([0, 0, 5, 1 << 20])),
                env_seed=int(r.integers(0, 1 << 31)),
                episodic=bool(kind == 'dynaq' and r.random() < 0.15),
                mask=bool(r.random() < 0.2),
                general=bool(kind != 'sr' and r.random() < 0.25),
                stream_rows=bool(kind == 'sr' and r.random() < 0.3),
                second=bool(r.random() < 0.4),
                mask_seed=int(r.integers(0, 1 << 31)))
    # (drawn last, so the fields above are what earlier versions of the sweep drew)
    case['extra_worlds'] = int(r.choice([0, 0, 0, 1, 2]))       # instance i lives in world i % W
    case['test_trials'] = int(r.choice([0, 0, 1, 3])) if kind != 'sr' else 0
    case['world_seed'] = int(r.integers(0, 1 << 31))
    # sessions cut into launches of `chunk` steps per instance (0: one launch, what train() does);
    # results may not depend on where the launches end (bench.py drives the kernels this way)
    case['chunk'] = int(r.choice([0, 0, 0, 1, 7, 37]))
    # per-instance hyper-parameters (cobel_param_set_t): instance i runs combination which[i]; it
    # must equal instance i of a run in which every instance uses that combination
    case['psets'] = None
    if r.random() < 0.2 and case['n'] > 1:
        k = int(r.integers(2, 4))
        case['psets'] = dict(
            combos=[(float(r.choice([0.99, 0.5, 0.1])), float(r.choice([0.99, 0.9, 0.5])),
                     float(r.choice([0.0, 0.1, 0.3, 1.0])), float(r.choice([0.9, 1.0, 0.25])))
                    for _ in range(k)],
            which=r.integers(0, k, case['n']))
    return case


def describe(c: dict) -> str:
    return ('seed %(seed)d %(kind)s %(H)dx%(W)d n=%(n)d trials=%(trials)d steps=%(steps)d B=%(batch)d '
            'a=%(alpha)g g=%(gamma)g e=%(eps)g mlr=%(model_lr)g base=%(base)d epi=%(episodic)d '
            'mask=%(mask)d general=%(general)d stream=%(stream_rows)d second=%(second)d '
            'worlds=1+%(extra_worlds)d test=%(test_trials)d chunk=%(chunk)d' % c
            + (' psets=%s' % (c['psets']['combos'],) if c.get('psets') else '')
            + ' terminals=%s rewards=%s' % (c['terminals'], c['rewards'].tolist()))


STATS = {}


def session(ag, env, c: dict, trials: int, steps: int, batch: int) -> None:
    """agent.train(...), optionally as budgeted launches (the steps of Agent._session written out)"""
    from cobel_amd import _lib
    kind = c['kind']
    if not c['chunk']:
        ag.train(env, trials, steps, batch) if kind != 'sr' else ag.train(env, trials, steps)
        return
    ag._bind(env)
    if kind == 'q':
        used = int(ag.inst[:, _lib.I_LOG_LEN].max().item())
        ag.reserve_replay(used + trials * steps)
    flags = _lib.F_LEARN | (_lib.F_MASK_ACTIONS if ag.mask_actions else 0)
    if kind == 'dynaq' and ag.episodic_replay:
        flags |= _lib.F_EPISODIC
    ag._env_in(env)
    flags |= ag._policy_in(ag.policy, env, False)
    target = ag.current_trial + trials
    ag.monitors.reserve(target, ag.n_envs, ag.track_instances)
    for _ in range(trials * steps + 1):
        ag._launch(env, ag.policy, flags, target, steps, c['chunk'], batch)
        if int(ag.inst[:, _lib.I_TRIAL].min().item()) >= target:
            break
    else:
        raise AssertionError('budgeted launches did not finish the session')
    ag.current_trial = target
    ag._policy_out(ag.policy)
    ag._env_out(env)


def run_case(c: dict):
    """-> list of mismatching items (empty = parity)"""
    import torch
    from cobel_amd.agent import SR, DynaQ, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    from oracle import c_oracle
    S = c['H'] * c['W']
    world = make_gridworld(c['H'], c['W'], terminals=c['terminals'], rewards=c['rewards'],
                           goals=c['terminals'], starting_states=c['starts'],
                           invalid_states=c['invalid'], invalid_transitions=c['trans'],
                           wind=c['wind'])
    if len(world['starting_states']) == 0:
        return []
    worlds = [world]
    wr = np.random.default_rng(c['world_seed'])
    for _ in range(c['extra_worlds']):       # same size, other walls / goals / rewards
        inv = sorted(set(int(x) for x in wr.integers(0, S, int(wr.integers(0, max(1, S // 5))))))
        term = sorted(set(int(x) for x in wr.integers(0, S, int(wr.integers(0, 3)))) - set(inv))
        rw = np.array([[int(wr.integers(0, S)), float(wr.choice([1.0, -0.5, 2.0]))]
                       for _ in range(int(wr.integers(0, 3)))], dtype=np.float64).reshape(-1, 2)
        w2 = make_gridworld(c['H'], c['W'], terminals=term, rewards=rw, goals=term, invalid_states=inv)
        if len(w2['starting_states']):
            worlds.append(w2)
    mask = None
    if c['mask']:
        m = np.random.default_rng(c['mask_seed']).random((S, 4)) < 0.8
        m[np.arange(S), np.random.default_rng(c['mask_seed'] + 1).integers(0, 4, S)] = True
        mask = m
    env = Gridworld(worlds if len(worlds) > 1 else world, n_envs=c['n'], seed=c['env_seed'],
                    instance_base=c['base'])
    ow = c_oracle.OracleWorld([dict(next=w['next'], reward=w['rewards'], terminal=w['terminals'],
                                    starts=w['starting_states']) for w in worlds])
    total = c['trials'] * (2 if c['second'] else 1)
    bad = []

    ps = c.get('psets')
    combos = ps['combos'] if ps else [(c['alpha'], c['gamma'], c['eps'], c['model_lr'])]
    which = np.asarray(ps['which']) if ps else np.zeros(c['n'], dtype=np.int64)
    col = lambda j: np.array([combos[k][j] for k in which]) if ps else combos[0][j]   # noqa: E731
    sel = [None]

    def cmp(name, got, want, per_instance=True):
        got, want = np.asarray(got), np.asarray(want)
        if ps and not per_instance:
            return                     # sums over instances mix the combinations
        if ps and got.shape == want.shape:
            got, want = got[sel[0]], want[sel[0]]
        if got.shape != want.shape:
            bad.append('%s: shape %s vs %s' % (name, got.shape, want.shape))
        elif not np.array_equal(got, want):
            where = np.argwhere(got != want)
            bad.append('%s: %d differ, first %s' % (name, len(where), [
                (tuple(int(x) for x in w), got[tuple(w)].item(), want[tuple(w)].item())
                for w in where[:4]]))

    if c['kind'] in ('dynaq', 'q'):
        cls = DynaQ if c['kind'] == 'dynaq' else QAgent
        tt = c['test_trials']
        test_eps = 0.0 if c['seed'] % 2 == 0 else 0.2
        extra = {}
        # one instance: the reference's per-step callbacks (single-step launches, logs from the
        # kernel's last-experience record) — compared with the oracle's per-step trace below
        steplog, triallog = [], []
        if c['n'] == 1 and not c['chunk']:
            extra['custom_callbacks'] = {
                'on_step_end': [lambda l: steplog.append((l['state'], l['action'], l['reward'],
                                                          l['next_state'], l['terminal'],
                                                          l.get('td', np.nan)))],
                'on_trial_end': [lambda l: triallog.append((l['steps'], l['trial_reward']))]}
        if c['kind'] == 'dynaq':
            from cobel_amd.memory import DynaQMemory
            extra['memory'] = DynaQMemory(S, 4, col(3))
        ag = cls(env.observation_space, env.action_space, EpsilonGreedy(col(2)),
                 EpsilonGreedy(test_eps) if tt else None,
                 learning_rate=col(0), gamma=col(1), **extra)
        ag.track_instances = True
        ag.track_responses = True
        ag.force_general = c['general']
        if c['kind'] == 'dynaq':
            ag.episodic_replay = c['episodic']
        if mask is not None:
            ag.mask_actions = True
            ag.action_mask = mask.copy()
        log_cap = total * c['steps'] if c['kind'] == 'q' else 0
        session(ag, env, c, c['trials'], c['steps'], c['batch'])
        if c['second']:
            session(ag, env, c, c['trials'], c['steps'], c['batch'])
        if tt:
            ag.test(env, tt, c['steps'])
        torch.cuda.synchronize()
        for k, (alpha, gamma, eps, mlr) in enumerate(combos):
            sel[0] = np.flatnonzero(which == k)
            o = c_oracle.TabOracle(ow, c['n'], c_oracle.AG_DYNAQ if c['kind'] == 'dynaq' else c_oracle.AG_Q,
                                   c['env_seed'], True, instance_base=c['base'], alpha=alpha,
                                   gamma=gamma, epsilon=eps, model_lr=mlr,
                                   trial_cap=total + tt, log_cap=log_cap, action_mask=mask)
            flags = c_oracle.F_LEARN | (c_oracle.F_EPISODIC if c['episodic'] else 0)
            tcap = total * c['steps'] if steplog else 0
            tr = o.run(c['trials'], c['steps'], c['batch'], flags=flags, trace_inst=0, trace_cap=tcap)
            if c['second']:
                tr = np.concatenate([tr, o.run(total, c['steps'], c['batch'], flags=flags,
                                               trace_inst=0, trace_cap=tcap)])
            if steplog:
                STATS['per-step callback runs'] = STATS.get('per-step callback runs', 0) + 1
                STATS['per-step log rows'] = STATS.get('per-step log rows', 0) + len(tr)
                mine = np.array(steplog, dtype=np.float64)[:len(tr)]
                cmp('per-step logs', mine, tr, per_instance=False)
                cmp('per-trial steps', np.array([t[0] for t in triallog][:total]), o.lat_trace[0, :total],
                    per_instance=False)
            q_trained = o.Q.copy()
            if tt:      # Agent.test: no learning, the test policy's own stream and counter
                saved = o.inst['ctr_policy'].copy()
                o.inst['ctr_policy'] = 0
                o.run(total + tt, c['steps'], 0, flags=c_oracle.F_TEST_STREAM, epsilon=test_eps)
                o.inst['ctr_policy'] = saved
                cmp('Q after test', o.Q, q_trained)
                cmp('test lat_trace', ag.monitors.lat_trace.cpu().numpy()[:, total:total + tt],
                    o.lat_trace[:, total:total + tt])
            cap = total + tt
            cmp('lat_sum', ag.monitors.lat_sum.cpu().numpy()[:cap].astype(np.uint64), o.lat_sum[:cap], per_instance=False)
            cmp('lat_cnt', ag.monitors.lat_cnt.cpu().numpy()[:cap].astype(np.uint64), o.lat_cnt[:cap], per_instance=False)
            cmp('resp_cnt', ag.monitors.resp_cnt.cpu().numpy()[:cap].astype(np.uint64), o.resp_cnt[:cap], per_instance=False)
            if not ps and not np.allclose(ag.monitors.reward_sum.cpu().numpy()[:cap], o.reward_sum[:cap],
                               rtol=1e-12, atol=1e-12):
                bad.append('reward_sum')
            cmp('Q', ag._q.cpu().numpy().astype(np.float64).reshape(o.Q.shape), o.Q)
            if c['kind'] == 'dynaq':
                cmp('M.states', np.asarray(ag.M.states).reshape(o.MS.shape), o.MS)
                cmp('M.terminals', np.asarray(ag.M.terminals).reshape(o.MT.shape), o.MT)
                cmp('M.rewards', np.asarray(ag.M.rewards, dtype=np.float64).reshape(o.MR.shape), o.MR)
            else:
                cmp('log_len', ag.inst[:, 6].cpu().numpy(), o.inst['log_len'].astype(np.int32))
            cmp('lat_trace', ag.monitors.lat_trace.cpu().numpy()[:, :total], o.lat_trace[:, :total])
            got = ag.inst.cpu().numpy()
            cmp('state', got[:, 0], o.inst['state'])
            cmp('trial', got[:, 2], o.inst['trial'])
            cmp('ctr_env', got[:, 3], o.inst['ctr_env'].astype(np.int32))
            cmp('ctr_policy', ag.policy.counter.cpu().numpy().astype(np.int64),
                o.inst['ctr_policy'].astype(np.int64))
            if c['kind'] == 'dynaq':
                cmp('ctr_memory', got[:, 5], o.inst['ctr_memory'].astype(np.int32))
    else:
        ag = SR(env.observation_space, env.action_space, EpsilonGreedy(col(2)),
                learning_rate=col(0), gamma=col(1))
        ag.track_instances = True
        ag.stream_rows = c['stream_rows']
        if mask is not None:
            ag.mask_actions = True
            ag.action_mask = mask.copy()
        session(ag, env, c, c['trials'], c['steps'], 0)
        if c['second']:
            session(ag, env, c, c['trials'], c['steps'], 0)
        torch.cuda.synchronize()
        for k, (alpha, gamma, eps, mlr) in enumerate(combos):
            sel[0] = np.flatnonzero(which == k)
            o = c_oracle.SROracle(ow, c['n'], c['env_seed'], True, instance_base=c['base'],
                                  alpha=alpha, gamma=gamma, epsilon=eps, trial_cap=total,
                                  action_mask=mask)
            o.run(c['trials'], c['steps'])
            if c['second']:
                o.run(total, c['steps'])
            cmp('SR', ag._sr.cpu().numpy().astype(np.float64), o.SR)
            cmp('T', ag._T.cpu().numpy().astype(np.int64), o.T)
            cmp('R', ag._rw.cpu().numpy().astype(np.float64), o.RW)
            cmp('lat_trace', ag.monitors.lat_trace.cpu().numpy()[:, :total], o.lat_trace[:, :total])
            got = ag.inst.cpu().numpy()
            cmp('state', got[:, 0], o.inst['state'])
            cmp('ctr_env', got[:, 3], o.inst['ctr_env'].astype(np.int32))
            cmp('ctr_policy', got[:, 4], o.inst['ctr_policy'].astype(np.int32))
    return bad


def main() -> int:
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    failed, t0 = [], time.time()
    kinds = {}
    for seed in range(first, first + count):
        c = draw_case(seed)
        kinds[c['kind']] = kinds.get(c['kind'], 0) + 1
        for key in ('psets', 'chunk', 'test_trials', 'extra_worlds', 'mask', 'general', 'episodic'):
            if c.get(key) is not None and c.get(key) is not False and c.get(key) != 0:
                STATS[key] = STATS.get(key, 0) + 1
        try:
            bad = run_case(c)
        except Exception as e:      # a refused configuration is a finding too
            bad = ['%s: %s' % (type(e).__name__, e)]
        if bad:
            failed.append(seed)
            print('MISMATCH', bad, describe(c), flush=True)
        if (seed - first) % 20 == 19:
            print('... %d cases, %d failing, %.0f s' % (seed - first + 1, len(failed), time.time() - t0),
                  flush=True)
    print('cases %d (%s), failing %d: %s; exercised: %s' % (count, kinds, len(failed), failed, STATS))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
