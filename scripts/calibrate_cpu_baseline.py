"""Calibration of the cpu_baseline 'port' (oracle/ref_loop.py) against the REAL reference, run in
the build container where both exist.  Prints env-steps/s of both on the C1, C3
and C4 analogues, single process, one core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
import gen_golden as G   # loads the reference through the import shim
from cobel.agent import DynaQ
from cobel.agent.sr import SR
from cobel.interface import Gridworld
from cobel.policy import EpsilonGreedy
from cobel.misc import gridworld_tools as gt
from oracle import ref_loop
from oracle.philox import TapeRNG, STREAM_ENV, STREAM_POLICY, STREAM_MEMORY

def time_pair(fn_a, count_a, fn_b, count_b, slice_s=2.0, rounds=5):
    """Rates of two loops measured in ALTERNATING slices (a, b, a, b, ...): load from other
    tenants of the sandbox then hits both sides alike — a session that timed one side after the
    other drifted to ratios of 1.2-1.4 for loops that do the same work."""
    spent, done = [0.0, 0.0], [0, 0]
    for _ in range(rounds):
        for k, (fn, count) in enumerate(((fn_a, count_a), (fn_b, count_b))):
            n0, t0 = count(), time.perf_counter()
            while time.perf_counter() - t0 < slice_s:
                fn()
            spent[k] += time.perf_counter() - t0
            done[k] += count() - n0
    return done[0] / spent[0], done[1] / spent[1]

rows = []
for name, world, kind, B, steps in (('C1 5x5 Dyna-Q B=32', gt.make_open_field(5, 5, 0, 1), 'dynaq', 32, 50),
                                    ('C3-analogue 32x32 Dyna-Q B=50', gt.make_open_field(32, 32, 0, 1), 'dynaq', 50, 200),
                                    ('C4-analogue 32x32 SR', gt.make_open_field(32, 32, 0, 1), 'sr', 0, 200)):
    S = world['states']
    # real reference, its own numpy Generators
    env = Gridworld(world, rng=np.random.default_rng(0))
    pol = EpsilonGreedy(0.1, rng=np.random.default_rng(1))
    cnt = [0]
    cb = {'on_step_end': [lambda logs: cnt.__setitem__(0, cnt[0] + 1)]}
    if kind == 'dynaq':
        ag = DynaQ(env.observation_space, env.action_space, pol, custom_callbacks=cb)
        ag.M.rng = np.random.default_rng(2)
        run_ref = lambda ag=ag, env=env, steps=steps, B=B: ag.train(env, 1, steps, B)
    else:
        ag = SR(env.observation_space, env.action_space, pol, custom_callbacks=cb)
        run_ref = lambda ag=ag, env=env, steps=steps: ag.train(env, 1, steps)
    tabs = dict(next=np.argmax(world['sas'], axis=2), reward=world['rewards'], terminal=world['terminals'],
                starts=world['starting_states'])
    renv = ref_loop.RefGridworld(tabs, np.random.default_rng(0))
    rpol = ref_loop.RefEpsilonGreedy(0.1, np.random.default_rng(1))
    tr = ref_loop.new_trace()
    if kind == 'dynaq':
        rag = ref_loop.RefDynaQ(S, 4, rpol, np.random.default_rng(2))
        run_port = lambda: rag.train(renv, 1, steps, B, trace=tr)
    else:
        rag = ref_loop.RefSR(S, 4, rpol)
        run_port = lambda: rag.train(renv, 1, steps, trace=tr)
    ref, port = time_pair(run_ref, lambda: cnt[0], run_port, lambda: len(tr['sarsn']))
    rows.append((name, ref, port, port / ref))
    print('%-32s reference %8.0f  port %8.0f  ratio %.2f' % rows[-1])

# SFMA (bench.py's C6): the reference's SFMA / SFMAMemory / DR against oracle/sfma_loop.py
from cobel.agent import SFMA
from cobel.memory import SFMAMemory
from cobel.memory.utils import DR
from oracle import sfma_loop

walls = [(3, 4), (4, 3), (8, 9), (9, 8), (13, 14), (14, 13), (18, 19), (19, 18)]
world = gt.make_gridworld(5, 5, terminals=[4], rewards=np.array([[4, 10]]), goals=[4],
                          invalid_transitions=walls)
world['starting_states'] = np.array([12])
env = Gridworld(world, rng=np.random.default_rng(0))
metric = DR(5, 5, world['sas'], 0.9, world['invalid_transitions'])
cnt = [0, 0]
cb = {'on_step_end': [lambda logs: cnt.__setitem__(0, cnt[0] + 1)],
      'on_replay_end': [lambda logs: cnt.__setitem__(1, cnt[1] + len(logs['replay']))]}
ag = SFMA(env.observation_space, env.action_space, EpsilonGreedy(0.1, rng=np.random.default_rng(1)),
          SFMAMemory(metric, 25, 4, rng=np.random.default_rng(2)), custom_callbacks=cb,
          rng=np.random.default_rng(3))
ag.M.mode = 'reverse'
ag.mask_actions = True
tabs = dict(next=np.argmax(world['sas'], axis=2), reward=world['rewards'], terminal=world['terminals'],
            starts=world['starting_states'])
renv = ref_loop.RefGridworld(tabs, np.random.default_rng(0))
mem = sfma_loop.RefSFMAMemory(metric.D, 25, 4, np.random.default_rng(2))
mem.mode = 'reverse'
rag = sfma_loop.RefSFMA(25, 4, ref_loop.RefEpsilonGreedy(0.1, np.random.default_rng(1)), mem,
                        rng=np.random.default_rng(3))
rag.mask_actions = True
ref, port = time_pair(lambda: ag.train(env, 1, 50, 32), lambda: cnt[0],
                      lambda: rag.train(renv, 1, 50, 32), lambda: len(rag.sarsn))
ref_replays = cnt[1] / max(cnt[0], 1)
print('%-32s reference %8.0f  port %8.0f  ratio %.2f   (reactivations per env step: %.2f vs %.2f)'
      % ('C6 5x5 SFMA DR reverse B=32', ref, port, port / ref, ref_replays,
         len(rag.replayed) / max(len(rag.sarsn), 1)))
