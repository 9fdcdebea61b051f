import sys, time
sys.path.insert(0, 'cobel-rl_amd')
import numpy as np, torch
from cobel_amd.agent import DynaQ, SFMA
from cobel_amd.interface import Gridworld
from cobel_amd.misc.gridworld_tools import make_open_field
from cobel_amd.policy import EpsilonGreedy
for cbs, name in (({'on_step_end': [lambda l: None]}, 'per-step'), ({'on_trial_end': [lambda l: None]}, 'per-trial'), (None, 'one launch')):
    env = Gridworld(make_open_field(5, 5, 0, 1), seed=1)
    ag = DynaQ(env.observation_space, env.action_space, EpsilonGreedy(0.1), custom_callbacks=cbs)
    ag.train(env, 5, 50, 32)
    torch.cuda.synchronize(); t = time.perf_counter(); s0 = ag.env_steps()
    ag.train(env, 100, 50, 32)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, '%.0f env-steps/s' % ((ag.env_steps() - s0) / dt), '%.1f ms' % (dt * 1e3))
