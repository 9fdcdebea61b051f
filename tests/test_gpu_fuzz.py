"""A fixed slice of the randomised parity sweep (scripts/fuzz_vs_oracle.py): random worlds (sizes
1 x 2 to 44 x 44, terminals, rewards of either sign, walls, blocked transitions, wind, start
lists), hyper-parameters, batch sizes 0..130, 1..257 instances, masks, episodic replay, second
sessions, forced general kernel / row-streaming SR kernel — every table, counter and per-trial
monitor of DynaQ / QAgent / SR bit for bit against the C oracle.  The sweep found, in its first
300 cases: QAgent sessions with batch_size 0 did not log their experiences (the reference does,
q.py:213), and SR's transition table was not initialised for worlds of fewer than four states."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('first', [0, 60, 120, 180])
def test_random_cases_match_the_oracle(first):
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    import fuzz_vs_oracle as fz
    failed = []
    for seed in range(first, first + 60):
        case = fz.draw_case(seed)
        bad = fz.run_case(case)
        if bad:
            failed.append((fz.describe(case), bad))
    assert not failed, failed[:3]


@pytest.mark.parametrize('first', [0, 50])
def test_random_sfma_cases_match_the_restatement(first):
    """scripts/fuzz_sfma.py: random worlds, metrics (DR / SR / Euclidean), the seven replay modes
    and every memory / agent switch; three instances per case re-run by oracle/sfma_loop.py —
    latencies, every reactivation with its TD error, Q, model tables, strengths, recency."""
    import fuzz_sfma as fz
    failed = []
    for seed in range(first, first + 50):
        case = fz.draw_case(seed)
        bad = fz.run_case(case)
        if bad:
            failed.append((fz.describe(case), bad))
    assert not failed, failed[:3]


def test_random_topologies_match_the_restatement():
    """scripts/fuzz_topology.py: QAgent on random graphs with 1..8 actions (four: wavefront
    kernels, otherwise the general kernel), replay batches 0..70, against oracle/ref_loop.py."""
    import fuzz_topology as fz
    failed = []
    for seed in range(0, 120):
        case = fz.draw_case(seed)
        bad = fz.run_case(case)
        if bad:
            failed.append((fz.describe(case), bad))
    assert not failed, failed[:3]


def test_random_stochastic_worlds_match_the_restatement():
    """scripts/fuzz_stochastic.py: gridworlds whose sas rows are edited into random distributions,
    Dyna-Q / Q-learning through the general kernel against oracle/ref_loop.py drawing successors
    the way the reference does."""
    import fuzz_stochastic as fz
    failed = []
    for seed in range(0, 100):
        case = fz.draw_case(seed)
        bad = fz.run_case(case)
        if bad:
            failed.append((fz.describe(case), bad))
    assert not failed, failed[:3]


def test_random_network_agent_runs_fused_equal_torch_loops():
    """scripts/fuzz_network_agents.py: DQN on random tracks, Dyna-DQN and Dyna-DSR on random
    gridworlds — the fused HIP loops against the PyTorch-ROCm loops of the same classes: rings /
    model tables, counters and monitors exactly, every network to float64 round-off."""
    import fuzz_network_agents as fz
    failed = []
    for seed in range(0, 40):
        case = fz.draw_case(seed)
        bad = fz.run_case(case)
        if bad:
            failed.append((fz.describe(case), bad))
    assert not failed, failed[:3]


def test_network_kernels_over_input_and_output_widths():
    """A slice of scripts/fuzz_networks.py: cobel_dqn_replay at input widths the fixed tests do not
    visit, cobel_mlp_forward / cobel_mlp_fit at random (inputs, outputs) pairs."""
    import torch
    import test_gpu_mlp
    import test_gpu_parity
    for n_in, dtype_name, ddqn in ((2, 'f64', True), (13, 'f64', False), (17, 'f32', True),
                                   (31, 'f64', True), (32, 'f64', False), (12, 'f32', True)):
        test_gpu_parity.test_fused_dqn_replay_equals_torch_path(torch, dtype_name, n_in, ddqn,
                                                                f32_atol=1e-4)
    for D, O, dtype_name in ((1, 1, 'f64'), (1, 32, 'f64'), (32, 1, 'f64'), (7, 29, 'f64'),
                             (31, 18, 'f32'), (19, 5, 'f64'), (30, 31, 'f64')):
        test_gpu_mlp.test_mlp_forward_and_fit_match_pytorch(torch, D, O, dtype_name, f32_atol=1e-4)


def test_sweep_regressions():
    """The two cases the sweep caught, spelled out."""
    import numpy as np
    import torch
    from cobel_amd.agent import SR, QAgent
    from cobel_amd.interface import Gridworld
    from cobel_amd.misc.gridworld_tools import make_gridworld
    from cobel_amd.policy import EpsilonGreedy
    # SR on a three-state corridor: T starts as the identity for every (state, action)
    world = make_gridworld(1, 3, terminals=[0], rewards=np.array([[0, 1.0]]), goals=[0])
    env = Gridworld(world, n_envs=5, seed=1)
    sr = SR(env.observation_space, env.action_space, EpsilonGreedy(1.0))
    sr.train(env, 1, 1)
    T = sr._T.cpu().numpy()
    untouched = T[:, 0, :]                    # the terminal state is never left
    assert (untouched == 0).all() and set(np.unique(T[:, 2, :])) <= {1, 2}
    assert (T[:, 1, :] <= 2).all() and (T[:, 1, 1] == 1).all() and (T[:, 1, 3] == 1).all()
    # QAgent without replay still fills its memory, and a later session replays from all of it
    world = make_gridworld(3, 3, terminals=[8], rewards=np.array([[8, 1.0]]), goals=[8])
    for n in (1, 70):                         # wave-per-instance and lane-per-instance kernels
        env = Gridworld(world, n_envs=n, seed=3)
        ag = QAgent(env.observation_space, env.action_space, EpsilonGreedy(0.3))
        ag.track_instances = True
        ag.train(env, 4, 12, 0)
        torch.cuda.synchronize()
        steps = ag.monitors.lat_trace[:, :4].cpu().numpy() + 1
        assert np.array_equal(ag.inst[:, 6].cpu().numpy(), steps.sum(axis=1))
        assert len(ag.M) == int(steps[0].sum())
        first = ag.M[0]
        assert set(first) == {'state', 'action', 'reward', 'next_state', 'terminal'}
        ag.log_experiences = False            # opt out: the log stops growing at batch_size 0
        ag.train(env, 2, 12, 0)
        assert len(ag.M) == int(steps[0].sum())
