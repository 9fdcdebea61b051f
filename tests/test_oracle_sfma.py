"""CPU tests: pin the SFMA restatement (oracle/sfma_loop.py) against the golden runs captured from
the real reference (tests/golden/sfma_traces.npz): similarity metrics, every replay mode and
switch, float64 == the reference as shipped, float32 == the reference with float32 Q / M.rewards."""
import numpy as np
import pytest

from conftest import SEED
from oracle import sfma_loop
from sfma_common import sfma_case, sfma_cases


@pytest.fixture(scope='module')
def Z(golden):
    return golden('sfma_traces')


def test_metrics_match_reference(Z):
    for wname in ('sfma_5x5', 'sfma_6x7'):
        w = {k: Z['world/%s/%s' % (wname, k)] for k in ('next', 'height', 'width',
                                                         'invalid_transitions')}
        W, H = int(w['width']), int(w['height'])
        nxt = w['next'].astype(np.int64)
        inv = [tuple(t) for t in w['invalid_transitions']]
        assert np.array_equal(sfma_loop.metric_euclidean(W, H), Z['metric/%s/Euclidean' % wname])
        # the inverses go through LAPACK: same library here, but keep a tolerance for other hosts
        np.testing.assert_allclose(sfma_loop.metric_sr(nxt, 0.9), Z['metric/%s/SR' % wname],
                                   rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(sfma_loop.metric_dr(W, H, nxt, 0.9, inv),
                                   Z['metric/%s/DR' % wname], rtol=1e-12, atol=1e-14)


def test_sfma_restatement_reproduces_reference(Z):
    names = sfma_cases(Z)
    assert len(names) == 20
    for name in names:
        g, world, D, opts = sfma_case(Z, name)
        inst, f32, trials, steps, B = [int(x) for x in g('cfg')]
        ag, env = sfma_loop.run_case(world, D, SEED, inst, bool(f32), opts['mode'], opts, trials,
                                     steps, B)
        sarsn = np.array(ag.sarsn, dtype=np.float64).reshape(-1, 5)
        assert np.array_equal(sarsn[:, 0], g('state')), name
        assert np.array_equal(sarsn[:, 1], g('action')), name
        assert np.array_equal(sarsn[:, 2], g('reward')), name
        assert np.array_equal(sarsn[:, 3], g('next_state')), name
        assert np.array_equal(sarsn[:, 4], g('nonterminal')), name
        assert np.array_equal(np.array(ag.tds), g('td')), name
        assert np.array_equal(np.array(ag.steps), g('steps')), name
        assert np.array_equal(np.array(ag.trial_reward), g('trial_reward')), name
        rp = np.array(ag.replayed, dtype=np.float64).reshape(-1, 8)
        for col, key in enumerate(('rp_trial', 'rp_kind', 'rp_state', 'rp_action', 'rp_reward',
                                   'rp_next', 'rp_nonterminal', 'rp_td')):
            assert np.array_equal(rp[:, col], g(key), equal_nan=True), (name, key)
        assert np.array_equal(np.array(ag.modes), g('replay_mode')), name
        assert np.array_equal(np.array(ag.td_trial), g('td_acc')), name
        assert np.array_equal(np.array(ag.Q_trial), g('Q_trial')), name
        assert np.array_equal(np.array(ag.Q, dtype=np.float64), g('Q')), name
        assert np.array_equal(np.array(ag.M.rewards, dtype=np.float64), g('M_rewards')), name
        assert np.array_equal(ag.M.states, g('M_states')), name
        assert np.array_equal(ag.M.terminals, g('M_terminals')), name
        for k in 'CTI':
            assert np.array_equal(getattr(ag.M, k), g(k)), (name, k)
        assert sfma_loop.MODES.index(ag.M.mode) == int(g('final_mode')), name
        ctr = [env.rng.index, ag.policy.rng.index, ag.M.rng.index, ag.rng.index]
        assert ctr == list(g('ctr')), name


def test_float32_run_tracks_float64_reference(Z):
    """float32 tables against the float64 reference: identical trajectories and replays on the
    fixture pairs, Q within 1e-6 relative to max(1, |Q|)."""
    for a, b in (('dr_default_f32', 'dr_default_f64'), ('dr_dynamic_f32', 'dr_dynamic_f64'),
                 ('w67_dr_reverse_f32', 'w67_dr_reverse_f64')):
        for k in ('state', 'action', 'steps', 'rp_state', 'rp_action', 'replay_mode'):
            assert np.array_equal(Z[a + '/' + k], Z[b + '/' + k]), (a, k)
        qa, qb = Z[a + '/Q'], Z[b + '/Q']
        assert np.max(np.abs(qa - qb) / np.maximum(1.0, np.abs(qb))) < 1e-6
