"""TEST INFRASTRUCTURE — not part of the product path.

Single-instance NumPy restatement of the reference's SFMA agent (Dyna-Q with Spatial structure
and Frequency-weighted Memory Access) on the compact world tables of the build, the checker for
``cobel_sfma_run`` and the CPU baseline of bench.py's SFMA leg.

Reference lines restated (relative to /root/reference/src/cobel):
  similarity metrics ....... memory/utils/metrics.py:28-61 (Euclidean), :63-105 (SR),
                             :107-268 (DR)
  memory store ............. memory/sfma.py:195-236
  replay (reactivation) .... memory/sfma.py:238-347, softmax :349-372
  random batches ........... memory/sfma.py:374-416
  agent loop ............... agent/sfma.py:233-334 (train), :336-396 (test)
  replay + TD .............. agent/sfma.py:398-458
Every expression that decides a dtype is spelled the way the reference spells it, so float32
``Q`` / ``M.rewards`` reproduce the reference's promotion behaviour under the same coercion:
online TD in float32, replayed TD in float64 stored into float32, accumulated |TD| float32 until
the first replayed update of a trial, strengths / recency / inhibition / priorities float64.

Pinned against tests/golden/sfma_traces.npz (captured from the real reference by
tests/golden/gen_golden.py: 20 runs over metrics, replay modes and switches).
"""
from __future__ import annotations

import numpy as np

from .ref_loop import RefEpsilonGreedy, RefGridworld  # noqa: F401  (re-exported for the tests)

MODES = ('default', 'reverse', 'forward', 'blend_forward', 'blend_reverse', 'interpolate',
         'sweeping')


# ---------------------------------------------------------------------------------------------
# similarity metrics on compact tables
# ---------------------------------------------------------------------------------------------
def policy_average(next_table: np.ndarray) -> np.ndarray:
    """``sum(sas, axis=1) / A`` for a deterministic world: T[s, next[s, a]] += 1 / A."""
    S, A = next_table.shape
    T = np.zeros((S, S))
    for a in range(A):          # same accumulation order as the sum over the action axis
        np.add.at(T, (np.arange(S), next_table[:, a]), 1.0)
    return T / A


def metric_euclidean(width: int, height: int) -> np.ndarray:
    """metrics.py:46-57: D[s1, s2] = exp(-|coords(s1) - coords(s2)|)."""
    S = width * height
    D = np.zeros((S, S))
    for s1 in range(S):
        c1 = np.array(divmod(s1, width))
        for s2 in range(s1, S):
            c2 = np.array(divmod(s2, width))
            D[s1, s2] = D[s2, s1] = np.exp(-np.sqrt(np.sum((c1 - c2) ** 2)))
    return D


def metric_sr(next_table: np.ndarray, gamma: float) -> np.ndarray:
    """metrics.py:97-105: (I - gamma * T)^-1 under the uniform policy."""
    T = policy_average(next_table)
    return np.linalg.inv(np.eye(T.shape[0]) - gamma * T)


def open_field_transitions(width: int, height: int) -> np.ndarray:
    """metrics.py:240-268: uniform-policy transition matrix of the wall-free grid."""
    S = width * height
    T = np.zeros((S, S))
    for s in range(S):
        h, w = divmod(s, width)
        for (hh, ww) in ((h, max(0, w - 1)), (max(0, h - 1), w), (h, min(width - 1, w + 1)),
                         (min(height - 1, h + 1), w)):
            T[s][hh * width + ww] += 0.25
    return T


def metric_dr(width: int, height: int, next_table: np.ndarray, gamma: float,
              invalid_transitions, T_default=None) -> np.ndarray:
    """metrics.py:176-214: default representation, corrected by a low-rank (Woodbury) update
    for the rows of the states that walls touch."""
    S = width * height
    T0 = open_field_transitions(width, height) if T_default is None else T_default
    D0 = np.linalg.inv(np.eye(S) - gamma * T0)
    T_new = policy_average(next_table)
    B = np.zeros(T_new.shape)
    if len(invalid_transitions) > 0:
        rows = np.unique(np.array(invalid_transitions)[:, 0])
        L, L0 = np.eye(S) - gamma * T_new, np.eye(S) - gamma * T0
        delta = L[rows] - L0[rows]
        alpha = np.linalg.inv(np.eye(rows.shape[0]) + np.matmul(delta, D0[:, rows]))
        B = np.matmul(np.matmul(D0[:, rows], alpha), np.matmul(delta, D0))
    return D0 - B


# ---------------------------------------------------------------------------------------------
# memory
# ---------------------------------------------------------------------------------------------
class RefSFMAMemory:
    """memory/sfma.py:142-416 with ``metric.D`` passed in as a matrix."""

    def __init__(self, D, n_states, n_actions, rng, decay_inhibition=0.9, decay_strength=1.0,
                 learning_rate=0.9, dtype=np.float64):
        self.D = np.asarray(D, dtype=np.float64)
        self.S, self.A, self.rng = n_states, n_actions, rng
        self.decay_inhibition, self.decay_strength = decay_inhibition, decay_strength
        self.decay_recency, self.learning_rate, self.beta = 0.9, learning_rate, 20
        self.reward_mod_local = self.reward_mod = self.state_mod = False
        self.rewards = np.zeros((n_states, n_actions), dtype=dtype)
        self.states = np.repeat(np.arange(n_states), n_actions).reshape(n_states, n_actions)
        self.terminals = np.zeros((n_states, n_actions), dtype=np.int64)
        self.C = np.zeros(n_states * n_actions)
        self.T = np.zeros(n_states * n_actions)
        self.I = np.zeros(n_states)
        self.C_step = self.I_step = 1.0
        self.R_threshold = 10.0 ** -6
        self.deterministic = self.recency = self.C_normalize = self.D_normalize = False
        self.R_normalize = True
        self.mode = 'default'
        self.reward_modulation, self.blend = 1.0, 0.1
        self.interpolation_fwd = self.interpolation_rev = 0.5

    # memory/sfma.py:195-236 (error modulation needs experience['td'], which SFMA.train has not
    # computed yet when it calls store — the reference raises KeyError; not restated)
    nan_ratings = 0

    def store(self, s, a, r, ns, nt):
        j = self.S * a + s
        self.rewards[s][a] += self.learning_rate * (r - self.rewards[s][a])
        self.states[s][a] = ns
        self.terminals[s][a] = nt
        self.C *= self.decay_strength
        self.C[j] += self.C_step
        self.T *= self.decay_recency
        self.T[j] = 1.0
        if self.reward_mod_local:
            self.C[j] += r * self.reward_modulation
        if self.reward_mod:
            self.C += r * np.tile(self.D[s], self.A) * self.reward_modulation
        if self.state_mod:
            self.C[[s + self.S * k for k in range(self.A)]] += 1.0

    def similarity(self, cur, nxt):
        """The mode-dependent similarity vector over experiences j = a * S + s (:284-307)."""
        flat_next = self.states.flatten(order='F')
        D = np.tile(self.D[cur], self.A)
        if self.D_normalize:
            D /= np.amax(D)
        if self.mode == 'forward':
            D = np.tile(self.D[nxt], self.A)
        elif self.mode == 'reverse':
            D = D[flat_next]
        elif self.mode == 'blend_forward':
            D += self.blend * np.tile(self.D[nxt], self.A)
        elif self.mode == 'blend_reverse':
            D += self.blend * D[flat_next]
        elif self.mode == 'interpolate':
            D = (self.interpolation_fwd * np.tile(self.D[nxt], self.A)
                 + self.interpolation_rev * D[flat_next])
        elif self.mode == 'sweeping':
            D = np.tile(self.D[nxt], self.A)[flat_next]
        return D

    def softmax(self, data, offset, beta):
        e = np.exp(data * beta) + offset
        if np.sum(e) == 0:
            e.fill(1)
        else:
            e /= np.sum(e)
        return e

    def replay(self, length, current_state=None):
        action = int(self.rng.integers(self.A))
        if current_state is None:
            P = np.clip(self.C, a_min=0, a_max=None) / np.sum(np.clip(self.C, a_min=0, a_max=None))
            exp = self.rng.choice(np.arange(0, P.shape[0]), p=P)
            current_state = exp % self.S
            action = int(exp / self.S)
        next_state = self.states[current_state, action]
        self.I *= 0
        out = []
        for _ in range(length):
            C = np.copy(self.C)
            if self.C_normalize:
                C /= np.amax(C)
            D = self.similarity(current_state, next_state)
            R = C * D * (1 - np.tile(self.I, self.A))
            if self.recency:
                R *= self.T
            R[R < self.R_threshold] = 0.0
            if np.sum(R) == 0.0:
                break
            if self.R_normalize:
                R /= np.amax(R)
            if np.isnan(R).any():
                # 0 / 0 from C_normalize while every strength is still zero.  The reference goes on:
                # argmax of NaNs is experience 0 (deterministic) or rng.choice raises ValueError.
                # The build ends the replay here instead; the count lets
                # the parity sweeps tell such runs apart.
                self.nan_ratings += 1
            exp = np.argmax(R)
            if not self.deterministic:
                probs = self.softmax(R, -1, self.beta)
                probs = probs / np.sum(probs)
                exp = self.rng.choice(np.arange(0, probs.shape[0]), p=probs)
            action = int(exp / self.S)
            current_state = exp - (action * self.S)
            next_state = self.states[current_state][action]
            self.I *= self.decay_inhibition
            self.I[current_state] = min(float(self.I[current_state] + self.I_step), 1.0)
            out.append([current_state, action, self.rewards[current_state][action], next_state,
                        self.terminals[current_state][action]])
        return out

    def retrieve_random_batch(self, n, mask):
        probs = np.ones(self.S * self.A) * mask.astype(int)
        probs /= np.sum(probs)
        idx = self.rng.choice(np.arange(self.S * self.A), n, p=probs)
        ss, aa = np.unravel_index(idx, (self.S, self.A), order='F')
        return [[s, a, self.rewards[s][a], self.states[s][a], self.terminals[s][a]]
                for s, a in zip(ss, aa)]


# ---------------------------------------------------------------------------------------------
# agent
# ---------------------------------------------------------------------------------------------
class RefSFMA:
    """agent/sfma.py:174-458.  ``trace`` collects what the reference hands to its callbacks."""

    def __init__(self, n_states, n_actions, policy, memory, policy_test=None, learning_rate=0.99,
                 gamma=0.99, rng=None, dtype=np.float64):
        self.policy = policy
        self.policy_test = policy if policy_test is None else policy_test
        self.rng = rng
        self.learning_rate, self.gamma = learning_rate, gamma
        self.Q = np.zeros((n_states, n_actions), dtype=dtype)
        self.M = memory
        self.action_mask = np.ones((n_states, n_actions), dtype=bool)
        self.mask_actions = False
        self.nb_replays = 1
        self.random = self.dynamic = self.start_replay = False
        self.td = 0.0
        self.current_trial = 0
        self.steps, self.trial_reward, self.sarsn, self.tds = [], [], [], []
        self.replayed, self.modes, self.td_trial, self.Q_trial = [], [], [], []

    def update_q(self, s, a, r, ns, nt):
        mask = np.arange(self.Q.shape[1])
        if self.mask_actions:
            mask = self.action_mask[ns]
        td = r
        td += self.gamma * nt * np.amax(self.Q[ns][mask])
        td -= self.Q[s][a]
        self.Q[s][a] += self.learning_rate * td
        self.td += np.abs(td)
        return td

    def replay(self, batch, state):
        if self.random:
            mask = np.ones(np.prod(self.Q.shape))
            if self.mask_actions:
                mask = np.copy(self.action_mask).flatten(order='F')
            exps = self.M.retrieve_random_batch(batch, mask)
        else:
            exps = self.M.replay(batch, state)
        for e in exps:
            e.append(self.update_q(*e))
        return exps

    def _select(self, pol, state):
        return pol.select_action(self.Q[state], self.action_mask[state] if self.mask_actions else None)

    def train(self, env, trials, steps, batch=32, no_replay=False):
        for _ in range(trials):
            last = None
            mode_log = self.M.mode
            state, _ = env.reset()
            if self.start_replay:
                for e in self.M.replay(batch, state):
                    self.replayed.append((self.current_trial, 1, *e, np.nan))
            trial_reward = 0
            for step in range(steps):
                action = self._select(self.policy, state)
                ns, reward, end, _, _ = env.step(action)
                nt = 1 - end
                self.M.store(state, int(action), float(reward), ns, nt)
                td = self.update_q(state, int(action), float(reward), ns, nt)
                self.sarsn.append((state, int(action), float(reward), ns, nt))
                self.tds.append(float(td))
                state = ns
                trial_reward += reward
                if end:
                    last = ns
                    break
            self.current_trial += 1
            if not no_replay:
                if self.dynamic:
                    p_mode = 1 / (1 + np.exp(-(self.td * 5 - 2)))
                    mode_log = ['reverse', 'default'][
                        self.rng.choice(np.arange(2), p=np.array([p_mode, 1 - p_mode]))]
                    self.M.mode = mode_log
                    self.td = 0.0
                for _ in range(self.nb_replays):
                    for e in self.replay(batch, last):
                        self.replayed.append((self.current_trial - 1, 0, *e))
                self.M.T.fill(0)
            self.steps.append(step)
            self.trial_reward.append(float(trial_reward))
            self.modes.append(MODES.index(mode_log))
            self.td_trial.append(float(self.td))
            self.Q_trial.append(np.array(self.Q, dtype=np.float64))

    def test(self, env, trials, steps):
        for _ in range(trials):
            state, _ = env.reset()
            trial_reward = 0
            for step in range(steps):
                # agent/sfma.py:369: test() draws from self.policy — policy_test is stored by the
                # constructor but never consulted (quirk of the reference, kept)
                action = self._select(self.policy, state)
                ns, reward, end, _, _ = env.step(action)
                self.sarsn.append((state, int(action), float(reward), ns, 1 - end))
                self.tds.append(0.0)
                state = ns
                trial_reward += reward
                if end:
                    break
            self.current_trial += 1
            self.steps.append(step)
            self.trial_reward.append(float(trial_reward))
            self.modes.append(MODES.index(self.M.mode))
            self.td_trial.append(float(self.td))
            self.Q_trial.append(np.array(self.Q, dtype=np.float64))


def run_case(world: dict, D, seed: int, inst: int, f32: bool, mode: str, opts: dict, trials: int,
             steps: int, batch: int, eps: float = 0.1):
    """Build env + policy + memory + agent on the build's streams and run the schedule the golden
    generator ran: train, optional no-replay train, optional test."""
    from .philox import (STREAM_AGENT, STREAM_AUX, STREAM_ENV, STREAM_MEMORY, STREAM_POLICY,
                         TapeRNG)
    dt = np.float32 if f32 else np.float64
    # (double_sub: a world whose rows are distributions draws one double per step from sub-stream 1
    #  of the env stream; a table world never asks for one)
    env = RefGridworld(world, TapeRNG(seed, inst, STREAM_ENV, double_sub=1))
    S = env.n_states
    mem = RefSFMAMemory(D, S, 4, TapeRNG(seed, inst, STREAM_MEMORY, double_sub=1), dtype=dt)
    pol = RefEpsilonGreedy(eps, TapeRNG(seed, inst, STREAM_POLICY))
    pol_test = (RefEpsilonGreedy(0.0, TapeRNG(seed, inst, STREAM_AUX))
                if opts.get('test_trials') else None)
    ag = RefSFMA(S, 4, pol, mem, pol_test, rng=TapeRNG(seed, inst, STREAM_AGENT), dtype=dt)
    mem.mode = mode
    for k in ('recency', 'C_normalize', 'D_normalize', 'R_normalize', 'deterministic',
              'reward_mod_local', 'reward_mod', 'state_mod', 'reward_modulation', 'beta',
              'decay_inhibition', 'decay_strength'):
        if k in opts:
            setattr(mem, k, opts[k])
    for k in ('dynamic', 'random', 'start_replay', 'nb_replays'):
        if k in opts:
            setattr(ag, k, opts[k])
    if opts.get('mask') is not None and opts.get('mask') is not False:
        ag.mask_actions = True
        ag.action_mask = np.asarray(opts['mask'], dtype=bool)
    ag.train(env, trials, steps, batch)
    if opts.get('noreplay_trials'):
        ag.train(env, opts['noreplay_trials'], steps, batch, True)
    if opts.get('test_trials'):
        ag.test(env, opts['test_trials'], steps)
    return ag, env
