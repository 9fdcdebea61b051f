"""TEST INFRASTRUCTURE — not part of the product path.

Single-instance NumPy restatement of the reference's hot loop, written against
the compact world tables the build uses (``next[S,4]``, ``reward[S]``,
``terminal[S]``, ``starts``) instead of the dense one-hot ``sas`` tensor.

It keeps the reference's per-step Python call pattern (one ``np.amax`` per TD
update, cumsum/searchsorted action draw, one vector draw per replay batch), so
it doubles as the single-core CPU baseline that ``bench.py`` times on the GPU
host.  Every expression that decides a dtype is spelled the way
the reference spells it, so running it with float32 tables reproduces the
promotion behaviour of the reference under the same coercion (NumPy >= 2):
online TD in float32, planning TD in float64 with a float32 store.

Reference lines restated (relative to /root/reference/src/cobel):
  env step / reset ........ interface/gridworld.py:115-126, :142
  epsilon-greedy .......... policy/greedy.py:58, :77-86
  model store / sample .... memory/dyna_q.py:92-96, :137-155
  Dyna-Q loop, TD ......... agent/dyna_q.py:164-215, :290-299, :327-330
  QAgent loop, replay ..... agent/q.py:183-228, :305-313, :353-354
  SR loop, update, q ...... agent/sr.py:155-197, :267-284, :302-308
  escape latency .......... monitor/behavior.py:82-85
Pinned against golden vectors captured from the real reference
(tests/golden/*.npz, made by tests/golden/gen_golden.py).
"""
from __future__ import annotations

import numpy as np


class RefGridworld:
    """Gridworld on compact tables (gridworld.py:76-145).  With ``world['sas']`` (dense
    ``[S, A, S]`` rows that are distributions, ``world['deterministic']`` off) the successor is
    DRAWN from the row as the reference does (gridworld.py:119-123): one double of the env stream
    per step, next to the integer draws of the trial starts."""

    def __init__(self, world: dict, rng) -> None:
        self.next = np.asarray(world['next'])
        self.reward = np.asarray(world['reward'], dtype=np.float64)
        self.terminal = np.asarray(world['terminal'])
        self.starts = np.asarray(world['starts'])
        self.sas = np.asarray(world['sas'], dtype=np.float64) if 'sas' in world else None
        self.rng = rng
        self.n_states, self.n_actions = self.next.shape
        self.current_state = 0
        self.reset()  # the constructor consumes one start draw (gridworld.py:89)

    def reset(self):
        self.current_state = int(self.starts[self.rng.integers(0, len(self.starts))])
        return self.current_state, {}

    def step(self, action):
        if self.sas is None:
            self.current_state = int(self.next[self.current_state, int(action)])
        else:
            self.current_state = int(self.rng.choice(
                np.arange(self.n_states), p=self.sas[self.current_state][int(action)]))
        s = self.current_state
        return s, self.reward[s], bool(self.terminal[s]), False, {}


class RefEpsilonGreedy:
    """greedy.py:40-88 with the Generator.choice draw written out."""

    def __init__(self, epsilon: float, rng) -> None:
        self.epsilon, self.rng = epsilon, rng

    def get_action_probs(self, v, mask=None):
        idx = np.arange(v.shape[0])
        vals = np.copy(v)
        p = np.zeros(v.shape)
        if mask is not None:
            assert np.sum(mask) > 0
            vals, idx = vals[mask], idx[mask]
        p[idx] = np.full(vals.shape, self.epsilon / vals.shape[0])
        best = np.amax(vals) == vals
        p[idx] += (1.0 - self.epsilon) * best / np.sum(best)
        return p

    def select_action(self, v, mask=None):
        cdf = np.cumsum(self.get_action_probs(v, mask))
        cdf /= cdf[-1]
        return int(cdf.searchsorted(self.rng.random(), side='right'))


class RefDynaQMemory:
    """Tabular world model (memory/dyna_q.py:62-157)."""

    def __init__(self, n_states, n_actions, rng, learning_rate=0.9, dtype=np.float64):
        self.S, self.A, self.rng, self.learning_rate = n_states, n_actions, rng, learning_rate
        self.rewards = np.zeros((n_states, n_actions), dtype=dtype)
        self.states = np.repeat(np.arange(n_states), n_actions).reshape(n_states, n_actions)
        self.terminals = np.zeros((n_states, n_actions), dtype=np.int64)

    def store(self, s, a, r, ns, nt):
        self.rewards[s, a] += self.learning_rate * (r - self.rewards[s, a])
        self.states[s, a] = ns
        self.terminals[s, a] = nt

    def sample(self, batch):
        flat = self.rng.integers(0, self.S * self.A, batch)
        pairs = np.array(np.unravel_index(flat, (self.S, self.A)))
        out = []
        for i in range(batch):       # one dictionary per experience, as the reference builds them
            s, a = pairs[0, i], pairs[1, i]
            exp = {'state': s, 'action': a, 'reward': self.rewards[s, a],
                   'next_state': self.states[s, a], 'terminal': self.terminals[s, a]}
            out.append((exp['state'], exp['action'], exp['reward'], exp['next_state'],
                        exp['terminal']))
        return out, flat


class _Tabular:
    def _td(self, s, a, r, ns, nt):
        td = r
        td += self.gamma * nt * np.amax(self.Q[ns])
        td -= self.Q[s][a]
        self.Q[s][a] += self.learning_rate * td
        return td


class RefDynaQ(_Tabular):
    """agent/dyna_q.py:106-330."""

    def __init__(self, n_states, n_actions, policy, mem_rng, learning_rate=0.99, gamma=0.99,
                 dtype=np.float64, policy_test=None):
        self.policy = policy
        self.policy_test = policy if policy_test is None else policy_test
        self.learning_rate, self.gamma = learning_rate, gamma
        self.Q = np.zeros((n_states, n_actions), dtype=dtype)
        self.M = RefDynaQMemory(n_states, n_actions, mem_rng, dtype=dtype)
        self.action_mask = np.ones((n_states, n_actions), dtype=bool)
        self.mask_actions = False
        self.episodic_replay = False
        self.current_trial = 0

    def replay(self, batch, trace=None):
        exps, flat = self.M.sample(batch)
        for s, a, r, ns, nt in exps:
            self._td(s, a, r, ns, nt)
        if trace is not None:
            trace['idx'].append(np.asarray(flat))

    def train(self, env, trials, steps, batch_size=32, no_replay=False, trace=None, learn=True):
        policy = self.policy if learn else self.policy_test
        for _ in range(trials):
            state, _ = env.reset()
            trial_reward = 0.0
            step = -1
            for step in range(steps):
                action = policy.select_action(
                    self.Q[state], self.action_mask[state] if self.mask_actions else None)
                ns, reward, end, _, _ = env.step(action)
                nt = 1 - end
                td = 0.0
                if learn:
                    self.M.store(state, action, float(reward), ns, nt)
                    td = self._td(state, action, float(reward), ns, nt)
                if trace is not None:
                    trace['sarsn'].append((state, action, float(reward), ns, nt))
                    trace['td'].append(float(td))
                state = ns
                if learn and not no_replay and not self.episodic_replay:
                    self.replay(batch_size, trace)
                trial_reward += reward
                if end:
                    break
            self.current_trial += 1
            if learn and not no_replay and self.episodic_replay:
                self.replay(batch_size, trace)
            if trace is not None:
                trace['steps'].append(step)
                trace['trial_reward'].append(float(trial_reward))
                if 'Q_trial' in trace:
                    trace['Q_trial'].append(self.Q.copy())

    def test(self, env, trials, steps, trace=None):
        self.train(env, trials, steps, trace=trace, learn=False)


class RefQAgent(_Tabular):
    """agent/q.py:115-354 for Discrete observations (dense zero table == lazy rows).
    `action_mask` / `mask_actions`: NOT in the reference's QAgent — the drop-in class offers the
    mask its Dyna-Q has (dyna_q.py:134-136, :180 `select_action(Q[state], action_mask[state])`),
    and this is that line restated so that the masked QAgent runs have a checker."""

    def __init__(self, n_states, n_actions, policy, replay_rng, learning_rate=0.9, gamma=0.8,
                 dtype=np.float64):
        self.policy, self.rng = policy, replay_rng
        self.learning_rate, self.gamma = learning_rate, gamma
        self.Q = np.zeros((n_states, n_actions), dtype=dtype)
        self.M: list = []
        self.current_trial = 0
        self.action_mask = np.ones((n_states, n_actions), dtype=bool)
        self.mask_actions = False

    def train(self, env, trials, steps=32, batch_size=32, trace=None):
        for _ in range(trials):
            state, _ = env.reset()
            trial_reward = 0.0
            step = -1
            for step in range(steps):
                action = self.policy.select_action(
                    self.Q[state], self.action_mask[state] if self.mask_actions else None)
                ns, reward, end, _, _ = env.step(action)
                exp = (state, action, float(reward), ns, 1 - end)
                self.M.append(exp)
                td = self._td(*exp)
                state = ns
                pick = self.rng.integers(0, len(self.M), batch_size)
                for i in pick:
                    self._td(*self.M[i])
                if trace is not None:
                    trace['sarsn'].append(exp)
                    trace['td'].append(float(td))
                    trace['idx'].append(np.asarray(pick))
                trial_reward += reward
                if end:
                    break
            self.current_trial += 1
            if trace is not None:
                trace['steps'].append(step)
                trace['trial_reward'].append(float(trial_reward))
                if 'Q_trial' in trace:
                    trace['Q_trial'].append(self.Q.copy())


class RefSR:
    """agent/sr.py:109-308 with ``transitions`` kept as the index table T[s,a]."""

    def __init__(self, n_states, n_actions, policy, learning_rate=0.1, gamma=0.99,
                 dtype=np.float64):
        self.policy = policy
        self.learning_rate, self.gamma = learning_rate, gamma
        self.S, self.A = n_states, n_actions
        self.SR = np.eye(n_states, dtype=dtype)
        self.T = np.repeat(np.arange(n_states), n_actions).reshape(n_states, n_actions)
        # sr.py:131-135: the one-hot [S, A, S] tensor the reference keeps; T is its index form
        self.transitions = np.zeros((n_states, n_actions, n_states))
        self.transitions[np.arange(n_states), :, np.arange(n_states)] = 1.0
        self.rewards = np.zeros(n_states, dtype=dtype)
        self.action_mask = np.ones((n_states, n_actions), dtype=bool)
        self.mask_actions = False
        self.current_trial = 0

    def retrieve_q(self, state):
        # sr.py:302-306: V = sum(SR * rewards, axis=1) over ALL rows (the reference's S^2 cost
        # pattern is kept so that this loop is a faithful CPU baseline); q[a] = V[T[s, a]]
        v = np.sum(self.SR * self.rewards, axis=1)
        q = []
        for a in range(self.A):      # sr.py:303-306: boolean lookup in the one-hot row
            q.append(np.mean(v[self.transitions[state][a] == 1]))
        return np.array(q)

    def update(self, s, a, r, ns, nt):
        d = r - self.rewards[ns]
        self.rewards[ns] += d * self.learning_rate
        self.T[s, a] = ns
        self.transitions[s][a] = np.eye(self.S)[ns]  # sr.py:274 (an S x S identity per update)
        td = np.eye(self.S)[s]                       # float64 whatever the table dtype
        if nt > 0:
            td += self.gamma * np.copy(self.SR[ns])
        else:
            td += self.gamma * np.eye(self.S)[ns]
        td -= np.copy(self.SR[s])
        self.SR[s] += self.learning_rate * td

    def train(self, env, trials, steps, trace=None):
        for _ in range(trials):
            state, _ = env.reset()
            trial_reward = 0.0
            step = -1
            for step in range(steps):
                mask = self.action_mask[state] if self.mask_actions else None
                q = self.retrieve_q(state)
                action = self.policy.select_action(q, mask)
                ns, reward, end, _, _ = env.step(action)
                self.update(state, action, float(reward), ns, 1 - end)
                if trace is not None:
                    trace['sarsn'].append((state, action, float(reward), ns, 1 - end))
                    trace['q'].append(np.asarray(q, dtype=np.float64))
                state = ns
                trial_reward += reward
                if end:
                    break
            self.current_trial += 1
            if trace is not None:
                trace['steps'].append(step)
                trace['trial_reward'].append(float(trial_reward))


def escape_latency_avg(latency: np.ndarray, max_steps: int) -> np.ndarray:
    """11-trial running nan-mean of monitor/behavior.py:82-85."""
    out = np.full(latency.shape, np.nan)
    for t in range(len(latency)):
        if np.isnan(latency[t]):
            continue
        m = np.nanmean(latency[max(0, t - 10):t + 1])
        out[t] = max_steps if np.isnan(m) else m
    return out


def new_trace(with_q: bool = False) -> dict:
    t = {'sarsn': [], 'td': [], 'idx': [], 'steps': [], 'trial_reward': [], 'q': []}
    if with_q:
        t['Q_trial'] = []
    return t
