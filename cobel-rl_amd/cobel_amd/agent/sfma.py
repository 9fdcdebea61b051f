"""SFMA agent — ``cobel.agent.SFMA`` (agent/sfma.py:16-474) on the fused HIP kernel.

Same constructor, ``train(interface, trials, steps, batch_size=32, no_replay=False)``, ``test``,
``predict_on_batch`` and attributes (``Q``, ``M``, ``learning_rate``, ``gamma``, ``action_mask``,
``mask_actions``, ``nb_replays``, ``random``, ``dynamic``, ``offline``, ``start_replay``, ``td``,
``current_trial``, ``stop``), plus the ``on_replay_begin`` / ``on_replay_end`` callbacks.

Behaviour kept from the reference, quirks included: ``test()`` draws its actions from ``policy``
(``policy_test`` is stored but never consulted, agent/sfma.py:369); ``offline`` is stored and
unused; the error-modulation switches of the memory raise ``KeyError('td')`` because ``train``
stores an experience before its TD error exists (agent/sfma.py:290-291, memory/sfma.py:225-232).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..spaces import Discrete
from .agent import Callbacks
from .tabular import TabularAgent

EVENT = np.dtype([('sa', '<u4'), ('next', '<u4'), ('reward', '<f4'), ('trial', '<i4'),
                  ('td', '<f8')])


class CallbacksSFMA(Callbacks):
    def on_replay_begin(self, logs: dict) -> dict:
        return self._fire('on_replay_begin', logs)

    def on_replay_end(self, logs: dict) -> dict:
        return self._fire('on_replay_end', logs)


class SFMA(TabularAgent):
    CallbacksSFMA = CallbacksSFMA
    describe_launch = None     # (cobel_tab_describe covers cobel_tab_run; SFMA has its own kernel)

    def __init__(self, observation_space, action_space, policy, memory, policy_test=None,
                 learning_rate: float = 0.99, gamma: float = 0.99, custom_callbacks=None,
                 rng=None) -> None:
        assert type(observation_space) is Discrete, 'SFMA requires a discrete observation space!'
        assert type(action_space) is Discrete, 'SFMA requires a discrete action space!'
        super().__init__(observation_space, action_space, policy, policy_test, learning_rate,
                         gamma, custom_callbacks)
        self.callbacks = CallbacksSFMA(self, custom_callbacks)
        self.rng = rng
        self.M = memory
        self.nb_replays = 1
        self.random = self.dynamic = self.offline = self.start_replay = False
        self.keep_replay_trace = False     # collect replayed experiences in `replay_events`
        self.force_general_kernel = False  # testing: skip the specialised kernels
        self.force_one_wave = False        # testing: general kernel with one wave per instance
        self.replay_events = []
        self._fired = 0
        self._trace = self._trace_len = self._cdf = self._cdf_key = None
        self.replays_done = None

    # -- tables ---------------------------------------------------------------------------------
    def _alloc_tables(self) -> None:
        lds = C.c_int32()
        _lib.check(_lib.lib().cobel_sfma_query(self.n_states, C.byref(lds)))
        self._q = torch.zeros((self.n_envs, self.n_states, 4), dtype=torch.float32,
                              device=self.device)
        self._q.copy_(torch.as_tensor(self._q_host, device=self.device).expand_as(self._q))
        self.M._bind(self.n_envs, self.device)
        self.replays_done = torch.zeros(1, dtype=torch.int64, device=self.device)

    @property
    def td(self):
        """The running sum of |TD| (agent/sfma.py:455); one value per instance when vectorised."""
        if self.M.state is None:
            return 0.0
        v = self.M.state[:, _lib.SI_TD_LO:_lib.SI_TD_HI + 1].contiguous().view(
            torch.float64).reshape(-1).cpu().numpy()
        return float(v[0]) if self.n_envs == 1 else v

    def _random_cdf(self):
        """cumsum(p) / cumsum(p)[-1] of retrieve_random_batch's masked uniform p
        (memory/sfma.py:393-397), summed sequentially on the host exactly as NumPy does."""
        mask = np.ones(4 * self.n_states)
        if self.mask_actions:
            mask = np.copy(np.asarray(self.action_mask, dtype=bool)).flatten(order='F')
        key = mask.tobytes()
        if key != self._cdf_key:
            probs = np.ones(4 * self.n_states) * mask.astype(int)
            probs /= np.sum(probs)
            cdf = np.cumsum(probs)
            cdf /= cdf[-1]
            self._cdf, self._cdf_key = torch.as_tensor(cdf, device=self.device), key
        return self._cdf

    # -- launch ---------------------------------------------------------------------------------
    def _launch(self, interface, pol, flags, trials_target, steps, budget, batch) -> None:
        M, mon = self.M, self.monitors
        if M.error_mod_local or M.error_mod:
            raise KeyError('td')       # what the reference's M.store raises inside train()
        run = _lib.SFMARun()
        run.q, run.model = _lib.ptr(self._q), _lib.ptr(M.table)
        run.strength, run.stamp = _lib.ptr(M.strength), _lib.ptr(M.stamp)
        run.inst, run.sfma_inst = _lib.ptr(self.inst), _lib.ptr(M.state)
        run.metric = _lib.ptr(M._metric_on(self.device, interface.handle.n_worlds))
        sf = 0
        for flag, on in ((_lib.SF_RANDOM, self.random), (_lib.SF_DYNAMIC, self.dynamic),
                         (_lib.SF_START_REPLAY, self.start_replay),
                         (_lib.SF_DETERMINISTIC, M.deterministic), (_lib.SF_RECENCY, M.recency),
                         (_lib.SF_C_NORMALIZE, M.C_normalize), (_lib.SF_D_NORMALIZE, M.D_normalize),
                         (_lib.SF_R_NORMALIZE, M.R_normalize),
                         (_lib.SF_REWARD_MOD_LOCAL, M.reward_mod_local),
                         (_lib.SF_REWARD_MOD, M.reward_mod), (_lib.SF_STATE_MOD, M.state_mod)):
            sf |= flag if on else 0
        if M.recency:
            tab = M._recency_table(self.device)
            run.recency_tab, run.recency_len = _lib.ptr(tab), tab.numel()
        if self.random:
            run.random_cdf = _lib.ptr(self._random_cdf())
        self._mask_dev = self._mask_bits() if (flags & _lib.F_MASK_ACTIONS) else None
        run.action_mask = _lib.ptr(self._mask_dev)
        run.lat_sum, run.lat_cnt = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt'))
        run.reward_sum = _lib.ptr(mon.raw('reward_sum'))
        run.resp_cnt = _lib.ptr(mon.raw('resp_cnt'))
        run.mon_stripes = mon.stripes
        run.lat_trace = _lib.ptr(mon.lat_trace)
        run.occupancy = _lib.ptr(mon.occupancy)
        run.steps_done, run.replays_done = _lib.ptr(mon.steps_done), _lib.ptr(self.replays_done)
        run.last_exp = _lib.ptr(self._last_exp) if budget == 1 else None
        want_trace = self.keep_replay_trace or self.callbacks.has('on_replay_begin',
                                                                  'on_replay_end')
        if want_trace and (flags & _lib.F_LEARN):
            trials_here = max(1, trials_target - int(self.inst[:, _lib.I_TRIAL].min().item()))
            cap = max(1, trials_here * (self.nb_replays + 1) * max(batch, 1))
            if self._trace is None or self._trace.shape[1] < cap * _lib.SFMA_EVENT_BYTES:
                self._trace = torch.zeros((self.n_envs, cap * _lib.SFMA_EVENT_BYTES),
                                          dtype=torch.uint8, device=self.device)
                self._trace_len = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
            self._trace_len.zero_()
            run.replay_trace, run.trace_len = _lib.ptr(self._trace), _lib.ptr(self._trace_len)
            run.trace_cap = self._trace.shape[1] // _lib.SFMA_EVENT_BYTES
        run.n, run.trial_cap = self.n_envs, mon.cap
        run.instance_base = interface.instance_base
        run.flags, run.sfma_flags = flags, sf
        run.trials_target, run.steps_per_trial, run.step_budget = trials_target, steps, budget
        run.batch, run.nb_replays = batch, self.nb_replays
        run.alpha, run.gamma, run.epsilon = self.learning_rate, self.gamma, pol.epsilon
        run.model_lr = M.learning_rate
        run.decay_inhibition, run.decay_strength = M.decay_inhibition, M.decay_strength
        run.c_step, run.i_step = M.C_step, M.I_step
        run.r_threshold, run.beta = M.R_threshold, M.beta
        run.reward_modulation, run.blend = M.reward_modulation, M.blend
        run.interp_fwd, run.interp_rev = M.interpolation_fwd, M.interpolation_rev
        run.seed = interface.seed
        M._sync_mode()
        self.inst[:, _lib.I_CTR_MEMORY] = M.counter
        _lib.check(_lib.lib().cobel_sfma_run(interface.handle.ptr, C.byref(run),
                                             _lib.current_stream(self.device)))
        M.counter.copy_(self.inst[:, _lib.I_CTR_MEMORY])
        if self.dynamic:
            M._read_mode()
        if run.replay_trace:
            self._collect_trace()

    def _collect_trace(self) -> None:
        lens = self._trace_len.cpu().numpy()
        cap = self._trace.shape[1] // _lib.SFMA_EVENT_BYTES
        assert int(lens.max(initial=0)) <= cap, 'replay trace overflow'
        raw = self._trace.cpu().numpy()
        for i, n in enumerate(lens):
            if n:
                ev = raw[i, : int(n) * _lib.SFMA_EVENT_BYTES].view(EVENT).copy()
                self.replay_events.append((i, ev))

    @staticmethod
    def decode_events(ev) -> list:
        """Trace records -> the experience dicts of ``logs['replay']``."""
        out = []
        for e in ev:
            sa = int(e['sa'])
            d = {'state': sa & 0xFFFF, 'action': (sa >> 16) & 0xFF, 'reward': e['reward'],
                 'next_state': int(e['next']), 'terminal': (sa >> 24) & 1}
            if not (sa >> 25) & 1:
                d['td'] = float(e['td'])
            out.append(d)
        return out

    # -- hooks of the shared trial driver ---------------------------------------------------------
    def _trial_logs(self, logs: dict) -> dict:
        logs['replay_mode'] = self.M.mode
        return logs

    def _after_trial(self, logs: dict) -> dict:
        """Per-trial launches (n_envs == 1 with callbacks): hand the replays of the trial to the
        on_replay_* callbacks in the order they happened."""
        logs['replay_mode'] = self.M.mode
        pending = self.replay_events[self._fired:]
        if not self.keep_replay_trace:
            self.replay_events = []
        self._fired = len(self.replay_events)
        for _, ev in pending:
            kinds = (ev['sa'] >> 25) & 1
            cuts = np.flatnonzero(np.diff(kinds)) + 1
            for part in np.split(ev, cuts):
                # a trial's start replay and its nb_replays end replays, `batch` events each at most
                logs = self.callbacks.on_replay_begin(logs)
                logs['replay'] = self.decode_events(part)
                logs = self.callbacks.on_replay_end(logs)
        return logs

    def train(self, interface, trials: int, steps: int, batch_size: int = 32,
              no_replay: bool = False) -> None:
        extra = (_lib.F_NO_REPLAY if no_replay else 0) | \
                (_lib.F_FORCE_WAVE if self.force_general_kernel else 0) | \
                ((_lib.F_FORCE_WAVE | _lib.F_NO_PREFETCH) if self.force_one_wave else 0)
        self._session(interface, trials, steps, batch_size, True, extra)

    def test(self, interface, trials: int, steps: int) -> None:
        # agent/sfma.py:369: the test loop selects with self.policy
        self._session(interface, trials, steps, 0, False, pol=self.policy)
