"""Shared implementation of the two tabular TD agents on ``cobel_tab_run``."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .agent import FusedAgent


class TabularAgent(FusedAgent):
    """Q table [N, S, A] float32 on device (A = 4 but for the general kernel) + launch plumbing
    for ``cobel_tab_run``."""

    agent_kind = _lib.AGENT_Q
    general_actions = True     # cobel_tab_run's general kernel takes any action count
    force_general = False      # True: always the general kernel (tests, A/B comparisons)
    extra_flags = 0            # COBEL_F_* testing switches ORed into every launch (F_NO_PWG, ...)

    def __init__(self, observation_space, action_space, policy, policy_test, learning_rate,
                 gamma, custom_callbacks) -> None:
        super().__init__(observation_space, action_space, policy, policy_test, custom_callbacks)
        self.learning_rate = learning_rate
        self.gamma = gamma
        self._q = None
        self._q_host = (np.zeros((self.n_states, self.n_actions), dtype=np.float32)
                        if self.n_states is not None else None)
        self._poses = None      # node poses when the observations are a Topology's (Box)

    # -- tables -----------------------------------------------------------------------------
    def _alloc_tables(self) -> None:
        self._q = torch.zeros((self.n_envs, self.n_states, self.n_actions), dtype=torch.float32,
                              device=self.device)
        # planning / replay batches the kernels evaluated (cobel_tab_run_t.batches_done): a Dyna-Q
        # batch is drawn every learning step, but evaluated only if it can change a table
        self.batches_done = torch.zeros(1, dtype=torch.int64, device=self.device)
        # work area of a launch (cobel_tab_run_t.scratch: ticket and slice counters of the
        # persistent-workgroup Dyna-Q kernel), one per agent: agents that share a world handle may
        # be in flight together
        # (zeroed once: its abort word is only ever raised by a launch, see check_launches)
        self._scratch = torch.zeros(_lib.tab_scratch_bytes(self.n_envs) // 4, dtype=torch.int32,
                                    device=self.device)
        if self._q_host is not None:
            self._q.copy_(torch.as_tensor(self._q_host, device=self.device).expand_as(self._q))

    @property
    def Q(self):
        """``(S, 4)`` float32 NumPy snapshot for one instance (the reference's ``agent.Q``), the
        device tensor ``[N, S, 4]`` when vectorised.  Assigning broadcasts to all instances."""
        if self._q is None:
            return self._q_host
        return self._q[0].cpu().numpy() if self.n_envs == 1 else self._q

    @Q.setter
    def Q(self, value) -> None:
        if self._q is None:
            self._q_host = np.array(value, dtype=np.float32).reshape(self.n_states, self.n_actions)
        else:
            v = torch.as_tensor(np.asarray(value, dtype=np.float32) if not torch.is_tensor(value)
                                else value, device=self.device).to(torch.float32)
            self._q.copy_(v.expand_as(self._q) if v.dim() == 2 else v)

    def predict_on_batch(self, batch):
        """Q-values of a batch of observations, ``[len(batch), 4]`` (dyna_q.py:303-317)."""
        if self._poses is not None:
            idx = self._node_index(batch)
        else:
            idx = np.array(batch).astype(int)
        if self._q is None:
            return self._q_host[idx]
        if self.n_envs == 1:
            return self._q[0][torch.as_tensor(idx, device=self.device)].cpu().numpy()
        return self._q[:, torch.as_tensor(idx, device=self.device)]

    def _node_index(self, batch) -> np.ndarray:
        """Pose observations -> node indices (the reference keys Q by tuple(pose), q.py:154-155;
        an unseen pose is a KeyError there as well)."""
        if len(batch) and isinstance(batch[0], dict):     # agent/q.py:156-158
            batch = [np.concatenate([np.asarray(o, dtype=np.float64).flatten()
                                     for _, o in b.items()]) for b in batch]
        obs = np.asarray(batch, dtype=np.float64).reshape(-1, self._poses.shape[1])
        hit = (obs[:, None, :] == self._poses[None, :, :]).all(axis=2)
        if not hit.any(axis=1).all():
            raise KeyError(tuple(obs[~hit.any(axis=1)][0]))
        return hit.argmax(axis=1)

    # -- launch -----------------------------------------------------------------------------
    def _extra(self, run: _lib.TabRun) -> None:
        pass

    def _model_lr(self):
        return 0.9

    def _launch(self, interface, pol, flags, trials_target, steps, budget, batch,
                describe=None) -> None:
        mon = self.monitors
        run = _lib.TabRun()
        run.q = _lib.ptr(self._q)
        run.inst = _lib.ptr(self.inst)
        self._mask_dev = self._mask_bits() if (flags & _lib.F_MASK_ACTIONS) else None
        run.action_mask = _lib.ptr(self._mask_dev)
        run.lat_sum, run.lat_cnt = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt'))
        run.reward_sum = _lib.ptr(mon.raw('reward_sum'))
        run.resp_cnt = _lib.ptr(mon.raw('resp_cnt'))
        run.mon_stripes = mon.stripes
        run.lat_trace = _lib.ptr(mon.lat_trace)
        run.occupancy = _lib.ptr(mon.occupancy)
        run.steps_done = _lib.ptr(mon.steps_done)
        run.batches_done = _lib.ptr(self.batches_done)
        run.scratch, run.scratch_bytes = _lib.ptr(self._scratch), self._scratch.numel() * 4
        run.last_exp = _lib.ptr(self._last_exp) if budget == 1 else None
        run.n, run.trial_cap = self.n_envs, mon.cap
        run.instance_base = interface.instance_base
        run.agent = self.agent_kind
        run.flags = flags | self.extra_flags | (_lib.F_TAB_GENERAL if self.force_general else 0)
        run.trials_target, run.steps_per_trial, run.step_budget = trials_target, steps, budget
        run.batch = batch
        run.seed = interface.seed
        self._hyper(run, self.learning_rate, self.gamma, pol.epsilon, self._model_lr())
        self._extra(run)
        if describe is not None:
            _lib.check(_lib.lib().cobel_tab_describe(interface.handle.ptr, C.byref(run), describe))
            return
        # (`launch_events`: a pair of torch.cuda.Event recorded right around the library call —
        #  bench.py times the kernel without the host's preparation of its arguments)
        #  (a LIST that gets one pair appended per launch: a train() call may issue several)
        ev = getattr(self, 'launch_events', None)
        pair = None
        if ev is not None:
            import torch
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        # (the entry point SURVEY.md section 8b names for this agent: cobel_tab_run with the kind checked)
        entry = (_lib.lib().cobel_dynaq_run if self.agent_kind == _lib.AGENT_DYNAQ
                 else _lib.lib().cobel_q_run)
        _lib.check(entry(interface.handle.ptr, C.byref(run), _lib.current_stream(self.device)))
        if pair is not None:
            pair[1].record()
            ev.append(pair)

    def check_launches(self) -> None:
        """Waits for this agent's launches and raises ``CobelHipError`` if a sliced launch of the
        persistent-workgroup kernel gave up waiting for a ring entry (``cobel_tab_scratch_check``:
        a lost producer wavefront ends in an error, not in a hung GPU)."""
        scratch = getattr(self, '_scratch', None)     # (SR / SFMA keep their own tables: no slices)
        if scratch is not None:
            _lib.check(_lib.lib().cobel_tab_scratch_check(
                _lib.ptr(scratch), scratch.numel() * 4, _lib.current_stream(self.device)))

    def env_steps(self) -> int:
        self.check_launches()
        return super().env_steps()

    def describe_launch(self, interface, pol, flags, trials_target, steps, budget, batch) -> dict:
        """Which kernel ``_launch`` would take with these arguments (``cobel_tab_describe``)."""
        out = (C.c_int32 * 4)()
        TabularAgent._launch(self, interface, pol, flags, trials_target, steps, budget, batch,
                             describe=out)
        return {'kernel': int(out[0]), 'lds_bytes': int(out[1]), 'workgroups_per_cu': int(out[2]),
                'instances_per_workgroup': int(out[3])}
