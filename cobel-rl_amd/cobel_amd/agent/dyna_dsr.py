"""Dyna-DSR — ``cobel.agent.dyna_q.DynaDSR`` (agent/dyna_q.py:711-1150) on PyTorch-ROCm.

A deep successor representation: one network per action maps an observation to the (discounted)
successor features of taking that action, a reward network maps successor features to a value;
``Q(s, a) = reward_net(sr_net_a(obs[s]))`` (:1013-1020).  Experiences come from the tabular world
model of Dyna-Q, sampled uniformly over all state-action pairs (``DynaQMemory.retrieve_batch``).
One replay (:1042-1150):

    future SR / value of every sampled next state under every action's TARGET network
    target_b = obs[s_b] (or obs[s'_b] with ``use_follow_up_state``)
             + gamma * ( obs[s'_b] * (1 - follow_up) * (1 - nonterminal_b) * (1 - ignore_terminality)
                         + SR_target[best_b or mean over actions (``use_DR``)](s'_b)
                           * min(nonterminal_b + ignore_terminality, 1) )
    each action's ONLINE network trains on the samples that took that action (none: no step)
    the reward network regresses the model's reward estimates from the next observations
    targets blend towards (``target_update`` < 1) or are periodically copied from the online nets

Vectorised like ``DynaDQN``: ``n_envs`` independent agents in lockstep; the 4 x n online (and
target) SR networks are one ``StackedTorchNetwork`` of 4 n instances (index ``4 i + a``), trained
in one pass with per-instance sample masks and per-instance Adam step counts, so every network
receives exactly the update it would compute alone.  With the 64-64 ReLU networks of the
reference's demo (demo/gridworld/demo_dyna_dsr.py) a lockstep step is five kernel launches
(``_run_fused``: cobel_dqn_act in world-model mode, two cobel_mlp_forward, two cobel_mlp_fit) and
a handful of elementwise torch kernels for the targets, replayed from a HIP graph; any other
network takes the PyTorch-ROCm loop (vmap + fused Adam), same results.  Same constructor, attributes and methods as
the reference (``models_online`` / ``models_target`` are dictionaries of single-instance views).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..spaces import Discrete
from .agent import DeviceMonitors
from .dqn import _capture
from .dyna_dqn import DynaDQN, _ModelMemory


class DynaDSR(DynaDQN):
    def __init__(self, observation_space, action_space, policy, model_sr, model_reward,
                 observations=None, policy_test=None, gamma: float = 0.99, memory=None,
                 custom_callbacks=None) -> None:
        assert type(observation_space) is Discrete, 'DynaDSR requires a discrete observation space!'
        assert type(action_space) is Discrete, 'DynaDSR requires a discrete action space!'
        super().__init__(observation_space, action_space, policy, model_sr, observations,
                         policy_test, gamma, memory, custom_callbacks)
        self.n_actions = int(action_space.n)
        # dyna_q.py:791-797: every action starts from a clone of model_sr (fresh optimizers)
        self.models_target = {a: model_sr.clone() for a in range(self.n_actions)}
        self.models_online = {a: model_sr.clone() for a in range(self.n_actions)}
        self.model_reward = model_reward
        self.use_DR = False
        self.use_follow_up_state = False
        self.ignore_terminality = True
        # the fused loop forms its regression targets in one launch (cobel_dsr_targets); False: the
        # same expressions as elementwise torch kernels (what the kernel is tested against)
        self.fused_targets = True
        self._reward_net = None

    # -- binding --------------------------------------------------------------------------------
    def _bind(self, interface, slots: int) -> None:
        if self.n_envs is None:
            self.n_envs, self.device = interface.n_envs, interface.device
            n, A = self.n_envs, self.n_actions
            proto = self.models_online[0]
            proto.set_device(self.device)
            self.model_reward.set_device(self.device)
            self._online = proto.replicate(n * A)
            self._target = self.models_target[0].clone()
            self._target.set_device(self.device)
            self._target = self._target.replicate(n * A)
            # the reference's clones share model_sr's weights; views of later edits are not tracked
            self._reward_net = self.model_reward.replicate(n)
            self.dtype = next(iter(self._online.params.values())).dtype
            self.monitors = DeviceMonitors(self.device, 1, 1, False)
            self.trial = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._bind_memory(interface, slots)

    def _stacks(self):
        A = self.n_actions
        return ([(self._online, self.models_online[a], a) for a in range(A)]
                + [(self._target, self.models_target[a], a) for a in range(A)]
                + [(self._reward_net, self.model_reward, 0)])

    def _adopt_user_weights(self) -> None:
        pass    # (per-action views: edits of the single networks between runs are not tracked)

    # -- values ---------------------------------------------------------------------------------
    def _make_capturable(self) -> None:
        self._online.make_capturable()
        self._reward_net.make_capturable()

    def _per_action(self, net, obs: torch.Tensor) -> torch.Tensor:
        """obs [n, B, D] -> successor features [n, A, B, F] of the n x A networks of ``net``."""
        n, A = self.n_envs, self.n_actions
        x = obs[:, None].expand(n, A, *obs.shape[1:]).reshape(n * A, *obs.shape[1:])
        out = net.forward(x)
        return out.reshape(n, A, *out.shape[1:])

    def _values(self, sr: torch.Tensor) -> torch.Tensor:
        """successor features [n, A, B, F] -> values [n, A, B] of the reward networks."""
        n, A, B, F = sr.shape
        return self._reward_net.forward(sr.reshape(n, A * B, F)).reshape(n, A, B)

    def _q_values(self, obs: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self._values(self._per_action(self._online, obs[:, None, :]))[:, :, 0]

    def retrieve_q(self, state):
        if self._online is None:
            q = np.zeros(self.n_actions)
            for a, model in self.models_online.items():
                sr = model.predict_on_batch(self.observations[state:(state + 1)])[0]
                q[a] = self.model_reward.predict_on_batch(np.array([sr]))[0][0]
            return q
        obs = torch.as_tensor(self.observations[int(state)], device=self.device).to(self.dtype)
        q = self._q_values(obs.expand(self.n_envs, -1))
        return q[0].cpu().numpy() if self.n_envs == 1 else q

    def predict_on_batch(self, batch):
        idx = np.array(batch).astype(int)
        if self._online is None:
            q = np.zeros((idx.shape[0], self.n_actions))
            for a, model in self.models_online.items():
                q[:, a] = self.model_reward.predict_on_batch(
                    model.predict_on_batch(self.observations[idx])).flatten()
            return q
        obs = torch.as_tensor(self.observations[idx], device=self.device).to(self.dtype)
        with torch.no_grad():
            q = self._values(self._per_action(
                self._online, obs[None].expand(self.n_envs, *obs.shape))).transpose(1, 2)
        return q[0].cpu().numpy() if self.n_envs == 1 else q

    # -- learning -------------------------------------------------------------------------------
    def replay(self, batch_size: int = 32, active=None):
        n, A = self.n_envs, self.n_actions
        s, a, r, ns, nt = self.M.sample(batch_size, active)
        table = self._table.to(self.dtype)
        states, next_states = table[s], table[ns]                   # [n, B, D]
        nonterminal = (nt != 0).to(self.dtype)                      # bool(experience['terminal'])
        follow, ignore = float(self.use_follow_up_state), float(self.ignore_terminality)
        with torch.no_grad():
            future_sr = self._per_action(self._target, next_states)            # [n, A, B, F]
            if self.use_DR:
                boot_sr = future_sr.mean(dim=1)
            else:
                best = self._values(future_sr).argmax(dim=1)                    # [n, B]
                boot_sr = torch.gather(
                    future_sr, 1, best[:, None, :, None].expand(n, 1, *future_sr.shape[2:]))[:, 0]
            bootstrap = next_states * ((1.0 - follow) * (1.0 - ignore)) * (1.0 - nonterminal)[..., None]
            bootstrap = bootstrap + boot_sr * torch.clamp(nonterminal + ignore, max=1.0)[..., None]
            targets = (next_states if self.use_follow_up_state else states) + self.gamma * bootstrap
        # every action's online network trains on the samples that took that action
        took = a[:, None, :] == torch.arange(A, device=self.device)[None, :, None]   # [n, A, B]
        took = took.reshape(n * A, -1)
        has = took.any(dim=1)
        if active is not None:
            has = has & active.repeat_interleave(A)
        expand = lambda x: x[:, None].expand(n, A, *x.shape[1:]).reshape(n * A, *x.shape[1:])  # noqa: E731
        self._online.train_on_device(expand(states), expand(targets), has, took)
        self._reward_net.train_on_device(next_states, r.to(self.dtype)[..., None], active)
        self.last_update += 1
        every = None if active is None else active.repeat_interleave(A)
        if self.target_update < 1.0:
            self._target.blend_from(self._online, self.target_update, every)
        elif self.last_update == self.target_update:
            self._target.copy_from(self._online, every)
            self.last_update = 0

    def train(self, interface, trials: int, steps: int = 32, batch_size: int = 32,
              no_replay: bool = False) -> None:
        assert not self.mask_actions and not self.episodic_replay, \
            'action masks / episodic replay are not part of the accelerated DynaDSR path'
        self._no_replay = no_replay
        self._run(interface, trials, steps, batch_size, True)

    # -- fused lockstep step ----------------------------------------------------------------------
    @staticmethod
    def _mlp_ptrs(net, names):
        """(weights, biases, exp_avg w/b, exp_avg_sq w/b) tensors of a stacked 3-layer network,
        with Adam's state created where the optimizer has not run yet."""
        opt, out = net.optimizer, {k: [] for k in ('w', 'b', 'mw', 'mb', 'vw', 'vb')}
        for name in names:
            for kind, kw, km, kv in (('.weight', 'w', 'mw', 'vw'), ('.bias', 'b', 'mb', 'vb')):
                p = net.params[name + kind]
                st = opt.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
                    st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                out[kw].append(p)
                out[km].append(st['exp_avg'])
                out[kv].append(st['exp_avg_sq'])
        return out

    @staticmethod
    def _step_counts(net):
        """Per-network Adam step counts (shared with the PyTorch path's fused optimizer kernel)."""
        net._diverged = True
        counts = getattr(net, '_steps', None)
        if counts is None:
            seen = [float(st['steps'].max()) if 'steps' in st else float(st.get('step', 0.0))
                    for st in net.optimizer.state.values()]
            counts = net._steps = torch.full((net.n,), max(seen, default=0.0), dtype=torch.float64,
                                             device=net.device)
        for st in net.optimizer.state.values():
            st['steps'] = counts
        return counts

    def _fused_loop_ok(self, interface, pol, batch_size: int) -> bool:
        """The run is one the MLP kernels cover: three-layer 64-64 ReLU networks (successor
        networks D -> D with D <= 32, reward network D -> 1), MSE, Adam, blended targets,
        batches of 32, an epsilon-greedy policy on a Gridworld's one-hot observations."""
        from ..interface.gridworld import Gridworld
        from ..policy.greedy import EpsilonGreedy
        if type(self) is not DynaDSR or self.fused_loop is False or self.use_graph is True \
                or not isinstance(interface, DynaDQN._ObsView) \
                or not isinstance(interface.env, Gridworld) or interface.env.handle.stochastic \
                or type(self.M) is not _ModelMemory \
                or self.M.A != 4 or interface.table.dtype != torch.float64 \
                or interface.table.dim() != 2 or self.mask_actions or self.episodic_replay \
                or type(pol) is not EpsilonGreedy or not (self.target_update < 1.0) \
                or getattr(self, '_no_replay', False) or batch_size != 32 \
                or self.dtype not in (torch.float64, torch.float32):
            return False
        shapes = []
        for net in (self._online, self._target, self._reward_net):
            names = net._mlp3_names() if net.fused_mlp else None
            if names is None or type(net.criterion) is not torch.nn.MSELoss \
                    or getattr(net.criterion, 'reduction', '') != 'none':
                return False
            w = [net.params[k + '.weight'] for k in names]
            shapes.append((w[0].shape[2], w[0].shape[1], w[1].shape[1], w[2].shape[1]))
        if not (self._online._fused_adam_ok() and self._reward_net._fused_adam_ok()):
            return False
        D = interface.table.shape[1]
        (d0, h0, h1, o0), tgt, (dr, hr0, hr1, orr) = shapes
        f64 = int(self.dtype == torch.float64)
        return tgt == shapes[0] and d0 == D and dr == o0 and orr == 1 and \
            _lib.lib().cobel_mlp_query(d0, h0, h1, o0, 32, f64, None) == _lib.OK and \
            _lib.lib().cobel_mlp_query(dr, hr0, hr1, orr, 32, f64, None) == _lib.OK

    def _run_fused(self, interface, pol, trials: int, steps: int, batch_size: int,
                   budget: int) -> None:
        """One lockstep step = cobel_dqn_act (select from Q, env.step, model store, trial
        bookkeeping, batch draw) -> successor features of the sampled next states under the four
        TARGET networks and their values under the reward network (two cobel_mlp_forward) -> the
        targets (elementwise torch) -> one optimisation step of every online successor network on
        the samples that took its action, target blend, and the successor features of the next
        observation (cobel_mlp_fit) -> one step of the reward network and Q of the next
        observation (cobel_mlp_fit).  Same streams, counters and arithmetic as ``replay``."""
        import ctypes as C
        n, A, dev, M, mon = self.n_envs, self.n_actions, self.device, self.M, self.monitors
        first, f64, dt = self.current_trial, int(self.dtype == torch.float64), self.dtype
        env, table = interface.env, interface.table.contiguous()
        D = table.shape[1]
        names_sr, names_r = self._online._mlp3_names(), self._reward_net._mlp3_names()
        O = self._online.params[names_sr[2] + '.weight'].shape[1]
        on, tg, rw = (self._mlp_ptrs(self._online, names_sr), self._mlp_ptrs(self._target, names_sr),
                      self._mlp_ptrs(self._reward_net, names_r))
        steps_sr, steps_r = self._step_counts(self._online), self._step_counts(self._reward_net)
        q = self._q_values(interface.observe().to(dt)).contiguous()
        step = torch.zeros(n, dtype=torch.int32, device=dev)
        trew = torch.zeros(n, dtype=torch.float64, device=dev)
        active = torch.ones(n, dtype=torch.uint8, device=dev)
        stepped = torch.zeros(n, dtype=torch.uint8, device=dev)
        unused_steps = torch.zeros(n, dtype=torch.float64, device=dev)
        si = torch.zeros((n, 32), dtype=torch.int32, device=dev)
        ni = torch.zeros_like(si)
        ba = torch.zeros((n, 32), dtype=torch.int64, device=dev)
        br = torch.zeros((n, 32), dtype=dt, device=dev)
        bt = torch.zeros_like(br)
        fsr = torch.zeros((n * A, 32, O), dtype=dt, device=dev)
        val = torch.zeros((n * A, 32, 1), dtype=dt, device=dev)
        y = torch.zeros((n, 32, O), dtype=dt, device=dev)
        took = torch.zeros((n * A, 32), dtype=torch.uint8, device=dev)
        train = torch.zeros(n * A, dtype=torch.uint8, device=dev)
        sr_next = torch.zeros((n * A, 1, O), dtype=dt, device=dev)
        arange_a = torch.arange(A, device=dev)[None, :, None]

        act = _lib.DQNAct()
        act.state, act.env_ctr = _lib.ptr(env.state), _lib.ptr(env.env_ctr)
        act.obs_table, act.q = _lib.ptr(table), _lib.ptr(q)
        act.policy_ctr, act.policy_stream = _lib.ptr(pol.counter), pol.stream
        act.is_float64, act.epsilon = f64, float(pol.epsilon)
        act.memory_ctr = _lib.ptr(M.counter)
        act.trial, act.step, act.trial_reward = _lib.ptr(self.trial), _lib.ptr(step), _lib.ptr(trew)
        act.active, act.adam_steps = _lib.ptr(active), _lib.ptr(unused_steps)
        act.lat_sum, act.lat_cnt = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt'))
        act.reward_sum, act.stepped = _lib.ptr(mon.raw('reward_sum')), _lib.ptr(stepped)
        act.n, act.n_obs, act.batch = n, D, 32
        act.steps_per_trial, act.trials_target = steps, first + trials
        act.trial_cap, act.mon_stripes = mon.cap, mon.stripes
        act.instance_base, act.seed = interface.instance_base, interface.seed
        act.model_rewards, act.model_states = _lib.ptr(M.rewards), _lib.ptr(M.states)
        act.model_nonterminal, act.model_lr = _lib.ptr(M.terminals), float(M.learning_rate)
        act.n_states = M.S
        act.batch_state_index, act.batch_next_index = _lib.ptr(si), _lib.ptr(ni)
        act.batch_actions, act.batch_rewards = _lib.ptr(ba), _lib.ptr(br)
        act.batch_nonterminal = _lib.ptr(bt)

        def fill(dst, tensors):
            for k in range(3):
                dst[k] = _lib.ptr(tensors[k])

        fwd_t, fwd_r = _lib.MLPForward(), _lib.MLPForward()
        fill(fwd_t.w, tg['w']); fill(fwd_t.b, tg['b'])           # noqa: E702
        fwd_t.active, fwd_t.act_div = _lib.ptr(stepped), A
        fwd_t.in_table, fwd_t.in_index, fwd_t.in_div = _lib.ptr(table), _lib.ptr(ni), A
        fwd_t.out, fwd_t.n, fwd_t.net_div = _lib.ptr(fsr), n * A, 1
        fwd_t.n_inputs, fwd_t.n_outputs, fwd_t.is_float64 = D, O, f64
        fill(fwd_r.w, rw['w']); fill(fwd_r.b, rw['b'])           # noqa: E702
        fwd_r.active, fwd_r.act_div = _lib.ptr(stepped), A
        fwd_r.in_dense, fwd_r.in_div = _lib.ptr(fsr), 1
        fwd_r.out, fwd_r.n, fwd_r.net_div = _lib.ptr(val), n * A, A
        fwd_r.n_inputs, fwd_r.n_outputs, fwd_r.is_float64 = O, 1, f64

        def optimiser(fit, net):
            g = net.optimizer.param_groups[0]
            fit.lr, (fit.beta1, fit.beta2) = float(g['lr']), (float(b) for b in g['betas'])
            fit.eps, fit.weight_decay = float(g['eps']), float(g['weight_decay'])

        fit_sr, fit_r = _lib.MLPFit(), _lib.MLPFit()
        for dst, key in ((fit_sr.w, 'w'), (fit_sr.b, 'b'), (fit_sr.m_w, 'mw'), (fit_sr.m_b, 'mb'),
                         (fit_sr.v_w, 'vw'), (fit_sr.v_b, 'vb')):
            fill(dst, on[key])
        fill(fit_sr.w_target, tg['w']); fill(fit_sr.b_target, tg['b'])   # noqa: E702
        optimiser(fit_sr, self._online)
        fit_sr.tau = float(self.target_update)
        fit_sr.steps, fit_sr.train = _lib.ptr(steps_sr), _lib.ptr(train)
        fit_sr.active, fit_sr.act_div = _lib.ptr(stepped), A
        fit_sr.in_table, fit_sr.in_index, fit_sr.in_div = _lib.ptr(table), _lib.ptr(si), A
        fit_sr.targets, fit_sr.tgt_div, fit_sr.sample_mask = _lib.ptr(y), A, _lib.ptr(took)
        fit_sr.ep_table, fit_sr.ep_index, fit_sr.ep_div = _lib.ptr(table), _lib.ptr(env.state), A
        fit_sr.ep_rows, fit_sr.ep_out = 1, _lib.ptr(sr_next)
        fit_sr.n, fit_sr.n_inputs, fit_sr.n_outputs, fit_sr.is_float64 = n * A, D, O, f64
        for dst, key in ((fit_r.w, 'w'), (fit_r.b, 'b'), (fit_r.m_w, 'mw'), (fit_r.m_b, 'mb'),
                         (fit_r.v_w, 'vw'), (fit_r.v_b, 'vb')):
            fill(dst, rw[key])
        optimiser(fit_r, self._reward_net)
        fit_r.tau = 0.0
        fit_r.steps, fit_r.active, fit_r.act_div = _lib.ptr(steps_r), _lib.ptr(stepped), 1
        fit_r.in_table, fit_r.in_index, fit_r.in_div = _lib.ptr(table), _lib.ptr(ni), 1
        fit_r.targets, fit_r.tgt_div = _lib.ptr(br), 1
        fit_r.ep_dense, fit_r.ep_div, fit_r.ep_rows = _lib.ptr(sr_next), 1, A
        fit_r.ep_out = _lib.ptr(q)
        fit_r.n, fit_r.n_inputs, fit_r.n_outputs, fit_r.is_float64 = n, O, 1, f64

        follow, ignore = float(self.use_follow_up_state), float(self.ignore_terminality)
        tab = table.to(dt)
        lib, world = _lib.lib(), env.handle.ptr

        # the targets between the forward passes and the fits: one launch (cobel_dsr_targets), or —
        # fused_targets = False, the form the kernel is tested against — the same expressions as
        # a dozen elementwise torch kernels
        tgt = _lib.DSRTargets()
        tgt.successor, tgt.value, tgt.table = _lib.ptr(fsr), _lib.ptr(val), _lib.ptr(table)
        tgt.state_index, tgt.next_index = _lib.ptr(si), _lib.ptr(ni)
        tgt.actions, tgt.nonterminal = _lib.ptr(ba), _lib.ptr(bt)
        tgt.targets, tgt.took, tgt.train = _lib.ptr(y), _lib.ptr(took), _lib.ptr(train)
        tgt.n, tgt.n_actions, tgt.n_outputs, tgt.is_float64 = n, A, O, f64
        tgt.use_dr, tgt.follow_up = int(bool(self.use_DR)), int(bool(self.use_follow_up_state))
        tgt.ignore_terminality, tgt.gamma = int(bool(self.ignore_terminality)), float(self.gamma)
        fused_targets = bool(self.fused_targets)

        def one_step() -> None:
            st = _lib.current_stream(dev)
            _lib.check(lib.cobel_dqn_act(world, C.byref(act), st))
            _lib.check(lib.cobel_mlp_forward(C.byref(fwd_t), st))
            _lib.check(lib.cobel_mlp_forward(C.byref(fwd_r), st))
            if fused_targets:
                _lib.check(lib.cobel_dsr_targets(C.byref(tgt), st))
                _lib.check(lib.cobel_mlp_fit(C.byref(fit_sr), st))
                _lib.check(lib.cobel_mlp_fit(C.byref(fit_r), st))
                return
            # agent/dyna_q.py:1079-1118, for all instances at once (as in ``replay``)
            future = fsr.view(n, A, 32, O)
            if self.use_DR:
                boot_sr = future.mean(dim=1)
            else:
                best = val.view(n, A, 32).argmax(dim=1)
                boot_sr = torch.gather(future, 1, best[:, None, :, None].expand(n, 1, 32, O))[:, 0]
            nxt, nonterminal = tab[ni.to(torch.int64)], (bt != 0).to(dt)
            boot = nxt * ((1.0 - follow) * (1.0 - ignore)) * (1.0 - nonterminal)[..., None]
            boot = boot + boot_sr * torch.clamp(nonterminal + ignore, max=1.0)[..., None]
            y.copy_((nxt if self.use_follow_up_state else tab[si.to(torch.int64)]) + self.gamma * boot)
            mine = (ba[:, None, :] == arange_a).reshape(n * A, 32)
            took.copy_(mine.to(torch.uint8))
            train.copy_(mine.any(dim=1).to(torch.uint8))
            _lib.check(lib.cobel_mlp_fit(C.byref(fit_sr), st))
            _lib.check(lib.cobel_mlp_fit(C.byref(fit_r), st))

        graph, done = None, 0
        per_graph = 8
        # (recording and instantiating a graph costs more than the steps of one trial replayed
        #  from it: the one-trial runs that trial hooks on a single instance force, _run, launch
        #  their steps directly)
        worth = not getattr(self, '_one_trial_runs', False)
        if self.fused_graph and self.use_graph is not False and worth and \
                (budget or steps) >= per_graph:
            one_step()
            done = 1
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                for _ in range(per_graph):
                    one_step()
        while True:
            chunk = (budget - done) if budget else min(steps, 64)
            left = chunk
            while graph is not None and left >= per_graph:
                graph.replay()
                left -= per_graph
                self.fused_graph_steps += per_graph
            for _ in range(left):
                one_step()
            done += chunk
            if budget or int(active.sum().item()) == 0:
                break
        self.last_update += done
        self.fused_steps += done

    # -- single-instance views of the stacked networks (reference attribute names) --------------
    def get_weights(self, action: int, target: bool = False, instance: int = 0):
        net = self._target if target else self._online
        if net is None:
            return (self.models_target if target else self.models_online)[action].get_weights()
        return net.get_weights(instance * self.n_actions + action)

    def get_reward_weights(self, instance: int = 0):
        if self._reward_net is None:
            return self.model_reward.get_weights()
        return self._reward_net.get_weights(instance)
