"""Deep Q-network agent — ``cobel.agent.DQN`` (agent/dqn.py:26-384) on PyTorch-ROCm.

Same constructor, ``train(interface, trials, steps, batch_size=32)``, ``test``, ``retrieve_q``,
``predict_on_batch``, ``replay`` and attributes (``model_online``, ``model_target``, ``M``,
``gamma``, ``target_update``, ``DDQN``, ``current_trial``, ``stop``).  ``n_envs`` independent
agent–environment pairs run in lockstep, each with its own copy of the network
(``StackedTorchNetwork``), its own replay ring and its own random streams; a step is

    Q(s) -> epsilon-greedy (cobel_eps_greedy*) -> env.step (cobel_env_step) -> store ->
    sample (cobel_rng_bounded_each) -> targets r + gamma * nt * max_a Q_target(s') -> one
    optimizer step on the online copies -> target blend w_t += tau (w_o - w_t)

with every tensor resident on the GPU (the reference crosses the host boundary at least five
times per step and blends the target on the host, dqn.py:346-371).  After every run
``model_online`` / ``model_target`` hold instance 0's trained weights; weights assigned to them
between runs are adopted by all instances.  Instances whose trial ends
reset immediately and stop once they have run ``trials`` trials, like the tabular kernels.
Requires array observations (``interface.observe()`` -> ``[N, D]``), e.g. ``Topology``.
"""
from __future__ import annotations

import contextlib
import gc

import numpy as np
import torch

from .. import _lib
from ..memory.dqn import DQNMemory
from .agent import Agent, DeviceMonitors


@contextlib.contextmanager
def _capture(graph):
    """``torch.cuda.graph`` with the garbage collector switched off for the duration: a collection
    in the middle of a capture can free tensors of earlier runs, and the allocator's event
    bookkeeping for them is not capturable (seen once as hipErrorStreamCaptureInvalidated in a
    bench run that had several agents behind it)."""
    was_enabled = gc.isenabled()      # (torch.cuda.graph itself collects once on entry)
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was_enabled:
            gc.enable()


class DQN(Agent):
    def __init__(self, observation_space, action_space, policy, model, gamma: float = 0.8,
                 memory=None, policy_test=None, custom_callbacks=None) -> None:
        super().__init__(observation_space, action_space, custom_callbacks)
        assert 1 <= int(action_space.n) <= _lib.MAX_ACTIONS
        self.n_actions = int(action_space.n)
        self.policy = policy
        self.policy_test = policy if policy_test is None else policy_test
        self.model_target = model
        self.model_online = model.clone()
        self.M = DQNMemory() if memory is None else memory
        self.target_update = 10 ** -2
        self.last_update = 0
        self.gamma = gamma
        self.DDQN = False
        # replay one captured lockstep step from a HIP graph wherever no look at the device is
        # needed: True / False, or None = from 256 instances on (fixed-budget runs: only if True)
        self.use_graph = None
        self.graph_replays = 0     # lockstep steps executed from a graph so far
        # two-kernel training step (cobel_dqn_act + cobel_dqn_replay) for 64-64 ReLU networks on a
        # Topology: None = whenever the run qualifies (and use_graph is not forced), False = never
        self.fused_loop = None
        self.fused_steps = 0       # lockstep steps executed by the two-kernel loop so far
        self.fused_graph = True    # record 16 steps of the two-kernel loop into a HIP graph
        self.fused_graph_steps = 0 # ... and how many steps were replayed from it so far
        self.n_envs = None
        self.monitors = None
        self._online = self._target = None

    # -- binding ------------------------------------------------------------------------------
    def _bind(self, interface, slots: int) -> None:
        if self.n_envs is None:
            self.n_envs, self.device = interface.n_envs, interface.device
            self.model_target.set_device(self.device)
            self.model_online.set_device(self.device)
            self._target = self.model_target.replicate(self.n_envs)
            self._online = self.model_online.replicate(self.n_envs)
            self.dtype = next(iter(self._online.params.values())).dtype
            # (striped like the fused kernels' monitors: the masked index_add_ below sends every
            #  instance that did NOT finish a trial to index 0 with a zero, and thousands of atomics on
            #  one address cost 0.3 ms per step at 8 192 instances)
            self.monitors = DeviceMonitors(self.device, 1, 1, False,
                                           stripes=16 if self.n_envs >= 1024 else 1)
            self.trial = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self._bind_memory(interface, slots)

    def _bind_memory(self, interface, slots: int) -> None:
        obs = interface.observe()
        self.M._bind(self.n_envs, obs.shape[1:], self.dtype, self.device, slots, interface.seed,
                     interface.instance_base)

    def _policy_bind(self, pol, interface, test: bool) -> None:
        if pol.seed is None:
            pol.seed = interface.seed
        if pol.stream is None:
            pol.stream = (_lib.STREAM_POLICY_TEST if (test and pol is not self.policy)
                          else _lib.STREAM_POLICY)
        if pol.counter is None or pol.counter.numel() != self.n_envs:
            pol.counter = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)

    def _select(self, pol, q: torch.Tensor, base: int, idle=None) -> torch.Tensor:
        n = self.n_envs
        u = torch.empty(n, dtype=torch.float64, device=self.device)
        st = _lib.current_stream(self.device)
        _lib.check(_lib.lib().cobel_rng_uniform(_lib.ptr(pol.counter), pol.seed, pol.stream, base,
                                                _lib.ptr(u), n, 1, st))
        act = torch.empty(n, dtype=torch.uint8, device=self.device)
        q = q.contiguous()
        if q.dtype not in (torch.float64, torch.float32):
            q = q.float()
        A = int(q.shape[1])
        if A == 4:
            fn = (_lib.lib().cobel_eps_greedy_f64 if q.dtype == torch.float64
                  else _lib.lib().cobel_eps_greedy)
            _lib.check(fn(_lib.ptr(q), None, _lib.ptr(u), float(pol.epsilon), _lib.ptr(act), None,
                          n, st))
        else:       # e.g. the six neighbours of a hexagonal Topology
            fn = (_lib.lib().cobel_eps_greedy_n_f64 if q.dtype == torch.float64
                  else _lib.lib().cobel_eps_greedy_n)
            _lib.check(fn(_lib.ptr(q), None, _lib.ptr(u), float(pol.epsilon), _lib.ptr(act), None,
                          n, A, st))
        if idle is not None:        # instances that are done do not consume their streams
            pol.counter -= idle.to(torch.int32)
        return act

    def _make_capturable(self) -> None:
        """Every optimizer that steps inside the captured iteration."""
        self._online.make_capturable()

    def _q_values(self, obs: torch.Tensor) -> torch.Tensor:
        """Q-values ``[N, A]`` of the per-instance online networks for observations ``[N, D]``."""
        return self._online.predict_on_device(obs[:, None, :])[:, 0]

    # -- reference surface ----------------------------------------------------------------------
    def retrieve_q(self, state):
        """Q-values of the online network(s) for an observation ``[D]`` or per-instance ``[N, D]``."""
        if self._online is None:
            return self.model_online.predict_on_batch(np.array([state]))[0]
        s = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state,
                            device=self.device).to(self.dtype)
        if s.dim() == 1:
            s = s.expand(self.n_envs, -1)
        q = self._online.predict_on_device(s[:, None, :])[:, 0]
        return q[0].cpu().numpy() if self.n_envs == 1 else q

    def predict_on_batch(self, batch):
        if self._online is None:
            return self.model_online.predict_on_batch(batch)
        b = torch.as_tensor(np.asarray(batch), device=self.device).to(self.dtype)
        q = self._online.predict_on_device(b[None].expand(self.n_envs, *b.shape).contiguous())
        return q[0].cpu().numpy() if self.n_envs == 1 else q

    def replay(self, batch_size: int = 32, active=None):
        """One optimisation step per instance on a sampled batch (dqn.py:331-384)."""
        states, actions, rewards, next_states, terminals = self.M.retrieve(batch_size)
        if active is not None:      # instances that are done do not consume their streams
            self.M.counter -= (~active & (self.M.size > 0)).to(torch.int32)
        if self.target_update < 1.0 and self._online.dqn_replay_fused(
                self._target, states, actions, rewards, next_states, terminals, self.gamma,
                self.DDQN, self.target_update, active):
            # 64-64 ReLU networks: the whole step in one kernel (cobel_dqn_replay)
            return {'states': states, 'actions': actions, 'rewards': rewards,
                    'next_states': next_states, 'terminals': terminals}
        with torch.no_grad():
            targets = self._online.forward(states).clone()
            boot = self._target.forward(next_states)
            pick = (self._online.forward(next_states) if self.DDQN else boot).argmax(dim=2)
            boot = torch.gather(boot, 2, pick[..., None])[..., 0]
            new = rewards + boot * terminals * self.gamma
            targets.scatter_(2, actions[..., None], new[..., None])
        if self.target_update < 1.0:    # optimizer step and target blend in one pass
            self._online.train_on_device(states, targets, active, blend_into=self._target,
                                         tau=self.target_update)
            return {'states': states, 'actions': actions, 'rewards': rewards,
                    'next_states': next_states, 'terminals': terminals}
        self._online.train_on_device(states, targets, active)
        if self.last_update >= self.target_update:
            self._target.copy_from(self._online, active)
            self.last_update = 1
        else:
            self.last_update += 1
        return {'states': states, 'actions': actions, 'rewards': rewards,
                'next_states': next_states, 'terminals': terminals}

    # -- the two-kernel training loop --------------------------------------------------------------
    def _fused_loop_ok(self, interface, pol, batch_size: int) -> bool:
        """The run is one the two fused kernels cover: a plain DQN (replay ring, blended target)
        with a 64-64 ReLU network and an epsilon-greedy policy on a Topology."""
        from ..interface.topology import Topology
        from ..policy.greedy import EpsilonGreedy
        net = self._online
        if not self._fused_setting_ok(interface) or self.fused_loop is False \
                or not 1 <= self.n_actions <= 8 \
                or self.use_graph is True \
                or type(pol) is not EpsilonGreedy or not (self.target_update < 1.0) \
                or getattr(self, '_no_replay', False) or not net.fused_mlp:
            return False
        names = net._mlp3_names()
        if names is None or self._target._mlp3_names() != names or not net._fused_adam_ok() \
                or type(net.criterion) is not torch.nn.MSELoss \
                or getattr(net.criterion, 'reduction', '') != 'none':
            return False
        w = [net.params[k + '.weight'] for k in names]
        if w[1].shape[2] != w[0].shape[1] or w[2].shape[2] != w[1].shape[1] \
                or self.dtype not in (torch.float64, torch.float32):
            return False
        return _lib.lib().cobel_dqn_replay_query(
            w[0].shape[2], w[0].shape[1], w[1].shape[1], w[2].shape[1], batch_size,
            int(self.dtype == torch.float64), None) == _lib.OK

    def _fused_setting_ok(self, interface) -> bool:
        """Agent class, environment and memory are the ones ``_fused_wire`` knows how to hand to
        the kernels."""
        from ..interface.topology import Topology
        return type(self) is DQN and isinstance(interface, Topology) and type(self.M) is DQNMemory

    def _fused_wire(self, interface, act, rep, batch_size: int):
        """Environment and memory side of the two launch descriptors; returns (world handle,
        env state tensor, observation table, tensors to keep alive)."""
        M, n, dev = self.M, self.n_envs, self.device
        slots = torch.zeros((n, batch_size), dtype=torch.int32, device=dev)
        for run in (act, rep):
            prefix = 'ring_' if run is act else ''
            setattr(run, prefix + 'states', _lib.ptr(M.states))
            setattr(run, prefix + 'next_states', _lib.ptr(M.next_states))
            setattr(run, prefix + 'actions', _lib.ptr(M.actions))
            setattr(run, prefix + 'rewards', _lib.ptr(M.rewards))
            setattr(run, prefix + 'nonterminal', _lib.ptr(M.terminals))
        act.ring_size, act.ring_head = _lib.ptr(M.size), _lib.ptr(M.head)
        act.memory_ctr, act.slots = _lib.ptr(M.counter), M.slots
        act.batch_slots = rep.batch_slots = _lib.ptr(slots)
        rep.ring_slots = M.slots
        act.env_ctr = _lib.ptr(interface.env_ctr)
        return interface.handle.ptr, interface.state, interface._pose_dev, [slots]

    def _run_fused(self, interface, pol, trials: int, steps: int, batch_size: int,
                   budget: int) -> None:
        """Two launches per lockstep step and nothing else: cobel_dqn_act (select, env.step,
        ring store, trial bookkeeping, batch draw) and cobel_dqn_replay (the optimisation step,
        plus the Q-values of the next observation for the next cobel_dqn_act).  Same streams,
        counters and arithmetic as the PyTorch loop in ``_run``; instances that have run their
        trials are skipped by both kernels, so the loop only looks at the device between chunks."""
        import ctypes as C
        n, dev, net, M, mon = self.n_envs, self.device, self._online, self.M, self.monitors
        first = self.current_trial
        f64 = self.dtype == torch.float64
        rep, act = _lib.DQNReplay(), _lib.DQNAct()
        world, state, table, keep = self._fused_wire(interface, act, rep, batch_size)
        q = self._q_values(interface.observe().to(self.dtype)).contiguous()
        step = torch.zeros(n, dtype=torch.int32, device=dev)
        trew = torch.zeros(n, dtype=torch.float64, device=dev)
        active = torch.ones(n, dtype=torch.uint8, device=dev)
        stepped = torch.zeros(n, dtype=torch.uint8, device=dev)
        # Adam step counts per instance (shared with the PyTorch path's fused optimizer kernel)
        net._diverged = True
        opt = net.optimizer
        counts = getattr(net, '_steps', None)
        if counts is None:
            seen = [float(st['steps'].max()) if 'steps' in st else float(st.get('step', 0.0))
                    for st in opt.state.values()]
            counts = net._steps = torch.full((n,), max(seen, default=0.0), dtype=torch.float64,
                                             device=dev)
        names = net._mlp3_names()
        for k, name in enumerate(names):
            for kind, dp, dt_, dm, dv in (('.weight', rep.w, rep.w_target, rep.m_w, rep.v_w),
                                          ('.bias', rep.b, rep.b_target, rep.m_b, rep.v_b)):
                p = net.params[name + kind]
                st = opt.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                    st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                st['steps'] = counts
                dp[k], dt_[k] = _lib.ptr(p), _lib.ptr(self._target.params[name + kind])
                dm[k], dv[k] = _lib.ptr(st['exp_avg']), _lib.ptr(st['exp_avg_sq'])
        w = [net.params[k + '.weight'] for k in names]
        group = opt.param_groups[0]
        rep.steps, rep.active = _lib.ptr(counts), _lib.ptr(stepped)
        rep.n, rep.batch = n, batch_size
        rep.n_inputs, rep.n_hidden1 = w[0].shape[2], w[0].shape[1]
        rep.n_hidden2, rep.n_actions = w[1].shape[1], w[2].shape[1]
        rep.is_float64, rep.ddqn = int(f64), int(bool(self.DDQN))
        rep.gamma, rep.lr = float(self.gamma), float(group['lr'])
        rep.beta1, rep.beta2 = (float(b) for b in group['betas'])
        rep.eps, rep.weight_decay = float(group['eps']), float(group['weight_decay'])
        rep.tau = float(self.target_update)
        rep.obs_index, rep.obs_table, rep.q_out = _lib.ptr(state), _lib.ptr(table), _lib.ptr(q)
        act.state = _lib.ptr(state)
        act.obs_table, act.q = _lib.ptr(table), _lib.ptr(q)
        act.policy_ctr, act.policy_stream = _lib.ptr(pol.counter), pol.stream
        act.is_float64, act.epsilon = int(f64), float(pol.epsilon)
        act.trial, act.step, act.trial_reward = _lib.ptr(self.trial), _lib.ptr(step), _lib.ptr(trew)
        act.active, act.adam_steps = _lib.ptr(active), _lib.ptr(counts)
        act.lat_sum, act.lat_cnt = _lib.ptr(mon.raw('lat_sum')), _lib.ptr(mon.raw('lat_cnt'))
        act.reward_sum = _lib.ptr(mon.raw('reward_sum'))
        act.stepped = _lib.ptr(stepped)
        act.n, act.n_obs, act.batch = n, table.shape[1], batch_size
        act.steps_per_trial, act.trials_target = steps, first + trials
        act.trial_cap, act.mon_stripes = mon.cap, mon.stripes
        act.instance_base, act.seed = interface.instance_base, interface.seed
        lib = _lib.lib()

        def pair() -> None:
            stream = _lib.current_stream(dev)        # (the capture stream while a graph records)
            _lib.check(lib.cobel_dqn_act(world, C.byref(act), stream))
            _lib.check(lib.cobel_dqn_replay(C.byref(rep), stream))

        # The two launches of a step take all their state from device buffers whose addresses never
        # change, so a chunk of steps is recorded ONCE into a HIP graph and replayed: the host
        # leaves the loop (one graph launch per 16 steps instead of 32 ctypes calls).
        per_graph = 16
        graph = None
        # (recording and instantiating a graph costs more than the steps of one trial replayed
        #  from it: the one-trial runs that trial hooks on a single instance force, _run, launch
        #  their steps directly)
        worth = not getattr(self, '_one_trial_runs', False)
        if self.fused_graph and self.use_graph is not False and worth and \
                (budget or steps) >= per_graph:
            pair()                                   # (lazy initialisations outside the capture)
            done_first = 1
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                for _ in range(per_graph):
                    pair()
        else:
            done_first = 0
        done = done_first
        while True:
            chunk = (budget - done) if budget else min(steps, 64)
            left = chunk
            while graph is not None and left >= per_graph:
                graph.replay()
                left -= per_graph
                self.fused_graph_steps += per_graph
            for _ in range(left):
                pair()
            done += chunk
            if budget or int(active.sum().item()) == 0:
                break
        self._fused_keep = (graph, act, rep, keep, q, step, trew, active, stepped)
        self.fused_steps += done

    def _run(self, interface, trials: int, steps: int, batch_size: int, learn: bool,
             budget: int = 0) -> None:
        """``budget`` > 0 stops after that many lockstep iterations (benchmarking).

        One environment instance with host callbacks keeps the reference's contract
        (agent/dqn.py:170-215): step hooks fire around every step with the reference's ``logs``
        keys, trial hooks after every trial, and ``agent.stop`` ends the session at the next trial
        boundary.  Vectorised runs fire the trial hooks once per ``train()`` call with per-trial
        means (a device cannot call Python between steps of thousands of instances)."""
        single = interface.n_envs == 1 and not budget
        if single and self.callbacks.has('on_step_begin', 'on_step_end'):
            self._run_hooks(interface, trials, steps, batch_size, learn)
        elif single and self.callbacks.has('on_trial_begin', 'on_trial_end'):
            self._one_trial_runs = True
            try:
                for t in range(trials):          # one trial per run: same streams, same kernels
                    self._run_core(interface, 1, steps, batch_size, learn, 0, session0=t)
                    if self.stop:
                        break
            finally:
                self._one_trial_runs = False
        else:
            self._run_core(interface, trials, steps, batch_size, learn, budget)

    def _run_hooks(self, interface, trials: int, steps: int, batch_size: int, learn: bool) -> None:
        """The reference's loop, step by step, for ONE instance with step hooks registered; the
        same kernels, streams and counters as the lockstep loop (so the outcome is identical)."""
        self._bind(interface, (int(self.M.size.max().item()) if torch.is_tensor(
            getattr(self.M, 'size', None)) else 0) + trials * steps)
        self._adopt_user_weights()
        pol = self.policy if learn else self.policy_test
        self._policy_bind(pol, interface, not learn)
        dev, mon = self.device, self.monitors
        for t in range(trials):
            trial = self.current_trial
            mon.reserve(trial + 1)
            logs = self.callbacks.on_trial_begin({'trial_reward': 0.0, 'trial': trial,
                                                  'trial_session': t})
            interface.reset()
            obs = interface.observe().to(self.dtype).clone()
            step = 0
            for step in range(steps):
                logs['step'] = step
                logs = self.callbacks.on_step_begin(logs)
                action = self._select(pol, self._q_values(obs), interface.instance_base)
                interface.step(action)
                nxt = interface.observe().to(self.dtype).clone()
                reward, done = float(interface._reward[0].item()), bool(interface._done[0].item())
                if learn:
                    self.M.store_batch(obs, action, interface._reward, nxt, ~interface._done.bool())
                    if not getattr(self, '_no_replay', False):
                        logs['replay'] = self.replay(batch_size)
                logs['trial_reward'] += reward
                if hasattr(interface, 'prev_state'):     # Dyna agents: integer gridworld states
                    s_log, ns_log = int(interface.prev_state[0].item()), int(interface.env.state[0].item())
                else:
                    s_log, ns_log = obs[0].cpu().numpy(), nxt[0].cpu().numpy()
                logs.update({'state': s_log, 'action': int(action[0].item()), 'reward': reward,
                             'next_state': ns_log, 'terminal': 1 - int(done)})
                logs = self.callbacks.on_step_end(logs)
                obs = nxt
                if done:
                    break
            mon.raw('lat_sum')[0, trial] += step
            mon.raw('lat_cnt')[0, trial] += 1
            mon.raw('reward_sum')[0, trial] += logs['trial_reward']
            self.current_trial += 1
            self.trial.fill_(self.current_trial)
            logs['step'] = logs['steps'] = step
            self.callbacks.on_trial_end(logs)
            if self.stop:
                break
        for stack, single_net, inst in self._stacks():
            stack.write_back(single_net, inst)

    def _run_core(self, interface, trials: int, steps: int, batch_size: int, learn: bool,
                  budget: int = 0, session0: int = 0) -> None:
        # the ring keeps everything up to `capacity` (memory/dqn.py:113-119): room for what is
        # stored already plus what this run can add
        bound = self.n_envs is not None
        stored = getattr(self.M, 'size', None)
        have = int(stored.max().item()) if torch.is_tensor(stored) else 0
        self._bind(interface, have + min(trials * steps, budget or trials * steps))
        if bound:
            self._adopt_user_weights()
        pol = self.policy if learn else self.policy_test
        self._policy_bind(pol, interface, not learn)
        n, dev = self.n_envs, self.device
        first = self.current_trial
        self.trial.fill_(first)
        self.monitors.reserve(first + trials)
        step = torch.zeros(n, dtype=torch.int32, device=dev)
        trew = torch.zeros(n, dtype=torch.float64, device=dev)
        for t in range(trials):
            self.callbacks.on_trial_begin({'trial_reward': 0.0, 'trial': first + t,
                                           'trial_session': session0 + t})
        obs, _ = interface.reset()
        if learn and self._fused_loop_ok(interface, pol, batch_size):
            self._run_fused(interface, pol, trials, steps, batch_size, budget)
            self._finish_run(first, trials, session0)
            return
        obs = interface.observe().to(self.dtype).clone()
        active = torch.ones(n, dtype=torch.bool, device=dev)
        zero64 = torch.zeros(n, dtype=torch.int64, device=dev)
        cap = self.monitors.cap
        mon = self.monitors
        stripe_off = (torch.arange(n, device=dev) % mon.stripes) * cap      # copy of each instance
        lat_sum, lat_cnt = mon.raw('lat_sum').view(-1), mon.raw('lat_cnt').view(-1)
        reward_sum = mon.raw('reward_sum').view(-1)
        all_active = True     # host-side knowledge; exact because it is refreshed every step

        def iteration() -> None:
            """One lockstep step of every instance; all state lives in tensors updated in place
            (so that the same sequence of launches can be replayed from a HIP graph)."""
            q = self._q_values(obs)
            idle = None if all_active else ~active
            action = self._select(pol, q, interface.instance_base, idle)
            interface.step(action)
            nxt = interface.observe().to(self.dtype).clone()
            reward, done = interface._reward, interface._done.bool()
            if learn:
                self.M.store_batch(obs, action, reward, nxt, (~done), None if all_active else active)
                if not getattr(self, '_no_replay', False):
                    self.replay(batch_size, None if all_active else active)
            trew.add_(torch.where(active, reward.to(torch.float64), torch.zeros_like(trew)))
            # trial ends, entirely masked (no host round trip per step)
            over = active & (done | (step + 1 >= steps))
            idx = torch.where(over, self.trial.to(torch.int64), zero64).clamp_(0, cap - 1)
            ok = over & (self.trial < cap)
            idx = idx + stripe_off
            lat_sum.index_add_(0, idx, torch.where(ok, step.to(torch.int64), zero64))
            lat_cnt.index_add_(0, idx, ok.to(torch.int64))
            reward_sum.index_add_(0, idx, torch.where(ok, trew, torch.zeros_like(trew)))
            self.trial.add_(over.to(torch.int32))
            trew.copy_(torch.where(over, torch.zeros_like(trew), trew))
            active.logical_and_(self.trial < first + trials)
            restart = over & active
            interface.reset(restart)
            nxt = torch.where(restart[:, None], interface.observe().to(self.dtype), nxt)
            step.copy_(torch.where(over, torch.zeros_like(step), step + active.to(torch.int32)))
            obs.copy_(nxt)

        if budget and learn and self.use_graph is True and budget > 4:
            # Fixed-budget runs never look at the device between steps: after a few eager steps
            # (lazy initialisations, optimizer state) ONE iteration is captured into a HIP graph and
            # replayed — ~180 kernel launches per step become one graph launch.
            for _ in range(3):
                iteration()
            self._make_capturable()
            iteration()
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                iteration()
            for _ in range(budget - 4):
                graph.replay()
            self.graph_replays += budget - 4
        else:
            # train() / test(): one look at the device per step keeps finished instances frozen
            # exactly.  While every instance still has more trials to run than steps in a chunk,
            # none can finish inside it (a step ends at most one trial), so the chunk needs no
            # look at all and is replayed from a HIP graph of one step.
            iters, graph = 0, None
            graphable = learn and budget == 0 and (self.use_graph or
                                                   (self.use_graph is None and n >= 256))
            while True:
                if graphable and all_active and iters >= 3:
                    rem = int((first + trials - self.trial).min().item())
                    k = min(rem - 1, 512)
                    if k >= 8:
                        if graph is None:
                            self._make_capturable()
                            iteration()
                            k -= 1
                            torch.cuda.synchronize(dev)
                            graph = torch.cuda.CUDAGraph()
                            with _capture(graph):
                                iteration()
                        for _ in range(k):
                            graph.replay()
                        iters += k
                        self.graph_replays += k
                        continue
                iteration()
                iters += 1
                if budget and iters >= budget:
                    break
                if not budget:
                    left = int(active.sum().item())
                    if left == 0:
                        break
                    all_active = left == n
        self._finish_run(first, trials, session0)

    def _stacks(self):
        """(stacked network, single network the user holds, instance) triples kept in step."""
        return [(self._online, self.model_online, 0), (self._target, self.model_target, 0)]

    def _adopt_user_weights(self) -> None:
        """Weights the user assigned to ``model_online`` / ``model_target`` since the last run
        (``set_weights``, an edited module) replace those of every instance."""
        for stack, single, inst in self._stacks():
            if not stack.matches(single, inst):
                stack.load_from(single)

    def _finish_run(self, first: int, trials: int, session0: int = 0) -> None:
        self.current_trial = first + trials
        for stack, single, inst in self._stacks():      # the trained networks, as attributes
            stack.write_back(single, inst)
        if self.callbacks.has('on_trial_end'):
            lat, rew = self.monitors.mean_latency(), self.monitors.mean_reward()
            cnt = self.monitors.lat_cnt.cpu().numpy()
            for t in range(trials):
                self.callbacks.on_trial_end({'trial_reward': float(rew[first + t]),
                                             'trial': first + t, 'trial_session': session0 + t,
                                             'steps': float(lat[first + t]),
                                             'count': int(cnt[first + t])})

    def train(self, interface, trials: int, steps: int, batch_size: int = 32) -> None:
        self._run(interface, trials, steps, batch_size, True)

    def test(self, interface, trials: int, steps: int) -> None:
        self._run(interface, trials, steps, 0, False)
