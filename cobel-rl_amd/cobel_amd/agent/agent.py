"""Agent base class, callbacks and the per-instance device state shared by all fused agents.

Surface of the reference's ``cobel.agent.agent`` (agent/agent.py:19-243): ``Agent.train /
test / predict_on_batch``, attributes ``current_trial``, ``stop``, ``callbacks``; ``Callbacks``
with the four hooks, each shallow-copying ``logs``, adding ``logs['agent']`` and merging dicts
returned by user callbacks.

Device hooks cannot call Python, so host callbacks fire at launch boundaries:
  * ``n_envs == 1`` and step callbacks registered  -> one launch per env step
    (``step_budget = 1``), logs identical in keys to the reference's;
  * ``n_envs == 1`` otherwise                      -> one launch per trial;
  * ``n_envs > 1``                                 -> one launch per ``train()`` call; monitors are
    reduced on device and ``on_trial_end`` then fires once per trial index with the mean
    ``steps`` / ``trial_reward`` over instances (and ``'count'`` = instances that ran it).
"""
from __future__ import annotations

import abc
import copy

import numpy as np
import torch

from .. import _lib


class Callbacks:
    def __init__(self, agent, custom_callbacks=None) -> None:
        self.agent = agent
        self.custom_callbacks = {} if custom_callbacks is None else custom_callbacks

    def _fire(self, name: str, logs: dict) -> dict:
        out = copy.copy(logs)
        out['agent'] = self.agent
        for cb in self.custom_callbacks.get(name, []):
            ret = cb(out)
            if type(ret) is dict:
                out.update(ret)
        return out

    def on_step_begin(self, logs: dict) -> dict:
        return self._fire('on_step_begin', logs)

    def on_step_end(self, logs: dict) -> dict:
        return self._fire('on_step_end', logs)

    def on_trial_begin(self, logs: dict) -> dict:
        return self._fire('on_trial_begin', logs)

    def on_trial_end(self, logs: dict) -> dict:
        return self._fire('on_trial_end', logs)

    def has(self, *names: str) -> bool:
        return any(len(self.custom_callbacks.get(n, [])) > 0 for n in names)


class Agent(abc.ABC):
    def __init__(self, observation_space, action_space, custom_callbacks=None) -> None:
        self.observation_space = observation_space
        self.action_space = action_space
        self.callbacks = Callbacks(self, custom_callbacks)
        self.current_trial = 0
        self.stop = False

    @abc.abstractmethod
    def train(self, interface, trials: int, steps: int) -> None:
        ...

    @abc.abstractmethod
    def test(self, interface, trials: int, steps: int) -> None:
        ...

    @abc.abstractmethod
    def predict_on_batch(self, batch):
        ...


class DeviceMonitors:
    """Per-trial reductions written by the kernels: escape latency (``logs['steps']``,
    monitor/behavior.py:82), trial reward, and per-state visit counts.

    The four per-trial arrays may be kept as ``stripes`` copies ``[stripes, cap]`` (workgroup b of a
    kernel adds into copy ``b % stripes``, see ``cobel_tab_run_t.mon_stripes``); ``lat_sum`` /
    ``lat_cnt`` / ``reward_sum`` / ``resp_cnt`` are the sums over the copies, ``raw(name)`` the
    striped tensors the kernels take.  With one stripe the attribute IS the tensor (in-place ops
    on it are seen by the kernels)."""

    _PER_TRIAL = {'lat_sum': torch.int64, 'lat_cnt': torch.int64, 'reward_sum': torch.float64,
                  'resp_cnt': torch.int64}

    def __init__(self, device, n_worlds: int, n_states: int, occupancy: bool = False,
                 responses: bool = False, stripes: int = 1) -> None:
        self.device = device
        self.cap = 0
        self.stripes = max(1, int(stripes))
        self._raw = {name: None for name in self._PER_TRIAL}
        self.responses = responses    # count rewarded trials (ResponseMonitor) — opt-in
        self.occupancy = (torch.zeros((n_worlds, n_states), dtype=torch.int64, device=device)
                          if occupancy else None)
        self.steps_done = torch.zeros(1, dtype=torch.int64, device=device)
        self.lat_trace = None
        self.collectives = 0          # all-gathers issued by all_reduce() (one per call)

    def raw(self, name: str):
        """The striped tensor ``[stripes, cap]`` handed to the kernels (or None)."""
        return self._raw[name]

    def _get(self, name: str):
        t = self._raw[name]
        if t is None:
            return None
        return t[0] if self.stripes == 1 else t.sum(dim=0)

    def _set(self, name: str, value) -> None:
        # (also what `monitors.lat_sum += x` ends in: the in-place result, a view of the raw tensor)
        if value is not None and value.dim() == 1:
            assert self.stripes == 1, 'assign striped monitors through raw tensors'
            value = value.reshape(1, -1)
        self._raw[name] = value

    lat_sum = property(lambda self: self._get('lat_sum'), lambda self, v: self._set('lat_sum', v))
    lat_cnt = property(lambda self: self._get('lat_cnt'), lambda self, v: self._set('lat_cnt', v))
    reward_sum = property(lambda self: self._get('reward_sum'),
                          lambda self, v: self._set('reward_sum', v))
    resp_cnt = property(lambda self: self._get('resp_cnt'), lambda self, v: self._set('resp_cnt', v))

    def reserve(self, trials: int, n_envs: int = 0, per_instance: bool = False) -> None:
        if trials > self.cap:
            for name, dtype in self._PER_TRIAL.items():
                if name == 'resp_cnt' and not self.responses:
                    continue
                new = torch.zeros((self.stripes, trials), dtype=dtype, device=self.device)
                old = self._raw[name]
                if old is not None:
                    new[:, : old.shape[1]] = old
                self._raw[name] = new
            if per_instance or self.lat_trace is not None:
                new = torch.full((n_envs, trials), -1, dtype=torch.int32, device=self.device)
                if self.lat_trace is not None:
                    new[:, : self.lat_trace.shape[1]] = self.lat_trace
                self.lat_trace = new
            self.cap = trials
        elif per_instance and self.lat_trace is None:
            self.lat_trace = torch.full((n_envs, self.cap), -1, dtype=torch.int32,
                                        device=self.device)

    def snapshot(self) -> 'MonitorSums':
        """This rank's sums on the host (stripes folded)."""
        return MonitorSums.unpack(self._pack().cpu().numpy()[None, :], self._layout())

    def _layout(self):
        """(name, elements) of the fused buffer, in order; every rank must agree on it."""
        lay = [('header', 2)]
        for name in ('lat_sum', 'lat_cnt', 'resp_cnt', 'reward_sum'):
            if self._raw[name] is not None:
                lay.append((name, self.cap))
        if self.occupancy is not None:
            lay.append(('occupancy', self.occupancy.numel()))
        lay.append(('steps_done', 1))
        return lay

    def _pack(self):
        """ONE int64 buffer holding every monitor of this rank: a header (capacity, layout
        digest), the per-trial arrays with their stripes folded, the float64 reward sums as
        their bit patterns, the visit counts and the step counter."""
        lay = self._layout()
        digest = sum((k + 1) * n for k, (_, n) in enumerate(lay)) + 1000003 * len(lay)
        parts = [torch.tensor([self.cap, digest], dtype=torch.int64, device=self.device)]
        for name, _ in lay[1:]:
            if name == 'occupancy':
                parts.append(self.occupancy.reshape(-1))
            elif name == 'steps_done':
                parts.append(self.steps_done.reshape(-1))
            elif name == 'reward_sum':
                parts.append(self._raw[name].sum(dim=0).view(torch.int64))
            else:
                parts.append(self._raw[name].sum(dim=0))
        return torch.cat(parts)

    def all_reduce(self) -> 'MonitorSums':
        """Sums of the monitor buffers over all ranks — the only collective of the path (RCCL on
        GPUs): ONE all-gather of the fused buffer per call, summed in rank order on every rank
        (integers exactly, the float64 reward sums in a fixed order, so all ranks hold the same
        bits).  The rank-local accumulators the kernels add into are left untouched: calling
        this again — a second monitor, a second reporting interval — counts nothing twice."""
        import torch.distributed as dist
        flat = self._pack()
        # (a process group of ONE rank still takes the collective: the same RCCL call, tensor types
        #  and unpacking as with eight — what a one-GPU box can rehearse of the multi-GPU path)
        if not (dist.is_available() and dist.is_initialized()):
            return MonitorSums.unpack(flat.cpu().numpy()[None, :], self._layout())
        world = dist.get_world_size()
        if dist.get_backend() == 'gloo':      # CPU rehearsals of the multi-GPU path
            flat = flat.cpu()
        gathered = torch.empty(world * flat.numel(), dtype=torch.int64, device=flat.device)
        dist.all_gather_into_tensor(gathered, flat)
        self.collectives += 1
        g = gathered.cpu().numpy().reshape(world, flat.numel())
        assert (g[:, :2] == g[0, :2]).all(), \
            'ranks disagree on the monitor layout (trial capacity / tracked monitors)'
        return MonitorSums.unpack(g, self._layout())

    def mean_latency(self) -> np.ndarray:
        return self.snapshot().mean_latency()

    def mean_response(self) -> np.ndarray:
        """Fraction of instances rewarded in each trial (ResponseMonitor's default response,
        monitor/behavior.py:286-289, averaged over instances)."""
        assert self.resp_cnt is not None, 'set agent.track_responses = True before training'
        return self.snapshot().mean_response()

    def mean_reward(self) -> np.ndarray:
        return self.snapshot().mean_reward()


class MonitorSums:
    """Monitor sums on the host — one rank's (``DeviceMonitors.snapshot``) or all ranks'
    (``DeviceMonitors.all_reduce``) — with the per-trial means the monitors report."""

    def __init__(self) -> None:
        self.lat_sum = self.lat_cnt = self.resp_cnt = self.reward_sum = self.occupancy = None
        self.steps_done = 0
        self.ranks = 1

    @classmethod
    def unpack(cls, rows: np.ndarray, layout) -> 'MonitorSums':
        """rows: int64 [ranks, elements] of fused buffers laid out as ``layout``."""
        out = cls()
        out.ranks = rows.shape[0]
        off = 0
        for name, n in layout:
            part = rows[:, off: off + n]
            off += n
            if name == 'header':
                continue
            if name == 'reward_sum':
                vals = np.ascontiguousarray(part).view(np.float64)
                acc = np.zeros(n, dtype=np.float64)
                for r in range(vals.shape[0]):        # fixed (rank) order
                    acc = acc + vals[r]
                out.reward_sum = acc
            elif name == 'steps_done':
                out.steps_done = int(part.sum())
            else:
                setattr(out, name, part.sum(axis=0))
        return out

    def _mean(self, s):
        c = self.lat_cnt
        with np.errstate(invalid='ignore', divide='ignore'):
            return np.where(c > 0, s / np.maximum(c, 1), np.nan)

    def mean_latency(self) -> np.ndarray:
        return self._mean(self.lat_sum)

    def mean_reward(self) -> np.ndarray:
        return self._mean(self.reward_sum)

    def mean_response(self) -> np.ndarray:
        assert self.resp_cnt is not None, 'set agent.track_responses = True before training'
        return self._mean(self.resp_cnt)


class FusedAgent(Agent):
    """Plumbing common to the agents whose whole loop runs in one kernel."""

    general_actions = False    # True: the agent also runs on action counts other than four

    def __init__(self, observation_space, action_space, policy, policy_test, custom_callbacks):
        super().__init__(observation_space, action_space, custom_callbacks)
        self.policy = policy
        self.policy_test = policy if policy_test is None else policy_test
        # Box observations (a Topology's poses): the state count is the node count of the
        # environment the agent is first trained on
        self.n_states = int(observation_space.n) if hasattr(observation_space, 'n') else None
        self.n_actions = int(action_space.n)
        assert self.n_actions == 4 or self.general_actions, \
            'this agent\'s kernels cover 4-action worlds'
        assert 1 <= self.n_actions <= _lib.MAX_ACTIONS
        self.action_mask = (np.ones((self.n_states, self.n_actions), dtype=bool)
                            if self.n_states is not None else None)
        self.mask_actions = False
        self.track_occupancy = False
        self.track_responses = False   # per-trial count of rewarded instances (ResponseMonitor)
        self.track_instances = False   # keep per-instance latency traces [N, trials]
        self.monitor_stripes = 16      # copies of the per-trial monitor arrays (from 4096 instances on)
        self.device = None
        self.n_envs = None
        self.inst = None
        self.monitors = None
        self._last_exp = None
        self._pset_key = self._pset_dev = self._pidx_dev = None
        self._pset_n = 0

    # -- hyper-parameters -----------------------------------------------------------------------
    def _hyper(self, run, alpha, gamma, epsilon, model_lr=None) -> None:
        """Hand the hyper-parameters to a run struct.  Scalars go in as they are.  If any of them
        is array-valued (one entry per instance) the distinct combinations become parameter sets
        and every instance gets the index of its set — the reference's grid search
        (optimizer/grid_search.py:173-262, one simulation per combination) folded onto the
        instance axis of one launch."""
        import ctypes as C
        names = ('alpha', 'gamma', 'epsilon') + (('model_lr',) if model_lr is not None else ())
        vals = [np.asarray(v, dtype=np.float64)
                for v in (alpha, gamma, epsilon) + ((model_lr,) if model_lr is not None else ())]
        if all(v.ndim == 0 for v in vals):
            for name, v in zip(names, vals):
                setattr(run, name, float(v))
            return
        n = self.n_envs
        for v in vals:
            assert v.ndim == 0 or v.shape == (n,), \
                'per-instance hyper-parameters need one entry per environment instance'
        cols = np.stack([np.broadcast_to(v, (n,)) for v in vals], axis=1)
        if model_lr is None:
            cols = np.concatenate([cols, np.full((n, 1), 0.9)], axis=1)
        key = cols.tobytes()
        if key != self._pset_key:
            uniq, inv = np.unique(cols, axis=0, return_inverse=True)
            assert len(uniq) <= 65535, 'at most 65535 distinct parameter combinations per launch'
            sets = (_lib.ParamSet * len(uniq))()
            for k, (a, g, e, m) in enumerate(uniq):
                _lib.check(_lib.lib().cobel_param_set_fill(float(a), float(g), float(e), float(m),
                                                           C.byref(sets[k])))
            raw = np.frombuffer(sets, dtype=np.uint8).copy()
            self._pset_dev = torch.as_tensor(raw, device=self.device)
            self._pidx_dev = torch.as_tensor(
                np.ascontiguousarray(inv.reshape(-1).astype(np.uint16)).view(np.int16),
                device=self.device)
            self._pset_key, self._pset_n = key, len(uniq)
        for name, v in zip(names, cols[0]):      # placeholders; the kernel reads the sets
            setattr(run, name, float(v))
        run.param_sets, run.param_index = _lib.ptr(self._pset_dev), _lib.ptr(self._pidx_dev)
        run.n_param_sets = self._pset_n

    # -- device state -------------------------------------------------------------------------
    def _bind(self, interface) -> None:
        """Allocate per-instance state on first contact with an environment."""
        if self.inst is not None:
            assert interface.n_envs == self.n_envs and interface.device == self.device, \
                'an agent stays bound to the instance count / device it first trained on'
            return
        self.n_envs, self.device = interface.n_envs, interface.device
        self._seed, self._instance_base = interface.seed, interface.instance_base
        if self.n_states is None:
            self.n_states = interface.handle.n_states
            # (a mask assigned before the first training run is kept: the reference's agents take
            #  theirs as an attribute set after construction, dyna_q.py:134-136)
            if self.action_mask is None:
                self.action_mask = np.ones((self.n_states, self.n_actions), dtype=bool)
            assert np.shape(self.action_mask) == (self.n_states, self.n_actions), \
                'action_mask must be (%d, %d)' % (self.n_states, self.n_actions)
            # the key an agent's table is indexed by: the pose, or (pre-rendered observations) the
            # flattened observation components in order
            keys = interface.observation_key_table() if hasattr(interface, 'observation_key_table') \
                else interface.pose
            self._poses = np.asarray(keys, dtype=np.float64)
            # The reference keys its table by the observation itself (agent/q.py:150-158): two nodes
            # with the same observation — "manually defined observations" may repeat — SHARE one
            # row there.  The tables here are indexed by node, so such a world would learn
            # differently without a word; refuse it instead.
            uniq = np.unique(self._poses.reshape(len(self._poses), -1), axis=0)
            if len(uniq) != len(self._poses):
                raise NotImplementedError(
                    '%d of the %d nodes share their observation with another node: the reference '
                    'would let them share a table row (agent/q.py:150-158), the device tables are '
                    'indexed by node' % (len(self._poses) - len(uniq), len(self._poses)))
        else:
            assert int(interface.observation_space.n) == self.n_states
        self.inst = torch.zeros((self.n_envs, _lib.I_WORDS), dtype=torch.int32, device=self.device)
        self.monitors = DeviceMonitors(self.device, interface.handle.n_worlds, self.n_states,
                                       self.track_occupancy, self.track_responses,
                                       self.monitor_stripes if self.n_envs >= 4096 else 1)
        self._last_exp = torch.zeros((self.n_envs, 6), dtype=torch.int32, device=self.device)
        self._alloc_tables()

    @abc.abstractmethod
    def _alloc_tables(self) -> None:
        ...

    @abc.abstractmethod
    def _launch(self, interface, pol, flags: int, trials_target: int, steps: int, budget: int,
                batch: int) -> None:
        ...

    def _mask_bits(self):
        m = np.asarray(self.action_mask, dtype=bool).reshape(self.n_states, self.n_actions)
        assert m.any(axis=1).all(), 'The action mask masks all actions!'
        # (one byte per state up to eight actions, one 32-bit word beyond: cobel_hip.h)
        bits = (m * (1 << np.arange(self.n_actions, dtype=np.int64))).sum(axis=1)
        bits = bits.astype(np.uint8 if self.n_actions <= 8 else np.uint32)
        if self.n_actions > 8:
            bits = bits.view(np.int32)
        return torch.as_tensor(bits, device=self.device)

    def _policy_in(self, pol, interface, test: bool) -> int:
        """Adopt the policy's stream + counter for a run; returns extra flags."""
        if pol.seed is None:
            pol.seed = interface.seed
        assert pol.seed == interface.seed, \
            'all streams of an instance derive from the environment seed'
        if pol.stream is None:
            pol.stream = (_lib.STREAM_POLICY_TEST if (test and pol is not self.policy)
                          else _lib.STREAM_POLICY)
        if pol.counter is None or pol.counter.numel() != self.n_envs:
            pol.counter = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self.inst[:, _lib.I_CTR_POLICY] = pol.counter
        return _lib.F_TEST_STREAM if pol.stream == _lib.STREAM_POLICY_TEST else 0

    def _policy_out(self, pol) -> None:
        pol.counter.copy_(self.inst[:, _lib.I_CTR_POLICY])

    def _env_in(self, interface) -> None:
        self.inst[:, _lib.I_STATE] = interface.state
        self.inst[:, _lib.I_CTR_ENV] = interface.env_ctr
        self.inst[:, _lib.I_FLAGS] = 0
        self.inst[:, _lib.I_STEP] = 0
        self.inst[:, _lib.I_TRIAL] = self.current_trial

    def _env_out(self, interface) -> None:
        interface.state.copy_(self.inst[:, _lib.I_STATE])
        interface.env_ctr.copy_(self.inst[:, _lib.I_CTR_ENV])

    def _trial_reward(self) -> np.ndarray:
        return self.inst[:, _lib.I_REWARD_LO:_lib.I_REWARD_HI + 1].contiguous().view(
            torch.float64).reshape(-1).cpu().numpy()

    def env_steps(self) -> int:
        """Env steps executed by this agent's kernels so far (this rank)."""
        return int(self.monitors.steps_done.item()) if self.monitors is not None else 0

    # -- the shared trial driver ----------------------------------------------------------------
    def _trial_logs(self, logs: dict) -> dict:
        """Agent-specific keys of the logs a trial starts with."""
        return logs

    def _after_trial(self, logs: dict) -> dict:
        """Called after the launch(es) of one trial in per-trial mode, before on_trial_end."""
        return logs

    def _session(self, interface, trials: int, steps: int, batch: int, learn: bool,
                 extra_flags: int = 0, pol=None) -> None:
        self._bind(interface)
        if pol is None:
            pol = self.policy if learn else self.policy_test
        flags = extra_flags | (_lib.F_LEARN if learn else 0)
        if self.mask_actions:
            flags |= _lib.F_MASK_ACTIONS
        self._env_in(interface)
        flags |= self._policy_in(pol, interface, not learn)
        first = self.current_trial
        self.monitors.reserve(first + trials, self.n_envs, self.track_instances)
        per_step = self.n_envs == 1 and self.callbacks.has('on_step_begin', 'on_step_end')
        per_trial = self.n_envs == 1 and (per_step or self.callbacks.has('on_trial_begin',
                                                                         'on_trial_end'))
        if not per_trial:
            for t in range(trials):
                self.callbacks.on_trial_begin({'trial_reward': 0.0, 'trial': first + t,
                                               'trial_session': t})
            self._launch(interface, pol, flags, first + trials, steps, 0, batch)
            self.current_trial = first + trials
            if self.callbacks.has('on_trial_end'):
                lat, rew = self.monitors.mean_latency(), self.monitors.mean_reward()
                cnt = self.monitors.lat_cnt.cpu().numpy()
                for t in range(trials):
                    self.callbacks.on_trial_end({
                        'trial_reward': float(rew[first + t]), 'trial': first + t,
                        'trial_session': t, 'steps': float(lat[first + t]),
                        'count': int(cnt[first + t])})
        else:
            for t in range(trials):
                logs = self.callbacks.on_trial_begin(self._trial_logs(
                    {'trial_reward': 0.0, 'trial': self.current_trial, 'trial_session': t}))
                target = self.current_trial + 1
                if per_step:
                    step = 0
                    while int(self.inst[0, _lib.I_TRIAL].item()) < target:
                        logs['step'] = step
                        logs = self.callbacks.on_step_begin(logs)
                        self._launch(interface, pol, flags, target, steps, 1, batch)
                        e = self._last_exp[0].cpu().numpy()
                        r = float(e[4:5].view(np.float32)[0])
                        logs['trial_reward'] += r
                        logs.update({'state': int(e[0]), 'action': int(e[1]), 'reward': r,
                                     'next_state': int(e[2]), 'terminal': int(e[3])})
                        if learn:
                            logs['td'] = float(e[5:6].view(np.float32)[0])
                        logs = self.callbacks.on_step_end(logs)
                        step += 1
                    logs['steps'] = step - 1
                else:
                    self._launch(interface, pol, flags, target, steps, 0, batch)
                    lat = self.monitors.lat_sum[self.current_trial].item()
                    logs['steps'] = int(lat)
                    logs['trial_reward'] = float(
                        self.monitors.reward_sum[self.current_trial].item())
                self.current_trial += 1
                logs = self._after_trial(logs)
                logs = self.callbacks.on_trial_end(logs)
                if self.stop:
                    break
        self._policy_out(pol)
        self._env_out(interface)
        # (a sliced launch that gave up waiting for a ring entry leaves incomplete tables: raise here,
        #  not when somebody happens to ask for the step count — the abort word is sticky)
        check = getattr(self, 'check_launches', None)
        if check is not None:
            check()
