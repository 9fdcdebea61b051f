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
receives exactly the update it would compute alone.  Same constructor, attributes and methods as
the reference (``models_online`` / ``models_target`` are dictionaries of single-instance views).
"""
from __future__ import annotations

import numpy as np
import torch

from ..spaces import Discrete
from .agent import DeviceMonitors
from .dyna_dqn import DynaDQN


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
