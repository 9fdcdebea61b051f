"""Tabular Dyna-Q — ``cobel.agent.DynaQ`` (agent/dyna_q.py:17-330) on the fused HIP kernel.

Same constructor, ``train(interface, trials, steps, batch_size=32, no_replay=False)``,
``test``, ``predict_on_batch`` and attributes (``Q``, ``M``, ``learning_rate``, ``gamma``,
``action_mask``, ``mask_actions``, ``episodic_replay``, ``current_trial``, ``stop``).  Tables are
float32; see include/cobel_hip.h for the exact arithmetic (bit-exact against the reference run
with float32 tables).  Up to 62 planning updates per step one wavefront plans a batch; larger
``batch_size`` values (the reference has no limit) run on the general kernel of ``cobel_tab_run``.
"""
from __future__ import annotations

from .. import _lib
from ..memory.dyna_q import DynaQMemory
from ..spaces import Discrete
from .tabular import TabularAgent


class DynaQ(TabularAgent):
    agent_kind = _lib.AGENT_DYNAQ
    # the model records and their digest are laid out for four actions (the reference's DynaQ takes
    # Discrete observations only — gridworlds — so other action counts cannot reach it either)
    general_actions = False

    def __init__(self, observation_space, action_space, policy, policy_test=None,
                 learning_rate: float = 0.99, gamma: float = 0.99, memory=None,
                 custom_callbacks=None) -> None:
        assert type(observation_space) is Discrete, 'DynaQ requires a discrete observation space!'
        assert type(action_space) is Discrete, 'DynaQ requires a discrete action space!'
        super().__init__(observation_space, action_space, policy, policy_test, learning_rate,
                         gamma, custom_callbacks)
        self.M = DynaQMemory(self.n_states, self.n_actions) if memory is None else memory
        self.episodic_replay = False

    def _alloc_tables(self) -> None:
        super()._alloc_tables()
        self.M._bind(self.n_envs, self.device, self._seed, self._instance_base)

    def _extra(self, run) -> None:
        run.model = _lib.ptr(self.M.table)
        run.model_index = _lib.ptr(self.M.index)
        self.inst[:, _lib.I_CTR_MEMORY] = self.M.counter

    def _model_lr(self):
        return self.M.learning_rate

    def _launch(self, *args) -> None:
        super()._launch(*args)
        self.M.counter.copy_(self.inst[:, _lib.I_CTR_MEMORY])

    def train(self, interface, trials: int, steps: int, batch_size: int = 32,
              no_replay: bool = False) -> None:
        assert batch_size >= 0
        extra = (_lib.F_NO_REPLAY if no_replay else 0) | \
                (_lib.F_EPISODIC if self.episodic_replay else 0)
        self._session(interface, trials, steps, batch_size, True, extra)

    def test(self, interface, trials: int, steps: int) -> None:
        self._session(interface, trials, steps, 0, False)
