"""Behavioural monitors of the reference (monitor/behavior.py:16-464: ``EscapeLatencyMonitor``,
``RewardMonitor``, ``ResponseMonitor``, ``TrajectoryMonitor``, ``QMonitor``) without the plotting.

``update(logs)`` has the reference's semantics (``latency_trace[trial] = logs['steps']`` and the
11-trial running nan-mean, behavior.py:82-85), so it can be registered under ``on_trial_end``
unchanged.  With vectorised agents the kernels reduce ``logs['steps']`` over instances on device
(``agent.monitors``); ``update_from_device`` ingests those sums — summed over ranks when running
on several GPUs (``DeviceMonitors.all_reduce`` returns the global sums and leaves the rank-local
accumulators alone, so any number of monitors and reporting intervals may ask) — and fills the
traces with per-trial means.
"""
from __future__ import annotations

import numpy as np


class Monitor:
    def __init__(self, widget=None) -> None:
        self.widget = widget

    def clear_plots(self) -> None:
        pass

    def refresh_visualization(self) -> None:
        pass


class EscapeLatencyMonitor(Monitor):
    def __init__(self, trials: int, max_steps: int, widget=None) -> None:
        super().__init__(widget)
        self.max_steps = max_steps
        self.latency_trace = np.full(trials, float('nan'), dtype='float')
        self.latency_trace_avg = np.copy(self.latency_trace)

    def update(self, logs: dict) -> None:
        trial = logs['trial']
        self.latency_trace[trial] = logs['steps']
        avg = np.nanmean(self.latency_trace[max(0, trial - 10): (trial + 1)])
        self.latency_trace_avg[trial] = self.max_steps if np.isnan(avg) else avg

    def update_from_device(self, monitors, reduce: bool = True) -> None:
        """Fill the traces from an agent's device-side reductions (mean over instances)."""
        sums = monitors.all_reduce() if reduce else monitors
        lat = sums.mean_latency()
        for trial in range(min(len(lat), len(self.latency_trace))):
            if not np.isnan(lat[trial]):
                self.update({'trial': trial, 'steps': lat[trial]})

    def get_trace(self):
        return self.latency_trace


class RewardMonitor(Monitor):
    def __init__(self, trials: int, reward_range=(0.0, 1.0), widget=None) -> None:
        super().__init__(widget)
        self.reward_range = reward_range
        self.reward_trace = np.full(trials, float('nan'), dtype='float')
        self.reward_trace_avg = np.copy(self.reward_trace)

    def update(self, logs: dict) -> None:
        trial = logs['trial']
        self.reward_trace[trial] = logs['trial_reward']
        avg = np.nanmean(self.reward_trace[max(0, trial - 10): (trial + 1)])
        self.reward_trace_avg[trial] = self.reward_range[0] if np.isnan(avg) else avg

    def update_from_device(self, monitors, reduce: bool = True) -> None:
        sums = monitors.all_reduce() if reduce else monitors
        rew = sums.mean_reward()
        for trial in range(min(len(rew), len(self.reward_trace))):
            if not np.isnan(rew[trial]):
                self.update({'trial': trial, 'trial_reward': rew[trial]})

    def get_trace(self):
        return self.reward_trace


class ResponseMonitor(Monitor):
    """Per-trial responses and their cumulative curve (behavior.py:212-301).  A response is what
    the user's callback put into ``logs['response']``; without one it is whether the trial was
    rewarded."""

    def __init__(self, trials: int, widget=None) -> None:
        super().__init__(widget)
        self.responses = np.full(trials, float('nan'), dtype='float')
        self.CRC = np.copy(self.responses)

    def update(self, logs: dict) -> None:
        trial = logs['trial']
        if 'response' in logs:
            self.responses[trial] = logs['response']
        else:
            self.responses[trial] = int(logs['trial_reward'] > 0)
        self.CRC[trial] = np.sum(self.responses[: (trial + 1)])

    def update_from_device(self, monitors, reduce: bool = True) -> None:
        """Vectorised runs: the response of a trial is the FRACTION of instances rewarded in it
        (mean of the per-instance default responses)."""
        sums = monitors.all_reduce() if reduce else monitors
        rate = sums.mean_response()
        for trial in range(min(len(rate), len(self.responses))):
            if not np.isnan(rate[trial]):
                self.update({'trial': trial, 'response': rate[trial]})

    def get_trace(self):
        return self.responses


class TrajectoryMonitor(Monitor):
    """Positions visited, one list per trial (behavior.py:304-385); register under
    ``on_step_end``."""

    def __init__(self, trials: int, env, widget=None) -> None:
        super().__init__(widget)
        self.env = env
        self.trajectory_trace: list = []
        self.current_trial = None

    def update(self, logs: dict) -> None:
        if self.current_trial != logs['trial_session']:
            self.current_trial = logs['trial_session']
            self.trajectory_trace.append([])
        self.trajectory_trace[-1].append(self.env.get_position())

    def get_trace(self):
        return self.trajectory_trace


class QMonitor(Monitor):
    """Q-function of a fixed set of observations after every update (behavior.py:388-464)."""

    def __init__(self, trials: int, observations, widget=None) -> None:
        super().__init__(widget)
        self.observations = observations
        self.q_trace: list = []

    def update(self, logs: dict) -> None:
        self.q_trace.append(logs['agent'].predict_on_batch(self.observations))

    def get_trace(self):
        return self.q_trace
