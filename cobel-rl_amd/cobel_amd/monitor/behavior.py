"""Behavioural monitors — ``EscapeLatencyMonitor`` and ``RewardMonitor`` of the reference
(monitor/behavior.py:14-209) without the plotting.

``update(logs)`` has the reference's semantics (``latency_trace[trial] = logs['steps']`` and the
11-trial running nan-mean, behavior.py:82-85), so it can be registered under ``on_trial_end``
unchanged.  With vectorised agents the kernels reduce ``logs['steps']`` over instances on device
(``agent.monitors``); ``update_from_device`` ingests those sums — after an all-reduce over ranks
when running on several GPUs — and fills the traces with per-trial means.
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
        if reduce:
            monitors.all_reduce()
        lat = monitors.mean_latency()
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
        if reduce:
            monitors.all_reduce()
        rew = monitors.mean_reward()
        for trial in range(min(len(rew), len(self.reward_trace))):
            if not np.isnan(rew[trial]):
                self.update({'trial': trial, 'trial_reward': rew[trial]})

    def get_trace(self):
        return self.reward_trace
