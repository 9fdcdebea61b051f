from .behavior import EscapeLatencyMonitor, Monitor, RewardMonitor  # noqa: F401
