from .behavior import (EscapeLatencyMonitor, Monitor, QMonitor, ResponseMonitor,  # noqa: F401
                       RewardMonitor, TrajectoryMonitor)
