from .dqn import DQNMemory  # noqa: F401
from .dyna_q import DynaQMemory  # noqa: F401
