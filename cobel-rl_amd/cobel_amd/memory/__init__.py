from .dqn import DQNMemory  # noqa: F401
from .dyna_q import DynaQMemory  # noqa: F401
from .sfma import SFMAMemory  # noqa: F401
