from .agent import Agent, Callbacks  # noqa: F401
from .dqn import DQN  # noqa: F401
from .dyna_dqn import DynaDQN  # noqa: F401
from .dyna_dsr import DynaDSR  # noqa: F401
from .dyna_q import DynaQ  # noqa: F401
from .q import QAgent  # noqa: F401
from .sfma import SFMA  # noqa: F401
from .sr import SR  # noqa: F401
