from .gridworld import Gridworld, WorldHandle  # noqa: F401
from .interface import Interface  # noqa: F401
