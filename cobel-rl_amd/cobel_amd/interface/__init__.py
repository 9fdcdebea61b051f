from .gridworld import Gridworld, WorldHandle  # noqa: F401
from .interface import Interface  # noqa: F401
from .topology import Topology  # noqa: F401
from .simulator import OfflineSimulator  # noqa: F401
