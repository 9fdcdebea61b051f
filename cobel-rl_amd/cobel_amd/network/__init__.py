from .network import Network  # noqa: F401
from .network_torch import StackedTorchNetwork, TorchNetwork  # noqa: F401
