from .network import Network  # noqa: F401
from .network_torch import (FlexibleTorchNetwork, StackedTorchNetwork,  # noqa: F401
                            TorchNetwork)
