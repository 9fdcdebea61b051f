from .greedy import EpsilonGreedy  # noqa: F401
from .policy import Policy  # noqa: F401
