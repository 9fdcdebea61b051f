"""Parameter search over simulations (``cobel.optimizer``): the abstract ``Optimizer`` and the
``GridSearchOptimizer`` with an additional vectorised mode in which every parameter combination
and every run becomes one environment instance of a single launch."""
from .optimizer import Optimizer  # noqa: F401
from .grid_search import GridSearchOptimizer, spread_over_instances  # noqa: F401
