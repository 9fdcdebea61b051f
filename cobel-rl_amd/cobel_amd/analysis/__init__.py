from .behavior_spatial import get_occupancy_map, match, occupancy_from_counts  # noqa: F401
from .utils import state_to_coordinates, states_to_coordinates  # noqa: F401
