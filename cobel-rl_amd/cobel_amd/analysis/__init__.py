from .behavior_spatial import get_occupancy_map, occupancy_from_counts  # noqa: F401
