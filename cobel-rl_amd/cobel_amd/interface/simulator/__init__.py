from .offline import OfflineSimulator  # noqa: F401
