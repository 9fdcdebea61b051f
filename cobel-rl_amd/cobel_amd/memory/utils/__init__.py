from .metrics import DR, SR, Euclidean, Metric  # noqa: F401
