from .dyna_q import DynaQMemory  # noqa: F401
