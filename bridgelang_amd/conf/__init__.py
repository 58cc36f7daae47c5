from .vla import VLAConfig, VLARegistry  # noqa: F401
