"""MI355X-native batched board evaluation for Batch-First (hot path only)."""
from . import spec  # noqa: F401

__all__ = ["spec"]
