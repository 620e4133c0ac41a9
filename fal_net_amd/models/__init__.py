"""`models` package of the reference (models/__init__.py:1-6); only FAL_netB is on the hot path."""
from .FAL_netB import FAL_netB  # noqa: F401

__all__ = ("FAL_netB",)
