"""`models` package of the reference (models/__init__.py:1-6): FAL_netA, FAL_netB (the hot path), FAL_netC."""
from .FAL_netA import FAL_netA  # noqa: F401
from .FAL_netB import FAL_netB  # noqa: F401
from .FAL_netC import FAL_netC  # noqa: F401

__all__ = ("FAL_netA", "FAL_netB", "FAL_netC")
