"""Drop-in alias so that `import models; models.__dict__['FAL_netB'](data, no_levels=49)` (the call the
reference's entry scripts make, Train_Stage1_K.py:171) resolves to the MI355X implementation."""
from fal_net_amd.models import FAL_netA, FAL_netB, FAL_netC  # noqa: F401

__all__ = ("FAL_netA", "FAL_netB", "FAL_netC")
