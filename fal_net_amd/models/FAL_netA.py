"""FAL_netA on MI355X (reference: models/FAL_netA.py): the FAL_net topology with separable 3x1 / 1x3 residual convs
(:73-76), its own channel table (:99-126), checkpoint keys under `BackBone.` (:183), no amask_conv, and the right
occlusion mask sampled with grid_sample's default align_corners=False (:264).  Same launch plan and kernels as FAL_netB
(fal_net_amd/models/FAL_netB.py), driven by the layer table in fal_net_amd/arch.py."""
from .FAL_netB import _make

__all__ = ["FAL_netA"]


def FAL_netA(data=None, no_levels=33, compute_dtype=None):
    """Factory with the reference's signature (models/FAL_netA.py:28-32)."""
    return _make("A", data, no_levels, compute_dtype)
