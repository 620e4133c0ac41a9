"""FAL_netC on MI355X (reference: models/FAL_netC.py): FAL_netB with a wider bottleneck (conv5/conv6 512 channels, iconv6
512, deconv5 256; :110-120) and checkpoint keys under `synth.` (:185).  Same launch plan and kernels as FAL_netB
(fal_net_amd/models/FAL_netB.py), driven by the layer table in fal_net_amd/arch.py."""
from .FAL_netB import _make

__all__ = ["FAL_netC"]


def FAL_netC(data=None, no_levels=33, compute_dtype=None):
    """Factory with the reference's signature (models/FAL_netC.py:29-33)."""
    return _make("C", data, no_levels, compute_dtype)
