"""Drop-in alias of the reference's `loss_functions` module (Train_Stage1_K.py:366 imports
`rec_loss_fnc, realEPE, smoothness, vgg` from it) -> MI355X implementation."""
from fal_net_amd.loss_functions import (EPE, Vgg19_pc, perceptual_loss, realEPE, rec_loss_fnc,  # noqa: F401
                                        set_compute_dtype, smoothness, vgg)
