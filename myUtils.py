"""Drop-in alias of the reference's `myUtils` (imported as `utils` by its entry scripts)."""
from fal_net_amd.myUtils import *  # noqa: F401,F403
from fal_net_amd.myUtils import kitti_error_names, width_to_baseline, width_to_focal  # noqa: F401
