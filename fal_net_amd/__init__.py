"""fal_net_amd: the FAL_netB hot path on MI355X (hand-written HIP kernels behind the reference's Python surface).

Hardware queues.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A training step uses three concurrent
streams (data gradients, weight gradients, label VGG) beside torch's own; when two of them land on one hardware queue they
serialize -- measured on MI355X: one run in four-to-eight loses 1.5-13 % (1185 / 1107 instead of 1280 pairs/s), none of sixteen
with 8 queues.  The variable is process-global and must be set before the HIP runtime initialises, so it is the APPLICATION's
call, not a library's: `bench.py` and the `Train_*` / `Test_KITTI` entry scripts set GPU_MAX_HW_QUEUES=8 themselves (a user
setting wins).  Importing this package changes nothing unless the host application opts in with FALNET_HW_QUEUES=<n>."""
import os as _os

if _os.environ.get("FALNET_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", _os.environ["FALNET_HW_QUEUES"])
