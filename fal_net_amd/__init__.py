"""fal_net_amd: the FAL_netB hot path on MI355X (hand-written HIP kernels behind the reference's Python surface).

Hardware queues.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A training step keeps four streams busy
(data gradients, two weight-gradient streams, label VGG) and a collective's stream beside them; two of them on one hardware queue
serialise (one run in four-to-eight lost 1.5-13 % with 4 queues), and with MANY queues active the data-parallel step falls off a
cliff (world-1 RCCL group, same box: 5.33 ms at 5 queues, 5.83 at 6, 6.64 at 7, 9.07 at 8; the single-GPU step is 5.12-5.13 at every
count from 4 to 8: profiles/r05_ab_dist_queues.txt).  So the scripts ask for FIVE queues, and the plan's stream self-test
(fal_net_amd/plan.py: stream_selftest) re-creates any stream that shares a queue with another stream of the step.  The variable is
process-global and must be set before the HIP runtime initialises, so it is the APPLICATION's call, not a library's: `bench.py` and
the `Train_*` / `Test_KITTI` entry scripts set GPU_MAX_HW_QUEUES=5 themselves (a user setting wins).  Importing this package changes
nothing unless the host application opts in with FALNET_HW_QUEUES=<n>."""
import os as _os

if _os.environ.get("FALNET_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", _os.environ["FALNET_HW_QUEUES"])
