"""fal_net_amd: the FAL_netB hot path on MI355X (hand-written HIP kernels behind the reference's Python surface).

Hardware queues.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and the chip's compute pipes run FOUR of them side by side.
A training step keeps exactly four streams busy (data gradients, two weight-gradient streams, the auxiliary stream: label VGG in forward, the gradient
buckets' all-reduces in a data-parallel backward -- fal_net_amd/train.py: enable_overlapped_allreduce).  Two of them on ONE hardware queue serialise
(one run in four-to-eight lost 1.5-13 % with 4 queues): the per-device stream self-test (fal_net_amd/plan.py: StepStreams.selftest) re-creates any stream
that shares a queue with another.  A FIFTH busy queue time-slices with one of the four: rounds 4-5 ran the collectives on the backend's own stream and
measured that as 0.3 ms of "exposed communication" on a world-1 RCCL group and as a cliff at more queues (box 1 of profiles/r05_ab_dist_queues.txt:
5.77 / 5.83 / 6.64 / 9.07 ms at 5 / 6 / 7 / 8 queues against 5.53-5.55 single-GPU); round 6 has no fifth stream (profiles/r06_ab_dist.txt: 5.14 vs 5.14 ms).
The scripts ask for FIVE queues (the four + the null stream's).  The variable is
process-global and must be set before the HIP runtime initialises, so it is the APPLICATION's call, not a library's: `bench.py` and
the `Train_*` / `Test_KITTI` entry scripts set GPU_MAX_HW_QUEUES=5 themselves (a user setting wins; a bucket hook under more than five queues is
flagged in the self-test's record).  Importing this package changes
nothing unless the host application opts in with FALNET_HW_QUEUES=<n>."""
import os as _os

if _os.environ.get("FALNET_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", _os.environ["FALNET_HW_QUEUES"])
