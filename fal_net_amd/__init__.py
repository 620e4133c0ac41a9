import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A step uses three concurrent streams (data gradients, weight
# gradients, label VGG) beside torch's own; when two of them land on one hardware queue they serialize -- measured on MI355X: one run
# in four-to-eight loses 1.5-13 % (1185 / 1107 instead of 1280 pairs/s), none of sixteen with 8 queues.  Must be set before the HIP
# runtime initialises (i.e. before the first CUDA call of the process); an explicit user setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
