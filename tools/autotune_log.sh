#!/bin/bash
# Candidate timings of every conv launch of the default bench (plan build with an empty cache): bash tools/autotune_log.sh > log   (GPU box)
FALNET_AUTOTUNE_CACHE=0 FALNET_AUTOTUNE_LOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep "^\[autotune\]"
