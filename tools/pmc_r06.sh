#!/bin/bash
# Round-6 counter rows (run on the GPU box): SQ counters + HBM bytes of the wave-streaming kernels and of the kernels they replace.
out=gpurun_out/r06_pmc_raw.txt
: > $out
run() { echo "### $*" >> $out; "$@" >> $out 2>&1; }
run bash tools/pmc.sh wv32 wgrad 32 32 256 512
WGRAD_VARIANT=0 run bash tools/pmc.sh wv32old wgrad 32 32 256 512
run bash tools/pmc.sh wv64 wgrad 32 49 256 512
run bash tools/pmc.sh cw27 conv 32 32 256 512 27
run bash tools/pmc.sh cw10 conv 32 32 256 512 10
run bash tools/pmc.sh r16 wgrad 64 64 128 256
run bash tools/pmc.sh v23 conv 64 64 256 512 23
run bash tools/pmc_mem.sh wv32 wgrad 32 32 256 512
run bash tools/pmc_mem.sh cw27 conv 32 32 256 512 27
cat $out | cut -c1-400
