#!/bin/bash
# One traced training step (run on the GPU box): rocprofv3 kernel trace of a short bench run -> per-kernel timeline + per-queue accounting.
# usage: bash tools/trace_step.sh <tag>   ->  gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_streams.txt
tag=${1:-trace}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f > gpurun_out/${tag}_timeline.txt
python3 tools/trace_streams.py $f > gpurun_out/${tag}_streams.txt
rm -rf gpurun_out/${tag}_trace
