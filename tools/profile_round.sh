#!/bin/bash
# Round profile artifacts (run on the GPU box): kernel stats of the bench command + HBM traffic PMC passes.
# usage: bash tools/profile_round.sh r01
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $BENCH > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $BENCH > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $BENCH > $out/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_profile.py $out $tag
mkdir -p gpurun_out/profiles && cp profiles/${tag}_* gpurun_out/profiles/
