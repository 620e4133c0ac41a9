#!/bin/bash
# Everything the round's evidence needs, in one GPU-box call:  bash tools/round_artifacts.sh r03
#   1. regenerate the autotune cache over every bench workload / dtype (tools/regen_cache.sh) -> gpurun_out/<tag>/
#   2. the full GPU test suite with that cache
#   3. rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the default bench (tools/profile_round.sh) -> profiles/<tag>_*
#   4. one traced step (timeline + per-queue accounting) and the per-launch table
tag=${1:-r03}
export FALNET_COMMIT=${FALNET_COMMIT:-unknown}
bash tools/regen_cache.sh $tag > gpurun_out/${tag}_regen.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
bash tools/trace_step.sh $tag
python bench.py --launch-table gpurun_out/${tag}_launches.txt > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
tail -3 gpurun_out/${tag}_pytest.txt; tail -9 gpurun_out/${tag}_regen.log | cut -c1-300; tail -1 gpurun_out/${tag}_bench_full.json | cut -c1-600
