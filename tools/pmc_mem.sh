#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <prof_one args...> : cache-path counters for one launch configuration
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcm_${tag}_1 -- python3 $GRAFT_REPO_ROOT/tools/prof_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcm_${tag}_2 -- python3 $GRAFT_REPO_ROOT/tools/prof_one.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcm_${tag}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if "conv3x3" in k or "wgrad3x3" in k or "conv_igemm" in k:
            print(k[:50], {c: f"{v/20:.4g}" for c, v in d.items()}, "(per launch)")
PY
