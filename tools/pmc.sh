#!/bin/bash
# usage: tools/pmc.sh <tag> <prof_one args...>   -> gpurun_out/pmc_<tag>_{1,2}/ ; prints per-kernel counter sums
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_1 -- python3 $GRAFT_REPO_ROOT/tools/prof_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_2 -- python3 $GRAFT_REPO_ROOT/tools/prof_one.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if "conv3x3" in k or "wgrad3x3" in k or "conv_igemm" in k or "falnet" in k or "_kernel" in k and "at::" not in k:
            print(k, {c: f"{v:.3g}" for c, v in d.items()})
PY
