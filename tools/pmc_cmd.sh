#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <kernel-name-substring> <python script + args...>   -> per-kernel SQ counter sums (two passes)
tag=$1; pat=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_1 -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_2 -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "$pat" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k] += 1
    for k, d in agg.items():
        n = cnt[k] / len(d)
        print(k, f"launches={n:.0f}", {c: f"{v / n:.4g}" for c, v in d.items()})
PY
