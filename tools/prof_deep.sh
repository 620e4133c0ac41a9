#!/bin/bash
# rocprofv3 kernel durations of the levels-5/6 conv kernel (variant 19) against the gather kernel with split-K on the bench layers (run on the GPU box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export BENCH_NOCHECK=1 BENCH_LAYERS=conv5_1,conv6_1,iconv6,deconv6 BENCH_VARIANTS=deep32,deep64
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/deep_prof -- python3 tools/bench_conv.py bf16 > gpurun_out/deep_prof.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/deep_prof/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'deep' in n or 'igemm' in n or 'splitk' in n:
        d[(n[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('LDS_Block_Size',''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, len(v), 'median %.1f us  min %.1f' % (v[len(v)//2], v[0]))
PY
