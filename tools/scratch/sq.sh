#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sq/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq/a.log 2>&1
echo rc=$?
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/sq/b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq/b.log 2>&1
echo rc=$?
cd $R
python3 - <<PY
import csv, glob, collections
for d in ('a', 'b'):
    f = glob.glob('gpurun_out/sq/%s/**/*counter_collection.csv' % d, recursive=True)
    if not f: print('no csv', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        name = r['Kernel_Name'].split('(')[0].replace('chomp::', '')
        if 'k_' in name:
            acc[name][r['Counter_Name']] += float(r['Counter_Value']); n[name].add(r['Dispatch_Id'])
    for k, v in acc.items():
        c = len(n[k]); print(k, c, {a.replace('SQ_', ''): round(b / c) for a, b in v.items()})
PY
