#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/flops
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/flops/avail.txt 2>&1
grep -o "SQ_INSTS_VALU[A-Z0-9_]*" $R/gpurun_out/flops/avail.txt | sort -u | tr '\n' ' '
echo
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $R/gpurun_out/flops/p1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/flops/p1.log 2>&1
echo rc=$?
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/flops/p1/**/*counter_collection.csv', recursive=True)
print(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    name = r['Kernel_Name'].split('(')[0]
    if 'chomp' in name:
        acc[name][r['Counter_Name']] += float(r['Counter_Value']); n[name].add(r['Dispatch_Id'])
for k, v in acc.items():
    c = len(n[k]); print(k, c, {a: b / c for a, b in v.items()})
PY
