#!/bin/bash
# per-kernel durations of Stage E on the 2^20 x 64 grid
tag=$1
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
export STAGE_E_ONE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/scratch/stage_e.py > $GRAFT_REPO_ROOT/gpurun_out/$tag/out.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/$tag/out.txt | tail -5
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if "power_" in r["Name"]:
        print('%-28s calls %4s avg %10.1f us  min %9.1f max %9.1f' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
