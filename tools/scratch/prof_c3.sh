#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/err.log
cd $R
python3 - <<PY
import csv, glob
f = glob.glob('$O/prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'chomp' in r['Name']:
        print('%-30s calls %5s avg %10.1f us min %10.1f max %10.1f  %5.1f%%' % (r['Name'].split('(')[0].replace('chomp::', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
PY
