#!/bin/bash
w=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$w; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/err.log
cd $R
python3 - <<PY
import csv, glob
f = glob.glob('$O/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'chomp' in r['Name'] and float(r['Percentage']) > 0.5:
        print('%-32s calls %4s avg %10.1f us  total %8.2f ms  %5.1f%%' % (r['Name'].split('(')[0].replace('chomp::', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
PY
