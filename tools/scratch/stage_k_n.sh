#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 16 64 1024; do
  O=$R/gpurun_out/kn$n; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/scratch/stage_k_n.py $n > $O/out.log 2>&1
  echo "== n=$n"
  python3 - <<PY
import csv, glob
f = glob.glob('$O/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'chomp' in r['Name']:
        print('%-30s calls %4s avg %9.1f us min %9.1f' % (r['Name'].split('(')[0].replace('chomp::','').replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
