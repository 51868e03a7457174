#!/bin/bash
# GPU box: kernel stats of one of the other workloads (c3 | c4 | c5).  usage: quick_other.sh <tag> <workload>...
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_$w.json 2> $O/err_$w.log || { echo "$w failed"; tail -5 $O/err_$w.log; exit 1; }
  cp $O/prof_$w/*/*kernel_stats.csv $O/kernel_stats_$w.csv
  python3 - <<PY
import csv, json
d = json.load(open('$O/bench_$w.json'))
print('== $w  %.4f ms/step  %.4g %s' % (d['ms_per_step'], d['value'], d['unit']))
for r in list(csv.DictReader(open('$O/kernel_stats_$w.csv')))[:14]:
    print('%-34s calls %4s avg %8.1f us  %5.1f%%' % (r['Name'].split('(')[0].replace('chomp::','').replace('void ','')[:34], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
done
