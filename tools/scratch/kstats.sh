#!/bin/bash
# usage: kstats.sh <tag>: profiled short bench only (no tests), prints chomp kernel times
tag=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/err.log
cd $R
python3 - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'chomp' in r['Name']:
        print('%-28s calls %4s avg %10.1f us  min %9.1f' % (r['Name'].split('(')[0].replace('chomp::','').replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
d=json.load(open('gpurun_out/$tag/bench.json')); print('value %.4g samples/s  ms/step %.4f' % (d['value'], d['ms_per_step']))
PY
