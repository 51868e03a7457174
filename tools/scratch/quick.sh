#!/bin/bash
# usage: quick.sh <tag>  -- GPU tests + profiled short bench, prints chomp kernel times
tag=$1
mkdir -p gpurun_out/$tag
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$tag/tests.log
grep -E "^(FAILED|ERROR|E  |[0-9]+ (passed|failed)|pytest rc)" gpurun_out/$tag/tests.log | cut -c1-300 | head -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$tag/err.log
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'chomp' in r['Name']:
        print('%-28s calls %4s avg %10.1f us  min %9.1f' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
d=json.load(open('gpurun_out/$tag/bench.json')); print('value %.4g samples/s  ms/step %.4f' % (d['value'], d['ms_per_step']), d['stage_split_rank0'], 'roof %.1f GB/s' % d['roofline']['achieved'])
PY
