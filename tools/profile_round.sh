#!/bin/bash
# Run on the GPU box (through gpurun): the default bench line, then the same command under
# rocprofv3 --kernel-trace --stats, and the C3 workload.  Results land in gpurun_out/<tag>/;
# copy the summaries you want judged into profiles/.
#   usage: tools/profile_round.sh <tag>
set -e
tag=${1:-round}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --workload c3 --steps 5 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_c2_profiled.json 2> $O/prof.err
cd $R
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats_c2.csv
python3 - <<PY
import csv, json
d = json.load(open('$O/bench_c2.json'))
print('C2 value %.4g samples/s  %.4f ms/step  roofline %.0f GB/s (frac %.3f, %.1f us)  cpu %.3g (1 core) / %.3g (%d cores)' % (
    d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'],
    d['cpu_baseline']['value'], d['cpu_baseline']['pool']['value'], d['cpu_baseline']['pool']['cores']))
d3 = json.load(open('$O/bench_c3.json'))
print('C3 value %.4g samples/s  %.3f ms/step  cpu %.3g' % (d3['value'], d3['ms_per_step'], d3['cpu_baseline']['value']))
for r in csv.DictReader(open('$O/kernel_stats_c2.csv')):
    if 'chomp' in r['Name']:
        print('%-30s calls %5s avg %9.1f us min %9.1f max %9.1f  %5.1f%%' % (r['Name'].split('(')[0].replace('chomp::', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
PY
