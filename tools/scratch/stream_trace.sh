#!/bin/bash
# per-launch durations of the Stage E kernels over the bench's back-to-back trains
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench.json 2> $O/err.log
cd $R
python3 - <<PY
import csv, glob, json, numpy
f = glob.glob('$O/prof/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_power' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
st = [(r['Kernel_Name'].split('(')[0].split('::')[-1], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
big = [i for i, s in enumerate(st) if s[0].startswith('k_power_stream')]
d = numpy.array([(st[i][2] - st[i][1]) / 1e3 for i in big])
print('k_power_stream launches', len(d), 'mean %.1f min %.1f max %.1f' % (d.mean(), d.min(), d.max()))
print('first 30:', numpy.round(d[:30], 1).tolist())
print('100-train tail:', numpy.round(d[80:101], 1).tolist())
print('last 45:', numpy.round(d[-45:], 1).tolist())
# gaps between prep end and stream start, stream end and lanes start
g1 = numpy.array([(st[i][1] - st[i-1][2]) / 1e3 for i in big if i > 0])
print('gap prep->stream mean %.2f us' % g1.mean())
d = json.load(open('$O/bench.json')); print(json.dumps(d['roofline']['whole_call']), d['roofline']['avg_launch_us'])
PY
