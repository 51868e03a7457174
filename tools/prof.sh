#!/bin/bash
# Profiling recipes for the GPU box (run through gpurun).  Output under gpurun_out/<tag>/;
# copy what should be judged into profiles/.
#   tools/prof.sh trace <tag> [bench args]   kernel timeline of the last timed step (durations + gaps)
#   tools/prof.sh stats <tag> [bench args]   rocprofv3 --kernel-trace --stats summary
#   tools/prof.sh sq    <tag> [bench args]   SQ counters per kernel (two PMC passes + fp64 op mix)
mode=$1; tag=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
ARGS="--no-cpu-baseline --no-roofline --steps 6 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
case $mode in
trace)
  rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py $ARGS > $O/bench.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
  cd $R
  python3 - <<PY
import csv, glob
f = glob.glob('$O/prof/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'chomp::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('chomp::', '').replace('void ', '')
# steps start at k_sigma_nodes: print the last complete one
starts = [i for i, r in enumerate(rows) if name(r).startswith('k_sigma_nodes') or name(r).startswith('k_proj_chi')]
if len(starts) >= 2:
    a, b = starts[-2], starts[-1]
    t0 = int(rows[a]['Start_Timestamp']); prev = t0
    print('%-28s %9s %9s %9s  %s' % ('kernel', 'start us', 'dur us', 'gap us', 'grid x wg, vgpr, lds'))
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%-28s %9.1f %9.1f %9.1f  %s x %s, %s, %s' % (name(r), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3,
              r.get('Grid_Size_X', '?') + ',' + r.get('Grid_Size_Y', '?') + ',' + r.get('Grid_Size_Z', '?'),
              r.get('Workgroup_Size_X', '?'), r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?')))
        prev = e
    print('step span %.1f us (first start -> next step start %.1f us)' % ((prev - t0) / 1e3, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
print(open('$O/bench.json').read().strip())
PY
  ;;
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py $ARGS > $O/bench.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
  cd $R
  cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
  python3 - <<PY
import csv
for r in csv.DictReader(open('$O/kernel_stats.csv')):
    if 'chomp' in r['Name']:
        print('%-30s calls %5s avg %10.1f us min %10.1f max %10.1f  %5.1f%%' % (r['Name'].split('(')[0].replace('chomp::', '').replace('void ', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
print(open('$O/bench.json').read().strip())
PY
  ;;
sq)
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/bench.py $ARGS > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; exit 1; }
  done
  cd $R
  python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ('p1', 'p2', 'p3'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('chomp::', '').replace('void ', '')
            if name.startswith('k_'):
                acc[name][r['Counter_Name']] += float(r['Counter_Value']); n[name][r['Counter_Name']].add(r['Dispatch_Id'])
out = {}
for k, v in acc.items():
    out[k] = {a.replace('SQ_', ''): b / max(1, len(n[k][a])) for a, b in v.items()}
    c = out[k]
    if 'INSTS_VALU' in c and 'BUSY_CYCLES' in c:
        f64 = 2 * c.get('INSTS_VALU_FMA_F64', 0) + c.get('INSTS_VALU_ADD_F64', 0) + c.get('INSTS_VALU_MUL_F64', 0) + c.get('INSTS_VALU_TRANS_F64', 0)
        c['fp64_flop_per_launch'] = 64 * f64
        c['valu_active_frac_of_wave_cycles'] = c.get('ACTIVE_INST_VALU', 0) / max(1.0, c.get('WAVE_CYCLES', 1))
    print(k, json.dumps({a: round(b, 3) if b < 10 else round(b) for a, b in c.items()}))
json.dump(out, open('$O/sq_counters.json', 'w'), indent=1)
PY
  ;;
esac
