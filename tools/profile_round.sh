#!/bin/bash
# Run on the GPU box (through gpurun): everything a round's profiles/ is made of.
#   the default bench line; the same command under rocprofv3 --kernel-trace --stats; the
#   other workloads' kernel stats; the SQ counters of Stage K (c2, c3); the TCC passes of
#   the streaming Stage-E kernel.  Results land in gpurun_out/<tag>/; copy the summaries you
#   want judged into profiles/.
#   usage: tools/profile_round.sh <tag>
tag=${1:-round}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-batch-scaling > $O/bench_c2_profiled.json 2> $O/prof_c2.err || { echo "c2 stats failed"; exit 1; }
cp $O/prof_c2/*/*kernel_stats.csv $O/kernel_stats_c2.csv
for w in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_${w}_profiled.json 2> $O/prof_$w.err || { echo "$w stats failed"; exit 1; }
  cp $O/prof_$w/*/*kernel_stats.csv $O/kernel_stats_$w.csv
  echo "$w stats done"
done
cd $R
bash tools/prof.sh sq $tag/sq_c2 --no-other-configs > $O/sq_c2.txt 2>&1 || { echo "sq c2 failed"; exit 1; }
bash tools/prof.sh sq $tag/sq_c3 --workload c3 --steps 3 --warmup 1 > $O/sq_c3.txt 2>&1 || { echo "sq c3 failed"; exit 1; }
bash tools/prof.sh sq $tag/sq_b1024d --workload batch --batch-n 1024 --batch-distinct 1 --steps 3 --warmup 1 > $O/sq_b1024d.txt 2>&1 || { echo "sq batch distinct failed"; exit 1; }
bash tools/prof.sh sq $tag/sq_b1024o --workload batch --batch-n 1024 --batch-distinct 0 --steps 3 --warmup 1 > $O/sq_b1024o.txt 2>&1 || { echo "sq batch one failed"; exit 1; }
cd /tmp
for b in "1024 1 b1024d" "1024 0 b1024o"; do set -- $b
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$3 -- python3 $R/bench.py --workload batch --batch-n $1 --batch-distinct $2 --steps 10 --warmup 2 > $O/bench_$3_profiled.json 2> $O/prof_$3.err || { echo "$3 stats failed"; exit 1; }
  cp $O/prof_$3/*/*kernel_stats.csv $O/kernel_stats_$3.csv
done
cd $R
echo "sq done"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_stage_e.py > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_stage_e.py > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
cd $R
python3 tools/pmc_parse.py $O/pmc_fetch $O/pmc_write $O/stage_e_pmc.json > $O/pmc_parse.log 2>&1
python3 tools/precision_sweep.py > $O/c5_precision_sweep.json 2> $O/precision_sweep.err || { echo "precision sweep failed"; exit 1; }
cp $O/pmc_fetch/*/*counter_collection.csv $O/stage_e_fetch_counter_collection.csv 2>/dev/null
cp $O/pmc_write/*/*counter_collection.csv $O/stage_e_write_counter_collection.csv 2>/dev/null
python3 - <<PY
import csv, json
d = json.load(open('$O/bench_default.json'))
r = d['roofline']
print('C2 value %.4g samples/s  %.4f ms/step | stage K %.3f ms, stage E %.4f ms' % (d['value'], d['ms_per_step'], d['stage_split_rank0']['stage_k_ms'], d['stage_split_rank0']['stage_e_ms']))
print('roofline k_power_stream %.0f GB/s (frac %.3f, %.1f us); whole call frac %.3f; registered grid frac %.3f' % (
    r['achieved'], r['frac'], r['avg_launch_us'], r['whole_call']['frac'], r['whole_call_registered_grid']['frac']))
if 'roofline_stage_k' in d:
    print('stage K %.2f TFLOP/s fp64 (frac %.3f)' % (d['roofline_stage_k']['achieved'], d['roofline_stage_k']['frac']))
print('cpu %.3g (1 core) / %.3g (%d cores)' % (d['cpu_baseline']['value'], d['cpu_baseline']['pool']['value'], d['cpu_baseline']['pool']['cores']))
for k, v in d.get('other_configs', {}).items():
    print('%s %.4f ms/step  %.4g samples/s' % (k, v['ms_per_step'], v['value']))
for b in d.get('batch_scaling', []):
    print('batch n=%4d distinct=%d  %.4f ms/step  %.3f us/epoch  %.4g samples/s  stage K frac %s' % (b['n_epoch'], b['distinct_cosmologies'], b['ms_per_step'], b['us_per_epoch'], b['samples_per_s'], b.get('stage_k_frac')))
for w in ('c2', 'c3', 'c4', 'c5', 'b1024d', 'b1024o'):
    print('--', w)
    for row in csv.DictReader(open('$O/kernel_stats_%s.csv' % w)):
        if 'chomp' in row['Name']:
            print('%-30s calls %5s avg %9.1f us min %9.1f max %9.1f  %5.1f%%' % (row['Name'].split('(')[0].replace('chomp::', '').replace('void ', ''), row['Calls'], float(row['AverageNs']) / 1e3, float(row['MinNs']) / 1e3, float(row['MaxNs']) / 1e3, float(row['Percentage'])))
PY
