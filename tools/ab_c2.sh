# per-kernel averages of the C2 step for the product library and for other builds of it:
#   bash tools/ab_c2.sh [build_exp/x.so ...]      (on the GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-roofline --steps 100 --warmup 10 $AB_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 $R/bench.py $ARGS > $O/p0.json 2>/dev/null
i=0
for so in "$@"; do i=$((i+1)); rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$i -- python3 $R/tools/ab_lib.py $R/$so $ARGS > $O/p$i.json 2>/dev/null; done
cd $R
python3 - "$@" <<PY
import csv,glob,sys,json
names=['product']+sys.argv[1:]
for i,tag in enumerate(names):
    f=glob.glob('$O/p%d/*/*kernel_stats.csv'%i)[0]
    ms=json.load(open('$O/p%d.json'%i))['ms_per_step']
    print('%-28s %.4f ms |'%(tag,ms), ' '.join('%s=%.1f'%(r['Name'].split('(')[0].replace('chomp::','').replace('void ','').replace('k_','')[:14], float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f)) if 'chomp' in r['Name']))
PY
