#!/bin/bash
# Development aid (GPU box): kernel timeline (all streams) of the last step of a bench workload.
#   usage: tools/dev_trace.sh <tag> <workload> [first-kernel-prefix]
tag=$1; w=$2; first=${3:-k_proj_chi}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace_$w.json 2> $O/trace_$w.err || { tail -3 $O/trace_$w.err; exit 1; }
cd $R
python3 tools/timeline.py $(ls $O/trace_$w/*/*kernel_trace.csv | head -1) $first
