#!/bin/bash
# two PMC passes of the Stage-E workload -> profiles/stage_e_pmc.json (run on the GPU box)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/fetch -- python3 $R/tools/pmc_stage_e.py > $R/gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/write -- python3 $R/tools/pmc_stage_e.py > $R/gpurun_out/pmc/write.log 2>&1
cd $R
python3 tools/pmc_parse.py gpurun_out/pmc/fetch gpurun_out/pmc/write gpurun_out/pmc/stage_e_pmc.json
