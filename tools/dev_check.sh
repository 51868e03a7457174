#!/bin/bash
# Development aid (GPU box, through gpurun): the new tests of the round, the default bench line
# with a digest of its new fields, and the two-rank rehearsal on one GPU.
#   usage: tools/dev_check.sh <tag> [pytest -k expression]
tag=${1:-dev}; kexp=${2:-"large_batch or soak_seed or precision_sweep"}
O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "$kexp" > $O/gputest.log 2>&1; rc=$?
tail -4 $O/gputest.log
test $rc -eq 0 || exit $rc
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["ms_per_step"], d["value"], {k: round(v, 4) for k, v in d.items() if k.startswith("ms_per_step_") or k == "slowest_step_index"})
print("dropin", {k: v for k, v in d["dropin"].items() if k != "workload"})
for k, v in d["other_configs"].items():
    print(k, round(v["ms_per_step"], 4), [round(v.get(x, 0), 4) for x in ("ms_per_step_median", "ms_per_step_p95", "ms_per_step_max")], v.get("slowest_step_index"))
for b in d["batch_scaling"]:
    print(b["n_epoch"], b["distinct_cosmologies"], round(b["ms_per_step"], 4), b.get("stage_k_frac"))
print("roofline", d["roofline"]["frac"], "stage K", d["roofline_stage_k"]["frac"], "cpu", d["cpu_baseline"]["value"])
PY
timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 3 --warmup 1 --no-other-configs --no-batch-scaling > $O/rehearse.json 2> $O/rehearse.err || { tail -5 $O/rehearse.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/rehearse.json"))
print("rehearsal:", d["scaling"], d["job"], "cpu_baseline" in d, d.get("weak_scaling", {}).get("rows_per_rank"))
PY
