"""Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ under this
round's names, and derive profiles/roundN_stage_k_counters.json (what bench.py's
roofline_stage_k reads) from the SQ counter passes.
    usage: python tools/collect_profiles.py <tag> [round]"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "round4"
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
STAGE_K = ("k_sigma_nodes", "k_sigma_lns", "k_epoch_probe", "k_nu_table", "k_mass_nodes", "k_halo_nodes",
           "k_halo_knots", "k_halo_knots_samples", "k_halo_knots_fast", "k_halo_knots_literal")
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, rnd + "_bench_default.json"))
for w in ("c2", "c3", "c4", "c5", "b1024d", "b1024o"):
    shutil.copy(os.path.join(src, "kernel_stats_%s.csv" % w), os.path.join(dst, "%s_kernel_stats_%s.csv" % (rnd, w)))
os.makedirs(os.path.join(dst, rnd + "_pmc"), exist_ok=True)
out = {}
for w, args, key in (("c2", "--no-cpu-baseline --no-roofline --no-other-configs", "c2"),
                     ("c3", "--workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline", "c3"),
                     ("b1024d", "--workload batch --batch-n 1024 --batch-distinct 1 --steps 3 --warmup 1", "batch_1024_distinct"),
                     ("b1024o", "--workload batch --batch-n 1024 --batch-distinct 0 --steps 3 --warmup 1", "batch_1024_one")):
    raw = json.load(open(os.path.join(src, "sq_" + w, "sq_counters.json")))
    shutil.copy(os.path.join(src, "sq_" + w, "sq_counters.json"), os.path.join(dst, rnd + "_pmc", "sq_counters_%s.json" % w))
    per = {}
    for name, c in sorted(raw.items()):
        if not name.split("<")[0] in STAGE_K:
            continue
        per[name] = {"fp64_flop": c.get("fp64_flop_per_launch", 0.0), "INSTS_VALU": c.get("INSTS_VALU", 0.0),
                     "valu_active_frac_of_wave_cycles": c.get("valu_active_frac_of_wave_cycles", 0.0),
                     "WAVES": c.get("WAVES", 0.0), "WAIT_INST_LDS": c.get("WAIT_INST_LDS", 0.0),
                     "LDS_BANK_CONFLICT": c.get("LDS_BANK_CONFLICT", 0.0)}
    out[key] = {"fp64_flop_per_step": sum(v["fp64_flop"] for v in per.values()),
              "valu_insts_per_step": sum(v["INSTS_VALU"] for v in per.values()),
              "kernels": sorted(per), "per_kernel": per,
              "source": "rocprofv3 --pmc (three passes: SQ issue counters, LDS/SMEM counters, "
                        "SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64) of `python bench.py %s`, per launch = per step; "
                        "tools/prof.sh sq; raw: profiles/%s_pmc/sq_counters_%s.json" % (args, rnd, w)}
json.dump(out, open(os.path.join(dst, rnd + "_stage_k_counters.json"), "w"), indent=1)
for f in ("stage_e_fetch_counter_collection.csv", "stage_e_write_counter_collection.csv"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, rnd + "_pmc", f))
if os.path.exists(os.path.join(src, "c5_precision_sweep.json")):
    shutil.copy(os.path.join(src, "c5_precision_sweep.json"), os.path.join(dst, rnd + "_c5_precision_sweep.json"))
if os.path.exists(os.path.join(src, "stage_e_pmc.json")):
    shutil.copy(os.path.join(src, "stage_e_pmc.json"), os.path.join(dst, "stage_e_pmc.json"))
log = os.path.join(R, "gpurun_out", tag + ".log")
if os.path.exists(log):
    lines = open(log).read().splitlines()
    start = max(i for i, l in enumerate(lines) if l.startswith("bench done")) if any(l.startswith("bench done") for l in lines) else 0
    open(os.path.join(dst, rnd + "_summary.txt"), "w").write("\n".join(lines[start:]) + "\n")
for w in out:
    print(w, "fp64 flop/step %.4g  VALU insts/step %.4g" % (out[w]["fp64_flop_per_step"], out[w]["valu_insts_per_step"]))
