#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_stage_e.py
into profiles/stage_e_pmc.json.  Corrections per MI355X_MICROARCH.md section HBM:
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
streaming read (doubled here); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import sys


def per_launch(d, counter):
    """Sum the counter over the Stage-E dispatches (k_power_prep, k_power_stream,
    k_power_grid_lanes) and divide by the number of chomp_power calls."""
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    total, calls, per_kernel = 0.0, set(), {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_power_" in name and r["Counter_Name"] == counter:
            v = float(r["Counter_Value"])
            total += v
            short = name.split("(")[0].split("::")[-1]
            per_kernel[short] = per_kernel.get(short, 0.0) + v
            if "k_power_grid_lanes" in name:
                calls.add(r["Dispatch_Id"])
    n = len(calls)
    return total / n, n, {k: v / n for k, v in per_kernel.items()}


fetch_kib, n1, fetch_k = per_launch(sys.argv[1], "FETCH_SIZE")
write_kib, n2, write_k = per_launch(sys.argv[2], "WRITE_SIZE")
nk, nz = 1 << 20, 64
out = {
    "kernel": "k_power_prep + k_power_stream + k_power_grid_lanes (one chomp_power call)",
    "nk": nk, "nz": nz, "calls_averaged": min(n1, n2),
    "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
    "FETCH_SIZE_KiB_raw_by_kernel": fetch_k, "WRITE_SIZE_KiB_raw_by_kernel": write_k,
    "fetch_bytes_corrected_x2": 2.0 * fetch_kib * 1024.0,
    "write_bytes": write_kib * 1024.0,
    "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
    "algorithmic_bytes_per_launch": 8.0 * nk + 8.0 * nk * nz,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
