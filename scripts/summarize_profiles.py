#!/usr/bin/env python
"""Condense gpurun_out/{prof,pmc1..4} (rocprofv3 CSV output of scripts/gpu_profile.sh) into the
tracked files under profiles/:  r<NN>_kernel_stats.csv (top kernels of the kernel trace) and
r<NN>_pmc_summary.json (per-dispatch averages of the PMC passes + derived HBM traffic)."""
import collections
import csv
import json
import os
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
os.makedirs(P, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(G, "prof", "bench_kernel_stats.csv"))))
with open(os.path.join(P, f"{ROUND}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline (1x MI355X)"])
    w.writerow(list(rows[0].keys()))
    for r in rows[:12]:
        r = dict(r)
        r["Name"] = r["Name"][:110]
        w.writerow(list(r.values()))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    path = os.path.join(G, d, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if "renderer" not in r["Kernel_Name"] and "splat" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pmc[k]["_VGPR"].append(float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]))
        pmc[k]["_scratch"].append(float(r["Scratch_Size"]))
        pmc[k]["_lds"].append(float(r["LDS_Block_Size"]))
out = {}
for k, v in pmc.items():
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    e = {c: round(a, 1) for c, a in avg.items()}
    # HBM traffic per launch as the microarch guide prescribes: (FETCH_SIZE [x2 on gfx950 for wide
    # coalesced reads -- NOT applied here: the access pattern is 16-byte gathers / 4-byte atomics,
    # uncalibrated] + WRITE_SIZE) * 1024 bytes
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        e["hbm_bytes_per_launch"] = round((avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg:
        e["note_mfma"] = "MFMA pipe busy cycles summed over 1024 SIMDs; divide by 1024*kernel_cycles for utilisation"
    out[k] = e
json.dump(out, open(os.path.join(P, f"{ROUND}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v[c] for c in v if c in ("hbm_bytes_per_launch", "SQ_INSTS_MFMA", "SQ_INSTS_VALU")} for k, v in out.items()}, indent=1))
