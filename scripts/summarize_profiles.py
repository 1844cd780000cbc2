#!/usr/bin/env python
"""Condense gpurun_out/{prof_<w>,pmc1..4_<w>} (rocprofv3 CSV output of scripts/gpu_profile.sh, w in cfg2 / cfg3 / cfg4 / small) into
the tracked files under profiles/:  r<NN>_kernel_stats_<w>.csv (top kernels of the kernel trace), r<NN>_bench_line_<w>.json
(the bench line of the traced run) and r<NN>_pmc_summary.json (per-dispatch averages of the PMC passes + derived traffic)."""
import collections
import csv
import glob
import json
import os
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
os.makedirs(P, exist_ok=True)
CMD = {"cfg2": "python bench.py --workload cfg2 --no-cpu-baseline --no-extras --steps 200 --warmup 3",
       "cfg3": "python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 50 --warmup 3",
       "cfg4": "python bench.py --workload cfg4 --no-cpu-baseline --no-extras --steps 5 --warmup 3",
       "small": "python bench.py --workload small --no-cpu-baseline --no-extras --steps 200 --warmup 3"}
out = {}
for w in ("cfg2", "cfg3", "cfg4", "small"):
    stats = glob.glob(os.path.join(G, f"prof_{w}", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(P, f"{ROUND}_kernel_stats_{w}.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow([f"# rocprofv3 --kernel-trace --stats -- {CMD[w]} (1x MI355X)"])
            wr.writerow(list(rows[0].keys()))
            for r in rows[:10]:
                r = dict(r)
                r["Name"] = r["Name"][:120]
                wr.writerow(list(r.values()))
    log = os.path.join(G, f"prof_{w}.txt")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(P, f"{ROUND}_bench_line_{w}.json"), "w"), indent=1)
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for path in glob.glob(os.path.join(G, f"{d}_{w}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                kn = r["Kernel_Name"]
                if not any(t in kn for t in ("renderer", "splat", "ray_embedding")):
                    continue
                k = kn.split("(")[0].replace("void ", "")
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc[k]["_VGPR"].append(float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]))
                pmc[k]["_scratch"].append(float(r["Scratch_Size"]))
    for k, v in pmc.items():
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        e = {c: round(a, 1) for c, a in avg.items()}
        # HBM-side traffic per launch as the microarch guide prescribes: (FETCH_SIZE + WRITE_SIZE) * 1024 bytes, separate
        # --pmc passes.  The guide's x2 correction of FETCH_SIZE applies to wide coalesced streaming reads; these kernels
        # issue 16-byte gathers and 64-byte atomic segments (uncalibrated), so it is NOT applied.
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            e["hbm_bytes_per_launch"] = round((avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        e["workload"] = w
        out[f"{w}: {k}"] = e
json.dump(out, open(os.path.join(P, f"{ROUND}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v[c] for c in v if c in ("hbm_bytes_per_launch", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "_scratch", "SQ_WAVES")}
                  for k, v in out.items()}, indent=1))
