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
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None  # workloads profiled in THIS run (gpurun_out keeps older rounds' directories)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
os.makedirs(P, exist_ok=True)
CMD = {"cfg2": "python bench.py --workload cfg2 --no-cpu-baseline --no-extras --steps 200 --warmup 3",
       "cfg3": "python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 50 --warmup 3",
       "cfg4": "python bench.py --workload cfg4 --no-cpu-baseline --no-extras --steps 5 --warmup 3",
       "small": "python bench.py --workload small --no-cpu-baseline --no-extras --steps 200 --warmup 3",
       "1080p_s128": "python bench.py --workload 1080p_s128 --no-cpu-baseline --no-extras --steps 5 --warmup 3",
       "cfg5": "python bench.py --workload cfg5 --no-cpu-baseline --no-extras --steps 3 --warmup 3",
       "refbench256": "python bench.py --workload refbench256 --no-cpu-baseline --no-extras --steps 20 --warmup 3",
       "cfg5_render": "python bench.py --workload cfg5_render --no-cpu-baseline --no-extras --steps 5 --warmup 3",
       "loop": "LP_LOOP=1 python bench.py --workload cfg2 --no-cpu-baseline --no-extras --steps 200 --warmup 3",
       "h64_222": "python bench.py --workload h64_222 --no-cpu-baseline --no-extras --steps 20 --warmup 3",
       "h64_example_112": "python bench.py --workload h64_example_112 --no-cpu-baseline --no-extras --steps 20 --warmup 3"}
out = {}
summary_path = os.path.join(P, f"{ROUND}_pmc_summary.json")
if ONLY and os.path.exists(summary_path):  # a partial re-profile keeps the other workloads' entries of this round
    out = {k: v for k, v in json.load(open(summary_path)).items() if v.get("workload") not in ONLY}
for w in CMD:
    if ONLY and w not in ONLY:
        continue
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
    trace_avg = {}  # kernel -> average duration (ns) in the kernel-trace run (no counters)
    if stats:
        for r in rows:
            trace_avg[r["Name"].split("(")[0].replace("void ", "").strip()] = float(r["AverageNs"])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    pmc_dur = collections.defaultdict(list)  # kernel -> durations (ns) of its dispatches in the pmc1 pass (the SQ_BUSY_CYCLES pass)
    for path in glob.glob(os.path.join(G, f"pmc1_{w}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            kn = r.get("Kernel_Name", "")
            if any(t in kn for t in ("renderer", "splat", "ray_embedding")) and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                pmc_dur[kn.split("(")[0].replace("void ", "").strip()].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for path in glob.glob(os.path.join(G, f"{d}_{w}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                kn = r["Kernel_Name"]
                if not any(t in kn for t in ("renderer", "splat", "ray_embedding")):
                    continue
                k = kn.split("(")[0].replace("void ", "").strip()
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if d == "pmc1" and r["Counter_Name"] == "SQ_BUSY_CYCLES" and r.get("Start_Timestamp") and r.get("End_Timestamp") \
                        and not glob.glob(os.path.join(G, f"pmc1_{w}", "**", "*kernel_trace.csv"), recursive=True):
                    pmc_dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
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
        if pmc_dur.get(k):  # duration of the kernel in the pass that counted SQ_BUSY_CYCLES: clock = busy / 32 SEs / duration
            e["pmc_duration_ns"] = round(sum(pmc_dur[k]) / len(pmc_dur[k]), 1)
        if k in trace_avg:
            e["trace_duration_ns"] = round(trace_avg[k], 1)
        if e.get("SQ_BUSY_CYCLES") and (e.get("pmc_duration_ns") or e.get("trace_duration_ns")):
            e["clock_ghz"] = round(e["SQ_BUSY_CYCLES"] / 32.0 / (e.get("pmc_duration_ns") or e["trace_duration_ns"]), 4)
        e["workload"] = w
        out[f"{w}: {k}"] = e  # ("loop" = the cfg-2 command with LP_LOOP=1: its entries never match bench.py's lookup for cfg2)
json.dump(out, open(summary_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v[c] for c in v if c in ("hbm_bytes_per_launch", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "_scratch", "SQ_WAVES")}
                  for k, v in out.items()}, indent=1))
