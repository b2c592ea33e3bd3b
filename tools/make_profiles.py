"""Digest gpurun_out/prof (written by tools/profile_run.sh on the GPU box) into profiles/rNN_*.

    python tools/make_profiles.py 1        # round number
"""
import csv, glob, json, os, shutil, sys
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
src, dst = "gpurun_out/prof", "profiles"
tag = f"r{rnd:02d}"

def newest(pattern):
    fs = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None

# kernel-trace stats
stats = newest("kt/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(f"{dst}/{tag}_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)

# PMC: per-dispatch counter values of the tower kernel
def counters(pattern):
    out = {}
    f = newest(pattern)
    if not f: return out
    for r in csv.DictReader(open(f)):
        if "tower_kernel" not in r.get("Kernel_Name", ""): continue
        out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in out.items()}
ctr = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    ctr.update(counters(f"{d}/**/*counter_collection.csv"))

bench = json.loads(open(f"{src}/bench_default.json").read().strip().splitlines()[-1])
cfg = bench["config"]
tower = [r for r in rows if "tower_kernel" in r["Name"]]
F, C, R, Bb = cfg["features"], cfg["filters"], cfg["residuals"], cfg["batch_per_gpu"]
summary = {
    "round": rnd,
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (default --steps 2000 --warmup 200 --prewarm 0.3)",
    "pmc_commands": ["rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --no-cpu-baseline",
                     "rocprofv3 --pmc WRITE_SIZE ... (same)",
                     "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS ... (same)"],
    "workload": cfg["workload"],
    "workload_key": {"features": F, "filters": C, "residuals": R, "batch": Bb, "dtype": bench["dtype"]},
    "counters": ctr,
}
if tower:
    t = tower[0]
    summary["tower_kernel"] = {"name": t["Name"], "calls": int(t["Calls"]), "avg_ns": float(t["AverageNs"]),
                               "min_ns": float(t["MinNs"]), "max_ns": float(t["MaxNs"])}
trace = newest("kt/**/*kernel_trace.csv")
if trace and tower:
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace)) if "tower_kernel" in r["Kernel_Name"]]
    d.sort()
    last = [x[1] for x in d[-2000:]]
    first = [x[1] for x in d[:200]]
    summary["tower_kernel"]["avg_ns_last_2000_dispatches"] = sum(last) / len(last)
    summary["tower_kernel"]["avg_ns_first_200_dispatches"] = sum(first) / len(first)
if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
    raw = ctr["FETCH_SIZE"]["mean"] * 1024
    wr = ctr["WRITE_SIZE"]["mean"] * 1024
    summary["hbm_traffic"] = {
        "fetch_bytes_raw": raw, "fetch_bytes_corrected": 2 * raw, "write_bytes": wr, "total_bytes_per_launch": 2 * raw + wr,
        "note": "FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads "
                "(MI355X_MICROARCH.md HBM section) so it is doubled; WRITE_SIZE is exact. Separate --pmc passes, no trace domains combined.",
        "algorithmic_bytes_per_launch": Bb * (64 * F * 4 + 4672 * 4 + 256 * 4),
    }
if "SQ_WAVE_CYCLES" in ctr:
    waves = 256 * 4
    wc = ctr["SQ_WAVE_CYCLES"]["mean"] / waves * 4          # counter is per SE-sampled quarter on this tool version? keep the r01 convention
    summary["derived"] = {
        "mfma_busy_cycles_per_simd": ctr["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (256 * 4),
        "mfma_busy_over_busy_cycles": ctr["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (256 * 4) / (ctr["SQ_BUSY_CYCLES"]["mean"] / 32) if ctr.get("SQ_BUSY_CYCLES") else None,
        "lds_bank_conflict_fraction": ctr["SQ_LDS_BANK_CONFLICT"]["mean"] / ctr["SQ_LDS_IDX_ACTIVE"]["mean"],
        "lds_active_cycles_per_cu": ctr["SQ_LDS_IDX_ACTIVE"]["mean"] / 256,
    }
json.dump(summary, open(f"{dst}/{tag}_summary.json", "w"), indent=1)

for name in ("bench_default", "bench_f16", "bench_f30", "bench_b2048", "bench_f32", "bench_10x128_b1024", "bench_20x256_b256_f16"):
    p = f"{src}/{name}.json"
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, f"{dst}/{tag}_{name}.json")
for name in ("fused_bench", "encode_bench", "host_path_bench", "clock_ramp", "selfplay_bench", "train_bench", "wide_bench"):
    p = f"{src}/{name}.txt"
    if os.path.exists(p): shutil.copy(p, f"{dst}/{tag}_{name}.txt")
lines = open(f"{src}/pytest_gpu.log").read().strip().splitlines()
open(f"{dst}/{tag}_pytest_gpu.log", "w").write("\n".join(lines[-3:]) + "\n")
print(json.dumps({k: summary[k] for k in ("tower_kernel", "hbm_traffic", "derived") if k in summary}, indent=1))
print(open(f"{src}/bench_default.json").read().strip())
