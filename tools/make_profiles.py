"""Digest gpurun_out/prof (written by tools/profile_run.sh on the GPU box) into profiles/rNN_*.

    python tools/make_profiles.py 2        # round number

Per profiled command: the rocprofv3 --kernel-trace --stats table, and the per-dispatch PMC values of its dominant kernel
(FETCH_SIZE / WRITE_SIZE in separate passes, the SQ counters in a third), corrected as MI355X_MICROARCH.md prescribes
(FETCH_SIZE x 2 for wide streaming reads on gfx950; KiB units).  rNN_summary.json is what bench.py reads back for
`roofline.traffic`: it carries the sha of the kernel sources it was collected on."""
import csv, glob, hashlib, json, os, shutil, sys

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
src, dst = "gpurun_out/prof", "profiles"
tag = f"r{rnd:02d}"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# device code of the whole-network forward kernel and its launcher (host-side engine code, other kernels: not part of it)
HEADLINE_KERNEL_FILES = ("kh_internal.h", "tower_common.h", "tower_mfma.hip", "tower8_mfma.hip")


def kernel_source_sha():
    h = hashlib.sha256()
    for name in HEADLINE_KERNEL_FILES:
        h.update(open(os.path.join(ROOT, "kami_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def stats_rows(d):
    f = newest(f"{d}/**/*kernel_stats.csv")
    return list(csv.DictReader(open(f))) if f else []


def counters(d, kernel_substr):
    out = {}
    f = newest(f"{d}/**/*counter_collection.csv")
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if kernel_substr not in r.get("Kernel_Name", ""):
            continue
        out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in out.items()}


def traffic(ctr, algorithmic):
    if "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
        return None
    raw, wr = ctr["FETCH_SIZE"]["mean"] * 1024, ctr["WRITE_SIZE"]["mean"] * 1024
    return {"fetch_bytes_raw": raw, "fetch_bytes_corrected": 2 * raw, "write_bytes": wr, "total_bytes_per_launch": 2 * raw + wr,
            "algorithmic_bytes_per_launch": algorithmic, "ratio": (2 * raw + wr) / algorithmic,
            "note": "FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads "
                    "(MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE is exact.  Separate --pmc passes, no trace domains combined."}


def derived(ctr, waves_per_launch):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in ctr:
        return None
    d = {"mfma_busy_cycles_per_simd": ctr["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (256 * 4)}
    if ctr.get("SQ_BUSY_CYCLES"):
        d["mfma_busy_over_busy_cycles"] = d["mfma_busy_cycles_per_simd"] / (ctr["SQ_BUSY_CYCLES"]["mean"] / 32)
    if ctr.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction"] = ctr["SQ_LDS_BANK_CONFLICT"]["mean"] / ctr["SQ_LDS_IDX_ACTIVE"]["mean"]
        d["lds_active_cycles_per_cu"] = ctr["SQ_LDS_IDX_ACTIVE"]["mean"] / 256
    return d


def write_stats(rows, name):
    if rows:
        with open(f"{dst}/{name}", "w") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)


sha = kernel_source_sha()
# ---- the headline command: python bench.py (configs[1], bf16, tower_kernel)
bench = json.loads(open(f"{src}/bench_default.json").read().strip().splitlines()[-1])
shutil.copy(f"{src}/bench_default.json", f"{dst}/{tag}_bench_default.json")
cfg = bench["config"]
F, C, R, Bb = cfg["features"], cfg["filters"], cfg["residuals"], cfg["batch_per_gpu"]
rows = stats_rows("kt")
write_stats(rows, f"{tag}_kernel_stats.csv")
ctr = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    ctr.update(counters(d, "tower8_kernelIDF16bLi8ELb0E") or counters(d, "tower8_kernel") or counters(d, "tower_kernel"))
summary = {
    "round": rnd, "kernel_source_sha16": sha,
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-variants --no-legs   (default --steps 2000 --warmup 200 --prewarm 0.3)",
    "pmc_commands": ["rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --repeats 1 --no-cpu-baseline --no-variants --no-legs",
                     "rocprofv3 --pmc WRITE_SIZE ... (same)", "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS ... (same)"],
    "workload": cfg["workload"],
    "workload_key": {"features": F, "filters": C, "residuals": R, "batch": Bb, "dtype": bench["dtype"]},
    "bench_line": {k: bench[k] for k in ("value", "ms_per_step", "roofline", "distribution") if k in bench},
    "counters": ctr,
}
tower = [r for r in rows if "tower8_kernelIDF16bLi8ELb0E" in r["Name"]] or [r for r in rows if "tower8_kernel" in r["Name"] or "tower_kernel" in r["Name"]]
if tower:
    t = tower[0]
    summary["tower_kernel"] = {"name": t["Name"], "calls": int(t["Calls"]), "avg_ns": float(t["AverageNs"]), "min_ns": float(t["MinNs"]), "max_ns": float(t["MaxNs"])}
    trace = newest("kt/**/*kernel_trace.csv")
    if trace:
        d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace)) if "tower8_kernel" in r["Kernel_Name"] or "tower_kernel" in r["Kernel_Name"])
        last = [x[1] for x in d[-2000:]]
        summary["tower_kernel"]["avg_ns_last_2000_dispatches"] = sum(last) / len(last)
tr = traffic(ctr, Bb * (64 * F * 4 + 4672 * 4 + 256 * 4))
if tr:
    summary["hbm_traffic"] = tr
dv = derived(ctr, 1024)
if dv:
    summary["derived"] = dv
json.dump(summary, open(f"{dst}/{tag}_summary.json", "w"), indent=1)

# ---- wide nets
wide = {}
for Cc, Rr, B, dt in ((128, 10, 1024, "bf16"), (256, 20, 256, "f16"), (256, 20, 2048, "f16")):
    wtag = f"w{Cc}_b{B}"
    rows = stats_rows(f"kt_{wtag}")
    if not rows:
        continue
    write_stats(rows, f"{tag}_wide_{Rr}x{Cc}_b{B}_{dt}_kernel_stats.csv")
    dominant = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    key = next((k for k in ("tower128_kernel", "tower2b_kernel", "tower2s_kernel", "conv4_mfma_kernel") if k in dominant["Name"]), "conv_mfma_kernel")
    ctr = {}
    for d in (f"pmcf_{wtag}", f"pmcw_{wtag}", f"pmcs_{wtag}"):
        ctr.update(counters(d, key))
    calls = {r["Name"]: int(r["Calls"]) for r in rows}
    once = [int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("policy_head4", "softmax", "tower128", "tower2b", "tower2s"))]
    fwd = min(once) if once else 1                         # kernels that run once per forward
    per_forward_ns = sum(float(r["TotalDurationNs"]) for r in rows if "fillBuffer" not in r["Name"] and "copyBuffer" not in r["Name"]) / fwd
    flops = (1152 * 119 * Cc + 2304 * Rr * Cc * Cc + 16512 * Cc + 1228800) * B
    entry = {"workload": f"{B} x (119x8x8) planes, {Rr}-block x {Cc}-filter net, {dt} (tools/wide_profile.py {Cc} {Rr} {B} {dt})",
             "forwards_profiled": fwd, "kernel_ns_per_forward": per_forward_ns,
             "tflops_from_kernel_durations": flops / per_forward_ns / 1e3, "frac_of_2500": flops / per_forward_ns / 1e3 / 2500.0,
             "dominant_kernel": {"name": dominant["Name"], "calls": int(dominant["Calls"]), "avg_ns": float(dominant["AverageNs"]),
                                 "share_of_kernel_time": float(dominant["TotalDurationNs"]) / sum(float(r["TotalDurationNs"]) for r in rows)},
             "counters_of_dominant_kernel_per_dispatch": ctr}
    # algorithmic HBM bytes of the dominant kernel's dispatch
    if key in ("tower128_kernel", "tower2b_kernel", "tower2s_kernel"):
        # fp32 planes in, residual stream out (T), every layer's packed weights once
        alg = B * 64 * 119 * 4 + B * 64 * Cc * 2 + (9 * 128 * Cc + Rr * 2 * 9 * Cc * Cc) * 2
        if key == "tower2s_kernel":
            # its own design adds: the planes read by both workgroups of a pair, and per layer but the last every board's
            # new image written once and read once through the exchange area
            entry["exchange_bytes_per_launch"] = 2 * Rr * 2 * B * 64 * Cc * 2
            entry["note"] = ("two workgroups per board pair exchange half an image per layer through global memory (sc1 stores, sc1 loads): "
                             "that traffic is the kernel's design, on top of the algorithmic bytes")
    elif key == "conv4_mfma_kernel":
        alg = B * 64 * Cc * 2 * 3                               # in, skip, out
    else:
        alg = B * 64 * Cc * 2 * 3
    tr = traffic(ctr, alg)
    if tr:
        entry["hbm_traffic_of_dominant_kernel"] = tr
    dv = derived(ctr, 1024)
    if dv:
        entry["derived"] = dv
    wide[f"{Rr}x{Cc}_b{B}_{dt}"] = entry
if wide:
    json.dump({"round": rnd, "kernel_source_sha16": sha, "configs": wide}, open(f"{dst}/{tag}_wide_summary.json", "w"), indent=1)

for name in ("wide_variants", "host_path_bench", "pinned_probe", "selfplay_bench", "train_bench", "train_bench_valu", "encode_bench"):
    p = f"{src}/{name}.txt"
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, f"{dst}/{tag}_{name}.txt")
print(json.dumps({k: summary[k] for k in ("tower_kernel", "hbm_traffic", "derived") if k in summary}, indent=1))
for k, v in wide.items():
    print(k, {kk: v[kk] for kk in ("kernel_ns_per_forward", "frac_of_2500")}, v["dominant_kernel"]["name"][:60], v.get("hbm_traffic_of_dominant_kernel", {}).get("ratio"), v.get("derived"))
