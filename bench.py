#!/usr/bin/env python3
"""bench.py — NN leaf-evals/s of the kami leaf-evaluation hot path on MI355X.

A "step" is one pass of the hot path (stem conv + residual tower + policy/value heads,
kami/nn/nn.cpp:59-91) over one synthetic batch of 512 x (119 x 8 x 8) planes already resident
in HBM.  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks), every rank evaluates its own disjoint batch: weak scaling, no data-path
collective (leaf evaluations are independent).

    python bench.py [--gpus N --steps K --warmup W] [--dtype bf16|f16|f32] [--batch 512]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md, dense


def cpu_baseline(F, Cc, R, batch):
    """Reference CPU evaluate() on this box's host cores, on a bounded sample (rank 0, N=1)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kami_ref")
    # the GPU box gives one GPU's job a 16-core share of the host (gpurun), whatever nproc says
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("KAMI_CPU_CORES", "16"))))
    iters = 40      # ~1 s of wall time on 16 threads = ~16 CPU-seconds, the bounded sample
    if os.path.exists(ref):
        try:
            out = subprocess.run([ref, "bench", str(F), str(Cc), str(R), str(batch), str(iters), str(cores)],
                                 capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            r = json.loads(line)
            return {"value": round(r["evals_per_s"], 1), "unit": "leaf-evals/s", "cores": r["threads"],
                    "kind": "reference",
                    "sample": f"{iters} calls of the unmodified reference NN::infer (libtorch CPU) on one "
                              f"uniform[0,1) batch of {batch}x(8x8x{F}), {R}x{Cc} net, after 1 warm-up call"}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"[bench] reference baseline unavailable ({e}); timing the oracle port\n")
    import numpy as np
    from oracle import pyoracle as ko
    from kami_amd import weights as W
    blob = W.random_weights(F, Cc, R, seed=0)
    n = min(batch, 256)
    x = np.random.default_rng(0).random((n, 8, 8, F), dtype=np.float32)
    ko.forward(blob, F, Cc, R, x[:8], want_logits=False)
    t0 = time.perf_counter()
    ko.forward(blob, F, Cc, R, x, want_logits=False)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "leaf-evals/s", "cores": ko.lib().ko_max_threads(),
            "kind": "port", "sample": f"one oracle forward over {n} boards of 8x8x{F}, {R}x{Cc} net (OpenMP)"}


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "NN leaf-evals/sec at batch 512 (119\u00d78\u00d78 planes), 1/2/4/8 MI355X"


def measured_traffic(F, Cc, R, B, dtype):
    """HBM bytes per launch of the forward kernel from the PMC passes kept under profiles/
    (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs, gfx950 corrections applied as
    MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside this process, so this
    is the committed measurement of the SAME command; None when the workload differs."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload_key", {})
        if w == {"features": F, "filters": Cc, "residuals": R, "batch": B, "dtype": dtype}:
            best = d["hbm_traffic"]["total_bytes_per_launch"]
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm", type=float, default=0.3, help="seconds of untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--dtype", default=os.environ.get("KAMI_BENCH_DTYPE", "bf16"))
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--features", type=int, default=119)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--residuals", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from kami_amd import NN, weights as W, _lib as L, dist as kd

    rank, local_rank, world = kd.env_rank()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU path)")
    # one rank per GPU (the driver's multi-GPU run); KAMI_DIST_BACKEND=gloo lets two ranks share
    # one GPU to rehearse the N > 1 control flow on a single-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("KAMI_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    local_rank = dev_index
    torch.cuda.set_device(dev_index)
    dist = kd.init(backend)         # RCCL; only the barrier and the max-over-ranks use it

    F, Cc, R, B = a.features, a.filters, a.residuals, a.batch
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=a.dtype, device=local_rank)
    nn.load_weights(W.random_weights(F, Cc, R, seed=20240607), 1)
    lib = L.load()

    g = torch.Generator(device="cuda")
    g.manual_seed(20240607 + rank)
    x = torch.rand((B, 8, 8, F), generator=g, device="cuda", dtype=torch.float32)   # test/nn.cpp:23-24 convention
    policy = torch.empty((B, 4672), device="cuda", dtype=torch.float32)
    vfull = torch.empty((B, 256), device="cuda", dtype=torch.float32)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def step():
        rc = lib.kh_infer_device(nn.handle, C.c_void_p(x.data_ptr()), B, C.c_void_p(policy.data_ptr()),
                                 C.c_void_p(vfull.data_ptr()), sp)
        if rc:
            raise RuntimeError(L.last_error())

    def barrier():
        kd.barrier(dist)
        torch.cuda.synchronize()

    # Clock pre-warm (untimed, before the W warm-up steps): an idle MI355X takes tens of milliseconds
    # of sustained load to reach the clocks it then holds — the same kernel measured 41 us in the
    # first 10 ms of a burst and 34.8 us from ~70 ms on (profiles/r01_clock_ramp.txt).  The metric
    # is sustained throughput, so every run, whatever K and W, starts from the settled state.
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < a.prewarm:
        for _ in range(200):
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = kd.max_over_ranks(dist, dt)
    assert bool(torch.isfinite(policy).all()) and abs(float(policy[0].sum()) - 1.0) < 1e-2

    # roofline of the dominant kernel (the forward pass), HIP events on the engine's own stream
    ms = C.c_float(0)
    iters = max(200, min(a.steps, 2000))
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < a.prewarm:      # the checks above let the clocks drop again
        lib.kh_time_infer_device(nn.handle, C.c_void_p(x.data_ptr()), B, C.c_void_p(policy.data_ptr()),
                                 C.c_void_p(vfull.data_ptr()), 500, C.byref(ms))
    rc = lib.kh_time_infer_device(nn.handle, C.c_void_p(x.data_ptr()), B, C.c_void_p(policy.data_ptr()),
                                  C.c_void_p(vfull.data_ptr()), iters, C.byref(ms))
    if rc:
        raise RuntimeError(L.last_error())
    flops = W.flops_per_eval(F, Cc, R) * B
    achieved = flops / (ms.value * 1e-3) / 1e12
    peak = PEAK_TFLOPS[a.dtype]

    if rank == 0:
        out = {
            "metric": baseline_metric(),
            "value": round(world * B * a.steps / dt, 1),
            "unit": "leaf-evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{B} random boards x ({F}x8x8) planes per GPU, {R}-block x {Cc}-filter net, "
                                   f"batched leaf evaluate() forward (BASELINE configs[1])",
                       "batch_per_gpu": B, "features": F, "filters": Cc, "residuals": R,
                       "parallelism": f"replicas x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 5),
                         "traffic": measured_traffic(F, Cc, R, B, a.dtype),
                         "kernel_ms": round(ms.value, 5),
                         "flops_per_launch": flops},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F, Cc, R, B)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
