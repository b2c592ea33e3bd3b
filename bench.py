#!/usr/bin/env python3
"""bench.py — NN leaf-evals/s of the kami leaf-evaluation hot path on MI355X.

A "step" is one pass of the hot path (stem conv + residual tower + policy/value heads,
kami/nn/nn.cpp:59-91) over one synthetic batch of 512 x (119 x 8 x 8) planes already resident
in HBM (BASELINE configs[1]).  N > 1: one process per GPU (torch.distributed / RCCL only for the
barrier and the max-over-ranks), every rank evaluates its own disjoint batch: weak scaling, no
data-path collective (leaf evaluations are independent).

    python bench.py [--gpus N --steps K --warmup W] [--dtype bf16|f16|f32] [--batch 512]

The ONE JSON line carries, besides the contract's fields:
  roofline      the dominant kernel's achieved TFLOP/s (HIP events on the stream it runs on) vs the dense MFMA peak
  distribution  p10 / median / p90 of >= 50 separately timed repeats of the same step
  variants      (N = 1) the other legs SURVEY §8d / BASELINE.md §4 define, each with its own workload label and roofline:
                exact fp32, f16, F=30 with the encoder fused into the kernel, configs[2] (10x128, batch 1024, bf16),
                configs[4]'s net on one GPU (20x256, f16, batch 256)
  end_to_end    (N = 1) what a caller of the host-buffer ABI sees, PCIe included: kh_infer (the legacy float ABI of
                NN::infer) from 1 and 4 threads, kh_encode_infer_legal (compact records in, legal priors out)
  cpu_baseline  (N = 1) the unmodified reference's NN::infer on this box's host cores, bounded sample
  value_f32 / roofline_f32   (N = 1) the same step in the REFERENCE's arithmetic (exact fp32 on the matrix cores)
  encode        (N = 1) the board -> plane encoder alone (Env::observe, env.h:202-262) against its HBM-write roofline
  configs0      (N = 1) BASELINE configs[0]'s shape on the engine: one game, 64 sims per move, batch-1 evaluations
`value` is always the configs[1] kernel-path number; nothing in variants / end_to_end replaces it.  The timed output is
checked after the timed region: sampled rows must equal, bit for bit, kh_infer_full of the same input.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys
import threading
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
            "kind": "port", "sample": f"one oracle forward over {n} boards of 8x8x{F}, {R}x{Cc} net (OpenMP); "
                                      "oracle/_ref/kami_ref (the reference build) was not present"}


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "NN leaf-evals/sec at batch 512 (119×8×8 planes), 1/2/4/8 MI355X"


# device code of the whole-network forward kernel and its launcher (host-side engine code, other kernels: not part of it)
HEADLINE_KERNEL_FILES = ("kh_internal.h", "tower_common.h", "tower_mfma.hip", "tower8_mfma.hip")


def kernel_source_sha():
    """sha-256 (first 16 hex digits) over the kernel sources: a PMC summary under profiles/ only speaks for the
    kernels it was collected on."""
    h = hashlib.sha256()
    for name in HEADLINE_KERNEL_FILES:
        h.update(open(os.path.join(ROOT, "kami_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(F, Cc, R, B, dtype):
    """HBM bytes per launch of the forward kernel from the PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE,
    separate rocprofv3 --pmc runs, gfx950 corrections applied as MI355X_MICROARCH.md prescribes).  Counters cannot be
    read from inside this process, so this is the committed measurement of the SAME command: (bytes, source) where
    source names the file and says whether the kernels are still the ones it was collected on; (None, None) when no
    summary matches the workload, and bytes None when the kernel sources have changed since."""
    best = (None, None)
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload_key", {})
        if w == {"features": F, "filters": Cc, "residuals": R, "batch": B, "dtype": dtype} and "hbm_traffic" in d:
            same = d.get("kernel_source_sha16") == sha
            src = {"file": os.path.relpath(path, ROOT), "kernel_source_sha16": d.get("kernel_source_sha16"),
                   "current_kernel_source_sha16": sha, "same_kernels": same}
            best = (d["hbm_traffic"]["total_bytes_per_launch"] if same else None, src)
    return best


def workload_label(B, F, R, Cc):
    names = {(512, 119, 6, 64): "BASELINE configs[1]", (1024, 119, 10, 128): "BASELINE configs[2]",
             (2048, 119, 20, 256): "BASELINE configs[4] per GPU"}
    tag = names.get((B, F, R, Cc), "not a BASELINE configuration as such")
    return (f"{B} random boards x ({F}x8x8) planes per GPU, {R}-block x {Cc}-filter net, "
            f"batched leaf evaluate() forward ({tag})")


def metric_label(B, F):
    base = baseline_metric()
    if (B, F) == (512, 119):
        return base
    return f"NN leaf-evals/sec at batch {B} ({F}×8×8 planes), MI355X"


class DeviceLeg:
    """One engine + device-resident synthetic inputs; timed through kh_time_infer_device (HIP events on the
    engine's own stream, the stream the kernels run on)."""

    def __init__(self, torch, lib, NN, W, dtype, F, Cc, R, B, seed, device):
        self.lib, self.B, self.F, self.Cc, self.R, self.dtype = lib, B, F, Cc, R, dtype
        self.nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dtype, device=device)
        self.nn.load_weights(W.random_weights(F, Cc, R, seed=20240607), 1)
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        self.x = torch.rand((B, 8, 8, F), generator=g, device="cuda", dtype=torch.float32)   # test/nn.cpp:23-24 convention
        self.policy = torch.empty((B, 4672), device="cuda", dtype=torch.float32)
        self.vfull = torch.empty((B, 256), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()                 # inputs complete before any engine stream touches them
        self.flops = W.flops_per_eval(F, Cc, R) * B

    def time_ms(self, iters):
        ms = C.c_float(0)
        rc = self.lib.kh_time_infer_device(self.nn.handle, C.c_void_p(self.x.data_ptr()), self.B, C.c_void_p(self.policy.data_ptr()),
                                           C.c_void_p(self.vfull.data_ptr()), iters, C.byref(ms))
        if rc:
            from kami_amd import _lib as L
            raise RuntimeError(L.last_error())
        return ms.value

    def settled_ms(self, prewarm, seconds):
        """prewarm seconds of untimed load (clock ramp), then launches for about `seconds`, averaged."""
        t0 = time.perf_counter()
        ms = self.time_ms(20)
        while time.perf_counter() - t0 < prewarm:
            ms = self.time_ms(max(20, int(0.05 / max(ms, 1e-3) * 1e3)))
        iters = max(20, int(seconds / (ms * 1e-3)))
        return self.time_ms(iters), iters

    def roofline(self, ms):
        achieved = self.flops / (ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[self.dtype]
        return {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 5),
                "kernel_ms": round(ms, 5), "flops_per_launch": self.flops}


def variant_legs(torch, lib, NN, W, L, device, prewarm):
    import numpy as np
    out = []
    specs = [("f32", 119, 64, 6, 512, "exact fp32 on v_mfma_f32_32x32x2_f32 (the reference's own precision)"),
             ("f16", 119, 64, 6, 512, "f16 operands, fp32 accumulate"),
             ("bf16", 119, 128, 10, 1024, "configs[2]: tower128_kernel — the 3x3 stack and both heads in ONE launch (layers_mfma.hip)"),
             ("f16", 119, 256, 20, 256, "configs[4]'s net at its per-GPU batch (2 048 boards on 8 GPUs): tower2s_kernel (two workgroups per board pair, output channels split) + policy_head4_kernel"),
             ("f16", 119, 256, 20, 2048, "configs[4]'s net at its evaluation batch: tower2b_kernel + policy_head4_kernel")]
    for dtype, F, Cc, R, B, note in specs:
        leg = DeviceLeg(torch, lib, NN, W, dtype, F, Cc, R, B, 7, device)
        ms, iters = leg.settled_ms(prewarm, 0.4)
        out.append({"workload": workload_label(B, F, R, Cc), "note": note, "dtype": dtype, "value": round(B / (ms * 1e-3), 1),
                    "unit": "leaf-evals/s", "ms_per_step": round(ms, 5), "launches_timed": iters, "roofline": leg.roofline(ms)})
        del leg
        torch.cuda.empty_cache()
    # F = 30 (the reference encoder's planes) with compact ingest: 80-byte records in, Env::observe inside the kernel
    B, F, Cc, R = 512, 30, 64, 6
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype="bf16", device=device)
    nn.load_weights(W.random_weights(F, Cc, R, seed=20240607), 1)
    rng = np.random.default_rng(0)
    boards = np.zeros(B, dtype=L.BOARD_DTYPE)
    boards["piece_occ"] = rng.integers(0, 2**63, (B, 6), dtype=np.uint64)
    boards["color_occ"] = rng.integers(0, 2**63, (B, 2), dtype=np.uint64)
    boards["ply"] = rng.integers(0, 300, B); boards["ctm"] = rng.integers(0, 2, B); boards["castle_rights"] = rng.integers(0, 16, B)
    d_b = torch.from_numpy(boards.view(np.uint8).reshape(B, 80)).cuda()
    pol = torch.empty((B, 4672), device="cuda"); vf = torch.empty((B, 256), device="cuda")
    st = torch.cuda.Stream()
    sp = C.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()

    def run(n):
        for _ in range(n):
            if lib.kh_encode_infer_device(nn.handle, C.c_void_p(d_b.data_ptr()), B, C.c_void_p(pol.data_ptr()), C.c_void_p(vf.data_ptr()), sp):
                raise RuntimeError(L.last_error())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm:
        run(200); st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 4000
    e0.record(st); run(K); e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / K
    flops = W.flops_per_eval(F, Cc, R) * B
    out.append({"workload": f"{B} compact board records (80 B each) per GPU, Env::observe fused into the forward kernel (F = 30 planes), "
                            f"{R}-block x {Cc}-filter net (kh_encode_infer_device; SURVEY 8f row 1)",
                "dtype": "bf16", "value": round(B / (ms * 1e-3), 1), "unit": "leaf-evals/s", "ms_per_step": round(ms, 5), "launches_timed": K,
                "roofline": {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": 2500.0, "unit": "TFLOP/s",
                             "frac": round(flops / (ms * 1e-3) / 1e12 / 2500.0, 5), "kernel_ms": round(ms, 5), "flops_per_launch": flops}})
    return out


def encode_leg(torch, lib, NN, L, device):
    """Env::observe (env.h:202-262) alone: compact records in HBM -> fp32 planes in HBM, HBM-write bound (80 B read +
    7 680 B written per position); at the headline batch and at 2^20 positions."""
    import numpy as np
    nn = NN(filters=8, residuals=0, device=device)
    out = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "bytes_per_position": 7760, "sizes": []}
    rng = np.random.default_rng(0)
    for N in (512, 1 << 20):
        boards = np.zeros(N, dtype=L.BOARD_DTYPE)
        boards["piece_occ"] = rng.integers(0, 2**63, (N, 6), dtype=np.uint64); boards["color_occ"] = rng.integers(0, 2**63, (N, 2), dtype=np.uint64)
        boards["ply"] = rng.integers(0, 400, N); boards["ctm"] = rng.integers(0, 2, N); boards["castle_rights"] = rng.integers(0, 16, N)
        d_b = torch.from_numpy(boards.view(np.uint8).reshape(N, 80)).cuda()
        d_x = torch.empty((N, 1920), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        ms = C.c_float(0)
        iters = 2000 if N <= 4096 else 20
        for _ in range(2):
            if lib.kh_time_encode_device(nn.handle, C.c_void_p(d_b.data_ptr()), N, C.c_void_p(d_x.data_ptr()), iters, C.byref(ms)):
                raise RuntimeError(L.last_error())
        gbs = N * 7760 / (ms.value * 1e-3) / 1e9
        out["sizes"].append({"positions": N, "us_per_launch": round(ms.value * 1e3, 3), "positions_per_s": round(N / (ms.value * 1e-3), 1),
                             "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4)})
        del d_b, d_x
    out["achieved"], out["frac"] = out["sizes"][-1]["achieved"], out["sizes"][-1]["frac"]
    out["note"] = "512 positions (3.9 MB) is one launch's latency, not bandwidth; the roofline figure is the 2^20-position stream"
    return out


def configs0_leg(NN, W, L, device):
    """BASELINE configs[0]'s shape — 1 self-play game, 64 MCTS sims per move, one leaf per evaluation — on the engine
    (the reference runs it with its CPU NN::infer: ~64 positions/s).  Host search + batch-1
    evaluations through kh_encode_infer_legal; the net is options.def.yml's default 2 x 64."""
    from kami_amd import search as S
    nn = NN(8, 8, 30, 4672, filters=64, residuals=2, dtype="bf16", device=device, value_mode=L.KH_VALUE_PER_SAMPLE0)
    nn.load_weights(W.random_weights(30, 64, 2, seed=1, peaky=5.0), 1)
    pool = S.Pool(nn, games=1, threads=1, nodes=64, leaves_per_tree=1, seed=1)
    pool.run(min_evals=2000, max_seconds=5.0)
    s0 = pool.run(min_evals=0, max_seconds=0.0)
    st = pool.run(min_evals=10**12, max_seconds=2.0)
    de, dm, dt = st.evals - s0.evals, st.moves - s0.moves, st.seconds - s0.seconds
    pool.close()
    return {"workload": "1 self-play game, 64 MCTS sims per move, batch-1 leaf evaluations, 2x64 net, one host thread (BASELINE configs[0]'s shape)",
            "positions_per_s": round(dm / dt, 1), "leaf_evals_per_s": round(de / dt, 1), "us_per_eval_call": round(dt / max(1, de) * 1e6, 2),
            "reference_cpu_positions_per_s": 64, "reference_source": "BASELINE.md section 2 (reference Selfplay, CPU NN::infer, survey container)"}


def pcie_probe():
    """What the host link of this box moves between page-locked host memory and HBM: each direction alone, and both at once
    (kh_infer needs both: planes in while policies go out)."""
    import torch
    n = 128 << 20
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory(); h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in = torch.empty(n, dtype=torch.uint8, device="cuda"); d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(h2d, d2h, reps=8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
        s1.synchronize(); s2.synchronize()
        return n * reps / (time.perf_counter() - t0) / 1e9
    timed(True, True, 2)
    a, b, c = timed(True, False), timed(False, True), timed(True, True)
    return {"call": "host link probe", "workload": "128 MiB copies between page-locked host memory and HBM", "h2d_GBps": round(a, 1), "d2h_GBps": round(b, 1),
            "h2d_GBps_both_directions": round(c, 1), "d2h_GBps_both_directions": round(c, 1)}


def end_to_end_legs(NN, W, L, device):
    """The host-buffer ABI, PCIe included (what an unmodified kami sees through NN::infer, and what this repository's
    search sends): pageable caller buffers, synchronous calls, evaluations per wall second."""
    import numpy as np
    out = []

    def rate(fn, n, seconds=0.8):
        fn(); fn()
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < seconds:
            fn(); k += 1
        return n * k / (time.perf_counter() - t0)

    def threaded(make_fn, n, T):
        res = [0.0] * T
        start = threading.Barrier(T)

        def work(i):
            fn = make_fn(i)
            fn(); start.wait()
            res[i] = rate(fn, n)
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        return sum(res)

    B, F = 512, 119
    link = pcie_probe()
    bound = min(link["h2d_GBps_both_directions"] * 1e9 / (64 * F * 4), link["d2h_GBps_both_directions"] * 1e9 / (4672 * 4 + 4))
    link["kh_infer_bound"] = (f"{bound / 1e6:.2f} M leaf-evals/s on THIS box's link with both directions busy ({64 * F * 4} B in, {4672 * 4 + 4} B out "
                              "per evaluation); SURVEY 7's 2.07 M/s assumed 63 GB/s per direction")
    out.append(link)
    nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16", device=device)
    nn.load_weights(W.random_weights(F, 64, 6, seed=1), 1)

    def make_infer(i):
        x = np.random.default_rng(i).random((B, 8, 8, F), dtype=np.float32)
        pol = np.empty((B, 4672), np.float32); val = np.empty(B, np.float32)
        return lambda: nn.infer(x, B, pol, val)
    for T in (1, 4):
        out.append({"call": "kh_infer", "workload": f"{B} x ({F}x8x8) fp32 planes in, full [4672] policy + value out per call, 6x64 bf16 "
                                                    "(the legacy float ABI of NN::infer, nn.cpp:155-187)",
                    "threads": T, "value": round(threaded(make_infer, B, T), 1), "unit": "leaf-evals/s",
                    "pcie_bound_leaf_evals_per_s": round(bound, 1)})
    pinned = []

    def make_infer_pinned(i):
        x = np.random.default_rng(i).random((B, 8, 8, F), dtype=np.float32)
        pol = np.empty((B, 4672), np.float32); val = np.empty(B, np.float32)
        nn.pin(x); nn.pin(pol); pinned.extend([x, pol])
        return lambda: nn.infer(x, B, pol, val)
    for T in (1, 2, 4):
        out.append({"call": "kh_infer, caller buffers registered once with kh_pin_buffer", "workload": "as above; plain DMA out of / into the caller's pages, "
                    "four chunks of the batch alternating between two streams", "threads": T,
                    "value": round(threaded(make_infer_pinned, B, T), 1), "unit": "leaf-evals/s"})
        for arr in pinned:
            nn.unpin(arr)
        pinned.clear()
    del nn
    F = 30
    nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16", device=device, value_mode=L.KH_VALUE_PER_SAMPLE0)
    nn.load_weights(W.random_weights(F, 64, 6, seed=1), 1)

    def make_legal(i):
        rng = np.random.default_rng(i)
        boards = np.zeros(B, dtype=L.BOARD_DTYPE)
        boards["piece_occ"] = rng.integers(0, 2**63, (B, 6), dtype=np.uint64)
        boards["color_occ"] = rng.integers(0, 2**63, (B, 2), dtype=np.uint64)
        boards["ply"] = rng.integers(0, 300, B); boards["ctm"] = rng.integers(0, 2, B)
        nact = rng.integers(10, 50, B)
        offs = np.concatenate([[0], np.cumsum(nact)]).astype(np.int32)
        acts = rng.integers(0, 4672, int(offs[-1])).astype(np.int32)
        return lambda: nn.infer_legal(boards, offs, acts)
    for T in (1, 4):
        out.append({"call": "kh_encode_infer_legal", "workload": f"{B} compact records (80 B) + ~30 legal actions each in, legal priors + value out, "
                                                                 "6x64 bf16, encoder and legal-move gather on the device (SURVEY 8f row 1)",
                    "threads": T, "value": round(threaded(make_legal, B, T), 1), "unit": "leaf-evals/s"})
    return out


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
    ap.add_argument("--repeats", type=int, default=60, help="separately timed repeats for p10 / median / p90")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the variants / end_to_end legs (N = 1 only anyway)")
    ap.add_argument("--no-legs", action="store_true", help="skip value_f32 / encode / configs0 too: the headline kernel alone (profiler passes)")
    a = ap.parse_args()

    import numpy as np
    import torch
    from kami_amd import NN, weights as W, _lib as L, dist as kd

    rank, local_rank, world = kd.env_rank()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU path)")
    # one rank per GPU (the driver's multi-GPU run); KAMI_DIST_BACKEND=gloo lets two ranks share
    # one GPU to rehearse the N > 1 control flow on a single-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("KAMI_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dist = kd.init(backend)         # RCCL; only the barrier and the max-over-ranks use it

    F, Cc, R, B = a.features, a.filters, a.residuals, a.batch
    lib = L.load()
    leg = DeviceLeg(torch, lib, NN, W, a.dtype, F, Cc, R, B, 20240607 + rank, dev_index)
    nn, x, policy, vfull = leg.nn, leg.x, leg.policy, leg.vfull
    # an explicit stream of our own: torch's current stream is the legacy default stream (handle 0), which the
    # engine would read as "use your own stream"
    stream = torch.cuda.Stream()
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
    # the timed entry point against the host-buffer call of the same engine (the one the parity tests pin to the
    # reference's fixtures and the oracle): sampled rows, bit for bit
    rows = sorted({0, 1, B // 2, B - 2, B - 1} | {int(i) for i in np.random.default_rng(1).integers(0, B, 11)})
    ref_p, ref_v, _ = nn.infer_full(x[rows].cpu().numpy(), want_logits=False)
    got_p, got_v = policy[rows].cpu().numpy(), vfull[rows].cpu().numpy()
    if not (np.array_equal(got_p.view(np.uint32), ref_p.view(np.uint32)) and np.array_equal(got_v.view(np.uint32), ref_v.view(np.uint32))):
        raise RuntimeError("bench: the timed kh_infer_device output differs from kh_infer_full on the same rows")

    # distribution: `repeats` separately synchronised bursts of the same step (after the timed region: a sync per
    # burst would perturb `value`), each long enough (>= 2 ms, whatever --steps was) that the synchronise does not dominate
    per = max(20, int(2.0 / max(dt / a.steps * 1e3, 1e-3)))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    rates = []
    for _ in range(max(1, a.repeats)):
        t1 = time.perf_counter()
        for _ in range(per):
            step()
        torch.cuda.synchronize()
        rates.append(B * per / (time.perf_counter() - t1))
    rates.sort()
    pct = lambda q: rates[min(len(rates) - 1, int(round(q * (len(rates) - 1))))]

    # roofline of the dominant kernel (the forward pass), HIP events on the engine's own stream
    ms, _ = leg.settled_ms(a.prewarm, max(200, min(a.steps, 2000)) * (dt / a.steps))

    if rank == 0:
        roof = leg.roofline(ms)
        traffic, source = measured_traffic(F, Cc, R, B, a.dtype)
        roof["traffic"] = traffic
        roof["traffic_source"] = source
        out = {
            "metric": metric_label(B, F),
            "value": round(world * B * a.steps / dt, 1),
            "unit": "leaf-evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": workload_label(B, F, R, Cc),
                       "batch_per_gpu": B, "features": F, "filters": Cc, "residuals": R,
                       "parallelism": f"replicas x{world}, no data-path collective"},
            "roofline": roof,
            "distribution": {"unit": "leaf-evals/s per GPU (this rank)", "repeats": len(rates), "steps_per_repeat": per,
                             "p10": round(pct(0.10), 1), "median": round(pct(0.50), 1), "p90": round(pct(0.90), 1)},
        }
        if world == 1 and not a.no_legs:
            del leg
            torch.cuda.empty_cache()
            f32 = DeviceLeg(torch, lib, NN, W, "f32", F, Cc, R, B, 7, dev_index)
            ms32, _ = f32.settled_ms(a.prewarm, 0.4)
            out["value_f32"] = round(B / (ms32 * 1e-3), 1)
            out["roofline_f32"] = f32.roofline(ms32)
            out["roofline_f32"]["note"] = "the same step in the reference's own arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32 (layers_mfma.hip)"
            del f32
            torch.cuda.empty_cache()
            out["encode"] = encode_leg(torch, lib, NN, L, dev_index)
            out["configs0"] = configs0_leg(NN, W, L, dev_index)
        if world == 1 and not a.no_variants and not a.no_legs:
            out["variants"] = variant_legs(torch, lib, NN, W, L, dev_index, a.prewarm)
            out["end_to_end"] = end_to_end_legs(NN, W, L, dev_index)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F, Cc, R, B)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
