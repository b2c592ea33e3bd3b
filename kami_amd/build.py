"""In-tree build of libkamihip.so (HIP kernels + C ABI) for gfx950.

    python -m kami_amd.build            # or kami_amd.build.build_all()

hipcc cross-compiles without a GPU.  Each .hip translation unit is compiled to an object
with hipcc and the objects are linked with the host C++ driver against ONE HIP runtime:
the libamdhip64.so that ships inside the torch wheel of this image, when torch is present,
so that a process which also imports torch (bench.py uses torch.distributed) never ends up
with two HIP runtimes; /opt/rocm/lib otherwise.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libkamihip.so")
ARCH = "gfx950"
SOURCES = ["kh_api.hip", "encode.hip", "forward_simple.hip", "tower_mfma.hip", "tower8_mfma.hip", "layers_mfma.hip", "train.hip"]
# MFMA results in arch VGPRs: the epilogues read them with VALU ops and would otherwise pay a
# v_accvgpr_read per value (the kernel runs one wave per SIMD, registers are not scarce).
EXTRA_FLAGS = {"tower_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "tower8_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# (tools/wide_stamps.py and tools/t128_stamps.py use a separate diagnostic object of layers_mfma.hip built with
#  -DKAMI_WIDE_DIAG — in-kernel s_memtime stamps — linked into csrc/build/libkamihip_diag.so by tools/build_diag.sh;
#  the shipped library never contains it, and tower_mfma.hip has no diagnostic variants any more.)
HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result", "-Wno-pass-failed",
               "-ffp-contract=fast"]


def _hip_runtime_dir() -> str:
    if os.environ.get("KAMI_HIP_LIBDIR"):
        return os.environ["KAMI_HIP_LIBDIR"]
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = os.path.join(os.path.dirname(spec.origin), "lib")
            if os.path.exists(os.path.join(d, "libamdhip64.so")):
                return d
    except Exception:
        pass
    return "/opt/rocm/lib"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "kami_hip.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        rt = _hip_runtime_dir()
        run(["g++", "-shared", "-o", LIB] + objs +
            [f"-L{rt}", "-lamdhip64", f"-Wl,-rpath,{rt}", "-lpthread"])
    build_search(force or bool(jobs), verbose)
    return LIB


SEARCH_LIB = os.path.join(HERE, "libkamisearch.so")


def build_search(force: bool = False, verbose: bool = False) -> str:
    """libkamisearch.so: the host side that feeds the engine (rules, MCTS, self-play pool; include/kami_search.h).
    Plain C++, linked against libkamihip.so for kh_encode_infer_legal."""
    host = os.path.join(HERE, "host")
    srcs = [os.path.join(host, f) for f in ("search_api.cpp", "mcts.h", "env.h", "chess.h")]
    srcs += [os.path.join(os.path.dirname(HERE), "include", f) for f in ("kami_search.h", "kami_hip.h")]
    if force or _stale(SEARCH_LIB, srcs + [LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(os.path.dirname(HERE), "include"), "-o", SEARCH_LIB, srcs[0],
               f"-L{HERE}", "-lkamihip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SEARCH_LIB


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
