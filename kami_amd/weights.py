"""Weight blobs for the kami leaf-evaluation engine.

A blob is the flat fp32 parameter set of the reference network in the canonical order
documented in include/kami_hip.h (kh_weight_count); tensor names and shapes are the
reference's (kami/nn/nn.cpp:20-23,45-56).  On disk a blob is a 32-byte little-endian header
(int32 magic 'KAMW', features, filters, residuals, generation, 3 reserved) followed by the
floats — the engine's own checkpoint format (the reference writes torch archives,
nn.cpp:189-202; reading those is a later row of SURVEY §8f).
"""
from __future__ import annotations

import math
import struct
from typing import List, Tuple

import numpy as np

MAGIC = 0x574D414B  # b"KAMW"
PSIZE = 4672
POLICY_MID = 128
POLICY_PLANES = 73
VALUE_WIDTH = 256


def tensor_specs(features: int, filters: int, residuals: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor in blob order."""
    F, C, R = features, filters, residuals
    specs: List[Tuple[str, Tuple[int, ...]]] = []

    def convbn(conv: str, bn: str, co: int, ci: int, k: int) -> None:
        specs.append((conv + ".weight", (co, ci, k, k)))
        specs.append((conv + ".bias", (co,)))
        for s in ("weight", "bias", "running_mean", "running_var"):
            specs.append((bn + "." + s, (co,)))

    convbn("conv1", "batchnorm1", C, F, 3)
    for i in range(R):
        convbn(f"residual{i}.conv1", f"residual{i}.batchnorm1", C, C, 3)
        convbn(f"residual{i}.conv2", f"residual{i}.batchnorm2", C, C, 3)
    convbn("policyconv", "pbatchnorm", POLICY_MID, C, 1)
    specs.append(("policyconv2.weight", (POLICY_PLANES, POLICY_MID, 1, 1)))
    specs.append(("policyconv2.bias", (POLICY_PLANES,)))
    convbn("valueconv", "vbatchnorm", 1, C, 1)
    specs.append(("valuefc.weight", (VALUE_WIDTH, 64)))
    specs.append(("valuefc.bias", (VALUE_WIDTH,)))
    return specs


def weight_count(features: int, filters: int, residuals: int) -> int:
    return sum(int(np.prod(s)) for _, s in tensor_specs(features, filters, residuals))


def flops_per_eval(features: int, filters: int, residuals: int) -> int:
    """Algorithmic FLOPs of one leaf evaluation (2*MAC, convs and FC only), SURVEY §8d."""
    F, C, R = features, filters, residuals
    return 1152 * F * C + 2304 * R * C * C + 16512 * C + 1228800


def random_weights(features: int, filters: int, residuals: int, seed: int = 0,
                   peaky: float = 1.0) -> np.ndarray:
    """Seeded synthetic parameters (there are no checkpoints to load offline):
    conv / linear weights and biases uniform(-k, k), k = 1/sqrt(fan_in); BatchNorm gamma and
    running_var in [0.9, 1.1], beta and running_mean in [-0.1, 0.1] so that BN folding is
    exercised.  peaky > 1 scales policyconv2.weight to give a non-flat policy."""
    rng = np.random.default_rng(seed)
    parts = []
    fan_in = 1
    for name, shape in tensor_specs(features, filters, residuals):
        n = int(np.prod(shape))
        leaf = name.rsplit(".", 1)[1]
        is_bn = "batchnorm" in name
        if is_bn:
            if leaf in ("weight", "running_var"):
                t = rng.uniform(0.9, 1.1, n)
            else:
                t = rng.uniform(-0.1, 0.1, n)
        else:
            if leaf == "weight":
                fan_in = int(np.prod(shape[1:]))
            k = 1.0 / math.sqrt(fan_in)
            t = rng.uniform(-k, k, n)
            if name == "policyconv2.weight":
                t = t * peaky
        parts.append(t.astype(np.float32))
    return np.concatenate(parts)


def split(blob: np.ndarray, features: int, filters: int, residuals: int) -> dict:
    out = {}
    off = 0
    for name, shape in tensor_specs(features, filters, residuals):
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError(f"blob has {blob.size} floats, expected {off}")
    return out


def save(path: str, blob: np.ndarray, features: int, filters: int, residuals: int,
         generation: int = 0) -> None:
    blob = np.ascontiguousarray(blob, dtype="<f4")
    if blob.size != weight_count(features, filters, residuals):
        raise ValueError("blob size does not match (features, filters, residuals)")
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", MAGIC, features, filters, residuals, generation, 0, 0, 0))
        f.write(blob.tobytes())


def load(path: str):
    """-> (blob, features, filters, residuals, generation)"""
    with open(path, "rb") as f:
        hdr = f.read(32)
        if len(hdr) != 32:
            raise ValueError("truncated weight file")
        magic, F, C, R, gen, _, _, _ = struct.unpack("<8i", hdr)
        if magic != MAGIC:
            raise ValueError("not a kami weight blob")
        if not (1 <= F <= 4096 and 1 <= C <= 1024 and 0 <= R <= 256):         # the engine's own limits (kh_checkpoint_read)
            raise ValueError("weight file header out of range")
        blob = np.frombuffer(f.read(), dtype="<f4").copy()
    if blob.size != weight_count(F, C, R):
        raise ValueError("weight file size does not match its header")
    return blob, F, C, R, gen
