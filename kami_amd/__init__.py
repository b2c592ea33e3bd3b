"""kami_amd — MI355X-native leaf-evaluation engine for kami's self-play path.

Only what the hot path needs: the HIP kernels + C ABI (csrc/, libkamihip.so), the host-side
mirror of the reference's `kami::NN` interface (nn.py, host/nn.h) and weight-blob helpers.
"""
from .nn import NN, KamiError, PSIZE, NFEATURES, OBSIZE, VALUE_WIDTH  # noqa: F401
from . import weights  # noqa: F401
