"""The oracle (oracle/kami_oracle.c) against vectors produced by the UNMODIFIED reference
(oracle/gen_golden.py -> oracle/_ref/kami_ref).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as ko
from kami_amd import weights as W

# fp32 restatement vs libtorch fp32 (different summation order only)
POLICY_RTOL = 2e-4     # relative, on probabilities
LOGP_ATOL = 2e-4       # absolute, on log-probabilities (== logit differences)
VALUE_ATOL = 2e-5


def test_weight_count_matches_reference_param_count():
    # SURVEY §8c [measured]: the reference's 6x64 F=30 module has 496 844 parameters;
    # the blob additionally carries running_mean/running_var for each BatchNorm.
    F, C, R = 30, 64, 6
    n_bn_buffers = 2 * (C * (1 + 2 * R) + 128 + 1)
    assert W.weight_count(F, C, R) == 496844 + n_bn_buffers
    assert ko.weight_count(F, C, R) == W.weight_count(F, C, R)


def test_observe_bit_exact_vs_reference(observe_fixture):
    f = observe_fixture
    fens = [s.decode() for s in f["fen"]]
    boards = ko.boards_from_fens(fens, f["ply"])
    got = ko.observe(boards).reshape(len(fens), -1)
    want = f["obs"].astype(np.float32)
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # coverage of the edge cases the fixture is meant to hold
    assert f["ply"].max() > 255
    assert (boards["ctm"] == 1).sum() > 100
    assert (boards["castle_rights"] == 0).any() and (boards["castle_rights"] == 15).any()
    assert (boards["halfmove_clock"] >= 50).any()


def test_observe_known_answers():
    # SURVEY §8c known-answer values measured on the reference
    start = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
    after_e4 = "rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 1"
    b = ko.boards_from_fens([start, after_e4], [0, 1])
    o = ko.observe(b).reshape(2, 64, 30)
    assert o[0].sum() == 992 and o[1].sum() == 1056
    assert list(o[0, 0, :18]) == [0] * 14 + [1, 2, 4, 8]
    assert o[0, 0, 21] == 1 and o[0, 4, 23] == 1 and o[0, 12, 18] == 1
    assert o[0, 59, 28] == 1 and o[0, 63, 27] == 1
    assert list(o[1, 0, :18]) == [1] + [0] * 13 + [4, 8, 1, 2]
    assert o[1, 4, 22] == 1 and o[1, 59, 29] == 1


def test_forward_matches_reference(net_fixture):
    d = net_fixture
    F, C, R = d["features"], d["filters"], d["residuals"]
    policy, vfull, _ = ko.forward(d["blob"], F, C, R, d["x"])
    rows = d["policy_rows"]
    ref_p = d["policy"]
    assert np.allclose(policy.sum(1), 1.0, atol=1e-4)
    np.testing.assert_allclose(np.log(policy[rows]), np.log(ref_p), atol=LOGP_ATOL, rtol=0)
    np.testing.assert_allclose(policy[rows], ref_p, rtol=POLICY_RTOL, atol=1e-12)
    np.testing.assert_allclose(vfull, d["value_full"], atol=VALUE_ATOL, rtol=0)


def test_infer_value_copyout_quirk(net_fixture):
    """nn.cpp:186 copies the first B floats of the flattened [B,256] value tensor."""
    d = net_fixture
    F, C, R = d["features"], d["filters"], d["residuals"]
    rc, policy, value = ko.infer(d["blob"], F, C, R, d["x"])
    assert rc == 0
    B = d["x"].shape[0]
    assert np.array_equal(d["value"], d["value_full"].reshape(-1)[:B])   # the fixture itself
    np.testing.assert_allclose(value, d["value"], atol=VALUE_ATOL, rtol=0)


def test_infer_nan_guard():
    F, C, R = 30, 8, 0
    blob = W.random_weights(F, C, R, seed=3)
    x = np.random.default_rng(0).random((2, 8, 8, F), dtype=np.float32)
    x[1, 0, 0, 0] = np.nan
    rc, _, _ = ko.infer(blob, F, C, R, x)
    assert rc == 4      # policy is checked first (nn.cpp:176)
