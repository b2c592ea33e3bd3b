import numpy as np

from kami_amd import weights as W


def test_roundtrip(tmp_path):
    F, C, R = 30, 8, 1
    blob = W.random_weights(F, C, R, seed=5)
    p = str(tmp_path / "w.bin")
    W.save(p, blob, F, C, R, generation=9)
    b2, F2, C2, R2, g = W.load(p)
    assert (F2, C2, R2, g) == (F, C, R, 9)
    assert np.array_equal(blob, b2)


def test_specs_and_flops():
    assert W.flops_per_eval(119, 64, 6) == 67682304
    assert W.flops_per_eval(30, 64, 6) == 61120512
    assert W.flops_per_eval(119, 128, 10) == 398376960
    names = [n for n, _ in W.tensor_specs(30, 8, 2)]
    assert names[0] == "conv1.weight" and "residual1.batchnorm2.running_var" in names
    d = W.split(W.random_weights(30, 8, 2, seed=1), 30, 8, 2)
    assert d["valuefc.weight"].shape == (256, 64)
    assert d["policyconv2.weight"].shape == (73, 128, 1, 1)


def test_random_weights_deterministic():
    a = W.random_weights(30, 8, 1, seed=3)
    b = W.random_weights(30, 8, 1, seed=3)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, W.random_weights(30, 8, 1, seed=4))
