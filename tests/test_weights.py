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


# ---------------------------------------------------------------------------------------------
# reference checkpoints (NN::write, nn.cpp:189-202): a libtorch archive read without libtorch

import os
import shutil
import zipfile

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLDEN, "ref_checkpoint_f30_c8_r1.pt")


def test_reference_checkpoint_both_readers_match_the_fixture_blob():
    """tests/golden/ref_checkpoint_f30_c8_r1.pt was written by the reference's own NN::write from the blob of
    the net_f30_c8_r1 fixture (oracle/gen_golden.py::gen_checkpoint).  The engine's C reader
    (kh_checkpoint_read, csrc/torch_archive.h) and the Python twin must both give that blob back bit for bit."""
    from kami_amd import torch_archive as TA
    from kami_amd.nn import read_checkpoint
    want = np.load(os.path.join(GOLDEN, "net_f30_c8_r1.npz"))["blob"]
    for reader in (read_checkpoint, TA.load_reference_checkpoint):
        blob, F, C, R, gen = reader(CKPT)
        assert (F, C, R, gen) == (30, 8, 1, 7)
        assert blob.dtype == np.float32 and np.array_equal(blob.view(np.uint32), want.view(np.uint32))
    tensors, attrs = TA.read_archive(CKPT)
    assert attrs["generation"] == 7 and tensors["batchnorm1.num_batches_tracked"].dtype == np.int64
    assert tensors["residual0.conv2.weight"].shape == (8, 8, 3, 3)


def test_checkpoint_reader_takes_the_engine_container_too(tmp_path):
    from kami_amd.nn import read_checkpoint
    blob = W.random_weights(30, 8, 2, seed=5)
    p = str(tmp_path / "w.bin")
    W.save(p, blob, 30, 8, 2, generation=4)
    b2, F, C, R, gen = read_checkpoint(p)
    assert (F, C, R, gen) == (30, 8, 2, 4) and np.array_equal(blob, b2)


def test_checkpoint_reader_refuses_what_it_does_not_know(tmp_path):
    """Not a checkpoint, a truncated archive, and an archive whose pickle asks for a call outside the two
    reconstructors libtorch's module pickler emits: all fail with an error, nothing is executed."""
    from kami_amd import KamiError, torch_archive as TA
    from kami_amd.nn import read_checkpoint
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"hello, not a checkpoint at all" * 10)
    with pytest.raises(KamiError):
        read_checkpoint(str(junk))
    cut = tmp_path / "cut.pt"
    cut.write_bytes(open(CKPT, "rb").read()[:50000])
    with pytest.raises(KamiError):
        read_checkpoint(str(cut))
    # same members, but data.pkl replaced by a pickle that REDUCEs os.system
    evil = tmp_path / "evil.pt"
    with zipfile.ZipFile(CKPT) as zin, zipfile.ZipFile(evil, "w", zipfile.ZIP_STORED) as zout:
        for item in zin.infolist():
            data = zin.read(item.filename)
            if item.filename.endswith("/data.pkl"):
                data = b"\x80\x02cos\nsystem\nX\x04\x00\x00\x00true\x85R."
            zout.writestr(item.filename, data)
    with pytest.raises(KamiError, match="refusing to call os.system"):
        read_checkpoint(str(evil))
    with pytest.raises(TA.ArchiveError, match="refusing to call"):
        TA.load_reference_checkpoint(str(evil))


def _hostile_archive(path, pkl, storage=b"\x00" * 16):
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("m/data.pkl", pkl)
        z.writestr("m/data/0", storage)
        z.writestr("m/version", b"3\n")


def _tensor_pickle(size, stride, offset=0):
    """data.pkl of a module with ONE fp32 tensor attribute 'w' over storage '0' (4 elements), sizes / strides as given."""
    def num(v):
        return b"\x8a\x08" + int(v).to_bytes(8, "little", signed=True)

    def tup(vals):
        return b"(" + b"".join(num(v) for v in vals) + b"t"
    s = lambda t: b"X" + len(t).to_bytes(4, "little") + t
    storage = b"(" + s(b"storage") + b"ctorch\nFloatStorage\n" + s(b"0") + s(b"cpu") + b"K\x04" + b"tQ"
    tensor = (b"ctorch._utils\n_rebuild_tensor_v2\n(" + storage + num(offset) + tup(size) + tup(stride) + b"\x89" +
              b"ccollections\nOrderedDict\n)R" + b"tR")
    return b"\x80\x02c__torch__.M\nM\n)\x81}(" + s(b"w") + tensor + b"ub."


@pytest.mark.parametrize("size,stride,offset,what", [
    ((1 << 40,), (1,), 0, "exceeds its storage"),            # more elements than the storage holds
    ((1 << 33, 1 << 33), (1 << 33, 1), 0, "exceeds"),        # a product that overflows 64 bits
    ((2, 2), (1,), 0, "stride|rank"),                        # rank mismatch
    ((2, 2), (1 << 20, 1), 0, "contiguous"),                 # a stride that would index far outside the storage
    ((4,), (1,), 3, "exceeds its storage"),                  # offset + numel past the end
    ((4,), (1,), -1, "exceeds|malformed"),
    ((-4,), (1,), 0, "negative|malformed"),
])
def test_checkpoint_readers_bound_every_number_from_the_file(tmp_path, size, stride, offset, what):
    """Sizes, strides and offsets of a tensor come from the file: both readers refuse anything that would read outside
    the tensor's storage (the file is untrusted input)."""
    from kami_amd import KamiError, torch_archive as TA
    from kami_amd.nn import read_checkpoint
    p = str(tmp_path / "hostile.pt")
    _hostile_archive(p, _tensor_pickle(size, stride, offset))
    with pytest.raises(KamiError, match=what):
        read_checkpoint(p)
    with pytest.raises(TA.ArchiveError, match=what):
        TA.read_archive(p)


def test_checkpoint_readers_survive_a_module_that_contains_itself(tmp_path):
    from kami_amd import KamiError, torch_archive as TA
    from kami_amd.nn import read_checkpoint
    s = lambda t: b"X" + len(t).to_bytes(4, "little") + t
    pkl = b"\x80\x02c__torch__.M\nM\n)\x81q\x00}(" + s(b"me") + b"h\x00ub."
    p = str(tmp_path / "cycle.pt")
    _hostile_archive(p, pkl)
    with pytest.raises(KamiError, match="too deep|already part of another"):
        read_checkpoint(p)
    with pytest.raises(TA.ArchiveError, match="too deep"):
        TA.read_archive(p)


def test_checkpoint_readers_survive_300000_nested_tuples(tmp_path):
    """A 300 KB data.pkl of ')' + 300 000 TUPLE1 + STOP nests a value 300 000 levels deep: the C++ reader's value graph
    is torn down recursively, so it bounds the nesting while parsing (it used to overflow the stack in the destructor
    after refusing the file); the Python twin refuses the root."""
    from kami_amd import KamiError, torch_archive as TA
    from kami_amd.nn import read_checkpoint
    p = str(tmp_path / "deep.pt")
    _hostile_archive(p, b"\x80\x02)" + b"\x85" * 300000 + b".")
    with pytest.raises(KamiError, match="nesting deeper"):
        read_checkpoint(p)
    with pytest.raises(TA.ArchiveError):
        TA.read_archive(p)
    # the same depth through lists, dicts and a module's state
    s = lambda t: b"X" + len(t).to_bytes(4, "little") + t
    for pkl in (b"\x80\x02" + b"]" * 100000 + b"a" * 99999 + b".",
                b"\x80\x02" + (b"}" + s(b"k")) * 5000 + b"N" + b"s" * 5000 + b"."):
        _hostile_archive(p, pkl)
        with pytest.raises(KamiError, match="nesting deeper|not a module"):
            read_checkpoint(p)
        with pytest.raises(TA.ArchiveError):
            TA.read_archive(p)


def test_checkpoint_readers_survive_mutated_archives(tmp_path):
    """Seeded byte mutations of the reference-written archive (biased to the pickle and the zip directory, some
    truncated): both readers either return a blob or raise their error — no crash, nothing else."""
    import random
    from kami_amd import KamiError, torch_archive as TA
    from kami_amd.nn import read_checkpoint
    src = open(CKPT, "rb").read()
    with zipfile.ZipFile(CKPT) as z:
        info = z.getinfo("ref_checkpoint_f30_c8_r1/data.pkl")
    pk0, pk1 = info.header_offset, info.header_offset + 30 + len(info.filename) + info.file_size + 64
    rnd = random.Random(20240607)
    p = str(tmp_path / "m.pt")
    accepted = refused = 0
    for _ in range(400):
        b = bytearray(src)
        region = rnd.choice(("pkl", "dir", "any"))
        for _ in range(rnd.choice((1, 1, 2, 4, 16))):
            i = rnd.randrange(pk0, pk1) if region == "pkl" else rnd.randrange(len(b) - 6000, len(b)) if region == "dir" else rnd.randrange(len(b))
            b[i] = rnd.randrange(256)
        if rnd.random() < 0.1:
            b = b[:rnd.randrange(len(b))]
        open(p, "wb").write(b)
        try:
            read_checkpoint(p); accepted += 1
        except KamiError:
            refused += 1
        try:
            TA.load_reference_checkpoint(p)
        except TA.ArchiveError:
            pass
    assert refused > 100 and accepted > 20


def test_weight_container_header_is_bounded_before_anything_is_sized(tmp_path):
    """A weight file whose header claims a huge network (or a negative one) is refused at once by both loaders: nothing
    is allocated or looped over by the header's numbers."""
    import struct, time
    from kami_amd import KamiError
    from kami_amd.nn import read_checkpoint
    for F, C, R in ((4096, 1024, 256), (30, 8, 2_000_000_000), (-1, 8, 1), (30, 0, 1)):
        p = str(tmp_path / "h.bin")
        open(p, "wb").write(struct.pack("<8i", W.MAGIC, F, C, R, 1, 0, 0, 0) + b"\x00" * 64)
        t0 = time.time()
        with pytest.raises(KamiError):
            read_checkpoint(p)
        with pytest.raises(ValueError):
            W.load(p)
        assert time.time() - t0 < 1.0



def _write_module_archive(path, tensors, ints):
    """A libtorch-style archive written by hand: `tensors` name -> fp32 array ('a.b.weight' nests modules a, b), `ints`
    top-level integer attributes.  Protocol-2 pickle of __torch__ module objects, one storage member per tensor."""
    s = lambda t: b"X" + len(t).to_bytes(4, "little") + t
    num = lambda v: b"\x8a\x08" + int(v).to_bytes(8, "little", signed=True)
    tup = lambda vals: b"(" + b"".join(num(v) for v in vals) + b"t"
    storages = {}

    def tensor(arr):
        key = str(len(storages)).encode()
        storages[key] = np.ascontiguousarray(arr, dtype=np.float32).tobytes()
        strides = [int(np.prod(arr.shape[i + 1:])) for i in range(arr.ndim)]
        storage = b"(" + s(b"storage") + b"ctorch\nFloatStorage\n" + s(key) + s(b"cpu") + num(arr.size) + b"tQ"
        return (b"ctorch._utils\n_rebuild_tensor_v2\n(" + storage + num(0) + tup(arr.shape) + tup(strides) + b"\x89" +
                b"ccollections\nOrderedDict\n)R" + b"tR")
    tree = {}
    for name, arr in tensors.items():
        cur = tree
        *path_, leaf = name.split(".")
        for part in path_:
            cur = cur.setdefault(part, {})
        cur[leaf] = arr

    def module(d, extra=()):
        body = b"".join(s(k.encode()) + (module(v) if isinstance(v, dict) else tensor(v)) for k, v in d.items())
        body += b"".join(s(k.encode()) + num(v) for k, v in extra)
        return b"c__torch__.M\nM\n)\x81}(" + body + b"ub"
    pkl = b"\x80\x02" + module(tree, tuple(ints.items())) + b"."
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("m/data.pkl", pkl)
        for key, raw in storages.items():
            z.writestr("m/data/" + key.decode(), raw)
        z.writestr("m/version", b"3\n")


def test_checkpoint_readers_compare_whole_shapes_and_bound_their_numbers(tmp_path):
    """A tensor with the right element count but another shape ([C, 9 F, 1, 1] for conv1.weight, a transposed valuefc) is
    not the network's: both readers compare whole shapes (the C reader used to match element counts only) and range-check
    the generation; the hand-written archive of the right shapes is read back bit for bit by both."""
    from kami_amd import KamiError, torch_archive as TA, weights as W
    from kami_amd.nn import read_checkpoint
    F, Cc, R = 30, 8, 1
    blob = W.random_weights(F, Cc, R, seed=2)

    def tensors():
        out, off = {}, 0
        for n, shape in W.tensor_specs(F, Cc, R):
            k = int(np.prod(shape)); out[n] = blob[off:off + k].reshape(shape).copy(); off += k
        return out
    p = str(tmp_path / "a.pt")
    _write_module_archive(p, tensors(), {"generation": 7})
    got, f, c, r, gen = read_checkpoint(p)
    assert (f, c, r, gen) == (F, Cc, R, 7) and np.array_equal(got.view(np.uint32), blob.view(np.uint32))
    got2 = TA.load_reference_checkpoint(p)
    assert np.array_equal(np.asarray(got2[0]).view(np.uint32), blob.view(np.uint32))
    for name, new_shape in (("conv1.weight", (Cc, 9 * F, 1, 1)), ("valuefc.weight", (64, 256)), ("policyconv2.weight", (73 * 128, 1, 1, 1)),
                            ("residual0.conv2.weight", (Cc * Cc, 3, 3))):
        t = tensors()
        t[name] = t[name].reshape(new_shape)
        _write_module_archive(p, t, {"generation": 7})
        with pytest.raises(KamiError, match="shape|rank|checkpoint"):
            read_checkpoint(p)
        with pytest.raises(TA.ArchiveError):
            TA.load_reference_checkpoint(p)
    _write_module_archive(p, tensors(), {"generation": 1 << 40})
    with pytest.raises(KamiError, match="generation out of range"):
        read_checkpoint(p)
