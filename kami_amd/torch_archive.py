"""Reader for the checkpoints the reference writes (kami/nn/nn.cpp:189-202, NN::write):

    serialize::OutputArchive a;  mod->save(a);  a.write("generation", IValue(generation));  a.save_to(path);

i.e. a libtorch archive: a zip whose `<root>/data.pkl` is a protocol-2 pickle describing a tree of
script-module objects whose leaves are tensors (`torch._utils._rebuild_tensor_v2` over a persistent
storage id) and whose tensor bytes are the stored zip members `<root>/data/<id>`.

Nothing from the file is executed: this module does not import `pickle` or `torch`.  It interprets the
handful of opcodes libtorch's pickler emits with its own stack machine, builds plain dicts / tuples, and
refuses anything else (an unknown opcode, an unexpected GLOBAL, a compressed tensor member).  The C++
twin for kami::NN::read is kami_amd/host/torch_archive.h.
"""
from __future__ import annotations

import struct
import zipfile
from zlib import error as zlib_error
from typing import Dict, Tuple

import numpy as np

from . import weights as W

_STORAGE_DTYPES = {
    "FloatStorage": "<f4", "DoubleStorage": "<f8", "HalfStorage": "<f2", "LongStorage": "<i8",
    "IntStorage": "<i4", "ShortStorage": "<i2", "CharStorage": "i1", "ByteStorage": "u1", "BoolStorage": "?",
}


class ArchiveError(ValueError):
    pass


class _Global:
    __slots__ = ("module", "name")

    def __init__(self, module: str, name: str):
        self.module, self.name = module, name

    def __repr__(self):
        return f"<global {self.module}.{self.name}>"


class _Object:
    """An instance of a script-module class (GLOBAL '__torch__... Module' + NEWOBJ + BUILD)."""
    __slots__ = ("cls", "state")

    def __init__(self, cls: _Global):
        self.cls, self.state = cls, {}


class _TensorRef:
    __slots__ = ("key", "dtype", "numel", "offset", "size", "stride")

    def __init__(self, key, dtype, numel, offset, size, stride):
        self.key, self.dtype, self.numel, self.offset, self.size, self.stride = key, dtype, numel, offset, size, stride


_MARK = object()


def _unpickle(data: bytes):
    """Interpret the opcode subset libtorch's Pickler writes (torch/csrc/jit/serialization/pickler.cpp)."""
    stack, memo = [], {}
    i, n = 0, len(data)

    def pop_mark():
        k = len(stack) - 1
        while k >= 0 and stack[k] is not _MARK:
            k -= 1
        if k < 0:
            raise ArchiveError("pickle: MARK missing")
        items = stack[k + 1:]
        del stack[k:]
        return items

    while i < n:
        op = data[i]; i += 1
        if op == 0x80:                                   # PROTO
            if data[i] > 5: raise ArchiveError(f"pickle protocol {data[i]}")
            i += 1
        elif op == 0x63:                                 # GLOBAL 'module name'
            e1 = data.index(b"\n", i); e2 = data.index(b"\n", e1 + 1)
            stack.append(_Global(data[i:e1].decode("latin-1"), data[e1 + 1:e2].decode("latin-1"))); i = e2 + 1
        elif op == 0x71: memo[data[i]] = stack[-1]; i += 1                                  # BINPUT
        elif op == 0x72: memo[struct.unpack_from("<I", data, i)[0]] = stack[-1]; i += 4     # LONG_BINPUT
        elif op == 0x68: stack.append(memo[data[i]]); i += 1                                # BINGET
        elif op == 0x6a: stack.append(memo[struct.unpack_from("<I", data, i)[0]]); i += 4   # LONG_BINGET
        elif op == 0x29: stack.append(())                                                  # EMPTY_TUPLE
        elif op == 0x7d: stack.append({})                                                  # EMPTY_DICT
        elif op == 0x5d: stack.append([])                                                  # EMPTY_LIST
        elif op == 0x28: stack.append(_MARK)                                               # MARK
        elif op == 0x58:                                                                   # BINUNICODE
            ln = struct.unpack_from("<I", data, i)[0]; i += 4
            stack.append(data[i:i + ln].decode("utf-8", "replace")); i += ln
        elif op == 0x4b: stack.append(data[i]); i += 1                                      # BININT1
        elif op == 0x4d: stack.append(struct.unpack_from("<H", data, i)[0]); i += 2         # BININT2
        elif op == 0x4a: stack.append(struct.unpack_from("<i", data, i)[0]); i += 4         # BININT
        elif op == 0x8a:                                                                   # LONG1
            ln = data[i]; i += 1
            stack.append(int.from_bytes(data[i:i + ln], "little", signed=True)); i += ln
        elif op == 0x47: stack.append(struct.unpack_from(">d", data, i)[0]); i += 8         # BINFLOAT
        elif op == 0x88: stack.append(True)
        elif op == 0x89: stack.append(False)
        elif op == 0x4e: stack.append(None)
        elif op == 0x74: stack.append(tuple(pop_mark()))                                   # TUPLE
        elif op == 0x85: stack[-1:] = [(stack[-1],)]                                       # TUPLE1
        elif op == 0x86: stack[-2:] = [(stack[-2], stack[-1])]                             # TUPLE2
        elif op == 0x87: stack[-3:] = [(stack[-3], stack[-2], stack[-1])]                  # TUPLE3
        elif op == 0x81:                                                                   # NEWOBJ
            args = stack.pop(); cls = stack.pop()
            if not isinstance(cls, _Global) or not cls.module.startswith("__torch__") or args != ():
                raise ArchiveError(f"pickle: NEWOBJ of {cls!r}")
            stack.append(_Object(cls))
        elif op == 0x51:                                                                   # BINPERSID
            pid = stack.pop()
            if (not isinstance(pid, tuple) or len(pid) != 5 or pid[0] != "storage" or not isinstance(pid[1], _Global)
                    or pid[1].module != "torch" or pid[1].name not in _STORAGE_DTYPES):
                raise ArchiveError(f"pickle: unexpected persistent id {pid!r}")
            stack.append(("storage", _STORAGE_DTYPES[pid[1].name], str(pid[2]), int(pid[4])))
        elif op == 0x52:                                                                   # REDUCE
            args = stack.pop(); fn = stack.pop()
            if not isinstance(fn, _Global):
                raise ArchiveError("pickle: REDUCE of a non-global")
            if (fn.module, fn.name) == ("collections", "OrderedDict") and args == ():
                stack.append({})
            elif (fn.module, fn.name) == ("torch._utils", "_rebuild_tensor_v2"):
                st, off, size, stride = args[0], args[1], args[2], args[3]
                if not (isinstance(st, tuple) and st[0] == "storage"):
                    raise ArchiveError("pickle: tensor without a storage")
                stack.append(_TensorRef(st[2], st[1], st[3], int(off), tuple(int(v) for v in size), tuple(int(v) for v in stride)))
            else:
                raise ArchiveError(f"pickle: refusing to call {fn!r}")
        elif op == 0x75:                                                                   # SETITEMS
            items = pop_mark(); d = stack[-1]
            if not isinstance(d, dict): raise ArchiveError("pickle: SETITEMS on a non-dict")
            for k in range(0, len(items), 2): d[items[k]] = items[k + 1]
        elif op == 0x73:                                                                   # SETITEM
            v = stack.pop(); k = stack.pop(); stack[-1][k] = v
        elif op == 0x65:                                                                   # APPENDS
            items = pop_mark(); stack[-1].extend(items)
        elif op == 0x61: v = stack.pop(); stack[-1].append(v)                               # APPEND
        elif op == 0x62:                                                                   # BUILD
            state = stack.pop(); obj = stack[-1]
            if not isinstance(obj, _Object) or not isinstance(state, dict):
                raise ArchiveError("pickle: BUILD on an unexpected object")
            obj.state = state
        elif op == 0x2e:                                                                   # STOP
            return stack.pop()
        else:
            raise ArchiveError(f"pickle: opcode 0x{op:02x} at {i - 1} is not one libtorch's module pickler writes")
    raise ArchiveError("pickle: no STOP")


def read_archive(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, object]]:
    """-> (tensors by dotted name, other attributes such as 'generation').  Whatever a malformed file trips over inside
    zipfile / struct / numpy comes out as ArchiveError."""
    try:
        return _read_archive(path)
    except ArchiveError:
        raise
    except (zipfile.BadZipFile, zlib_error, struct.error, IndexError, KeyError, ValueError, EOFError, OverflowError,
            NotImplementedError, TypeError, AttributeError, RecursionError) as e:
        raise ArchiveError(f"malformed archive: {type(e).__name__}: {e}") from None


def _read_archive(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, object]]:
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") and n.count("/") == 1]
        if len(pkl) != 1:
            raise ArchiveError("not a libtorch module archive (no <root>/data.pkl)")
        root = pkl[0].split("/")[0]
        top = _unpickle(z.read(pkl[0]))
        if not isinstance(top, _Object):
            raise ArchiveError("archive root is not a module")
        tensors: Dict[str, np.ndarray] = {}
        attrs: Dict[str, object] = {}
        storages: Dict[str, bytes] = {}

        def walk(obj: _Object, prefix: str, depth: int = 0):
            if depth > 32:                                   # a memo reference can make a state contain its own object
                raise ArchiveError("module tree too deep (a cycle?)")
            for k, v in obj.state.items():
                name = prefix + str(k)
                if isinstance(v, _Object):
                    walk(v, name + ".", depth + 1)
                elif isinstance(v, _TensorRef):
                    member = f"{root}/data/{v.key}"
                    if v.key not in storages:
                        if member not in names:
                            raise ArchiveError(f"storage {v.key} missing")
                        info = z.getinfo(member)
                        if info.compress_type != zipfile.ZIP_STORED:
                            raise ArchiveError(f"{member}: tensor data is compressed")
                        storages[v.key] = z.read(member)
                    raw = np.frombuffer(storages[v.key], dtype=v.dtype)
                    # sizes, strides and offset come from the file: contiguous tensors only (what module.save writes), every
                    # element inside the storage — nothing here may index memory by an unchecked number
                    if len(v.stride) != len(v.size) or any(d < 0 for d in v.size) or v.offset < 0:
                        raise ArchiveError(f"{name}: malformed size / stride / offset")
                    numel = 1
                    for d in v.size:
                        numel *= d
                    want = 1
                    for d, st in zip(reversed(v.size), reversed(v.stride)):
                        if d != 1 and st != want:
                            raise ArchiveError(f"{name} is not contiguous")
                        want *= d
                    if v.offset + numel > raw.size:
                        raise ArchiveError(f"{name} exceeds its storage ({raw.size} elements)")
                    tensors[name] = raw[v.offset:v.offset + numel].reshape(v.size).copy()
                elif prefix == "":
                    attrs[name] = v

        walk(top, "")
    return tensors, attrs


def load_reference_checkpoint(path: str):
    """A checkpoint written by the reference's NN::write -> (blob, features, filters, residuals, generation),
    the blob in the canonical order of include/kami_hip.h.  The BatchNorm `num_batches_tracked` counters
    (momentum is fixed, nn.cpp never reads them) are not part of the blob and are dropped."""
    tensors, attrs = read_archive(path)
    if "conv1.weight" not in tensors or "generation" not in attrs:
        raise ArchiveError("not a kami checkpoint (no conv1.weight / generation)")
    C, F = int(tensors["conv1.weight"].shape[0]), int(tensors["conv1.weight"].shape[1])
    R = 0
    while f"residual{R}.conv1.weight" in tensors:
        R += 1
    parts = []
    for name, shape in W.tensor_specs(F, C, R):
        if name not in tensors:
            raise ArchiveError(f"checkpoint has no tensor {name}")
        t = tensors[name]
        if tuple(t.shape) != tuple(shape) or t.dtype != np.float32:
            raise ArchiveError(f"{name}: shape {t.shape} / {t.dtype}, expected {shape} float32")
        parts.append(np.ascontiguousarray(t).ravel())
    extra = [k for k in tensors if k not in {n for n, _ in W.tensor_specs(F, C, R)} and not k.endswith("num_batches_tracked")]
    if extra:
        raise ArchiveError(f"checkpoint has tensors this network does not: {extra[:4]}")
    return np.concatenate(parts), F, C, R, int(attrs["generation"])
