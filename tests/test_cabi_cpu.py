"""CPU-side checks of the C-ABI: the library loads, exports every symbol the header
declares, and its pure-host entry points behave.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from control_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kkt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(kkt_[a-z0-9_]+)\s*\(", src)
    # function-pointer typedefs are not exported symbols
    typedefs = set(re.findall(r"\(\*\s*(kkt_[a-z0-9_]+)\s*\)", src))
    return sorted(set(names) - typedefs)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"libkkt.so does not export {name}"
    # and the ctypes table binds exactly the header
    assert sorted(_lib.SIGNATURES) == declared


def test_shard_range_partitions_rows():
    lib = _lib.load()
    for m in (1, 7, 9, 64, 128):
        for world in (1, 2, 3, 8):
            if world > m:
                continue
            covered = []
            for r in range(world):
                lo, hi = C.c_int(), C.c_int()
                assert lib.kkt_shard_range(m, r, world, C.byref(lo), C.byref(hi)) == 0
                assert hi.value - lo.value in (m // world, m // world + 1)
                covered += list(range(lo.value, hi.value))
            assert covered == list(range(m))
    lo, hi = C.c_int(), C.c_int()
    assert lib.kkt_shard_range(4, 5, 4, C.byref(lo), C.byref(hi)) == -1


def test_create_fails_loudly_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.kkt_create(C.byref(h), 0)
    if rc == 0:                      # a GPU is present: nothing to check here
        lib.kkt_destroy(h)
        pytest.skip("GPU present")
    assert rc == -2
    assert b"no HIP device" in lib.kkt_last_error(None)


def test_null_handle_is_an_argument_error():
    lib = _lib.load()
    assert lib.kkt_finalize(None) == -1
    assert lib.kkt_local_size(None) == -1
