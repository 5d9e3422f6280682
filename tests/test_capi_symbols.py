"""The C-ABI library loads without a GPU and exports every function that
include/suchtree_hip.h declares; host-only entry points work; device entry
points fail cleanly (no compute is attempted without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from suchtree_amd import _capi, build as st_build
from suchtree_amd import synth


@pytest.fixture(scope="module")
def lib():
    st_build.build()
    return _capi.load()


def _declared():
    text = open(os.path.join(ROOT, "include", "suchtree_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert set(names) == set(_capi.SYMBOLS)
    for name in names:
        assert getattr(lib, name) is not None


def test_kernels_are_gfx950_code_objects():
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_canopy" in blob and b"k_walk" in blob


def test_host_depths(lib):
    parent, _ = synth.balanced_tree(8)
    out = np.zeros(len(parent), dtype=np.int32)
    td = ctypes.c_int32(0)
    rc = lib.st_host_depths(parent.ctypes.data_as(ctypes.c_void_p), len(parent),
                            out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(td))
    assert rc == 0 and td.value == 9 and out.max() == 8 and out[255] == 0
    bad = np.array([1, 2, 0], dtype=np.int32)
    rc = lib.st_host_depths(bad.ctypes.data_as(ctypes.c_void_p), 3, None, None)
    assert rc == _capi.ST_ERR_TREE and "root" in _capi.last_error()


def test_argument_errors(lib):
    h = ctypes.c_void_p()
    assert lib.st_tree_create(None, None, 0, 0, 0, ctypes.byref(h)) == _capi.ST_ERR_ARG
    assert lib.st_distances_host(None, None, 1, 2, 1, None, None, None) == _capi.ST_ERR_ARG
    assert lib.st_tree_info_get(None, None) == _capi.ST_ERR_ARG
    lib.st_tree_destroy(None)   # no-op


def test_abi_version_marker_and_sized_info(lib):
    """The header's ST_API_VERSION is what the library answers and what the ctypes binding expects; the sized info getter and
    the probe diagnostic refuse bad arguments (no GPU needed for that)."""
    import re
    from conftest import ROOT
    header = open(os.path.join(ROOT, "include", "suchtree_hip.h")).read()
    declared = int(re.search(r"#define\s+ST_API_VERSION\s+(\d+)", header).group(1))
    assert lib.st_api_version() == declared == _capi.API_VERSION
    buf = ctypes.create_string_buffer(256)
    assert lib.st_tree_info_get_sized(None, buf, 256) == _capi.ST_ERR_ARG
    assert lib.st_probe_last_choice(None, None, None) == _capi.ST_ERR_ARG
    # the binding's struct is the header's: same fields, same order (a reordered or missing field would shift every value)
    body = re.search(r"typedef struct st_tree_info \{(.*?)\} st_tree_info;", header, re.S).group(1)
    fields = re.findall(r"\b(?:int32_t|int64_t)\s+(\w+);", body)
    assert fields == [name for name, _ in _capi.TreeInfo._fields_], (fields, [n for n, _ in _capi.TreeInfo._fields_])
    sizes = {"int32_t": 4, "int64_t": 8}
    assert ctypes.sizeof(_capi.TreeInfo) == sum(sizes[t] for t in re.findall(r"\b(int32_t|int64_t)\s+\w+;", body))


def test_without_gpu_create_fails_cleanly(lib):
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    parent, dist = synth.balanced_tree(4)
    with pytest.raises(_capi.HipBackendError):
        _capi.DeviceTree(parent, dist)
    with pytest.raises(_capi.TreeStructureError):
        _capi.DeviceTree(np.array([1, 2, 0], dtype=np.int32), np.zeros(3, dtype=np.float32))
