import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def gopher_flat():
    from suchtree_amd.newick import flat_tree_from_newick
    return flat_tree_from_newick(open(golden_path("test.tree")).read())


@pytest.fixture(scope="session")
def ml_arrays():
    z = np.load(golden_path("ml_tree.npz"))
    return z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)


@pytest.fixture(scope="session")
def nj_arrays():
    z = np.load(golden_path("nj_tree.npz"))
    return z["parent"], z["distance"], z["leaf_ids"].astype(np.int64)


# ---- host emulator of the product's table construction + pair functions (tests/emu) ----
class EmuInfo(ctypes.Structure):
    _fields_ = [("n_leaves", ctypes.c_int64)] + [
        (k, ctypes.c_int32) for k in
        "root depth has_canopy canopy_nodes understory_max record_bytes parity pad".split()]


class Emulator:
    def __init__(self):
        emu_dir = os.path.join(ROOT, "tests", "emu")
        lib = os.path.join(emu_dir, "libst_emu.so")
        srcs = [os.path.join(emu_dir, "emulator.cpp"),
                os.path.join(ROOT, "suchtree_amd", "csrc", "tree_prep.cpp")]
        deps = srcs + [os.path.join(ROOT, "suchtree_amd", "csrc", h) for h in ("tree_prep.h", "pair_math.h")]
        if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-o", lib] + srcs)
        self.lib = ctypes.CDLL(lib)
        self.lib.emu_last_error.restype = ctypes.c_char_p

    def run(self, parent, dist, pairs, strategy, want_dist=True):
        parent = np.ascontiguousarray(parent, np.int32)
        dist = np.ascontiguousarray(dist, np.float32)
        pairs = np.ascontiguousarray(pairs, np.int64)
        n = len(pairs)
        d = np.zeros(n)
        m = np.zeros(n, np.int32)
        info = EmuInfo()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        rc = self.lib.emu_distances(p(parent), p(dist), ctypes.c_int64(len(parent)), {"walk": 1, "canopy": 2}[strategy],
                                    p(pairs), ctypes.c_int64(n), p(d) if want_dist else None, p(m),
                                    ctypes.byref(info))
        if rc != 0:
            raise RuntimeError("emulator rc=%d: %s" % (rc, self.lib.emu_last_error().decode()))
        return d, m, info


@pytest.fixture(scope="session")
def emulator():
    return Emulator()


def oracle_both(parent, dist, pairs, threads=None):
    """(distances, MRCA ids) of the oracle on all host cores: distances by its own pthread entry point, MRCA ids by
    one OracleTree per Python thread over contiguous chunks (the C call releases the GIL; the visited-list scratch
    is per instance).  Test infrastructure: the same functions, only faster on deep trees."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleTree
    pairs = np.asarray(pairs)
    threads = threads or min(32, len(os.sched_getaffinity(0)))      # (hundreds of threads cost more than they save here)
    n = len(pairs)
    k = max(1, min(threads, n // 256))
    want_d = OracleTree(parent, dist).distances_mt(pairs, k)
    bounds = [n * i // k for i in range(k + 1)]
    with ThreadPoolExecutor(k) as ex:
        parts = list(ex.map(lambda i: OracleTree(parent, dist).mrca_bulk(pairs[bounds[i]:bounds[i + 1]]), range(k)))
    return want_d, np.concatenate(parts) if parts else np.zeros(0, np.int32)


def assert_bits_equal(got, want, what="distances"):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    same = got.view(np.int64) == want.view(np.int64)
    if not same.all():
        bad = np.flatnonzero(~same)[:5]
        raise AssertionError("%s differ at %d of %d positions, first %s: got %s want %s"
                             % (what, (~same).sum(), same.size, bad, got[bad], want[bad]))
