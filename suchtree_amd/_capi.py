"""ctypes binding of libsuchtree_hip.so (the C ABI in include/suchtree_hip.h).

Loading fails loudly (``HipBackendError``) when the library has not been built:
there is no CPU fallback anywhere in this package.
"""
import ctypes
import os
import threading

import numpy as np

from .exceptions import HipBackendError, InvalidNodeError, TreeStructureError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsuchtree_hip.so")

ST_OK, ST_ERR_ARG, ST_ERR_HIP, ST_ERR_BOUNDS, ST_ERR_NOMEM, ST_ERR_TREE, ST_ERR_MEASURE_ONLY = 0, 1, 2, 3, 4, 5, 6
STRATEGY = {"auto": 0, "walk": 1, "canopy": 2}
STRATEGY_NAME = {v: k for k, v in STRATEGY.items()}
BIG_BATCH_KERNEL = {0: "walk", 1: "canopy", 3: "canopy_sorted", 4: "walk_sorted", 5: "canopy_ladder"}      # ST_KERNEL_*

# ST_TABLE_* bits of st_tree_info.dropped_tables, in the order a table budget drops them
DROPPED_TABLES = ((1, "lineage_len"), (2, "lineage_sum"), (4, "tree_rmq"), (8, "rec_i"), (16, "rec_a4"), (32, "ranks"), (64, "canopy"))


def dropped_table_names(bits):
    return [name for bit, name in DROPPED_TABLES if bits & bit]


class TreeOptions(ctypes.Structure):
    """st_tree_options (include/suchtree_hip.h)."""
    _fields_ = [("table_budget_bytes", ctypes.c_int64), ("reserved", ctypes.c_int64 * 7)]


API_VERSION = 6      # include/suchtree_hip.h: ST_API_VERSION

# every symbol include/suchtree_hip.h declares (tests check the .so exports them all)
SYMBOLS = (
    "st_api_version", "st_tree_info_get_sized", "st_last_error", "st_device_count", "st_tree_create", "st_tree_create_multi", "st_tree_create_ex", "st_host_table_plan", "st_tree_devices",
    "st_host_chunk_plan", "st_host_chunk_owner", "st_tree_destroy", "st_tree_info_get",
    "st_distances_host", "st_distances_host_i32", "st_distances_device", "st_distances_device_f32", "st_distances_device_wire", "st_unpack_mrca24_device", "st_fault_check", "st_probe_last_choice", "st_tree_set_strategy",
    "st_tree_set_option", "st_triangle_device", "st_triangle_host", "st_grid_host", "st_knn_host",
    "st_quartets_host", "st_graph_matrices_host", "st_newick_open", "st_newick_fill", "st_newick_close",
    "st_host_depths", "st_link_sample_pairs", "st_bucket_moments", "st_device_malloc", "st_device_free", "st_memcpy_h2d", "st_memcpy_d2h",
    "st_device_synchronize",
)


class TreeInfo(ctypes.Structure):
    _fields_ = [
        ("n_nodes", ctypes.c_int64),
        ("n_leaves", ctypes.c_int64),
        ("root", ctypes.c_int32),
        ("depth", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("strategy", ctypes.c_int32),
        ("canopy_nodes", ctypes.c_int32),
        ("understory_max", ctypes.c_int32),
        ("record_bytes", ctypes.c_int32),
        ("n_devices", ctypes.c_int32),
        ("device_bytes", ctypes.c_int64),
        ("lineage_entries", ctypes.c_int64),
        ("big_batch_kernel", ctypes.c_int32),
        ("tuned", ctypes.c_int32),
        ("host_wire_bytes_in", ctypes.c_int32),
        ("host_wire_bytes_out", ctypes.c_int32),
        ("a_side_bytes", ctypes.c_int32),
        ("dropped_tables", ctypes.c_int32),
        ("table_budget_bytes", ctypes.c_int64),
        ("b_table_bytes_per_leaf", ctypes.c_int32),
        ("ladder_sums", ctypes.c_int32),
        ("ladder_sums_max_pairs", ctypes.c_int64),
    ]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_}
        d["strategy"] = STRATEGY_NAME.get(d["strategy"], str(d["strategy"]))
        d["big_batch_kernel"] = BIG_BATCH_KERNEL.get(d["big_batch_kernel"], str(d["big_batch_kernel"]))
        d["dropped_tables"] = dropped_table_names(d["dropped_tables"])
        return d


_lib = None
_lock = threading.Lock()
_gpu_pid = None     # pid of the process in which this module first put a tree on a GPU


def _check_fork():
    """HIP cannot be used in a child forked after the parent initialised the GPU runtime (its
    worker threads and its device context do not survive fork).  The reference's documented
    recipe -- module-level trees used from multiprocessing.Pool workers
    (docs/examples/SuchTree_examples.md:462-497) -- therefore works when the parent has not
    touched the GPU before the pool forks (uploads are lazy: each worker uploads at its first
    query), or with the 'spawn' start method (trees pickle as their flat arrays)."""
    if _gpu_pid is not None and _gpu_pid != os.getpid():
        raise HipBackendError(
            "this process was forked after its parent (pid %d) had initialised the GPU; HIP cannot be used "
            "in a forked child. Use multiprocessing.get_context('spawn'), or fork the workers before the "
            "first GPU query (a SuchTree uploads lazily, at its first query in each process)." % _gpu_pid)


def _preload_hip_runtime():
    """Make this library and PyTorch agree on ONE HIP runtime, whichever is imported first.

    PyTorch-ROCm wheels bundle their own libamdhip64.so with the same SONAME as the system
    one.  If this library pulled in /opt/rocm's copy first, a later ``import torch`` would
    mix two ROCm stacks in one process and see no device.  So when torch is installed but
    not yet imported, its bundled runtime is loaded first (RTLD_GLOBAL); libsuchtree_hip.so's
    NEEDED libamdhip64.so.7 then resolves to it, exactly as when torch was imported first.
    """
    import sys
    if "torch" in sys.modules or os.environ.get("SUCHTREE_AMD_SYSTEM_HIP", "0") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass   # fall back to the system runtime through the library's RPATH


def load():
    """Load the HIP library once; raise HipBackendError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        autobuild = os.environ.get("SUCHTREE_AMD_AUTOBUILD", "1") != "0"
        try:
            from . import build as _build
            is_stale = os.path.exists(LIB_PATH) and _build.stale()
        except OSError:      # sources not shipped: nothing to compare with
            is_stale = False
        if (not os.path.exists(LIB_PATH) or is_stale) and autobuild:
            # source checkout without the built library, or sources newer than it (the C ABI's
            # struct layout and argtypes are mirrored by hand below): compile (hipcc, gfx950)
            try:
                from . import build as _build
                _build.build()
            except Exception as e:   # no hipcc / compile error: reported below
                build_error = e
            else:
                build_error = None
        else:
            build_error = None
        if os.path.exists(LIB_PATH) and is_stale and build_error is not None:
            import warnings
            warnings.warn("libsuchtree_hip.so is older than its sources and rebuilding failed (%s); "
                          "loading the stale library" % build_error, RuntimeWarning)
        if not os.path.exists(LIB_PATH):
            if build_error is not None:
                raise HipBackendError("libsuchtree_hip.so is not built and building it failed: %s" % build_error)
            raise HipBackendError(
                "libsuchtree_hip.so is not built (%s). Build it with `python -m suchtree_amd.build` "
                "(hipcc, gfx950); this package has no CPU fallback." % LIB_PATH)
        _preload_hip_runtime()
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise HipBackendError("cannot load %s: %s" % (LIB_PATH, e)) from e
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.st_last_error.argtypes = []
        L.st_last_error.restype = ctypes.c_char_p
        L.st_device_count.argtypes = [ctypes.POINTER(i32)]
        L.st_tree_create.argtypes = [vp, vp, i64, i32, i32, ctypes.POINTER(vp)]
        L.st_tree_create_multi.argtypes = [vp, vp, i64, ctypes.POINTER(i32), i32, i32, ctypes.POINTER(vp)]
        L.st_tree_create_ex.argtypes = [vp, vp, i64, ctypes.POINTER(i32), i32, i32, ctypes.POINTER(TreeOptions), ctypes.POINTER(vp)]
        L.st_host_table_plan.argtypes = [vp, vp, i64, i32, i64, ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.st_tree_devices.argtypes = [vp, ctypes.POINTER(i32), i32, ctypes.POINTER(i32)]
        L.st_host_chunk_plan.argtypes = [i64, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.st_host_chunk_owner.argtypes = [i64, i32, i64, ctypes.POINTER(i32), ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.st_tree_destroy.argtypes = [vp]
        L.st_tree_destroy.restype = None
        L.st_tree_info_get.argtypes = [vp, ctypes.POINTER(TreeInfo)]
        L.st_tree_info_get_sized.argtypes = [vp, vp, i64]
        L.st_api_version.argtypes = []
        if L.st_api_version() != API_VERSION:
            raise HipBackendError("libsuchtree_hip.so speaks ABI version %d, this binding %d (include/suchtree_hip.h: ST_API_VERSION); "
                                  "rebuild the library" % (L.st_api_version(), API_VERSION))
        L.st_distances_host.argtypes = [vp, vp, i64, i64, i64, vp, vp, ctypes.POINTER(i64)]
        L.st_distances_host_i32.argtypes = [vp, vp, i64, i64, i64, vp, vp, ctypes.POINTER(i64)]
        L.st_distances_device.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
        L.st_distances_device_f32.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
        L.st_distances_device_wire.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
        L.st_unpack_mrca24_device.argtypes = [i32, vp, i64, vp, vp]
        L.st_fault_check.argtypes = [vp, vp, ctypes.POINTER(i64)]
        L.st_probe_last_choice.argtypes = [vp, vp, ctypes.POINTER(i32)]
        L.st_tree_set_strategy.argtypes = [vp, i32]
        L.st_tree_set_option.argtypes = [vp, ctypes.c_char_p, i64]
        L.st_triangle_device.argtypes = [vp, vp, i64, i64, i64, i64, vp, vp, vp]
        L.st_triangle_host.argtypes = [vp, vp, i64, i64, i64, i64, vp, vp, ctypes.POINTER(i64)]
        L.st_grid_host.argtypes = [vp, vp, i64, vp, i64, i32, i64, i64, vp, vp, ctypes.POINTER(i64)]
        L.st_knn_host.argtypes = [vp, vp, i64, vp, i64, i32, i32, vp, vp, ctypes.POINTER(i64)]
        L.st_quartets_host.argtypes = [vp, vp, i64, i64, i64, vp, ctypes.POINTER(i64)]
        L.st_graph_matrices_host.argtypes = [i32, i64, i64, vp, vp, vp, vp, vp]
        L.st_newick_open.argtypes = [ctypes.c_char_p, i64, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                     ctypes.POINTER(i64), ctypes.POINTER(i64),
                                     ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        L.st_newick_fill.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.st_newick_close.argtypes = [vp]
        L.st_newick_close.restype = None
        L.st_host_depths.argtypes = [vp, i64, vp, ctypes.POINTER(ctypes.c_int32)]
        L.st_link_sample_pairs.argtypes = [ctypes.POINTER(ctypes.c_uint64), vp, i64, i64, vp, vp]
        L.st_bucket_moments.argtypes = [vp, i64, i64, vp, vp]
        L.st_device_malloc.argtypes = [i32, i64, ctypes.POINTER(vp)]
        L.st_device_free.argtypes = [i32, vp]
        L.st_memcpy_h2d.argtypes = [i32, vp, vp, i64]
        L.st_memcpy_d2h.argtypes = [i32, vp, vp, i64]
        L.st_device_synchronize.argtypes = [i32]
        for name in SYMBOLS:
            if name not in ("st_last_error", "st_tree_destroy", "st_newick_close"):
                getattr(L, name).restype = i32
        _lib = L
    return _lib


class MeasureOnly(HipBackendError):
    """A host-path call made under the handle option ``measure`` (2 / 4) skipped part of the pipeline: its time counts,
    its results do not (bench_legs.host_path_leg)."""


def last_error():
    return load().st_last_error().decode("utf-8", "replace")


def check(rc, tree_size=None, bad_id=None):
    """Map a C return code onto the package's exceptions."""
    if rc == ST_OK:
        return
    msg = last_error()
    if rc == ST_ERR_BOUNDS:
        raise InvalidNodeError(bad_id, tree_size)
    if rc == ST_ERR_TREE:
        raise TreeStructureError(msg)
    if rc == ST_ERR_ARG:
        raise ValueError(msg)
    if rc == ST_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == ST_ERR_MEASURE_ONLY:
        raise MeasureOnly(msg)
    raise HipBackendError(msg)


def graph_matrices(n, u, v, w, device=0, want_adjacency=True, want_laplacian=True):
    """Dense adjacency / Laplacian (n x n float64) of an undirected weighted edge list, on the GPU."""
    L = load()
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    adj = np.empty((n, n), dtype=np.float64) if want_adjacency else None
    lap = np.empty((n, n), dtype=np.float64) if want_laplacian else None
    check(L.st_graph_matrices_host(int(device), int(n), int(len(u)), _ptr(u), _ptr(v), _ptr(w), _ptr(adj), _ptr(lap)))
    return adj, lap


def newick_native(text):
    """Parse with the native ingest; returns a dict of arrays or None when the native
    parser declines (the pure-Python parser then handles -- and diagnoses -- the input)."""
    try:
        L = load()
    except HipBackendError:
        return None
    try:
        raw = text.encode("ascii")
    except UnicodeEncodeError:
        return None
    h = ctypes.c_void_p()
    n, nl, nb = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    root, depth = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = L.st_newick_open(raw, len(raw), ctypes.byref(h), ctypes.byref(n), ctypes.byref(nl), ctypes.byref(nb),
                          ctypes.byref(root), ctypes.byref(depth))
    if rc != ST_OK:
        return None
    try:
        n, nl, nb = n.value, nl.value, nb.value
        out = {k: np.empty(n, dtype=np.int32) for k in ("parent", "left", "right")}
        out["support"] = np.empty(n, dtype=np.float32)
        out["distance"] = np.empty(n, dtype=np.float32)
        out["leaf_ids"] = np.empty(nl, dtype=np.int32)
        names = ctypes.create_string_buffer(max(nb, 1))
        offs = np.empty(nl + 1, dtype=np.int64)
        check(L.st_newick_fill(h, _ptr(out["parent"]), _ptr(out["left"]), _ptr(out["right"]),
                               _ptr(out["support"]), _ptr(out["distance"]), _ptr(out["leaf_ids"]),
                               ctypes.cast(names, ctypes.c_void_p), _ptr(offs)))
        blob = names.raw[:nb].decode("ascii")
        o = offs.tolist()
        out["names"] = [blob[o[i]:o[i + 1]] for i in range(nl)]
        out["root"], out["depth"] = int(root.value), int(depth.value)
        return out
    finally:
        L.st_newick_close(h)


def link_sample_pairs(state, linklist, count):
    """(query_a, query_b, new_state): `count` link-pair draws of the reference's xorshift64* generator
    (st_link_sample_pairs; MuchTree.pyx:2937-2949, 3025-3038).  Host only."""
    ll = np.ascontiguousarray(linklist, dtype=np.int64)
    qa = np.empty((int(count), 2), dtype=np.int64)
    qb = np.empty((int(count), 2), dtype=np.int64)
    st = ctypes.c_uint64(int(state) & 0xFFFFFFFFFFFFFFFF)
    check(load().st_link_sample_pairs(ctypes.byref(st), ll.ctypes.data, int(ll.shape[0]), int(count), qa.ctypes.data, qb.ctypes.data))
    return qa, qb, int(st.value)


def bucket_moments(dist, sums, sumsq):
    """sums[i] += d, sumsq[i] += pow(d, 2.0) over the rows of `dist` (buckets, n), in place (st_bucket_moments)."""
    d = np.ascontiguousarray(dist, dtype=np.float64)
    assert sums.dtype == np.float64 and sumsq.dtype == np.float64 and sums.flags.c_contiguous and sumsq.flags.c_contiguous
    check(load().st_bucket_moments(d.ctypes.data, int(d.shape[0]), int(d.shape[1]), sums.ctypes.data, sumsq.ctypes.data))


def host_chunk_map(n, n_devices):
    """[(device_index, first_pair, n_pairs)] for every pipeline chunk of an n-pair host batch
    dealt over n_devices GPUs (st_host_chunk_plan / st_host_chunk_owner; no GPU needed)."""
    L = load()
    chunk, count = ctypes.c_int64(0), ctypes.c_int64(0)
    check(L.st_host_chunk_plan(int(n), int(n_devices), ctypes.byref(chunk), ctypes.byref(count)))
    out = []
    for c in range(count.value):
        d, first, m = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(L.st_host_chunk_owner(int(n), int(n_devices), c, ctypes.byref(d), ctypes.byref(first), ctypes.byref(m)))
        out.append((int(d.value), int(first.value), int(m.value)))
    return out


class _LentBlock:
    """An ordinary (pageable) block on loan to one numpy array; see RecyclePool."""

    def __init__(self, pool, block, n, dtype):
        self._pool, self._block = pool, block
        self.__array_interface__ = {"data": (block.ctypes.data, False), "shape": (n,), "typestr": np.dtype(dtype).str, "version": 3}

    def __del__(self):
        try:
            self._pool._give_back(self._block)
        except Exception:     # interpreter shutdown
            pass


class RecyclePool:
    """Recycled ordinary memory for LARGE result arrays.

    What a fresh numpy array costs beyond 32 MiB -- where glibc stops recycling freed blocks and
    maps / unmaps every one -- is the kernel zeroing its pages on first touch and tearing them
    down on release: for 5e7 pairs (600 MB of float64 + int32) three times as long as the
    computation (``profiles/host_path_r02.jsonl``: 1.2e9 pairs/s for
    "call, drop the result" against 4.8e9 into arrays that are reused).  Result arrays of that
    size are therefore handed out as views of blocks that come back here when the array and all
    its views are gone, and go out again, resident, with the next call.  The memory is ordinary:
    readable in ``fork()`` children, swappable, and returned to the system by ``trim()``.
    ``SUCHTREE_AMD_RECYCLE_MB`` caps what the pool may hold (default 2048; 0 switches it off);
    beyond the cap callers get plain ``np.empty`` arrays.
    """

    MIN_BYTES = 32 << 20

    def __init__(self, budget_bytes=None):
        if budget_bytes is None:
            budget_bytes = int(os.environ.get("SUCHTREE_AMD_RECYCLE_MB", "2048")) << 20
        self.budget = int(budget_bytes)
        self.total = 0
        self._free = {}
        # re-entrant: the garbage collector may run a lent block's __del__ (-> _give_back) on this
        # very thread while it is inside array() / _give_back()
        self._lock = threading.RLock()

    @staticmethod
    def _size_class(nbytes):
        cap = 1 << 21
        while cap < nbytes and cap < (1 << 26):
            cap <<= 1
        if cap < nbytes:                                   # beyond 64 MiB: multiples of 64 MiB
            cap = (nbytes + (1 << 26) - 1) >> 26 << 26
        return cap

    def array(self, n, dtype):
        """A 1-D array of ``n`` items in a recycled block, or None (too small, over budget)."""
        nbytes = int(n) * np.dtype(dtype).itemsize
        if nbytes < self.MIN_BYTES or self.budget <= 0:
            return None
        cap = self._size_class(nbytes)
        with self._lock:
            stack = self._free.get(cap)
            block = stack.pop() if stack else None
            if block is None:
                if self.total + cap > self.budget:
                    return None
                self.total += cap
        if block is None:
            block = np.empty(cap, dtype=np.uint8)
        return np.asarray(_LentBlock(self, block, int(n), dtype))

    def _give_back(self, block):
        with self._lock:
            self._free.setdefault(block.nbytes, []).append(block)

    def trim(self):
        """Release every block that is not on loan."""
        with self._lock:
            free, self._free = self._free, {}
            for cap, blocks in free.items():
                self.total -= cap * len(blocks)


_recycle_pool = None


def recycle_pool():
    global _recycle_pool
    if _recycle_pool is None:
        _recycle_pool = RecyclePool()
    return _recycle_pool


def device_count():
    c = ctypes.c_int(0)
    rc = load().st_device_count(ctypes.byref(c))
    if rc != ST_OK:
        return 0
    return int(c.value)


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def host_table_plan(parent, distance, strategy="auto", table_mb=0):
    """What a tree would get on the device under a table budget (MiB; 0 = SUCHTREE_AMD_TABLE_MB if set, else none),
    computed on the host (no GPU): ``{"device_bytes", "dropped_tables", "family"}`` (st_host_table_plan)."""
    L = load()
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    distance = np.ascontiguousarray(distance, dtype=np.float32)
    b, d, f = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
    check(L.st_host_table_plan(_ptr(parent), _ptr(distance), int(parent.shape[0]), STRATEGY[strategy], int(float(table_mb) * (1 << 20)),
                               ctypes.byref(b), ctypes.byref(d), ctypes.byref(f)))
    return {"device_bytes": int(b.value), "dropped_tables": dropped_table_names(int(d.value)), "family": STRATEGY_NAME.get(int(f.value))}


class DeviceTree:
    """Owner of one ``st_tree`` handle: the tree resident in one GPU's HBM, or -- with
    ``devices=[...]`` -- replicated on several GPUs of the node (``st_tree_create_multi``),
    the host-buffer entry points then dealing their chunks over all of them."""

    def __init__(self, parent, distance, device=0, strategy="auto", devices=None, table_mb=None):
        """``table_mb``: budget (MiB) for the tree's device tables (st_tree_options.table_budget_bytes; default: the
        environment's SUCHTREE_AMD_TABLE_MB, else none): accelerator tables are left out, in a stated order, until the
        rest fits -- ``info()["dropped_tables"]`` names them; results are the same bits."""
        global _gpu_pid
        _check_fork()
        L = load()
        self._lib = L
        self._h = ctypes.c_void_p()
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        distance = np.ascontiguousarray(distance, dtype=np.float32)
        if parent.ndim != 1 or parent.shape != distance.shape:
            raise ValueError("parent and distance must be 1-D arrays of equal length")
        if strategy not in STRATEGY:
            raise ValueError("strategy must be one of %s" % sorted(STRATEGY))
        self.size = int(parent.shape[0])
        self._pid = os.getpid()
        if devices is not None or table_mb is not None:
            devices = [int(d) for d in (devices if devices is not None else [device])]
            if not devices:
                raise ValueError("devices must not be empty")
            arr = (ctypes.c_int * len(devices))(*devices)
            opts = TreeOptions()
            if table_mb is not None:
                if table_mb < 0:
                    raise ValueError("table_mb must be >= 0")
                opts.table_budget_bytes = int(float(table_mb) * (1 << 20))
            rc = L.st_tree_create_ex(_ptr(parent), _ptr(distance), self.size, arr, len(devices),
                                     STRATEGY[strategy], ctypes.byref(opts), ctypes.byref(self._h))
            device = devices[0]
        else:
            rc = L.st_tree_create(_ptr(parent), _ptr(distance), self.size, int(device),
                                  STRATEGY[strategy], ctypes.byref(self._h))
            devices = [int(device)]
        check(rc)
        self.device = int(device)
        self.devices = list(devices)
        if _gpu_pid is None:
            _gpu_pid = self._pid

    def close(self):
        h, self._h = self._h, ctypes.c_void_p()
        if h and self._pid == os.getpid():     # a handle inherited through fork is not ours to destroy
            self._lib.st_tree_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise HipBackendError("tree handle is closed")
        if self._pid != os.getpid():
            _check_fork()
            raise HipBackendError("tree handle belongs to another process (pid %d)" % self._pid)
        return self._h

    def info(self):
        ti = TreeInfo()
        check(self._lib.st_tree_info_get_sized(self.handle, ctypes.byref(ti), ctypes.sizeof(ti)))
        return ti.as_dict()

    def set_strategy(self, strategy):
        check(self._lib.st_tree_set_strategy(self.handle, STRATEGY[strategy]))

    def set_option(self, name, value):
        check(self._lib.st_tree_set_option(self.handle, name.encode(), int(value)))

    def _out(self, buf, n, dtype, want):
        if not want:
            return None
        if buf is None:
            arr = recycle_pool().array(n, dtype)
            if arr is not None:
                return arr
            return np.empty(n, dtype=dtype)
        if buf.dtype != dtype or buf.shape != (n,) or not buf.flags.c_contiguous:
            raise ValueError("output buffer must be a contiguous %s array of shape (%d,)" % (np.dtype(dtype).name, n))
        return buf

    def distances_host(self, pairs, want_dist=True, want_mrca=False, out_dist=None, out_mrca=None):
        """pairs: int64 (n,2) ndarray with any strides (multiples of 8 bytes).
        ``out_dist`` / ``out_mrca``: optional preallocated result arrays (reused buffers
        avoid the page-fault cost of fresh memory on every call)."""
        n = int(pairs.shape[0])
        out_d = self._out(out_dist, n, np.float64, want_dist)
        out_m = self._out(out_mrca, n, np.int32, want_mrca)
        if n == 0:
            return out_d, out_m
        item = pairs.dtype.itemsize
        if pairs.dtype not in (np.int64, np.int32):
            raise ValueError("pairs must be int64 or int32")
        if pairs.strides[0] % item or pairs.strides[1] % item:
            pairs = np.ascontiguousarray(pairs)
        s0, s1 = pairs.strides[0] // item, pairs.strides[1] // item
        if s0 < 0 or s1 < 0:   # negative strides: the base pointer is not the lowest address
            pairs = np.ascontiguousarray(pairs)
            s0, s1 = 2, 1
        bad = ctypes.c_int64(0)
        fn = self._lib.st_distances_host if item == 8 else self._lib.st_distances_host_i32
        rc = fn(self.handle, _ptr(pairs), n, s0, s1, _ptr(out_d), _ptr(out_m),
                                         ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
        return out_d, out_m

    def triangle_host(self, ids, k_begin=0, k_count=None, want_dist=True, want_mrca=False,
                      out_dist=None, out_mrca=None):
        """Lower-triangle all-pairs over ``ids`` (1-D int64): pair k = (ids[j], ids[i]),
        k = i(i-1)/2 + j; returns the slice [k_begin, k_begin + k_count)."""
        ids = np.asarray(ids)
        if ids.ndim != 1:
            raise ValueError("ids must be 1-D")
        if ids.dtype != np.int64 or ids.strides[0] % 8 or ids.strides[0] < 0:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
        m = int(ids.shape[0])
        total = m * (m - 1) // 2
        if k_count is None:
            k_count = total - k_begin
        out_d = self._out(out_dist, k_count, np.float64, want_dist)
        out_m = self._out(out_mrca, k_count, np.int32, want_mrca)
        bad = ctypes.c_int64(0)
        rc = self._lib.st_triangle_host(self.handle, _ptr(ids) if m else None, m,
                                        ids.strides[0] // 8 if m else 1, int(k_begin), int(k_count),
                                        _ptr(out_d), _ptr(out_m), ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
        return out_d, out_m

    def grid_host(self, row_ids, col_ids, symmetric=False, e_begin=0, e_count=None, want_dist=True,
                  want_mrca=False, out_dist=None, out_mrca=None):
        """Element e = r * len(col_ids) + c of the grid is the pair (row_ids[r], col_ids[c]);
        ``symmetric`` (same list twice): below the diagonal the mirror image's argument order,
        i.e. the full range is the symmetric matrix of pairwise_distances, flattened."""
        row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        col_ids = np.ascontiguousarray(col_ids, dtype=np.int64)
        if row_ids.ndim != 1 or col_ids.ndim != 1:
            raise ValueError("id lists must be 1-D")
        total = int(row_ids.shape[0]) * int(col_ids.shape[0])
        if e_count is None:
            e_count = total - e_begin
        out_d = self._out(out_dist, e_count, np.float64, want_dist)
        out_m = self._out(out_mrca, e_count, np.int32, want_mrca)
        bad = ctypes.c_int64(0)
        rc = self._lib.st_grid_host(self.handle, _ptr(row_ids) if len(row_ids) else None, len(row_ids),
                                    _ptr(col_ids) if len(col_ids) else None, len(col_ids), int(bool(symmetric)),
                                    int(e_begin), int(e_count), _ptr(out_d), _ptr(out_m), ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
        return out_d, out_m

    KNN_MAX_K = 256

    def knn_host(self, queries, cands, k, skip_self=False):
        """(index int64 (q,k) into cands, dist float64 (q,k)): the k nearest candidates of every
        query, selected on the GPU; -1 / NaN where fewer than k candidates exist."""
        queries = np.ascontiguousarray(queries, dtype=np.int64)
        cands = np.ascontiguousarray(cands, dtype=np.int64)
        if queries.ndim != 1 or cands.ndim != 1:
            raise ValueError("queries and cands must be 1-D")
        q = int(queries.shape[0])
        idx = np.empty((q, int(k)), dtype=np.int64)
        dist = np.empty((q, int(k)), dtype=np.float64)
        bad = ctypes.c_int64(0)
        rc = self._lib.st_knn_host(self.handle, _ptr(queries) if q else None, q, _ptr(cands) if len(cands) else None,
                                   len(cands), int(k), int(bool(skip_self)), _ptr(idx), _ptr(dist), ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
        return idx, dist

    def quartets_host(self, quartets):
        """quartets: int64 (n,4) ndarray (any non-negative strides); returns int64 (n,4)."""
        n = int(quartets.shape[0])
        flat = recycle_pool().array(4 * n, np.int64)      # (large results: recycled blocks)
        out = flat.reshape(n, 4) if flat is not None else np.empty((n, 4), dtype=np.int64)
        if n == 0:
            return out
        if quartets.strides[0] % 8 or quartets.strides[1] % 8 or quartets.strides[0] < 0 or quartets.strides[1] < 0:
            quartets = np.ascontiguousarray(quartets)
        bad = ctypes.c_int64(0)
        rc = self._lib.st_quartets_host(self.handle, _ptr(quartets), n, quartets.strides[0] // 8,
                                        quartets.strides[1] // 8, _ptr(out), ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
        return out

    def triangle_device(self, d_ids, m, k_begin, k_count, d_out_dist=0, d_out_mrca=0, stream=0, id_stride=1):
        rc = self._lib.st_triangle_device(self.handle, ctypes.c_void_p(d_ids), int(m), int(id_stride),
                                          int(k_begin), int(k_count), ctypes.c_void_p(d_out_dist or None),
                                          ctypes.c_void_p(d_out_mrca or None), ctypes.c_void_p(stream or None))
        check(rc)

    def distances_device(self, d_pairs, n, d_out_dist=0, d_out_mrca=0, stream=0, stride0=2, stride1=1,
                         f32=False):
        """Raw device pointers (ints); enqueues on ``stream`` without synchronising.
        ``f32``: d_out_dist is a float32 buffer (the values are float32 sums)."""
        fn = self._lib.st_distances_device_f32 if f32 else self._lib.st_distances_device
        rc = fn(self.handle, ctypes.c_void_p(d_pairs), int(n), int(stride0),
                                           int(stride1), ctypes.c_void_p(d_out_dist or None),
                                           ctypes.c_void_p(d_out_mrca or None),
                                           ctypes.c_void_p(stream or None))
        check(rc)

    def distances_device_wire(self, d_pairs, n, d_out_dist=0, d_out_mrca24=0, stream=0, stride0=2, stride1=1):
        """The wire format of result slices that travel: float32 distances, MRCA ids packed as 24 bits each
        (st_distances_device_wire; trees of fewer than 2^24 nodes)."""
        check(self._lib.st_distances_device_wire(self.handle, ctypes.c_void_p(d_pairs), int(n), int(stride0), int(stride1),
                                                 ctypes.c_void_p(d_out_dist or None), ctypes.c_void_p(d_out_mrca24 or None),
                                                 ctypes.c_void_p(stream or None)))

    def unpack_mrca24_device(self, d_packed, n, d_out_mrca, stream=0):
        check(self._lib.st_unpack_mrca24_device(int(self.devices[0]), ctypes.c_void_p(d_packed), int(n), ctypes.c_void_p(d_out_mrca),
                                                ctypes.c_void_p(stream or None)))

    def probe_last_choice(self, stream=0):
        """0 / 1: the batch probe gave this handle's last probed batch to the scalar ladder / tile-sorted walk kernel; -1: none yet."""
        c = ctypes.c_int32(-1)
        check(self._lib.st_probe_last_choice(self.handle, ctypes.c_void_p(stream or None), ctypes.byref(c)))
        return int(c.value)

    def fault_check(self, stream=0):
        bad = ctypes.c_int64(0)
        rc = self._lib.st_fault_check(self.handle, ctypes.c_void_p(stream or None), ctypes.byref(bad))
        check(rc, tree_size=self.size, bad_id=int(bad.value))
