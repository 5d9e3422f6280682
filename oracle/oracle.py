"""ctypes front-end of the CPU oracle (oracle/suchtree_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of suchtree_oracle.c.  Importable
from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg;
never from ``suchtree_amd``.

Also carries ``py_mrca`` / ``py_distances``: a pure-Python restatement of the
same reference lines (MuchTree.pyx:911-943, 999-1030) used on tiny inputs to
cross-check the C file itself, and ``RefTree``: the REFERENCE's own compiled
``_distances`` / ``_mrca`` (oracle/_ref/libref_hotpath.so, see ref_harness.c),
against which the restatement is checked bit for bit (tests/test_oracle_ref.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

NODE_DTYPE = np.dtype(
    [
        ("parent", np.int32),
        ("left_child", np.int32),
        ("right_child", np.int32),
        ("support", np.float32),
        ("distance", np.float32),
    ],
    align=False,
)
assert NODE_DTYPE.itemsize == 20  # MuchTree.pyx:55-60


def build(force=False):
    """Compile liboracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "suchtree_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.oracle_fill_nodes.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        L.oracle_fill_nodes.restype = None
        L.oracle_depth.argtypes = [vp, i64]
        L.oracle_depth.restype = ctypes.c_uint
        L.oracle_mrca.argtypes = [vp, vp, i32, i32]
        L.oracle_mrca.restype = i32
        L.oracle_distances_n.argtypes = [vp, i64, vp, vp, i64, i64, vp]
        L.oracle_distances_n.restype = None
        L.oracle_distance.argtypes = [vp, vp, i32, i32]
        L.oracle_distance.restype = ctypes.c_float
        L.oracle_mrca_bulk.argtypes = [vp, i64, vp, vp, i64, i64, vp]
        L.oracle_mrca_bulk.restype = None
        L.oracle_linked_pairs.argtypes = [vp, i64, vp, vp]
        L.oracle_linked_pairs.restype = None
        L.oracle_quartets.argtypes = [vp, i64, vp, vp, vp]
        L.oracle_quartets.restype = None
        L.oracle_distances_mt.argtypes = [vp, i64, i64, vp, i64, i64, vp, i32]
        L.oracle_distances_mt.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleTree:
    """The reference's in-memory tree (20-byte AoS) plus its ``depth``."""

    def __init__(self, parent, distance, left=None, right=None, support=None):
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        distance = np.ascontiguousarray(distance, dtype=np.float32)
        n = parent.shape[0]
        if left is None or right is None:
            left, right = children_from_parent(parent)
        left = np.ascontiguousarray(left, dtype=np.int32)
        right = np.ascontiguousarray(right, dtype=np.int32)
        sup = None if support is None else np.ascontiguousarray(support, dtype=np.float32)
        self.size = n
        self.nodes = np.zeros(n, dtype=NODE_DTYPE)
        lib().oracle_fill_nodes(_p(self.nodes), n, _p(parent), _p(left), _p(right),
                                None if sup is None else _p(sup), _p(distance))
        self.depth = int(lib().oracle_depth(_p(self.nodes), n))
        roots = np.flatnonzero(parent == -1)
        self.root = int(roots[0]) if len(roots) else -1

    def _visited(self):
        return np.zeros(max(self.depth, 1) + 1, dtype=np.int64)

    @staticmethod
    def _ids(pairs):
        ids = np.asarray(pairs)
        if ids.dtype != np.int64:
            ids = ids.astype(np.int64)
        assert ids.ndim == 2 and ids.shape[1] == 2
        assert ids.strides[0] % 8 == 0 and ids.strides[1] % 8 == 0
        return ids, ids.strides[0] // 8, ids.strides[1] // 8

    def mrca(self, a, b):
        return int(lib().oracle_mrca(_p(self.nodes), _p(self._visited()), int(a), int(b)))

    def distance(self, a, b):
        return float(lib().oracle_distance(_p(self.nodes), _p(self._visited()), int(a), int(b)))

    def distances(self, pairs):
        ids, s0, s1 = self._ids(pairs)
        out = np.zeros(ids.shape[0], dtype=np.float64)
        if ids.shape[0]:
            lib().oracle_distances_n(_p(self.nodes), ids.shape[0], _p(self._visited()),
                                     _p(ids), s0, s1, _p(out))
        return out

    def mrca_bulk(self, pairs):
        ids, s0, s1 = self._ids(pairs)
        out = np.zeros(ids.shape[0], dtype=np.int32)
        if ids.shape[0]:
            lib().oracle_mrca_bulk(_p(self.nodes), ids.shape[0], _p(self._visited()),
                                   _p(ids), s0, s1, _p(out))
        return out

    def quartets(self, quartets):
        q = np.ascontiguousarray(quartets, dtype=np.int64)
        assert q.ndim == 2 and q.shape[1] == 4
        out = np.zeros_like(q)
        if q.shape[0]:
            lib().oracle_quartets(_p(self.nodes), q.shape[0], _p(self._visited()), _p(q), _p(out))
        return out

    def distances_mt(self, pairs, n_threads):
        ids, s0, s1 = self._ids(pairs)
        out = np.zeros(ids.shape[0], dtype=np.float64)
        rc = lib().oracle_distances_mt(_p(self.nodes), ids.shape[0], self.depth, _p(ids),
                                       s0, s1, _p(out), int(n_threads))
        if rc != 0:
            raise RuntimeError("oracle_distances_mt failed rc=%d" % rc)
        return out


# ---- the reference's own compiled hot path (oracle/_ref/libref_hotpath.so: oracle/ref_harness.c, `make -C oracle ref`) ----
_REF_PATH = os.path.join(_HERE, "_ref", "libref_hotpath.so")
_REFERENCE_C = "/root/reference/SuchTree/MuchTree.c"
_ref = None


def build_ref(force=False):
    """Compile oracle/_ref/libref_hotpath.so from the reference's generated C where it lies (needs /root/reference: this
    container only; the GPU box uses the prebuilt file).  Returns its path, or None when it neither exists nor can be built."""
    if os.path.exists(_REFERENCE_C) and (force or not os.path.exists(_REF_PATH) or
                                         os.path.getmtime(_REF_PATH) < os.path.getmtime(os.path.join(_HERE, "ref_harness.c"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _REF_PATH if os.path.exists(_REF_PATH) else None


def ref_lib():
    """The library, or None when it is not there (nothing under /root/reference and no prebuilt file)."""
    global _ref
    if _ref is None:
        try:
            path = build_ref()
            if path is None:
                return None
            L = ctypes.CDLL(path)
        except (OSError, subprocess.CalledProcessError):      # (unbuildable or unloadable here: callers fall back to the restatement)
            return None
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        L.ref_hotpath_distances.argtypes = [vp, vp, i64, i32, vp, i64, i64, i64, vp, ctypes.c_int]
        L.ref_hotpath_mrca.argtypes = [vp, vp, i64, i32, vp, i64, i64, i64, vp]
        L.ref_hotpath_quartets.argtypes = [vp, vp, i64, i32, vp, i64, vp]
        _ref = L
    return _ref


class RefTree:
    """The REFERENCE's compiled ``SuchTree._distances`` / ``_mrca`` (MuchTree.c as shipped, see ref_harness.c) on flat
    arrays: the same method names as OracleTree, so that either can check the other or the GPU.  ``depth`` is the
    reference's (nodes on the longest leaf-to-root path, MuchTree.pyx:218-225): it sizes the ``visited`` scratch."""

    def __init__(self, parent, distance, depth=None):
        if ref_lib() is None:
            raise RuntimeError("oracle/_ref/libref_hotpath.so is not available (make -C oracle ref needs /root/reference)")
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.distance = np.ascontiguousarray(distance, dtype=np.float32)
        self.size = self.parent.shape[0]
        self.depth = int(depth) if depth is not None else OracleTree(self.parent, self.distance).depth

    def distances(self, pairs, n_threads=1):
        ids = np.asarray(pairs)
        if ids.dtype != np.int64:
            ids = ids.astype(np.int64)
        assert ids.ndim == 2 and ids.shape[1] == 2 and ids.strides[0] >= 0 and ids.strides[1] >= 0
        out = np.zeros(ids.shape[0], dtype=np.float64)
        if ids.shape[0]:
            rc = ref_lib().ref_hotpath_distances(_p(self.parent), _p(self.distance), self.size, self.depth, _p(ids), ids.shape[0],
                                                 ids.strides[0], ids.strides[1], _p(out), int(n_threads))
            if rc != 0:
                raise RuntimeError("ref_hotpath_distances failed rc=%d" % rc)
        return out

    def distances_mt(self, pairs, n_threads):
        return self.distances(pairs, n_threads)

    def quartets(self, quartets):
        q = np.ascontiguousarray(quartets, dtype=np.int64)
        assert q.ndim == 2 and q.shape[1] == 4
        out = np.zeros_like(q)
        if q.shape[0]:
            rc = ref_lib().ref_hotpath_quartets(_p(self.parent), _p(self.distance), self.size, self.depth, _p(q), q.shape[0], _p(out))
            if rc != 0:
                raise RuntimeError("ref_hotpath_quartets failed rc=%d" % rc)
        return out

    def mrca_bulk(self, pairs):
        ids = np.asarray(pairs)
        if ids.dtype != np.int64:
            ids = ids.astype(np.int64)
        assert ids.ndim == 2 and ids.shape[1] == 2 and ids.strides[0] >= 0 and ids.strides[1] >= 0
        out = np.zeros(ids.shape[0], dtype=np.int32)
        if ids.shape[0]:
            rc = ref_lib().ref_hotpath_mrca(_p(self.parent), _p(self.distance), self.size, self.depth, _p(ids), ids.shape[0],
                                            ids.strides[0], ids.strides[1], _p(out))
            if rc != 0:
                raise RuntimeError("ref_hotpath_mrca failed rc=%d" % rc)
        return out


def linked_pairs(linklist):
    """(ids_a, ids_b) of linked_distances(), MuchTree.pyx:2909-2925."""
    ll = np.ascontiguousarray(linklist, dtype=np.int64)
    L = ll.shape[0]
    size = L * (L - 1) // 2
    ids_a = np.zeros((size, 2), dtype=np.int64)
    ids_b = np.zeros((size, 2), dtype=np.int64)
    lib().oracle_linked_pairs(_p(ll), L, _p(ids_a), _p(ids_b))
    return ids_a, ids_b


def children_from_parent(parent):
    """left/right child arrays for an in-order-numbered strictly binary tree:
    the left child has the smaller id (in-order puts the left subtree first)."""
    n = len(parent)
    left = np.full(n, -1, dtype=np.int32)
    right = np.full(n, -1, dtype=np.int32)
    for c in range(n):
        p = int(parent[c])
        if p < 0:
            continue
        if c < p:
            left[p] = c
        else:
            right[p] = c
    return left, right


# ---- pure-Python restatement (tiny inputs only) -----------------------------

def py_mrca(parent, a, b):
    """MuchTree.pyx:999-1030 in plain Python."""
    visited = []
    n = a
    while True:
        visited.append(n)
        n = int(parent[n])
        if n == -1:
            break
    n = b
    while True:
        for v in visited:
            if v == n:
                return v
        n = int(parent[n])
        if n == -1:
            return -1


def py_distances(parent, distance, pairs):
    """MuchTree.pyx:911-943 in plain Python (float32 accumulator)."""
    out = np.zeros(len(pairs), dtype=np.float64)
    dist32 = np.asarray(distance, dtype=np.float32)
    for i, (a, b) in enumerate(pairs):
        a, b = int(a), int(b)
        m = py_mrca(parent, a, b)
        d = np.float32(0)
        n = a
        while n != m:
            d = np.float32(d + dist32[n])
            n = int(parent[n])
        n = b
        while n != m:
            d = np.float32(d + dist32[n])
            n = int(parent[n])
        out[i] = float(d)
    return out


# ---- graph matrices of the two-tree graph (dense, numpy; small inputs only) ----
# Independent restatement of the reference's own dense-block construction; shares no code
# with suchtree_amd.linked (which builds an edge list for the GPU assembly).

def tree_adjacency(parent, left, right, distance, start_node, epsilon):
    """SuchTree.adjacency_matrix, MuchTree.pyx:1750-1813: breadth-first node list of the
    subtree below ``start_node`` and the dense symmetric matrix of its edge lengths
    (float32 branch lengths widened to float64; a zero length becomes ``epsilon``)."""
    to_visit = [int(start_node)]                      # :1777-1788
    for current_id in to_visit:
        if int(left[current_id]) != -1:
            to_visit.append(int(left[current_id]))
            to_visit.append(int(right[current_id]))
    node_ids = np.array(to_visit)
    node_count = len(to_visit)
    adj_matrix = np.zeros((node_count, node_count), dtype=float)   # :1790-1791
    for i in range(node_count):                       # :1794-1811
        node_id = node_ids[i]
        parent_id = int(parent[node_id])
        if parent_id == -1:
            continue
        d = float(np.float32(distance[node_id]))
        if d == 0:
            d += epsilon
        parent_idx = np.where(node_ids == parent_id)[0]
        if len(parent_idx) > 0:
            parent_idx = parent_idx[0]
            adj_matrix[i, parent_idx] = d
            adj_matrix[parent_idx, i] = d
    return adj_matrix, node_ids


def linked_adjacency(tree_a, tree_b, linklist, root_a, root_b, epsilon_a, epsilon_b):
    """SuchLinkedTrees.adjacency with deletions = additions = swaps = 0, MuchTree.pyx:3081-3131.
    ``tree_x`` = (parent, left, right, distance) flat arrays; ``linklist`` rows are
    [TreeB leaf id, TreeA leaf id] (MuchTree.pyx:2862-2874)."""
    ta_aj, ta_ids = tree_adjacency(*tree_a, root_a, epsilon_a)
    tb_aj, tb_ids = tree_adjacency(*tree_b, root_b, epsilon_b)
    ta_node_ids, tb_node_ids = ta_ids.tolist(), tb_ids.tolist()
    ll = np.array(linklist)
    ta_links = [ta_node_ids.index(x) for x in ll[:, 1]]                       # :3106
    tb_links = [tb_node_ids.index(x) + ta_aj.shape[0] for x in ll[:, 0]]      # :3107
    aj = np.zeros((ta_aj.shape[0] + tb_aj.shape[0], ta_aj.shape[1] + tb_aj.shape[1]))   # :3110-3111
    aj[0:ta_aj.shape[0], 0:ta_aj.shape[1]] = ta_aj / ta_aj.max()              # :3114
    aj[ta_aj.shape[0]:, ta_aj.shape[1]:] = tb_aj / tb_aj.max()                # :3115
    ta_mean = np.mean(ta_aj.flatten()[ta_aj.flatten() > epsilon_a])           # :3118
    tb_mean = np.mean(tb_aj.flatten()[tb_aj.flatten() > epsilon_b])           # :3119
    link_mean = (ta_mean / ta_aj.max() + tb_mean / tb_aj.max()) / 2.0         # :3120
    for i, j in zip(tb_links, ta_links):                                      # :3125-3127
        aj[i, j] = link_mean
        aj[j, i] = link_mean
    return aj


def linked_laplacian(aj):
    """SuchLinkedTrees.laplacian, MuchTree.pyx:3133-3145."""
    lp = np.zeros(aj.shape)
    np.fill_diagonal(lp, aj.sum(axis=0))
    lp = lp - aj
    return lp


# ---- callers of the path restated for the navigation tests (tests/test_gpu_navigation.py) -------------------------
def leaves_below(left, right, node):
    """MuchTree.pyx:429-466 (get_leaves): leaf ids below `node` in the order the reference's to_visit list grows."""
    to_visit, out = [int(node)], []
    for current in to_visit:
        if left[current] == -1:
            out.append(current)
        else:
            to_visit.append(int(left[current]))
            to_visit.append(int(right[current]))
    return out


def preorder(left, right, root):
    """MuchTree.pyx:1505-1540 (traverse_preorder)."""
    stack, out = [int(root)], []
    while stack:
        current = stack.pop()
        if right[current] != -1:
            stack.append(int(right[current]))
        if left[current] != -1:
            stack.append(int(left[current]))
        out.append(current)
    return out


def distance_to_root(parent, distance, node):
    """MuchTree.pyx:826-847 (_get_distance_to_root): C float accumulator, ends at the first length equal to -1."""
    d, i = np.float32(0.0), int(node)
    while True:
        d_i = distance[i]
        if d_i == -1:
            break
        d = np.float32(d + d_i)
        i = int(parent[i])
    return float(d)


def relative_evolutionary_divergence(tree, left, right, root):
    """MuchTree.pyx:303-330, statement by statement: RED[root] = 0, then in pre-order
    P + (a / (a + b)) * (1 - P) with a = distance(node, parent), b = np.mean of the list of distance(node, leaf) over
    get_leafs(node).  `tree` is an OracleTree (its distances are the reference's float32 ordered sums); the per-node
    lists are evaluated with one oracle batch per node instead of one call per leaf -- same values, same order.
    The reference holds no golden for RED: what pins this restatement is that it is the quoted statements over an
    oracle pinned elsewhere, and the properties its docstring states (0 at the root, 1 at every leaf)."""
    red = {int(root): 0}
    parent = tree.nodes["parent"]
    for node in preorder(left, right, root)[1:]:
        P = red[int(parent[node])]
        a = tree.distance(node, int(parent[node]))
        leaves = leaves_below(left, right, node)
        pairs = np.stack((np.full(len(leaves), node, dtype=np.int64), np.array(leaves, dtype=np.int64)), axis=1)
        b = np.mean([float(x) for x in tree.distances(pairs)])
        if a + b == 0:
            raise Exception("node {n} : a={a}, b={b}".format(n=node, a=a, b=b))
        red[node] = P + (a / (a + b)) * (1 - P)
    return red


def xorshift64star(state, n):
    """MuchTree.pyx:2937-2949 (_random_int): one draw; returns (new state, draw)."""
    mask = 0xFFFFFFFFFFFFFFFF
    state ^= state >> 12
    state ^= (state << 25) & mask
    state ^= state >> 27
    return state, ((state * 2685821657736338717) & mask) % n


def sample_linked_distances(tree_a, tree_b, linklist, seed, sigma=0.001, buckets=64, n=4096, maxcycles=100):
    """MuchTree.pyx:2951-3079 loop by loop, in the arithmetic of the compiled extension (SuchTree/MuchTree.c:65197-65505:
    doubles for the bucket moments, C floats for the four accumulators over the buckets, pow / powf for squares and
    roots).  `tree_a` / `tree_b` are OracleTrees, `seed` the generator's state.  Returns (result or None, state)."""
    import math
    f32 = np.float32
    sums_a, sums_b = [0.0] * buckets, [0.0] * buckets
    sumsq_a, sumsq_b = [0.0] * buckets, [0.0] * buckets
    samples = [0] * buckets
    dev_a, dev_b = [0.0] * buckets, [0.0] * buckets
    all_a, all_b = [], []
    cycles = 0
    root = lambda x: math.pow(x, 0.5) if x >= 0 else float("nan")      # C pow: NaN, no exception
    while True:
        for i in range(buckets):
            qa, qb = np.zeros((n, 2), dtype=np.int64), np.zeros((n, 2), dtype=np.int64)
            for j in range(n):
                seed, l1 = xorshift64star(seed, len(linklist))
                seed, l2 = xorshift64star(seed, len(linklist))
                qa[j] = (linklist[l1][1], linklist[l2][1])
                qb[j] = (linklist[l1][0], linklist[l2][0])
            da, db = tree_a.distances(qa), tree_b.distances(qb)
            all_a.extend(da.tolist()); all_b.extend(db.tolist())
            for j in range(n):
                sums_a[i] += float(da[j]); sums_b[i] += float(db[j])
                sumsq_a[i] += math.pow(float(da[j]), 2.0); sumsq_b[i] += math.pow(float(db[j]), 2.0)
            samples[i] += n
            dev_a[i] = root(sumsq_a[i] / float(samples[i]) - math.pow(sums_a[i] / float(samples[i]), 2.0))
            dev_b[i] = root(sumsq_b[i] / float(samples[i]) - math.pow(sums_b[i] / float(samples[i]), 2.0))
        deviation_a = deviation_b = sq_a = sq_b = f32(0.0)
        for i in range(buckets):
            deviation_a = f32(float(deviation_a) + dev_a[i])
            deviation_b = f32(float(deviation_b) + dev_b[i])
            sq_a = f32(float(sq_a) + math.pow(dev_a[i], 2.0))
            sq_b = f32(float(sq_b) + math.pow(dev_b[i], 2.0))
        ma, mb = f32(deviation_a / f32(buckets)), f32(deviation_b / f32(buckets))
        deviation_a = f32(root(float(f32(f32(sq_a / f32(buckets)) - f32(ma * ma)))))
        deviation_b = f32(root(float(f32(f32(sq_b / f32(buckets)) - f32(mb * mb)))))
        cycles += 1
        if deviation_a < sigma and deviation_b < sigma:
            break
        if cycles >= maxcycles:
            return None, seed
    L = len(linklist)
    return {"TreeA": np.array(all_a), "TreeB": np.array(all_b), "n_pairs": (L * (L - 1)) / 2, "n_samples": n * buckets * cycles,
            "deviation_a": float(deviation_a), "deviation_b": float(deviation_b)}, seed
