"""SuchLinkedTrees: two trees and the links between their leaves.

Host-side mirror of the parts of the reference's ``SuchLinkedTrees``
(/root/reference/SuchTree/MuchTree.pyx:2525-3208) that feed or call the bulk
distance path: the link table and link list (:2560-2874), subsetting
(:2876-2898), ``linked_distances`` (:2900-2934) and the adjacency / Laplacian
assembly (:3081-3145).  The O(L^2) pair enumeration of ``linked_distances``
never exists in memory here: the kernels derive each pair from its index
(``st_triangle_host``).  ``sample_linked_distances`` keeps the reference's algorithm and
stop rule with numpy's generator in place of its xorshift64*; the igraph export is out of scope.
"""
from typing import Dict

import numpy as np

from . import _capi
from .suchtree import SuchTree


class SuchLinkedTrees:
    def __init__(self, tree_a, tree_b, link_matrix):
        # trees: Newick path / string / URL, or existing SuchTree objects (pyx:2592-2609)
        if isinstance(tree_a, str):
            self._tree_a = SuchTree(tree_a)
        elif type(tree_a) == SuchTree:
            self._tree_a = tree_a
        else:
            raise Exception("unknown input for tree", type(tree_a))
        if isinstance(tree_b, str):
            self._tree_b = SuchTree(tree_b)
        elif type(tree_b) == SuchTree:
            self._tree_b = tree_b
        else:
            raise Exception("unknown input for tree", type(tree_b))
        A, B = self._tree_a, self._tree_b

        # the link matrix (a pandas DataFrame: rows = TreeA leaves, columns = TreeB leaves)
        if not link_matrix.shape == (A.num_leaves, B.num_leaves):
            raise Exception("link_matrix shape must match tree leaf counts")
        if not set(link_matrix.axes[0]) == set(A.leaves.keys()):
            raise Exception("axis[0] does not match TreeA leaf names")
        if not set(link_matrix.axes[1]) == set(B.leaves.keys()):
            raise Exception("axis[1] does not match TreeB leaf names")

        self._row_ids = np.array(list(A.leaves.values()))
        self._col_ids = np.array(list(B.leaves.values()))
        self._row_names = list(A.leaves.keys())
        self._col_names = list(B.leaves.keys())
        self._n_rows = A.num_leaves
        self._n_cols = B.num_leaves

        # reverse map for row ids (pyx:2629-2632)
        self._row_map = np.zeros(A.size, dtype=int)
        for n, i in enumerate(self._row_ids):
            self._row_map[i] = n
        # leaf id -> link-table column (the reference stores it in the leaf's right_child, pyx:1993-2003)
        self._col_of_leaf_b = {int(leaf): i for i, leaf in enumerate(self._col_ids)}
        self._row_of_leaf_a = {int(leaf): i for i, leaf in enumerate(self._row_ids)}
        for i, leaf in enumerate(self._col_ids):      # TreeB's leaves know their columns (pyx:2639)
            B.link_leaf(int(leaf), i)

        # the link table: per TreeB column (in TreeB leaf order) the linked TreeA leaf ids, in the
        # row order of the DataFrame (pyx:2637-2653)
        values = link_matrix.T.reindex(self._col_names)
        row_leaf = np.array([A.leaves[name] for name in values.columns], dtype=np.int64)
        mask = values.to_numpy() > 0
        self._table = [row_leaf[mask[i]] for i in range(self._n_cols)]
        self._n_links = int(sum(len(c) for c in self._table))

        # by default, the subset is the whole table (pyx:2655-2666)
        self._subset_a_root = A.root_node
        self._subset_b_root = B.root_node
        self._subset_a_size = len(self._row_ids)
        self._subset_b_size = len(self._col_ids)
        self._subset_rows = np.array(range(self._subset_a_size))
        self._subset_columns = np.array(range(self._subset_b_size))
        self._subset_a_leafs = self._row_ids
        self._subset_b_leafs = self._col_ids
        self._np_linklist = np.ndarray((self._n_links, 2), dtype=int)
        self._subset_n_links = 0
        self._seed = int(np.random.randint(0xFFFFFFFFFFFFFFFF >> 1))      # xorshift64* state (pyx:2572)
        self._build_linklist()

    # ------------------------------------------------------------ properties
    TreeA = property(lambda self: self._tree_a)
    TreeB = property(lambda self: self._tree_b)
    n_links = property(lambda self: self._n_links)
    n_cols = property(lambda self: self._n_cols)
    n_rows = property(lambda self: self._n_rows)
    col_ids = property(lambda self: self._col_ids)
    row_ids = property(lambda self: self._row_ids)
    col_names = property(lambda self: self._col_names)
    row_names = property(lambda self: self._row_names)
    subset_columns = property(lambda self: self._subset_columns)
    subset_rows = property(lambda self: self._subset_rows)
    subset_a_leafs = property(lambda self: self._subset_a_leafs)
    subset_b_leafs = property(lambda self: self._subset_b_leafs)
    subset_a_size = property(lambda self: self._subset_a_size)
    subset_b_size = property(lambda self: self._subset_b_size)
    subset_a_root = property(lambda self: self._subset_a_root)
    subset_b_root = property(lambda self: self._subset_b_root)
    subset_n_links = property(lambda self: self._subset_n_links)

    @property
    def linklist(self) -> np.ndarray:
        """(n_links, 2) array: column 0 = TreeB leaf id, column 1 = TreeA leaf id (pyx:2838-2874)."""
        return self._np_linklist[: self._subset_n_links, :]

    @property
    def linkmatrix(self) -> np.ndarray:
        """Boolean link matrix of the current subset (pyx:2812-2836)."""
        table = np.zeros((self._subset_a_size, self._subset_b_size), dtype=bool)
        in_a = set(int(x) for x in self._subset_a_leafs)
        for col in self._subset_columns:
            for m in self._table[col]:
                if int(m) in in_a:
                    table[self._row_map[m], col] = True
        return table

    def _build_linklist(self) -> None:
        """pyx:2846-2874: columns in subset order; within a column the table order, kept when the
        TreeA leaf is in the current row subset."""
        in_a = set(int(x) for x in self._subset_a_leafs)
        k = 0
        for col in self._subset_columns:
            for m in self._table[col]:
                if int(m) in in_a:
                    self._np_linklist[k, 0] = self._col_ids[col]
                    self._np_linklist[k, 1] = m
                    k += 1
        self._subset_n_links = k

    def get_column_leafs(self, col, as_row_ids=False) -> np.ndarray:
        col_id = self._col_names.index(col) if isinstance(col, str) else col
        if col_id > self._n_cols:
            raise Exception("col_id out of bounds", col_id)
        column = np.array(self._table[col_id], dtype=int)
        return self._row_map[column] if as_row_ids else column

    def get_column_links(self, col) -> np.ndarray:
        col_id = self._col_names.index(col) if isinstance(col, str) else col
        if col_id > self._n_cols:
            raise Exception("col_id out of bounds", col_id)
        column = np.zeros(self._n_rows, dtype=bool)
        column[self._row_map[np.array(self._table[col_id], dtype=int)]] = True
        return column

    # ------------------------------------------------------------- subsetting
    @staticmethod
    def _leaves_below(tree: SuchTree, node_id: int) -> np.ndarray:
        """Breadth-first leaf order of the reference's get_leaves (pyx:427-462)."""
        left, right = tree._flat.left, tree._flat.right
        to_visit = [int(node_id)]
        out = []
        for cur in to_visit:
            if left[cur] == -1:
                out.append(cur)
            else:
                to_visit.append(int(left[cur]))
                to_visit.append(int(right[cur]))
        return np.array(out, dtype=int)

    def subset_b(self, node_id) -> None:
        """Subset the link matrix to leaves descended from node_id in TreeB (pyx:2876-2886)."""
        if node_id > self._tree_b.size or node_id < 0:
            raise Exception("Node ID out of bounds.", node_id)
        self._subset_b_leafs = self._leaves_below(self._tree_b, node_id)
        self._subset_columns = np.array([self._col_of_leaf_b[int(x)] for x in self._subset_b_leafs], dtype=int)
        self._subset_b_size = len(self._subset_columns)
        self._subset_b_root = node_id
        self._build_linklist()

    def subset_a(self, node_id) -> None:
        """Subset the link matrix to leaves descended from node_id in TreeA (pyx:2888-2898)."""
        if node_id > self._tree_a.size or node_id < 0:
            raise Exception("Node ID out of bounds.", node_id)
        self._subset_a_leafs = self._leaves_below(self._tree_a, node_id)
        self._subset_rows = np.array([self._row_of_leaf_a[int(x)] for x in self._subset_a_leafs], dtype=int)
        self._subset_a_size = len(self._subset_rows)
        self._subset_a_root = node_id
        self._build_linklist()

    # ----------------------------------------------------- the distance caller
    def linked_distances(self) -> Dict[str, object]:
        """Distances in both trees for all pairs of links (pyx:2900-2934).

        Pair k = i(i-1)/2 + j (j < i) is (link j, link i): in TreeA the leaves
        ``(linklist[j,1], linklist[i,1])``, in TreeB ``(linklist[j,0], linklist[i,0])``.
        The id arrays are returned like the reference does, but the distances do not
        depend on them being materialised: each tree's kernel launch generates its pairs
        from the link-list column.
        """
        ll = np.ascontiguousarray(self.linklist, dtype=np.int64)
        L = ll.shape[0]
        size = (L * (L - 1)) // 2
        d_a, _ = self._tree_a._device_tree().triangle_host(ll[:, 1])
        d_b, _ = self._tree_b._device_tree().triangle_host(ll[:, 0])
        rows, cols = np.tril_indices(L, -1)
        ids_a = np.stack([ll[cols, 1], ll[rows, 1]], axis=1)
        ids_b = np.stack([ll[cols, 0], ll[rows, 0]], axis=1)
        return {"TreeA": d_a, "TreeB": d_b, "ids_A": ids_a, "ids_B": ids_b,
                "n_pairs": size, "n_samples": size, "deviation_a": None, "deviation_b": None}

    def sample_linked_distances(self, sigma=0.001, buckets=64, n=4096, maxcycles=100, seed=None):
        """Monte-Carlo form of :meth:`linked_distances` (pyx:2951-3079): cycles of ``buckets`` x ``n`` random link
        pairs, distances in both trees, until the spread of the per-bucket standard deviations falls below ``sigma``
        in both trees (``None`` after ``maxcycles`` cycles).

        The reference's algorithm in the reference's arithmetic: link pairs from its xorshift64* generator
        (``st_link_sample_pairs``; the generator's state lives in the object as it does there and starts at a random
        value -- ``seed=`` sets it, an extension, so that a run can be repeated), the running sums in doubles
        element by element, the four bucket accumulators in C floats, ``pow`` for squares and roots
        (SuchTree/MuchTree.c:65197-65505).  What differs is the batching: one cycle is one launch per tree
        (``buckets * n`` pairs) instead of ``buckets`` calls of ``n`` pairs.
        """
        import math
        if seed is not None:
            self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        ll = np.ascontiguousarray(self.linklist, dtype=np.int64)
        if ll.shape[0] < 1:
            raise ValueError("no links in the current subset")
        buckets, n = int(buckets), int(n)
        sums_a, sums_b = np.zeros(buckets), np.zeros(buckets)
        sumsq_a, sumsq_b = np.zeros(buckets), np.zeros(buckets)
        samples = 0
        all_a, all_b = [], []
        cycles = 0

        def c_pow_half(x):      # C pow(x, 0.5): NaN for negative x, no exception
            return math.pow(x, 0.5) if x >= 0 else float("nan")

        f32 = np.float32
        while True:
            query_a, query_b, self._seed = _capi.link_sample_pairs(self._seed, ll, buckets * n)
            d_a = self._tree_a.distances_bulk(query_a).reshape(buckets, n)
            d_b = self._tree_b.distances_bulk(query_b).reshape(buckets, n)
            all_a.append(d_a.ravel())
            all_b.append(d_b.ravel())
            # sums[i] += d[i, j]; sumsq[i] += pow(d[i, j], 2.0), element by element onto the running values (pyx:3044-3048)
            _capi.bucket_moments(d_a, sums_a, sumsq_a)
            _capi.bucket_moments(d_b, sums_b, sumsq_b)
            samples += n
            dev_a = [c_pow_half(float(sumsq_a[i]) / float(samples) - math.pow(float(sums_a[i]) / float(samples), 2.0)) for i in range(buckets)]
            dev_b = [c_pow_half(float(sumsq_b[i]) / float(samples) - math.pow(float(sums_b[i]) / float(samples), 2.0)) for i in range(buckets)]
            # C floats: x = (float)((double)x + y)
            acc_a = acc_b = sq_a = sq_b = f32(0)
            for i in range(buckets):
                acc_a = f32(float(acc_a) + dev_a[i])
                acc_b = f32(float(acc_b) + dev_b[i])
                sq_a = f32(float(sq_a) + math.pow(dev_a[i], 2.0))
                sq_b = f32(float(sq_b) + math.pow(dev_b[i], 2.0))
            # (float)pow((double)(sq / (float)buckets - powf(acc / (float)buckets, 2.0f)), 0.5)
            mean_a, mean_b = f32(acc_a / f32(buckets)), f32(acc_b / f32(buckets))
            deviation_a = f32(c_pow_half(float(f32(f32(sq_a / f32(buckets)) - f32(mean_a * mean_a)))))
            deviation_b = f32(c_pow_half(float(f32(f32(sq_b / f32(buckets)) - f32(mean_b * mean_b)))))
            cycles += 1
            if deviation_a < sigma and deviation_b < sigma:
                break
            if cycles >= maxcycles:
                return None
        return {"TreeA": np.concatenate(all_a), "TreeB": np.concatenate(all_b),
                "n_pairs": (self._subset_n_links * (self._subset_n_links - 1)) / 2,
                "n_samples": n * buckets * cycles,
                "deviation_a": float(deviation_a), "deviation_b": float(deviation_b)}

    # ------------------------------------------- adjacency / Laplacian assembly
    @staticmethod
    def _tree_adjacency(tree: SuchTree, from_node):
        """SuchTree.adjacency_matrix (pyx:1750-1813) for the subtree below from_node."""
        r = tree.adjacency_matrix(int(from_node))
        return r["adjacency_matrix"], r["node_ids"]

    def _graph_edges(self, deletions=0, additions=0, swaps=0):
        """Edge list (u, v, weight) and size of the two-tree graph of pyx:3081-3131: tree edges
        normalised by each tree's longest edge, link edges at the mean of the two trees' mean
        (non-epsilon) normalised edge lengths; optional random link perturbations as the reference."""
        ta_aj, ta_ids = self._tree_adjacency(self._tree_a, self._subset_a_root)
        tb_aj, tb_ids = self._tree_adjacency(self._tree_b, self._subset_b_root)
        ta_node_ids, tb_node_ids = ta_ids.tolist(), tb_ids.tolist()
        ll = np.array(self.linklist)
        for _ in range(1, deletions):
            ll = np.delete(ll, np.random.randint(len(ll)), axis=0)
        for _ in range(1, swaps):
            x, y = np.random.choice(range(len(ll)), size=2, replace=False)
            ll[x, 1], ll[y, 1] = ll[y, 1], ll[x, 1]
        for _ in range(1, additions):
            a = np.random.choice(list(self._tree_a.leaves.values()))
            b = np.random.choice(list(self._tree_b.leaves.values()))
            ll = np.concatenate((ll, np.array([[b, a]])), axis=0)
        na, nb = ta_aj.shape[0], tb_aj.shape[0]
        a_index = {x: i for i, x in enumerate(ta_node_ids)}
        b_index = {x: i for i, x in enumerate(tb_node_ids)}
        ta_links = [a_index[int(x)] for x in ll[:, 1]]
        tb_links = [b_index[int(x)] + na for x in ll[:, 0]]
        ta_mean = np.mean(ta_aj.flatten()[ta_aj.flatten() > self._tree_a.polytomy_epsilon])
        tb_mean = np.mean(tb_aj.flatten()[tb_aj.flatten() > self._tree_b.polytomy_epsilon])
        link_mean = (ta_mean / ta_aj.max() + tb_mean / tb_aj.max()) / 2.0
        ua, va = np.nonzero(np.triu(ta_aj))
        ub, vb = np.nonzero(np.triu(tb_aj))
        u = np.concatenate([ua, ub + na, np.array(tb_links, dtype=np.int64)])
        v = np.concatenate([va, vb + na, np.array(ta_links, dtype=np.int64)])
        w = np.concatenate([ta_aj[ua, va] / ta_aj.max(), tb_aj[ub, vb] / tb_aj.max(),
                            np.full(len(ta_links), link_mean)])
        return na + nb, u, v, w

    def _matrices(self, deletions, additions, swaps, on_gpu, want_adjacency, want_laplacian):
        n, u, v, w = self._graph_edges(deletions, additions, swaps)
        if on_gpu:
            from . import _capi
            return _capi.graph_matrices(n, u, v, w, device=self._tree_a._device,
                                        want_adjacency=want_adjacency, want_laplacian=want_laplacian)
        aj = np.zeros((n, n))
        aj[u, v] = w
        aj[v, u] = w
        lp = None
        if want_laplacian:
            lp = np.zeros(aj.shape)
            np.fill_diagonal(lp, aj.sum(axis=0))
            lp = lp - aj
        return (aj if want_adjacency else None), lp

    def adjacency(self, deletions=0, additions=0, swaps=0, on_gpu=True) -> np.ndarray:
        """Graph adjacency matrix of both (subsetted) trees plus the link edges (pyx:3081-3131).
        The dense matrix is assembled on the GPU (``st_graph_matrices_host``); ``on_gpu=False``
        assembles it with numpy exactly as the reference does."""
        return self._matrices(deletions, additions, swaps, on_gpu, True, False)[0]

    def laplacian(self, deletions=0, additions=0, swaps=0, on_gpu=True) -> np.ndarray:
        """Graph Laplacian L = D - A of the current subset (pyx:3133-3145)."""
        return self._matrices(deletions, additions, swaps, on_gpu, False, True)[1]

    def spectrum(self, deletions=0, additions=0, swaps=0, on_gpu=True) -> np.ndarray:
        """Eigenvalues of the Laplacian by LAPACK's dsyev, as the reference calls it (pyx:3147-3173: jobz 'N', uplo 'U',
        workspace (4 + 2) N; scipy's LAPACK, the library the reference's `cython_lapack.dsyev` binds); the LAPACK info
        code if the solver fails.  The eigen-solve stays on the host."""
        lp = self.laplacian(deletions=deletions, additions=additions, swaps=swaps, on_gpu=on_gpu)
        try:
            from scipy.linalg.lapack import dsyev
        except ImportError:      # no scipy: numpy's symmetric solver (dsyevd), same values to rounding
            return np.linalg.eigvalsh(lp)
        w, _, info = dsyev(lp, compute_v=0, lower=0, lwork=6 * lp.shape[0])
        return w if info == 0 else info

    def to_igraph(self, deletions=0, additions=0, swaps=0):
        """The current subgraph as a weighted, labelled igraph object (pyx:3175-3198); igraph must be installed."""
        try:
            from igraph import ADJ_UNDIRECTED, Graph
        except ImportError:
            raise Exception("igraph package not installed.")
        g = Graph.Weighted_Adjacency(self.adjacency(deletions=deletions, additions=additions, swaps=swaps).tolist(),
                                     mode=ADJ_UNDIRECTED)
        na = len(list(self._tree_a.get_descendants(self._subset_a_root)))
        nb = len(list(self._tree_b.get_descendants(self._subset_b_root)))
        g.vs["color"] = ["#e1e329ff"] * na + ["#24878dff"] * nb
        g.vs["label"] = ["h" + str(i) for i in range(na)] + ["g" + str(i) for i in range(nb)]
        g.vs["tree"] = [0] * na + [1] * nb
        return g

    def dump_table(self) -> None:
        """Print the link matrix, one line per TreeB column (pyx:3200-3208)."""
        for i in range(self._n_cols):
            print("column", i, ":", ",".join(str(int(x)) for x in self._table[i]))
