"""The SuchTree class surface for the bulk distance / MRCA path.

Host-side mirror of the reference's extension type
(/root/reference/SuchTree/MuchTree.pyx:89-2518): same constructor dispatch,
property names (and deprecated aliases), method names, argument conventions,
return types, exceptions and warnings for everything on the hot path; the
reference's navigation methods around it (node tests, lineages, traversals,
bipartitions, RED) come from navigate.py.  All arithmetic happens in
libsuchtree_hip.so on the GPU; nothing here computes a distance or an MRCA on
the CPU, and a missing library or GPU raises ``HipBackendError``.
"""
import os
from itertools import chain
from numbers import Integral
from typing import Dict, List, Tuple, Union
from urllib.parse import urlparse
from warnings import warn

import numpy as np

from . import _capi
from .exceptions import InvalidNodeError, NodeNotFoundError
from .navigate import TreeNavigation
from .newick import EPSILON, FlatTree, flat_tree_from_arrays, flat_tree_from_newick

_NAMES_EXT = False      # not looked for yet


def _names_ext():
    """The optional _names extension (host-side name -> id loop of distances_by_name), or None."""
    global _NAMES_EXT
    if _NAMES_EXT is False:
        try:
            from . import _names
            _NAMES_EXT = _names
        except ImportError:
            try:                                    # not built yet: a one-second gcc job
                from . import build as _build
                _build.build_names_ext()
                from . import _names
                _NAMES_EXT = _names
            except Exception:                       # noqa: BLE001 -- no compiler, read-only tree, ...
                _NAMES_EXT = None
    return _NAMES_EXT


def _deprecation_warning(old_name: str, new_name: str, version: str = "2.0") -> None:
    # wording of MuchTree.pyx:33-42
    warn(
        f"{old_name} is deprecated and will be removed in SuchTree {version}. "
        f"Use {new_name} instead.",
        DeprecationWarning,
        stacklevel=3,
    )


class SuchTree(TreeNavigation):
    """Immutable phylogenetic tree resident in GPU memory.

    ``tree_input`` follows the reference (MuchTree.pyx:138-155): a URL, a Newick
    string, or a path to a Newick file.  As an extension it may also be a
    :class:`~suchtree_amd.newick.FlatTree` or a ``(parent, distance[, leaf_names])``
    tuple of flat arrays in the reference's in-order numbering.

    ``device``: HIP device index the tree is uploaded to (on first use).
    ``devices``: list of HIP device indices instead: the tree is replicated on all of them and
    the bulk methods taking host arrays (``distances_bulk``, ``pairwise_distances``, ...) deal
    their work over every listed GPU from this one process.
    ``strategy``: ``'auto'`` | ``'canopy'`` | ``'walk'`` kernel family.
    """

    def __init__(self, tree_input, device: int = 0, strategy: str = "auto", devices=None, table_mb=None):
        self._epsilon = EPSILON
        if isinstance(tree_input, FlatTree):
            flat = tree_input
        elif isinstance(tree_input, tuple):
            flat = flat_tree_from_arrays(*tree_input)
        elif isinstance(tree_input, str):
            if urlparse(tree_input).scheme in ("http", "https", "ftp"):
                from urllib.request import urlopen
                with urlopen(tree_input) as fh:
                    text = fh.read().decode("utf-8")
            elif all(["(" in tree_input,
                      ")" in tree_input,
                      tree_input.count("(") == tree_input.count(")"),
                      tree_input.endswith(";")]):
                text = tree_input
            else:
                with open(tree_input) as fh:
                    text = fh.read()
            flat = flat_tree_from_newick(text)
        else:
            raise TypeError("tree_input must be a str, FlatTree or (parent, distance) tuple")
        self._flat = flat
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and not self._devices:
            raise ValueError("devices must not be empty")
        self._device = int(device) if self._devices is None else self._devices[0]
        if strategy not in _capi.STRATEGY:
            raise ValueError("strategy must be one of %s" % sorted(_capi.STRATEGY))
        self._strategy = strategy
        self._table_mb = table_mb      # budget (MiB) of the device tables: _capi.DeviceTree
        self._dev_tree = None

    # ------------------------------------------------------------------ device
    def _device_tree(self) -> "_capi.DeviceTree":
        """Upload on first use; raises HipBackendError without a usable GPU."""
        if self._dev_tree is not None and self._dev_tree._pid != os.getpid():
            # inherited through fork: the handle belongs to the parent's GPU context.  Drop it
            # (without destroying it); the upload below raises the explanatory HipBackendError
            # if the parent had initialised the GPU, which a resident tree implies.
            self._dev_tree = None
        if self._dev_tree is None:
            self._dev_tree = _capi.DeviceTree(self._flat.parent, self._flat.distance,
                                              device=self._device, strategy=self._strategy,
                                              devices=self._devices, table_mb=self._table_mb)
        return self._dev_tree

    def to_device(self) -> "SuchTree":
        """Force the upload now (otherwise it happens at the first query)."""
        self._device_tree()
        return self

    def device_info(self) -> dict:
        """Kernel family, canopy / understory geometry and HBM footprint."""
        return self._device_tree().info()

    def close(self) -> None:
        """Release the GPU copy (it is re-created on the next query)."""
        if self._dev_tree is not None:
            self._dev_tree.close()
            self._dev_tree = None

    # trees travel between processes (spawn / pickle) as their flat arrays; every process uploads
    # its own copy.  The reference's users parallelise with fork pools
    # (docs/examples/SuchTree_examples.md:462-497): that works here as long as the pool forks
    # before the parent's first GPU query (uploads are lazy); a child forked later gets a
    # HipBackendError saying so instead of a hang (see _capi._check_fork).
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_dev_tree"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._dev_tree = None

    def __repr__(self) -> str:
        return "<SuchTree %d nodes, %d leaves, depth %d, device %d%s>" % (
            self.size, self.num_leaves, self.depth, self._device,
            "" if self._dev_tree is None else ", resident")

    # -------------------------------------------------------------- properties
    @property
    def size(self) -> int:
        """The number of nodes in the tree."""
        return self._flat.size

    @property
    def depth(self) -> int:
        """The maximum depth of the tree (nodes on the longest leaf-to-root path)."""
        return self._flat.depth

    @property
    def num_leaves(self) -> int:
        return self._flat.num_leaves

    @property
    def leaves(self) -> Dict[str, int]:
        """Dictionary mapping leaf names to node IDs."""
        return self._flat.leaves

    @property
    def leaf_nodes(self) -> Dict[int, str]:
        """Dictionary mapping leaf node IDs to names."""
        return self._flat.leaf_nodes

    @property
    def root_node(self) -> int:
        return self._flat.root

    @property
    def internal_nodes(self) -> np.ndarray:
        return self._flat.internal_nodes

    @property
    def all_nodes(self) -> np.ndarray:
        return np.concatenate((np.array(list(self.leaves.values())),
                               np.array(list(self.internal_nodes))))

    @property
    def leaf_node_ids(self) -> np.ndarray:
        return np.array(list(self.leaves.values()))

    @property
    def leaf_names(self) -> list:
        return list(self.leaves.keys())

    @property
    def polytomy_epsilon(self) -> float:
        return self._epsilon

    @polytomy_epsilon.setter
    def polytomy_epsilon(self, new_epsilon: float) -> None:
        # like the reference (MuchTree.pyx:298-301) this does not rewrite stored lengths
        self._epsilon = new_epsilon

    # deprecated aliases (MuchTree.pyx:2374-2414)
    @property
    def length(self) -> int:
        _deprecation_warning("length property", "size")
        return self.size

    @property
    def leafs(self) -> dict:
        _deprecation_warning("leafs property", "leaves")
        return self.leaves

    @property
    def leafnodes(self) -> dict:
        _deprecation_warning("leafnodes property", "leaf_nodes")
        return self.leaf_nodes

    @property
    def n_leafs(self) -> int:
        _deprecation_warning("n_leafs property", "num_leaves")
        return self.num_leaves

    @property
    def root(self) -> int:
        _deprecation_warning("root property", "root_node")
        return self.root_node

    @property
    def polytomy_distance(self) -> float:
        _deprecation_warning("polytomy_distance property", "polytomy_epsilon")
        return self.polytomy_epsilon

    @polytomy_distance.setter
    def polytomy_distance(self, value: float) -> None:
        _deprecation_warning("polytomy_distance property", "polytomy_epsilon")
        self.polytomy_epsilon = value

    # ------------------------------------------------------------- validation
    def _validate_node(self, node: Union[int, str]) -> int:
        """MuchTree.pyx:2255-2284."""
        if isinstance(node, str):
            if node not in self.leaves:
                raise NodeNotFoundError(node)
            return self.leaves[node]
        if not isinstance(node, Integral):
            raise TypeError("Node must be int or str, got {t}".format(t=str(type(node))))
        node_id = int(node)
        if node_id < 0 or node_id >= self.size:
            raise InvalidNodeError(node_id, self.size)
        return node_id

    def _validate_node_pair(self, a, b) -> Tuple[int, int]:
        return self._validate_node(a), self._validate_node(b)

    # -------------------------------------------------- cheap harness lookups
    def get_parent(self, node: Union[int, str]) -> int:
        return int(self._flat.parent[self._validate_node(node)])

    def get_children(self, node: Union[int, str]) -> Tuple[int, int]:
        i = self._validate_node(node)
        return int(self._flat.left[i]), int(self._flat.right[i])

    def get_support(self, node: Union[int, str]) -> float:
        return float(self._flat.support[self._validate_node(node)])

    def is_leaf(self, node: Union[int, str]) -> bool:
        return bool(self._flat.left[self._validate_node(node)] == -1)

    def get_ancestors(self, node: Union[int, str]):
        i = self._validate_node(node)
        parent = self._flat.parent
        while True:
            p = int(parent[i])
            if p == -1:
                break
            yield p
            i = p

    # ----------------------------------------------------------- the hot path
    def _coerce_pairs(self, pairs) -> np.ndarray:
        """Input conventions of distances_bulk (MuchTree.pyx:889-894)."""
        if not isinstance(pairs, np.ndarray):
            pairs = np.array(pairs, dtype=np.int64)
        if pairs.ndim != 2 or pairs.shape[1] != 2:
            # (a 1-D array raises IndexError here, exactly like the reference's formatting)
            shape = str((pairs.shape[0], pairs.shape[1]))
            raise ValueError("Expected (n, 2) array, got shape {shape}".format(shape=shape))
        if pairs.dtype != np.int64:
            if not np.issubdtype(pairs.dtype, np.integer):
                raise ValueError("Buffer dtype mismatch, expected 'long' but got '%s'" % pairs.dtype.name)
            # more permissive than the reference (which only takes int64): int32 goes to the
            # library as it is, other integer widths are widened
            if pairs.dtype != np.int32:
                pairs = pairs.astype(np.int64)
        if pairs.shape[0] == 0:
            pairs.max()   # the reference fails here: ValueError (zero-size array to reduction ...)
        return pairs

    def distances_bulk(self, pairs) -> np.ndarray:
        """Patristic distances for an (n, 2) array of node-id pairs.

        Stands in for MuchTree.pyx:872-909 + ``_distances`` (:911-943): float64
        array holding the reference's float32 ordered sums.
        Raises ValueError for a wrong shape, InvalidNodeError for an id outside
        ``[0, size)`` (the id reported follows MuchTree.pyx:897-903).
        """
        pairs = self._coerce_pairs(pairs)
        dist, _ = self._device_tree().distances_host(pairs, want_dist=True, want_mrca=False)
        return dist

    def common_ancestors_bulk(self, pairs) -> np.ndarray:
        """MRCA node ids (int32) for an (n, 2) array of node-id pairs.

        The reference has no bulk MRCA call; this equals a loop over
        ``common_ancestor`` (MuchTree.pyx:1128-1149) and uses the same kernels.
        """
        pairs = self._coerce_pairs(pairs)
        _, mrca = self._device_tree().distances_host(pairs, want_dist=False, want_mrca=True)
        return mrca

    def distances_and_ancestors_bulk(self, pairs) -> Tuple[np.ndarray, np.ndarray]:
        """(distances float64[n], MRCA ids int32[n]) from one kernel launch."""
        pairs = self._coerce_pairs(pairs)
        return self._device_tree().distances_host(pairs, want_dist=True, want_mrca=True)

    def distance(self, a: Union[int, str], b: Union[int, str]) -> float:
        """Patristic distance between two nodes (MuchTree.pyx:852-870, 981-997)."""
        node_a, node_b = self._validate_node_pair(a, b)
        pairs = np.array([[node_a, node_b]], dtype=np.int64)
        dist, _ = self._device_tree().distances_host(pairs, want_dist=True, want_mrca=False)
        return float(dist[0])

    def distances_by_name(self, pairs: List[Tuple[str, str]]) -> List[float]:
        """Distances for (leaf_name, leaf_name) tuples (MuchTree.pyx:945-979)."""
        if not isinstance(pairs, list):
            raise TypeError("pairs must be a list of tuples")
        leaves = self.leaves
        names_ext = _names_ext()
        if names_ext is not None and pairs:
            # fastest path: the lookup loop in C (csrc/names_ext.c); -1 = something unexpected,
            # left to the code below
            ids = np.empty((len(pairs), 2), dtype=np.int64)
            if names_ext.lookup_pairs(pairs, leaves, ids) == 0:
                return self.distances_bulk(ids).tolist()
        try:
            # fast path: the dict lookups run at C speed over the flattened names, no per-element
            # checks (keys are str only, so anything that is not a known leaf name raises here
            # and is diagnosed by the loop below, which owns the error messages)
            if set(map(len, pairs)) - {2}:
                raise ValueError("a pair does not have two elements")
            ids = np.fromiter(map(leaves.__getitem__, chain.from_iterable(pairs)),
                              dtype=np.int64, count=2 * len(pairs))
        except (KeyError, TypeError, ValueError):
            ids = None
        if ids is not None:
            # (an empty list becomes the 1-D empty array of the reference and fails the shape check)
            return self.distances_bulk(ids.reshape(-1, 2) if len(pairs) else ids).tolist()
        node_pairs = []
        for i, (name_a, name_b) in enumerate(pairs):
            if not isinstance(name_a, str) or not isinstance(name_b, str):
                raise TypeError("Pair {i}: both elements must be strings".format(i=str(i)))
            if name_a not in leaves:
                raise NodeNotFoundError(name_a)
            if name_b not in leaves:
                raise NodeNotFoundError(name_b)
            node_pairs.append((leaves[name_a], leaves[name_b]))
        pairs_array = np.array(node_pairs, dtype=np.int64)
        return self.distances_bulk(pairs_array).tolist()

    def common_ancestor(self, a: Union[int, str], b: Union[int, str]) -> int:
        """Most recent common ancestor of two nodes (MuchTree.pyx:1128-1149)."""
        node_a, node_b = self._validate_node_pair(a, b)
        pairs = np.array([[node_a, node_b]], dtype=np.int64)
        _, mrca = self._device_tree().distances_host(pairs, want_dist=False, want_mrca=True)
        return int(mrca[0])

    # ------------------------------------------- callers of the path (all-pairs, kNN)
    def _node_ids(self, nodes) -> np.ndarray:
        if nodes is None:
            return self.leaf_node_ids
        return np.array([self._validate_node(node) for node in nodes])

    def pairwise_distances(self, nodes: List[Union[int, str]] = None) -> np.ndarray:
        """Symmetric matrix of all pairwise distances (MuchTree.pyx:1084-1124).

        The reference materialises the n(n-1)/2 pairs as a Python list, calls
        ``distances_bulk`` and scatters the result into the matrix in a Python loop.  Here the
        whole (n, n) matrix is generated on the device (``st_grid_host``, symmetric grid) and
        streamed straight into the result array: entry [i, j] with i < j is d(ids[i], ids[j])
        in that argument order, exactly as the reference computes it, entry [j, i] is the same
        value, the diagonal is d(x, x) = 0.
        """
        node_ids = self._node_ids(nodes)
        n = len(node_ids)
        distance_matrix = np.zeros((n, n), dtype=float)
        if n > 1:
            ids = np.asarray(node_ids, dtype=np.int64)
            self._device_tree().grid_host(ids, ids, symmetric=True, out_dist=distance_matrix.reshape(-1))
        return distance_matrix

    def distance_matrix(self, nodes: list = None) -> dict:
        """MuchTree.pyx:1919-1956."""
        if nodes is None:
            node_ids = self.leaf_node_ids
            node_names = [self.leaf_nodes[nid] for nid in node_ids]
        else:
            node_ids = np.array([self._validate_node(node) for node in nodes])
            node_names = [self.leaf_nodes[int(i)] if self.is_leaf(int(i)) else f"node_{i}" for i in node_ids]
        return {"distance_matrix": self.pairwise_distances(nodes), "node_ids": node_ids,
                "node_names": node_names}

    def nearest_neighbors(self, node: Union[int, str], k: int = 1, from_nodes: list = None) -> list:
        """k nearest neighbours of a node (MuchTree.pyx:1032-1082)."""
        if k <= 0:
            raise ValueError("k must be positive")
        query_node_id = self._validate_node(node)
        if from_nodes is None:
            if self.is_leaf(query_node_id):
                from_node_ids = [nid for nid in self.leaf_node_ids if nid != query_node_id]
            else:
                from_node_ids = self.leaf_node_ids
            from_nodes_orig = [self.leaf_nodes[nid] for nid in from_node_ids]
        else:
            from_node_ids = [self._validate_node(n) for n in from_nodes]
            from_nodes_orig = from_nodes.copy()
        if len(from_node_ids) == 0:
            # the reference hands np.array([], dtype=int64) to distances_bulk here, whose shape check
            # raises (pyx:1072, 892-894): same call, same exception
            self.distances_bulk(np.array([], dtype=np.int64))
        cands = np.asarray(from_node_ids, dtype=np.int64)
        dev = self._device_tree()
        if k <= dev.KNN_MAX_K:
            # distances and the selection of the k smallest both happen on the GPU (st_knn_host);
            # ties go to the candidate listed first (the reference's argsort leaves them unspecified)
            idx, dist = dev.knn_host(np.array([query_node_id], dtype=np.int64), cands, k)
            return [(from_nodes_orig[int(i)], d) for i, d in zip(idx[0], dist[0]) if i >= 0]
        distances, _ = dev.grid_host(np.array([query_node_id], dtype=np.int64), cands)
        sorted_indices = np.argsort(distances, kind="stable")
        return [(from_nodes_orig[i], distances[i]) for i in sorted_indices[:k]]

    def nearest_neighbors_bulk(self, nodes, k: int = 1, from_nodes: list = None):
        """``nearest_neighbors`` for many query nodes in one call (no reference counterpart):
        returns ``(neighbor_ids int64 (q, k), distances float64 (q, k))``, neighbours ascending
        by distance, -1 / NaN where a query has fewer than k candidates.  With the default
        candidate set (all leaves) a leaf query is not its own neighbour, as in
        ``nearest_neighbors`` (MuchTree.pyx:1058-1062)."""
        if k <= 0:
            raise ValueError("k must be positive")
        queries = np.array([self._validate_node(n) for n in nodes], dtype=np.int64)
        if from_nodes is None:
            cands, skip_self = np.asarray(self.leaf_node_ids, dtype=np.int64), True
        else:
            cands, skip_self = np.array([self._validate_node(n) for n in from_nodes], dtype=np.int64), False
        dev = self._device_tree()
        if k > dev.KNN_MAX_K:
            raise ValueError("k must be at most %d" % dev.KNN_MAX_K)
        idx, dist = dev.knn_host(queries, cands, k, skip_self=skip_self)
        ids = np.where(idx >= 0, cands[np.maximum(idx, 0)], -1) if len(cands) else idx
        return ids, dist

    # --------------------------------------------------- quartet topologies (MRCA caller)
    def quartet_topologies_bulk(self, quartets) -> np.ndarray:
        """(n, 4) node ids re-ordered so columns (0,1) and (2,3) are sisters (MuchTree.pyx:1271-1376)."""
        if not isinstance(quartets, np.ndarray):
            quartets = np.array(quartets, dtype=np.int64)
        if quartets.ndim != 2 or quartets.shape[1] != 4:
            raise ValueError(f"Expected (n, 4) array, got shape {quartets.shape}")
        if quartets.dtype != np.int64:
            if not np.issubdtype(quartets.dtype, np.integer):
                raise ValueError("Buffer dtype mismatch, expected 'long' but got '%s'" % quartets.dtype.name)
            quartets = quartets.astype(np.int64)
        if quartets.shape[0] == 0:
            quartets.max()
        return self._device_tree().quartets_host(quartets)

    def quartet_topology(self, a, b, c, d) -> frozenset:
        """Topology of one quartet as a frozenset of two sister-pair frozensets (MuchTree.pyx:1202-1248)."""
        nodes = [a, b, c, d]
        node_ids = [self._validate_node(node) for node in nodes]
        has_strings = any(isinstance(node, str) for node in nodes)
        w, x, y, z = (int(v) for v in self.quartet_topologies_bulk(np.array([node_ids], dtype=np.int64))[0])
        if has_strings:
            ln = self.leaf_nodes
            return frozenset((frozenset((ln[w], ln[x])), frozenset((ln[y], ln[z]))))
        return frozenset((frozenset((w, x)), frozenset((y, z))))

    def quartet_topologies_by_name(self, quartets) -> list:
        """MuchTree.pyx:1378-1422."""
        quartet_ids = []
        for i, (a, b, c, d) in enumerate(quartets):
            if not all(isinstance(name, str) for name in (a, b, c, d)):
                raise TypeError(f"Quartet {i}: all elements must be strings")
            try:
                quartet_ids.append([self.leaves[a], self.leaves[b], self.leaves[c], self.leaves[d]])
            except KeyError as e:
                raise NodeNotFoundError(str(e).strip("'"))
        topologies = self.quartet_topologies_bulk(np.array(quartet_ids, dtype=np.int64))
        ln = self.leaf_nodes
        return [frozenset((frozenset((ln[a], ln[b])), frozenset((ln[c], ln[d]))))
                for a, b, c, d in topologies.tolist()]

    # deprecated wrappers (MuchTree.pyx:2447-2459)
    def distances(self, pairs):
        _deprecation_warning("distances()", "distances_bulk()")
        return self.distances_bulk(pairs)

    def mrca(self, a: Union[int, str], b: Union[int, str]) -> int:
        _deprecation_warning("mrca()", "common_ancestor()")
        return self.common_ancestor(a, b)

    def get_quartet_topology(self, a, b, c, d):
        _deprecation_warning("get_quartet_topology()", "quartet_topology()")
        return self.quartet_topology(a, b, c, d)

    def quartet_topologies(self, quartets):
        _deprecation_warning("quartet_topologies()", "quartet_topologies_bulk()")
        return self.quartet_topologies_bulk(quartets)
