"""Tree navigation around the distance path: node tests, lineages, subtrees, traversals, bipartitions, and the
relative evolutionary divergence -- the last one a caller of the bulk distance path.

Mixed into :class:`suchtree_amd.SuchTree`.  Everything but `relative_evolutionary_divergence` is host bookkeeping
over the flat arrays (`parent`, `left`, `right`, `distance`) with the reference's names, argument conventions,
orders of enumeration, return types and exceptions (/root/reference/SuchTree/MuchTree.pyx, cited per method).
No patristic distance and no MRCA is computed here: `relative_evolutionary_divergence`, `to_networkx_nodes` and
`relationships` send their pairs through the bulk path, `distance_to_root` and `path_between_nodes` go through
`distance` / `common_ancestor` (the GPU).  (`traverse_with_distances` carries the reference's own running sum of
branch lengths down its traversal -- Python floats, top-down: bookkeeping of that generator, not the path's float32
sums.)
"""
from collections import deque
from typing import Dict

import numpy as np

from .exceptions import InvalidNodeError


def _deprecated(old_name, new_name):
    from .suchtree import _deprecation_warning
    _deprecation_warning(old_name, new_name)


class TreeNavigation:
    """Methods of the reference's SuchTree that walk the node table (`self._flat`)."""

    RED = None      # cache of relative_evolutionary_divergence (MuchTree.pyx:316-318)

    # ------------------------------------------------------------------ validation helpers
    def _is_leaf(self, node_id: int) -> bool:
        return bool(self._flat.left[node_id] == -1)

    def _validate_leaf_node(self, node) -> int:
        """MuchTree.pyx:2297-2324."""
        node_id = self._validate_node(node)
        if not self._is_leaf(node_id):
            raise InvalidNodeError(node_id, message="Node {n} is not a leaf node".format(n=str(node_id)))
        return node_id

    def _validate_internal_node(self, node) -> int:
        """MuchTree.pyx:2326-2348."""
        node_id = self._validate_node(node)
        if self._is_leaf(node_id):
            raise InvalidNodeError(node_id, message="Node {n} is not an internal node".format(n=str(node_id)))
        return node_id

    def _convert_to_leaf_names(self, node_ids) -> list:
        """MuchTree.pyx:2350-2370."""
        names = []
        for node_id in node_ids:
            if not self._is_leaf(node_id):
                raise InvalidNodeError(node_id, message="Node {n} is not a leaf".format(n=str(node_id)))
            names.append(self.leaf_nodes[int(node_id)])
        return names

    def _start_node(self, from_node) -> int:
        return self.root_node if from_node is None else self._validate_node(from_node)

    # ------------------------------------------------------------------ subtrees (breadth-first, left child first)
    def _breadth_first(self, node_id: int):
        """Node ids of the subtree in the order the reference's `to_visit` lists grow (MuchTree.pyx:415-427)."""
        left, right = self._flat.left, self._flat.right
        order = [node_id]
        for current in order:
            l = int(left[current])
            if l != -1:
                order.append(l)
                order.append(int(right[current]))
        return order

    def get_descendants(self, node_id):
        """Generator over the subtree's node ids, the starting node included (MuchTree.pyx:396-427)."""
        yield from self._breadth_first(self._validate_node(node_id))

    def get_leaves(self, node) -> np.ndarray:
        """Leaf ids below a node, in breadth-first order (MuchTree.pyx:429-466)."""
        left = self._flat.left
        return np.array([x for x in self._breadth_first(self._validate_node(node)) if left[x] == -1], dtype=int)

    def get_internal_nodes(self, from_node=-1) -> np.ndarray:
        """Internal node ids from a node (default: the root), breadth-first (MuchTree.pyx:560-587, the definition
        that is in force; the earlier one at :483-521 takes None for the root -- both are accepted)."""
        start = self.root_node if from_node is None or from_node == -1 else self._validate_node(from_node)
        left = self._flat.left
        return np.array([x for x in self._breadth_first(start) if left[x] != -1], dtype=int)

    def get_nodes(self, from_node=-1) -> np.ndarray:
        """All node ids from a node (default: the root), breadth-first (MuchTree.pyx:589-613)."""
        start = self.root_node if from_node is None or from_node == -1 else self._validate_node(from_node)
        return np.array(self._breadth_first(start), dtype=int)

    # ------------------------------------------------------------------ node tests
    def is_internal(self, node) -> bool:
        """MuchTree.pyx:636-655."""
        return not self._is_leaf(self._validate_node(node))

    def is_ancestor(self, ancestor, descendant) -> int:
        """1 if `ancestor` is an ancestor of `descendant`, -1 the other way round, 0 neither (MuchTree.pyx:664-730)."""
        a, b = self._validate_node_pair(ancestor, descendant)
        parent = self._flat.parent
        i = b
        while True:
            n = int(parent[i])
            if n == -1:
                break
            if n == a:
                return 1
            i = n
        i = a
        while True:
            n = int(parent[i])
            if n == -1:
                break
            if n == b:
                return -1
            i = n
        return 0

    def is_descendant(self, descendant, ancestor) -> bool:
        """True if `descendant` lies below `ancestor` (MuchTree.pyx:684-701)."""
        a, b = self._validate_node_pair(ancestor, descendant)
        return self.is_ancestor(a, b) == 1

    def is_root(self, node) -> bool:
        """MuchTree.pyx:732-746."""
        return self._validate_node(node) == self.root_node

    def is_sibling(self, node1, node2) -> bool:
        """MuchTree.pyx:748-772."""
        a, b = self._validate_node_pair(node1, node2)
        if a == self.root_node or b == self.root_node:
            return False
        pa, pb = int(self._flat.parent[a]), int(self._flat.parent[b])
        return pa == pb and pa != -1

    def has_children(self, node) -> bool:
        """MuchTree.pyx:774-788."""
        return self.is_internal(node)

    def has_parent(self, node) -> bool:
        """MuchTree.pyx:790-804."""
        return not self.is_root(node)

    # ------------------------------------------------------------------ one lineage
    def distance_to_root(self, node) -> float:
        """Sum of the branch lengths from a node up to the root, accumulated in float32 from the node upwards
        (MuchTree.pyx:808-847).  The reference's loop ends at the first node whose length equals -1 -- the root's --
        so this is the path kernel's distance(node, that node): a's side of the pair in the same order, b's side empty.
        Finding that node is bookkeeping over the parent table; the sum is taken on the GPU."""
        node_id = self._validate_node(node)
        dist, parent = self._flat.distance, self._flat.parent
        stop = node_id
        while dist[stop] != -1:
            stop = int(parent[stop])
        return self.distance(node_id, stop)

    def distances_to_root_bulk(self, nodes=None) -> np.ndarray:
        """`distance_to_root` of many nodes (default: all, by id) as ONE batch of the path kernel -- an extension:
        the reference's callers (`to_networkx_nodes`, MuchTree.pyx:2066) loop over `distance_to_root`."""
        ids = np.arange(self.size, dtype=np.int64) if nodes is None else np.array([self._validate_node(x) for x in nodes], dtype=np.int64)
        dist, parent = self._flat.distance, self._flat.parent
        stops = np.full(len(ids), self.root_node, dtype=np.int64)
        if np.count_nonzero(dist == -1) > 1:      # a branch of length -1 somewhere: the reference's loop ends there
            for k, x in enumerate(ids):
                stop = int(x)
                while dist[stop] != -1:
                    stop = int(parent[stop])
                stops[k] = stop
        if len(ids) == 0:
            return np.zeros(0)
        return self.distances_bulk(np.stack((ids, stops), axis=1))

    def path_between_nodes(self, a, b) -> list:
        """Node ids from a to b through their common ancestor (MuchTree.pyx:1421-1461)."""
        node_a, node_b = self._validate_node_pair(a, b)
        if node_a == node_b:
            return [node_a]
        mrca = self.common_ancestor(node_a, node_b)
        parent = self._flat.parent
        path_a, current = [], node_a
        while current != mrca:
            path_a.append(current)
            current = int(parent[current])
        path_b, current = [], node_b
        while current != mrca:
            path_b.append(current)
            current = int(parent[current])
        return path_a + [mrca] + list(reversed(path_b))

    # ------------------------------------------------------------------ traversals
    def traverse_inorder(self, include_distances: bool = True):
        """Left subtree, node, right subtree from the root (MuchTree.pyx:1465-1503)."""
        left, right, dist = self._flat.left, self._flat.right, self._flat.distance
        current, stack = self.root_node, []
        while True:
            if current != -1:
                stack.append(current)
                current = int(left[current])
            elif stack:
                current = stack.pop()
                yield (current, float(dist[current])) if include_distances else current
                current = int(right[current])
            else:
                break

    def traverse_preorder(self, from_node=None):
        """Node, left subtree, right subtree (MuchTree.pyx:1505-1540)."""
        left, right = self._flat.left, self._flat.right
        stack = [self._start_node(from_node)]
        while stack:
            current = stack.pop()
            r, l = int(right[current]), int(left[current])
            if r != -1:
                stack.append(r)
            if l != -1:
                stack.append(l)
            yield current

    def traverse_postorder(self, from_node=None):
        """Left subtree, right subtree, node (MuchTree.pyx:1542-1584)."""
        left, right = self._flat.left, self._flat.right
        stack, last_visited, current = [], None, self._start_node(from_node)
        while stack or current != -1:
            if current != -1:
                stack.append(current)
                current = int(left[current])
            else:
                peek = stack[-1]
                r = int(right[peek])
                if r != -1 and last_visited != r:
                    current = r
                else:
                    yield peek
                    last_visited = stack.pop()

    def traverse_levelorder(self, from_node=None):
        """Level by level, left child first (MuchTree.pyx:1586-1620)."""
        left, right = self._flat.left, self._flat.right
        queue = deque([self._start_node(from_node)])
        while queue:
            current = queue.popleft()
            yield current
            l, r = int(left[current]), int(right[current])
            if l != -1:
                queue.append(l)
            if r != -1:
                queue.append(r)

    def traverse_leaves_only(self, from_node=None):
        """Leaves in pre-order (MuchTree.pyx:1622-1646)."""
        for node_id in self.traverse_preorder(self._start_node(from_node)):
            if self._is_leaf(node_id):
                yield node_id

    def traverse_internal_only(self, from_node=None):
        """Internal nodes in pre-order (MuchTree.pyx:1648-1672)."""
        for node_id in self.traverse_preorder(self._start_node(from_node)):
            if not self._is_leaf(node_id):
                yield node_id

    def traverse_with_depth(self, from_node=None):
        """(node, edges below the starting node) in pre-order (MuchTree.pyx:1674-1706)."""
        left, right = self._flat.left, self._flat.right
        stack = [(self._start_node(from_node), 0)]
        while stack:
            current, depth = stack.pop()
            yield (current, depth)
            r, l = int(right[current]), int(left[current])
            if r != -1:
                stack.append((r, depth + 1))
            if l != -1:
                stack.append((l, depth + 1))

    def traverse_with_distances(self, from_node=None):
        """(node, length of its branch, summed lengths of the branches ABOVE it counted from the starting node) in
        pre-order; the sums are Python floats accumulated downwards (MuchTree.pyx:1708-1748)."""
        left, right, dist = self._flat.left, self._flat.right, self._flat.distance
        stack = [(self._start_node(from_node), 0.0)]
        while stack:
            current, dist_to_root = stack.pop()
            dist_to_parent = float(dist[current])
            yield (current, dist_to_parent, dist_to_root)
            next_dist = dist_to_root + (dist_to_parent if dist_to_parent != -1 else 0)
            r, l = int(right[current]), int(left[current])
            if r != -1:
                stack.append((r, next_dist))
            if l != -1:
                stack.append((l, next_dist))

    # ------------------------------------------------------------------ bipartitions
    def bipartition(self, node, by_id: bool = False) -> frozenset:
        """The two leaf sets below the children of an internal node (MuchTree.pyx:1151-1186)."""
        node_id = self._validate_internal_node(node)
        l, r = self.get_children(node_id)
        if by_id:
            return frozenset((frozenset(self.get_leaves(l)), frozenset(self.get_leaves(r))))
        return frozenset((frozenset(self._convert_to_leaf_names(self.get_leaves(l))),
                          frozenset(self._convert_to_leaf_names(self.get_leaves(r)))))

    def bipartitions(self, by_id: bool = False):
        """Generator over the bipartitions of all internal nodes, breadth-first (MuchTree.pyx:1188-1204)."""
        for node_id in self.get_internal_nodes():
            yield self.bipartition(int(node_id), by_id=by_id)

    # ------------------------------------------------------------------ the tree as a graph
    def adjacency_matrix(self, from_node=None) -> dict:
        """Dense symmetric matrix of branch lengths over the subtree's nodes in breadth-first order; a zero length
        becomes the polytomy epsilon (MuchTree.pyx:1750-1813)."""
        node_ids = np.array(self._breadth_first(self._start_node(from_node)), dtype=int)
        index = {int(x): i for i, x in enumerate(node_ids)}
        adj = np.zeros((len(node_ids), len(node_ids)), dtype=float)
        parent, dist = self._flat.parent, self._flat.distance
        for i, x in enumerate(node_ids):
            p = int(parent[x])
            if p == -1 or p not in index:      # the root; the starting node of a subtree
                continue
            d = float(dist[x])
            if d == 0:
                d += self.polytomy_epsilon
            adj[i, index[p]] = d
            adj[index[p], i] = d
        return {"adjacency_matrix": adj, "node_ids": node_ids}

    def laplacian_matrix(self, from_node=None) -> dict:
        """diag(column sums) - adjacency (MuchTree.pyx:1815-1854)."""
        result = self.adjacency_matrix(self._start_node(from_node))
        adj = result["adjacency_matrix"]
        lap = np.zeros(adj.shape, dtype=float)
        np.fill_diagonal(lap, adj.sum(axis=0))
        return {"laplacian": lap - adj, "node_ids": result["node_ids"]}

    def incidence_matrix(self, from_node=None) -> dict:
        """Nodes x edges, +1 at an edge's parent and -1 at its child, edges in breadth-first order of their child
        (MuchTree.pyx:1856-1917).  Like the reference it only works from the root: the edge above the starting node
        of a subtree has no parent row (the reference's `np.where(...)[0][0]` raises IndexError there too)."""
        node_ids = np.array(self._breadth_first(self._start_node(from_node)), dtype=int)
        index = {int(x): i for i, x in enumerate(node_ids)}
        parent = self._flat.parent
        edges = [(int(parent[x]), int(x)) for x in node_ids if parent[x] != -1]
        incidence = np.zeros((len(node_ids), len(edges)), dtype=int)
        for k, (p, c) in enumerate(edges):
            if p not in index:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            incidence[index[p], k] = 1
            incidence[index[c], k] = -1
        return {"incidence_matrix": incidence, "node_ids": node_ids, "edge_list": edges}

    def degree_sequence(self, from_node=None) -> dict:
        """Number of edges at every node of the subtree (MuchTree.pyx:1958-1989)."""
        result = self.adjacency_matrix(from_node)
        degrees = np.sum(result["adjacency_matrix"] > 0, axis=1)
        return {"degrees": degrees, "node_ids": result["node_ids"],
                "max_degree": degrees.max(), "min_degree": degrees.min()}

    # ------------------------------------------------------------------ exports
    def to_networkx_nodes(self, from_node=None):
        """(node id, attributes) for networkx, breadth-first (MuchTree.pyx:2026-2080).  The reference walks every
        node's lineage twice (distance to root, depth); here the distances to the root are one batch of the path
        kernel and the depths one pass over the tree."""
        order = self._breadth_first(self._start_node(from_node))
        to_root = dict(zip(order, self.distances_to_root_bulk(order).tolist()))
        depth = dict(self.traverse_with_depth())
        flat = self._flat
        for node_id in order:
            attributes = {}
            if self._is_leaf(node_id):
                attributes["type"] = "leaf"
                attributes["label"] = self.leaf_nodes[node_id]
            else:
                attributes["type"] = "internal"
                attributes["label"] = f"node_{node_id}"
            support = float(flat.support[node_id])
            if support != -1:
                attributes["support"] = support
            distance = float(flat.distance[node_id])
            if distance != -1:
                attributes["distance_to_parent"] = distance
            attributes["distance_to_root"] = to_root[node_id]
            attributes["depth"] = depth[node_id]
            yield (node_id, attributes)

    def to_networkx_edges(self, from_node=None):
        """(child id, parent id, attributes) for networkx, breadth-first (MuchTree.pyx:2082-2122)."""
        flat = self._flat
        for node_id in self._breadth_first(self._start_node(from_node)):
            parent_id = int(flat.parent[node_id])
            if parent_id == -1:
                continue
            distance = float(flat.distance[node_id])
            attributes = {"weight": distance, "length": distance}
            if not self._is_leaf(node_id):
                support = float(flat.support[node_id])
                if support != -1:
                    attributes["support"] = support
            yield (node_id, parent_id, attributes)

    def to_networkx_graph(self, from_node=None):
        """MuchTree.pyx:2124-2156."""
        try:
            import networkx as nx
        except ImportError:
            raise ImportError("NetworkX is required for to_networkx_graph()")
        G = nx.Graph()
        for node_id, attributes in self.to_networkx_nodes(from_node):
            G.add_node(node_id, **attributes)
        for child_id, parent_id, attributes in self.to_networkx_edges(from_node):
            G.add_edge(child_id, parent_id, **attributes)
        return G

    def to_newick(self, from_node=None, include_support: bool = True, include_distances: bool = True) -> str:
        """Newick text of the tree or a subtree: leaf names, supports behind the closing bracket, lengths as Python
        prints the float32 values (MuchTree.pyx:2180-2229).  Iterative: the reference's recursion stops at Python's
        recursion limit on trees a few hundred levels deep."""
        start = self._start_node(from_node)
        flat = self._flat
        text = {}
        for node_id in self.traverse_postorder(start):
            l, r = int(flat.left[node_id]), int(flat.right[node_id])
            if l == -1:
                result = self.leaf_nodes[node_id]
            else:
                result = "(" + text.pop(l) + "," + text.pop(r) + ")"
                if include_support:
                    support = float(flat.support[node_id])
                    if support != -1:
                        result += str(support)
            if include_distances and node_id != start:
                distance = float(flat.distance[node_id])
                if distance != -1:
                    result += ":" + str(distance)
            text[node_id] = result
        return text[start] + ";"

    # ------------------------------------------------------------------ relative evolutionary divergence
    @property
    def relative_evolutionary_divergence(self) -> Dict[int, float]:
        """RED of every node (MuchTree.pyx:303-330): 0 at the root, then in pre-order P + (a / (a + b)) * (1 - P)
        with P the parent's RED, a = distance(node, parent) and b = the mean of distance(node, leaf) over the leaves
        below the node in `get_leaves` order.

        The reference computes its sum(leaves below every node) distances one Python call at a time; here they are
        ONE `distances_bulk` batch (n - 1 node-parent pairs and, for every node, a pair per leaf below it -- the
        leaves' depths summed: 6e6 pairs on a 54,000-leaf tree 370 levels deep), and the means are taken per node
        with numpy's own `mean` over the same values in the same order, so the result is the reference's bit for bit.
        Cached as `self.RED`, like the reference."""
        if getattr(self, "RED", None):
            return self.RED
        flat = self._flat
        n = self.size
        parent = flat.parent.astype(np.int64)
        left = flat.left
        root = self.root_node
        order = np.fromiter(self.traverse_preorder(), dtype=np.int64, count=n)
        # leaves below every node in get_leaves order = breadth-first = by (level, left to right): the leaves of a
        # subtree are a contiguous run of the left-to-right leaf sequence, stably sorted by depth
        inorder = np.fromiter(self.traverse_inorder(include_distances=False), dtype=np.int64, count=n)
        is_leaf = left[inorder] == -1
        leaf_seq = inorder[is_leaf]                                   # leaves left to right
        leaves_before = np.cumsum(is_leaf) - is_leaf                  # per in-order position
        pos = np.empty(n, dtype=np.int64)
        pos[inorder] = np.arange(n)
        # a subtree's in-order positions are contiguous: [first, last]; found bottom-up over the pre-order
        first, last = pos.copy(), pos.copy()
        for x in order[::-1]:
            p = parent[x]
            if p >= 0:
                if first[x] < first[p]:
                    first[p] = first[x]
                if last[x] > last[p]:
                    last[p] = last[x]
        lo = leaves_before[first]                                     # run of leaf_seq below node x: [lo, hi)
        hi = leaves_before[last] + is_leaf[last]
        depth = np.zeros(n, dtype=np.int64)
        for x in order[1:]:
            depth[x] = depth[parent[x]] + 1
        count = hi - lo
        seg_node = np.repeat(np.arange(n, dtype=np.int64), count)
        within = np.arange(int(count.sum()), dtype=np.int64) - np.repeat(np.cumsum(count) - count, count)
        leaf = leaf_seq[np.repeat(lo, count) + within]
        sort = np.lexsort((within, depth[leaf], seg_node))          # per node: by level, then left to right
        leaf = leaf[sort]
        others = order[1:]
        pairs = np.concatenate((np.stack((others, parent[others]), axis=1),
                                np.stack((seg_node, leaf), axis=1))).astype(np.int64)
        dist = self.distances_bulk(pairs)
        a = np.zeros(n)
        a[others] = dist[:n - 1]
        leaf_dist = dist[n - 1:]
        b = np.zeros(n)
        start = np.cumsum(count) - count
        for c in np.unique(count):      # numpy's mean per run length: a contiguous row is summed as a 1-D array is
            nodes = np.flatnonzero(count == c)
            rows = np.ascontiguousarray(leaf_dist[start[nodes][:, None] + np.arange(c)[None, :]])
            b[nodes] = rows.mean(axis=1)
        red = {root: 0}
        for x in order[1:]:
            x = int(x)
            if a[x] + b[x] == 0:
                raise Exception("node {n} : a={a}, b={b}".format(n=x, a=a[x], b=b[x]))
            P = red[int(parent[x])]
            red[x] = P + (a[x] / (a[x] + b[x])) * (1 - P)
        self.RED = red
        return self.RED

    # ------------------------------------------------------------------ leftovers of the reference's surface
    def relationships(self):
        """pandas DataFrame of all leaf pairs: distance, MRCA, and the distances of a, b and the MRCA to the root and of
        a and b to the MRCA (MuchTree.pyx:2158-2178, the definition that works; the later one at :2515 forwards to a
        method that does not exist).  Every column is one batch of the path kernels where the reference loops."""
        import pandas as pd
        from itertools import combinations
        from random import sample
        pairs = [sample([a, b], 2) for a, b in combinations(self.leaves.keys(), 2)]
        first, second = [p[0] for p in pairs], [p[1] for p in pairs]
        ids = np.array([(self.leaves[a], self.leaves[b]) for a, b in pairs], dtype=np.int64).reshape(-1, 2)
        distances, mrca = (self.distances_and_ancestors_bulk(ids) if len(ids) else (np.zeros(0), np.zeros(0, dtype=np.int32)))
        mrca = [int(m) for m in mrca]
        mrca_to_root = self.distances_to_root_bulk(mrca).tolist()
        a_to_root = self.distances_to_root_bulk(ids[:, 0].tolist()).tolist()
        b_to_root = self.distances_to_root_bulk(ids[:, 1].tolist()).tolist()
        return pd.DataFrame({"a": first, "b": second, "distance": distances.tolist(),
                             "a_to_root": a_to_root, "b_to_root": b_to_root, "mrca": mrca, "mrca_to_root": mrca_to_root,
                             "a_to_mrca": [x - m for x, m in zip(a_to_root, mrca_to_root)],
                             "b_to_mrca": [x - m for x, m in zip(b_to_root, mrca_to_root)]})

    def dump_array(self) -> None:
        """Print the node table (MuchTree.pyx:2231-2240)."""
        f = self._flat
        for n in range(self.size):
            print("id : %d ->" % n)
            print("   distance    : %0.3f" % f.distance[n])
            print("   parent      : %d" % f.parent[n])
            print("   left child  : %d" % f.left[n])
            print("   right child : %d" % f.right[n])

    def link_leaf(self, leaf_id: int, col_id: int) -> None:
        """Attach a leaf to a column of a SuchLinkedTrees link matrix (MuchTree.pyx:1993-2003; the reference keeps the
        column index in the leaf's unused right-child slot, here it is a dict beside the tree)."""
        leaf_id = int(leaf_id)
        if not (0 <= leaf_id < self.size) or not self._is_leaf(leaf_id):
            raise Exception("Cannot link non-leaf node.", leaf_id)
        if leaf_id not in self.leaf_nodes:
            raise Exception("Unknown leaf id.", leaf_id)
        if getattr(self, "_linked_columns", None) is None:
            self._linked_columns = {}
        self._linked_columns[leaf_id] = int(col_id)

    def get_links(self, leaf_ids) -> np.ndarray:
        """Column ids for an array of leaf ids (MuchTree.pyx:2005-2014); -1 for a leaf that was never linked (the
        reference reads the leaf's right child there, which is -1)."""
        if not set(int(x) for x in leaf_ids) <= set(self.leaves.values()):
            raise Exception("Unknown leaf id(s).", leaf_ids)
        cols = getattr(self, "_linked_columns", None) or {}
        return np.array([cols.get(int(x), -1) for x in leaf_ids], dtype=int)

    # ------------------------------------------------------------------ deprecated names (MuchTree.pyx:2416-2497)
    def get_lineage(self, node):
        _deprecated("get_lineage()", "get_ancestors()")
        return self.get_ancestors(node)

    def get_descendant_nodes(self, node):
        _deprecated("get_descendant_nodes()", "get_descendants()")
        return self.get_descendants(node)

    def get_leafs(self, node):
        _deprecated("get_leafs()", "get_leaves()")
        return self.get_leaves(node)

    def is_internal_node(self, node) -> bool:
        _deprecated("is_internal_node()", "is_internal()")
        return self.is_internal(node)

    def get_distance_to_root(self, node) -> float:
        _deprecated("get_distance_to_root()", "distance_to_root()")
        return self.distance_to_root(node)

    def get_bipartition(self, node, by_id: bool = False):
        _deprecated("get_bipartition()", "bipartition()")
        return self.bipartition(node, by_id=by_id)

    def in_order(self, distances: bool = True):
        _deprecated("in_order()", "traverse_inorder()")
        return self.traverse_inorder(include_distances=distances)

    def pre_order(self):
        _deprecated("pre_order()", "traverse_preorder()")
        return self.traverse_preorder()

    def adjacency(self, node: int = -1):
        """(the reference's wrapper, MuchTree.pyx:2499-2502, names a variable it does not have; this one forwards)"""
        _deprecated("adjacency()", "adjacency_matrix()")
        return self.adjacency_matrix(None if node == -1 else node)

    def laplacian(self, node: int = -1):
        _deprecated("laplacian()", "laplacian_matrix()")
        return self.laplacian_matrix(None if node == -1 else node)

    def nodes_data(self):
        _deprecated("nodes_data()", "to_networkx_nodes()")
        return self.to_networkx_nodes()

    def edges_data(self):
        _deprecated("edges_data()", "to_networkx_edges()")
        return self.to_networkx_edges()
