"""Newick ingest with the reference's node numbering.

The reference builds its flat node table from a dendropy tree
(/root/reference/SuchTree/MuchTree.pyx:138-228): parse, ``resolve_polytomies()``,
number the nodes by in-order traversal, then fill parent / children / support /
distance per id.  dendropy is a third-party dependency that is not vendored in
the reference, so this module restates the parts of its *published behaviour*
that decide node ids and edge lengths:

* Newick tokens: ``( ) , : ;``, ``[comments]`` are dropped, single-quoted
  labels with ``''`` as an escaped quote, underscores kept verbatim
  (``preserve_underscores=True``, MuchTree.pyx:141).
* A label on a leaf is its taxon name; a label on an internal node is kept
  as the node label (``suppress_internal_node_taxa=True``) and becomes
  ``support`` when it parses as a float (MuchTree.pyx:207-210).
* ``resolve_polytomies()`` without an rng: for every node with more than two
  children, repeatedly detach its first two children, hang them under a new
  zero-length node and append that node as the last child.
* In-order traversal: left subtree, node, right subtree; only defined for
  nodes with zero or two children.

The numbering is pinned by the dendropy-produced outputs printed in the
reference's docs (tests/test_newick.py).

Output is a :class:`FlatTree` of numpy arrays -- the SoA form of the
reference's 20-byte ``Node`` records (MuchTree.pyx:55-60).
"""
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .exceptions import TreeStructureError

# tiny nonzero length given to missing / zero-length edges (MuchTree.pyx:136,188-194)
EPSILON = float(np.finfo(np.float64).eps)

_TOKEN = re.compile(
    r"""\s*(?:
        (?P<comment>\[[^\]]*\])
      | (?P<quoted>'(?:[^']|'')*')
      | (?P<punct>[(),:;])
      | (?P<word>[^\s()\[\],:;'][^\s()\[\],:;]*)
    )""",
    re.X,
)


@dataclass
class FlatTree:
    """SoA image of the reference's node table plus the name maps."""

    parent: np.ndarray          # int32[N], root = -1
    left: np.ndarray            # int32[N], leaf = -1
    right: np.ndarray           # int32[N], leaf = -1
    support: np.ndarray         # float32[N], -1 when absent
    distance: np.ndarray        # float32[N], root = -1.0
    root: int
    depth: int                  # nodes on the longest leaf->root path
    leaves: Dict[str, int] = field(default_factory=dict)      # name -> id, in-order
    leaf_nodes: Dict[int, str] = field(default_factory=dict)  # id -> name
    internal_nodes: Optional[np.ndarray] = None

    @property
    def size(self) -> int:
        return int(self.parent.shape[0])

    @property
    def num_leaves(self) -> int:
        return len(self.leaf_nodes)


def _tokenize(text: str):
    pos, n = 0, len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if m is None:
            if text[pos:].strip() == "":
                return
            raise TreeStructureError("Newick syntax error near offset %d: %r" % (pos, text[pos:pos + 20]))
        pos = m.end()
        kind = m.lastgroup
        if kind == "comment":
            continue
        tok = m.group(kind)
        if kind == "quoted":
            yield "label", tok[1:-1].replace("''", "'")
        elif kind == "punct":
            yield tok, tok
        else:
            yield "label", tok


def parse_newick(text: str):
    """Parse the first tree of a Newick string into parallel lists.

    Returns ``(children, label, length, root)`` where ``children[i]`` is the
    ordered child list of node ``i`` (parse order ids), ``label[i]`` a str or
    None and ``length[i]`` a float or None.
    """
    children: List[List[int]] = [[]]
    parent: List[int] = [-1]
    label: List[Optional[str]] = [None]
    length: List[Optional[float]] = [None]
    cur = 0
    want_length = False
    closed = False
    seen_any = False

    def new_child(p):
        children.append([])
        parent.append(p)
        label.append(None)
        length.append(None)
        k = len(children) - 1
        children[p].append(k)
        return k

    for kind, tok in _tokenize(text):
        seen_any = True
        if want_length:
            if kind != "label":
                raise TreeStructureError("Newick syntax error: expected a branch length, got %r" % tok)
            try:
                length[cur] = float(tok)
            except ValueError:
                raise TreeStructureError("Newick syntax error: bad branch length %r" % tok)
            want_length = False
        elif kind == "(":
            cur = new_child(cur)
        elif kind == ",":
            if parent[cur] < 0:
                raise TreeStructureError("Newick syntax error: ',' outside parentheses")
            cur = new_child(parent[cur])
        elif kind == ")":
            if parent[cur] < 0:
                raise TreeStructureError("Newick syntax error: unbalanced ')'")
            cur = parent[cur]
        elif kind == ":":
            want_length = True
        elif kind == ";":
            closed = True
            break
        else:
            label[cur] = tok
    if not seen_any:
        raise TreeStructureError("empty Newick input")
    if cur != 0 or want_length:
        raise TreeStructureError("Newick syntax error: unbalanced parentheses")
    del closed  # a missing ';' on the last tree is tolerated
    return children, label, length, 0


def _resolve_polytomies(children, label, length):
    """dendropy's deterministic ``resolve_polytomies()`` (limit=2, no rng)."""
    n0 = len(children)
    for node in range(n0):
        ch = children[node]
        while len(ch) > 2:
            c1, c2 = ch[0], ch[1]
            children.append([c1, c2])
            label.append(None)
            length.append(0.0)
            nn = len(children) - 1
            del ch[0:2]
            ch.append(nn)


def _inorder(children, root):
    """In-order node list (iterative; trees can be 10^5+ levels deep)."""
    order = []
    stack = [(root, 0)]
    while stack:
        node, state = stack.pop()
        ch = children[node]
        if not ch:
            order.append(node)
        elif len(ch) == 2:
            if state == 0:
                stack.append((node, 1))
                stack.append((ch[0], 0))
            else:
                order.append(node)
                stack.append((ch[1], 0))
        else:
            # dendropy: "In-order traversal only supported for binary trees"
            raise TypeError("In-order traversal only supported for binary trees")
    return order


def node_depths(parent: np.ndarray) -> np.ndarray:
    """Edges from every node to the root, by pointer doubling (O(N log depth))."""
    p = np.asarray(parent, dtype=np.int64).copy()
    d = (p >= 0).astype(np.int64)
    for _ in range(70):
        idx = np.flatnonzero(p >= 0)
        if idx.size == 0:
            break
        tgt = p[idx]
        d[idx] = d[idx] + d[tgt]
        p[idx] = p[tgt]
    else:
        raise TreeStructureError("parent array contains a cycle")
    return d


def flat_tree_from_newick(text: str, native: bool = True) -> FlatTree:
    """Newick text -> :class:`FlatTree` (MuchTree.pyx:138-228).

    Uses the native ingest of libsuchtree_hip.so (csrc/newick_parse.cpp, host code, no GPU
    needed) when it is available and accepts the input; anything it declines -- every
    syntax error included -- goes through the pure-Python path below, which is the
    reference statement of the behaviour and the owner of the error messages.
    """
    if native:
        try:
            from . import _capi
            got = _capi.newick_native(text)
        except Exception:
            got = None
        if got is not None:
            leaf_list = got["leaf_ids"].tolist()
            names = got["names"]
            return FlatTree(parent=got["parent"], left=got["left"], right=got["right"],
                            support=got["support"], distance=got["distance"], root=got["root"],
                            depth=got["depth"], leaves=dict(zip(names, leaf_list)),
                            leaf_nodes=dict(zip(leaf_list, names)),
                            internal_nodes=np.flatnonzero(got["left"] >= 0))
    return _flat_tree_from_newick_py(text)


def _flat_tree_from_newick_py(text: str) -> FlatTree:
    """MuchTree.pyx:157-228 on top of :func:`parse_newick` (pure Python)."""
    children, label, length, root0 = parse_newick(text)
    _resolve_polytomies(children, label, length)
    order = _inorder(children, root0)
    size = len(order)
    if size != len(children):
        raise TreeStructureError("tree has unreachable nodes")

    node_id = np.empty(size, dtype=np.int64)
    node_id[np.asarray(order, dtype=np.int64)] = np.arange(size, dtype=np.int64)

    parent = np.full(size, -1, dtype=np.int32)
    left = np.full(size, -1, dtype=np.int32)
    right = np.full(size, -1, dtype=np.int32)
    support = np.full(size, -1.0, dtype=np.float32)
    dist64 = np.full(size, -1.0, dtype=np.float64)

    leaves: Dict[str, int] = {}
    leaf_nodes: Dict[int, str] = {}
    internal = []
    for nid, k in enumerate(order):
        ch = children[k]
        if not ch:
            name = label[k]
            if name is None:
                raise TreeStructureError("leaf without a name (node %d)" % nid)
            leaves[name] = nid
            leaf_nodes[nid] = name
        else:
            internal.append(nid)
            l, r = int(node_id[ch[0]]), int(node_id[ch[1]])
            left[nid], right[nid] = l, r
            parent[l] = nid
            parent[r] = nid
            lab = label[k]
            if lab is not None:
                try:
                    support[nid] = float(lab)
                except ValueError:
                    pass
        if k != root0:
            e = length[k]
            # missing and zero lengths both become epsilon (MuchTree.pyx:188-194)
            dist64[nid] = EPSILON if not e else e
    root = int(node_id[root0])
    with np.errstate(over="ignore"):
        distance = dist64.astype(np.float32)   # the reference stores C floats (pyx:60,215)

    depths = node_depths(parent)
    leaf_ids = np.fromiter(leaf_nodes.keys(), dtype=np.int64, count=len(leaf_nodes))
    depth = int(depths[leaf_ids].max()) + 1 if leaf_ids.size else 0

    return FlatTree(parent=parent, left=left, right=right, support=support, distance=distance,
                    root=root, depth=depth, leaves=leaves, leaf_nodes=leaf_nodes,
                    internal_nodes=np.array(internal))


def flat_tree_from_arrays(parent, distance, leaf_names=None, support=None) -> FlatTree:
    """Adopt precomputed flat arrays (in-order ids, strictly binary).

    ``leaf_names``: optional sequence of names for the leaves in increasing id
    order; default ``L0, L1, ...``.
    """
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    distance = np.ascontiguousarray(distance, dtype=np.float32)
    size = parent.shape[0]
    if distance.shape[0] != size:
        raise TreeStructureError("parent and distance differ in length")
    roots = np.flatnonzero(parent < 0)
    if roots.size != 1:
        raise TreeStructureError("expected exactly one root, found %d" % roots.size)
    root = int(roots[0])
    if size and (parent.max() >= size):
        raise TreeStructureError("parent id out of range")
    ids = np.arange(size, dtype=np.int64)
    has_parent = parent >= 0
    is_left = has_parent & (ids < parent)
    is_right = has_parent & (ids > parent)
    if np.any(has_parent & (ids == parent)):
        raise TreeStructureError("node is its own parent")
    left = np.full(size, -1, dtype=np.int32)
    right = np.full(size, -1, dtype=np.int32)
    left[parent[is_left]] = ids[is_left]
    right[parent[is_right]] = ids[is_right]
    n_left = np.bincount(parent[is_left], minlength=size)
    n_right = np.bincount(parent[is_right], minlength=size)
    if np.any(n_left > 1) or np.any(n_right > 1) or np.any(n_left != n_right):
        raise TreeStructureError("arrays do not describe an in-order-numbered strictly binary tree")
    leaf_ids = np.flatnonzero(left < 0)
    if leaf_names is None:
        names = ["L%d" % i for i in range(leaf_ids.size)]
    else:
        names = list(leaf_names)
        if len(names) != leaf_ids.size:
            raise TreeStructureError("leaf_names has %d entries for %d leaves" % (len(names), leaf_ids.size))
    leaf_list = leaf_ids.tolist()
    leaves = dict(zip(names, leaf_list))
    leaf_nodes = dict(zip(leaf_list, names))
    depths = node_depths(parent)
    if np.any(depths[has_parent] <= 0):
        raise TreeStructureError("parent array contains a cycle")
    depth = int(depths[leaf_ids].max()) + 1 if leaf_ids.size else 0
    sup = np.full(size, -1.0, dtype=np.float32) if support is None else np.ascontiguousarray(support, dtype=np.float32)
    return FlatTree(parent=parent, left=left, right=right, support=sup, distance=distance,
                    root=root, depth=depth, leaves=leaves, leaf_nodes=leaf_nodes,
                    internal_nodes=np.flatnonzero(left >= 0))
