"""Synthetic trees and pair batches for the BASELINE configs (numpy only).

The flat arrays are generated directly in the reference's numbering (in-order
ids of a strictly binary tree, /root/reference/SuchTree/MuchTree.pyx:171-180),
so no Newick round trip is needed for million-leaf trees; ``to_newick`` exists
to prove on small instances that the Newick loader gives the same arrays.
"""
import numpy as np


def balanced_tree(levels, seed=42, low=0.01, high=1.0, decimals=6):
    """Perfectly balanced binary tree with 2**levels leaves.

    In-order ids: leaves are the even ids, a node at height h (leaf = 0) has
    id = (2k+1) * 2**h - 1.  Branch lengths are uniform(low, high) rounded to
    ``decimals`` (what a Newick file with that many decimals would carry) and
    then stored as float32 like the reference does (MuchTree.pyx:60,215).

    Returns ``(parent:int32[N], distance:float32[N])`` with N = 2**(levels+1)-1.
    """
    if levels < 0 or levels > 26:
        raise ValueError("levels must be in [0, 26]")
    n = (1 << (levels + 1)) - 1
    ids = np.arange(n, dtype=np.int64)
    # height of id i = number of trailing one bits of i (== trailing zeros of i+1)
    ip1 = ids + 1
    h = np.zeros(n, dtype=np.int64)
    low_bit = ip1 & -ip1
    h = np.log2(low_bit.astype(np.float64)).astype(np.int64)
    k = (ip1 >> h) >> 1          # i + 1 = (2k+1) * 2**h
    # parent sits at height h+1: left child when k is even, right child when odd
    step = np.left_shift(np.int64(1), h)
    parent = np.where(k % 2 == 0, ids + step, ids - step)
    root = (1 << levels) - 1
    parent[root] = -1
    rng = np.random.default_rng(seed)
    lengths = np.round(rng.uniform(low, high, size=n), decimals)
    lengths[root] = -1.0
    return parent.astype(np.int32), lengths.astype(np.float32)


def complete_tree(n_leaves, seed=44, low=0.01, high=1.0, decimals=6):
    """Complete binary tree with ``n_leaves`` leaves (BASELINE config 4: 100,000 leaves, seed
    44): heap shape -- node k has children 2k and 2k+1, every level full except the last,
    which is filled from the left -- numbered in order like every tree of the reference.
    Branch lengths as in :func:`balanced_tree`.  Returns ``(parent:int32[N], distance:float32[N])``,
    N = 2*n_leaves - 1."""
    if n_leaves < 1:
        raise ValueError("n_leaves must be >= 1")
    n = 2 * n_leaves - 1
    heap = np.arange(n + 1, dtype=np.int64)           # heap positions 1..n (0 unused)
    size = np.zeros(2 * n + 2, dtype=np.int64)        # subtree sizes, zero beyond the heap
    size[1:n + 1] = 1
    level_lo = 1 << (int(n).bit_length() - 1)
    while level_lo >= 1:                              # bottom-up, one level at a time
        k = heap[level_lo:min(2 * level_lo, n + 1)]
        size[k] = 1 + size[2 * k] + size[2 * k + 1]
        level_lo >>= 1
    start = np.zeros(n + 1, dtype=np.int64)           # first in-order id of each subtree
    level_lo = 1
    while level_lo <= n:                              # top-down
        k = heap[level_lo:min(2 * level_lo, n + 1)]
        k = k[2 * k + 1 <= n]                         # internal nodes have both children
        start[2 * k] = start[k]
        start[2 * k + 1] = start[k] + size[2 * k] + 1
        level_lo <<= 1
    ids = start[1:] + size[2 * heap[1:]]              # in-order id of heap node k (k = 1..n)
    parent = np.full(n, -1, dtype=np.int64)
    parent[ids[1:]] = ids[(heap[2:] >> 1) - 1]
    rng = np.random.default_rng(seed)
    lengths = np.round(rng.uniform(low, high, size=n), decimals)
    lengths[ids[0]] = -1.0
    return parent.astype(np.int32), lengths.astype(np.float32)


def caterpillar_tree(n_leaves, seed=7):
    """Maximally unbalanced (ladder) tree: the worst case for depth.

    Leaves 0,2,4,...; internal node 2k+1 joins everything left of it with leaf
    2k+2; the root is the last internal node.
    """
    if n_leaves < 1:
        raise ValueError("n_leaves must be >= 1")
    n = 2 * n_leaves - 1
    parent = np.full(n, -1, dtype=np.int64)
    ids = np.arange(n, dtype=np.int64)
    if n_leaves > 1:
        parent[0] = 1
        odd = ids[1::2]                 # internal nodes 1,3,...,n-2
        parent[odd] = odd + 2
        parent[odd + 1] = odd           # leaf 2k+2 hangs off internal 2k+1
        parent[n - 2] = -1              # the last internal node is the root
    rng = np.random.default_rng(seed)
    lengths = np.round(rng.uniform(0.01, 1.0, size=n), 6)
    lengths[parent < 0] = -1.0
    return parent.astype(np.int32), lengths.astype(np.float32)


def random_binary_tree(n_leaves, seed=0, zero_fraction=0.0):
    """Random strictly binary tree (uniform random split sizes), in-order ids.

    ``zero_fraction`` of the edges get the reference's epsilon length (what a
    resolved polytomy looks like, MuchTree.pyx:188-194).
    """
    rng = np.random.default_rng(seed)
    n = 2 * n_leaves - 1
    parent = np.full(n, -1, dtype=np.int64)
    # iterative: (lo, hi, parent) covers in-order id range [lo, hi] with an odd count
    stack = [(0, n - 1, -1)]
    while stack:
        lo, hi, par = stack.pop()
        if lo == hi:
            parent[lo] = par
            continue
        leaves = (hi - lo) // 2 + 1
        left_leaves = int(rng.integers(1, leaves))      # 1 .. leaves-1
        node = lo + 2 * left_leaves - 1
        parent[node] = par
        stack.append((lo, node - 1, node))
        stack.append((node + 1, hi, node))
    lengths = rng.uniform(0.0001, 2.0, size=n)
    if zero_fraction > 0:
        eps = np.finfo(np.float64).eps
        lengths[rng.random(n) < zero_fraction] = eps
    lengths[parent < 0] = -1.0
    return parent.astype(np.int32), lengths.astype(np.float32)


def random_binary_tree_levels(n_leaves, seed=0):
    """Random strictly binary tree (uniform random split sizes, as ``random_binary_tree``; another random stream),
    in-order ids, built one level of splits at a time with array operations: tens of millions of nodes in seconds."""
    rng = np.random.default_rng(seed)
    n = 2 * n_leaves - 1
    parent = np.full(n, -1, dtype=np.int64)
    lo = np.zeros(1, dtype=np.int64)
    hi = np.full(1, n - 1, dtype=np.int64)
    par = np.full(1, -1, dtype=np.int64)
    while lo.size:
        single = lo == hi
        parent[lo[single]] = par[single]
        lo, hi, par = lo[~single], hi[~single], par[~single]
        if not lo.size:
            break
        leaves = (hi - lo) // 2 + 1
        left = 1 + (rng.random(lo.size) * (leaves - 1)).astype(np.int64)      # 1 .. leaves - 1
        left = np.minimum(np.maximum(left, 1), leaves - 1)
        node = lo + 2 * left - 1
        parent[node] = par
        lo, hi, par = np.concatenate([lo, node + 1]), np.concatenate([node - 1, hi]), np.concatenate([node, node])
    lengths = rng.uniform(0.0001, 2.0, size=n)
    lengths[parent < 0] = -1.0
    return parent.astype(np.int32), lengths.astype(np.float32)


def skewed_tree(rng, n_leaves, skew):
    """Random strictly binary tree whose split sizes are skewed towards caterpillars (skew -> 1) or towards
    balance (skew -> 0); in-order ids.  ``rng``: a numpy Generator (the tree is a function of its state).  With
    ``default_rng(5)`` and 1,000,000 leaves: skew 0.8 -> depth 173 (512-byte records), 0.9 -> depth 338 (the canopy
    family refuses it: the walk family's tables serve it)."""
    n = 2 * n_leaves - 1
    parent = np.full(n, -1, dtype=np.int64)
    stack = [(0, n - 1, -1)]
    while stack:
        lo, hi, par = stack.pop()
        if lo == hi:
            parent[lo] = par
            continue
        leaves = (hi - lo) // 2 + 1
        if rng.random() < skew:
            left = 1 if rng.random() < 0.5 else leaves - 1
        else:
            left = int(rng.integers(max(1, leaves // 2 - leaves // 8), min(leaves - 1, leaves // 2 + leaves // 8) + 1))
        left = min(max(left, 1), leaves - 1)
        node = lo + 2 * left - 1
        parent[node] = par
        stack.append((lo, node - 1, node))
        stack.append((node + 1, hi, node))
    dist = rng.uniform(1e-4, 3.0, size=n)
    dist[rng.random(n) < 0.1] = np.finfo(np.float64).eps
    dist[parent < 0] = -1.0
    return parent.astype(np.int32), dist.astype(np.float32)


def random_leaf_pairs(n_leaves, n_pairs, seed=3):
    """Uniform random leaf-id pairs, int64 (n,2) C-order (leaves = even ids)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_leaves, size=(n_pairs, 2), dtype=np.int64) * 2


def to_newick(parent, distance, names=None):
    """Newick text of a flat in-order tree (small trees; used by tests)."""
    parent = np.asarray(parent)
    n = len(parent)
    left = [-1] * n
    right = [-1] * n
    root = -1
    for c in range(n):
        p = int(parent[c])
        if p < 0:
            root = c
        elif c < p:
            left[p] = c
        else:
            right[p] = c
    leaf_ids = [i for i in range(n) if left[i] < 0]
    if names is None:
        names = {i: "L%d" % k for k, i in enumerate(leaf_ids)}
    out = []
    stack = [(root, 0)]
    while stack:
        node, state = stack.pop()
        if left[node] < 0:
            out.append(names[node])
            out.append(":%r" % float(np.float64(distance[node])))
        elif state == 0:
            out.append("(")
            stack.append((node, 1))
            stack.append((left[node], 0))
        elif state == 1:
            out.append(",")
            stack.append((node, 2))
            stack.append((right[node], 0))
        else:
            out.append(")")
            if node != root:
                out.append(":%r" % float(np.float64(distance[node])))
    return "".join(out) + ";"
