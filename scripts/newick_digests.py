#!/usr/bin/env python3
"""Both Newick parsers (suchtree_amd/newick.py, csrc/newick_parse.cpp) over EVERY tree file
under /root/reference/data; writes tests/golden/newick_digests.json (build container only).

Per file: node / leaf counts, root id, reference depth, sha256 of the parent and float32
distance arrays and of the leaf-name order, and whether the native parser took the file and
produced identical arrays.  An id-assignment divergence between the two parsers, or a change
of either parser's numbering on any of the reference's 327 trees, shows up as a changed
digest (tests/test_newick.py::test_reference_data_digests re-checks them when the reference
checkout is present).  Data only: no reference source text is stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_DATA = "/root/reference/data"

from suchtree_amd import _capi  # noqa: E402
from suchtree_amd.newick import flat_tree_from_newick  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def digest_file(path):
    text = open(path).read()
    try:
        t = flat_tree_from_newick(text, native=False)
    except Exception as e:   # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
    d = {"nodes": int(t.size), "leaves": int(t.num_leaves), "root": int(t.root), "depth": int(t.depth),
         "parent": sha(t.parent), "distance": sha(t.distance),
         "leaf_names": hashlib.sha256("\n".join(t.leaves.keys()).encode()).hexdigest()[:24]}
    nat = _capi.newick_native(text)
    if nat is None:
        d["native"] = "declined"
    else:
        same = (np.array_equal(nat["parent"], t.parent) and np.array_equal(nat["distance"].view(np.int32), t.distance.view(np.int32))
                and nat["names"] == list(t.leaves.keys()) and nat["root"] == t.root and nat["depth"] == t.depth
                and np.array_equal(nat["left"], t.left) and np.array_equal(nat["right"], t.right))
        d["native"] = "identical" if same else "DIFFERENT"
    return d


def main():
    out = {}
    for base, _, files in sorted(os.walk(REF_DATA)):
        for f in sorted(files):
            if f.endswith(".tree"):
                p = os.path.join(base, f)
                out[os.path.relpath(p, REF_DATA)] = digest_file(p)
    dst = os.path.join(ROOT, "tests", "golden", "newick_digests.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    kinds = {}
    for v in out.values():
        k = "error" if "error" in v else v["native"]
        kinds[k] = kinds.get(k, 0) + 1
    print(len(out), "files:", kinds, "->", dst)


if __name__ == "__main__":
    main()
