"""Regenerate tests/golden/ from the reference checkout (build container only).

    python scripts/make_golden.py            # needs /root/reference

What is written (all of it DATA -- inputs and expected outputs -- never source):
  test.tree, test.matrix   verbatim copies of the reference's own test fixtures
                           (SuchTree/tests/test.tree, SuchTree/tests/test.matrix;
                           the matrix is the only numeric known-answer file the
                           reference holds for this path)
  support_*.tree           the reference's three fixtures for node supports (SuchTree/tests/support_{int,float,comment}.tree)
  host.tree                data/bigtrees/host.tree, the tree behind the known
                           answers printed in docs/examples/SuchTree_examples.md
  known_answers.json       the values printed in the reference docs, with citations
  gopher_louse/, fish_worm/  the two-tree co-phylogeny data sets of BASELINE config 5 and
                           the notebooks' printed answers (data/gopher-louse, data/fish-worm)
  ml_nj_leaf_map.npz        for every leaf of ml.tree (leaf_ids order) the id of the same taxon in nj.tree
  ml_tree.npz / nj_tree.npz  flat arrays (parent:int32, distance:float32, leaf ids)
                           of data/bigtrees/{ml,nj}.tree produced by
                           suchtree_amd.newick -- BASELINE config 2's tree, shipped as
                           arrays because /root/reference does not exist on the GPU box
  oracle_digests.json      sha256 of oracle outputs on seeded pair batches, so the
                           oracle built on another machine can be checked against
                           the oracle built here

The reference extension itself is NOT imported: MuchTree.pyx:3 needs dendropy,
which this image lacks, and writing a stand-in for it is not allowed.  The
oracle (oracle/suchtree_oracle.c) is therefore pinned by the reference's own
known answers above, not by outputs of the reference.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from suchtree_amd.newick import flat_tree_from_newick  # noqa: E402
from oracle.oracle import OracleTree  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "SuchTree/tests/test.tree"), os.path.join(OUT, "test.tree"))
    shutil.copyfile(os.path.join(REF, "SuchTree/tests/test.matrix"), os.path.join(OUT, "test.matrix"))
    shutil.copyfile(os.path.join(REF, "data/bigtrees/host.tree"), os.path.join(OUT, "host.tree"))
    for f in ("support_int.tree", "support_float.tree", "support_comment.tree"):      # SuchTree/tests: node supports
        shutil.copyfile(os.path.join(REF, "SuchTree/tests", f), os.path.join(OUT, f))
    for f in ("test.tree", "test.matrix", "host.tree", "support_int.tree", "support_float.tree", "support_comment.tree"):
        os.chmod(os.path.join(OUT, f), 0o644)

    known = {
        "host_tree_leaves": {
            "cite": "docs/examples/SuchTree_examples.md:99-112 (T.leaves printed by the reference with real dendropy)",
            "value": {
                "Tropheus_moorii": 0, "Lobochilotes_labiatus": 2, "Tanganicodus_irsacae": 4,
                "Cyprichromis_coloratus": 6, "Haplotaxodon_microlepis": 8, "Perissodus_microlepis": 10,
                "Plecodus_straeleni": 12, "Xenotilapia_flavipinnis": 14, "Triglachromis_otostigma": 16,
                "Reganochromis_calliurus": 18, "Trematochromis_benthicola": 20,
                "Lepidiolamprologus_profundicola": 22, "Neolamprologus_buescheri": 24,
                "Chalinochromis_brichardi": 26},
        },
        "host_tree_distances": [
            {"cite": "docs/examples/SuchTree_examples.md:136-138", "a": 12, "b": 26, "printed_6dp": "0.388425"},
            {"cite": "docs/examples/SuchTree_examples.md:156-158", "a": "Reganochromis_calliurus",
             "b": "Haplotaxodon_microlepis", "printed_6dp": "0.270743"},
        ],
        "bigtrees_sizes": {
            "cite": "docs/benchmarks.md:45-48, data/bigtrees/README.md:5-6",
            "nodes": 108653, "leaves": 54327,
        },
    }
    known["gopher_louse_linklist"] = {
        "cite": "data/gopher-louse/Gopher-Louse.ipynb cell 18 (SLT.linklist printed by the reference; compare as a "
                "set: the 2016 build listed columns in another order)",
        "value": [[10, 4], [16, 10], [32, 28], [22, 18], [26, 22], [0, 26], [30, 26], [20, 14], [2, 28], [4, 12],
                  [8, 2], [28, 24], [12, 6], [24, 20], [14, 8], [18, 16], [6, 0]]}
    known["gopher_louse_linked_distances"] = {
        "cite": "data/gopher-louse/Gopher-Louse.ipynb cell 9: pearsonr / kendalltau of result['TreeA'] vs "
                "result['TreeB'] from SLT.linked_distances() (136 link pairs)",
        "pearson_r": 0.49018498968585178, "kendall_tau": 0.20975684102929301}
    known["fish_worm_sizes"] = {"cite": "data/fish-worm/fish-worm.ipynb cell 24",
                                "links": 191, "hosts": 21, "guests": 191}
    for sub, names in (("gopher-louse", ("gopher.tree", "lice.tree", "links.csv")),
                       ("fish-worm", ("host.tree", "guest.tree", "links.csv"))):
        dst = os.path.join(OUT, sub.replace("-", "_"))
        os.makedirs(dst, exist_ok=True)
        for f in names:
            shutil.copyfile(os.path.join(REF, "data", sub, f), os.path.join(dst, f))
            os.chmod(os.path.join(dst, f), 0o644)
    with open(os.path.join(OUT, "known_answers.json"), "w") as fh:
        json.dump(known, fh, indent=1, sort_keys=True)

    digests = {}
    leaves_of = {}
    for name in ("ml", "nj"):
        t = flat_tree_from_newick(open(os.path.join(REF, "data/bigtrees/%s.tree" % name)).read())
        leaves_of[name] = dict(t.leaves)
        leaf_ids = np.array(list(t.leaves.values()), dtype=np.int32)
        np.savez_compressed(os.path.join(OUT, "%s_tree.npz" % name), parent=t.parent, distance=t.distance,
                            leaf_ids=leaf_ids, depth=np.int32(t.depth), root=np.int32(t.root))
        O = OracleTree(t.parent, t.distance)
        rng = np.random.default_rng(2)
        pairs = rng.choice(leaf_ids.astype(np.int64), size=(20000, 2))
        digests[name] = {
            "pairs": "default_rng(2).choice(leaf_ids.astype(int64), size=(20000, 2))",
            "size": int(t.size), "depth": int(t.depth), "root": int(t.root),
            "parent_sha256": digest(t.parent), "distance_sha256": digest(t.distance),
            "dist_sha256": digest(O.distances(pairs)), "mrca_sha256": digest(O.mrca_bulk(pairs)),
        }
    # the same taxa in both trees (docs/examples/SuchTree_examples.md:296-352 compares their distances by name):
    # for every leaf of ml.tree, in the order of ml_tree.npz's leaf_ids, the id of the leaf of that name in nj.tree
    assert set(leaves_of["ml"]) == set(leaves_of["nj"])
    np.savez_compressed(os.path.join(OUT, "ml_nj_leaf_map.npz"),
                        nj_id_of_ml_leaf=np.array([leaves_of["nj"][k] for k in leaves_of["ml"]], dtype=np.int32))
    t = flat_tree_from_newick(open(os.path.join(OUT, "test.tree")).read())
    O = OracleTree(t.parent, t.distance, t.left, t.right, t.support)
    allp = np.array([(a, b) for a in range(t.size) for b in range(t.size)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "gopher_all_pairs.npz"), parent=t.parent, distance=t.distance,
                        pairs=allp, dist=O.distances(allp), mrca=O.mrca_bulk(allp))
    with open(os.path.join(OUT, "oracle_digests.json"), "w") as fh:
        json.dump(digests, fh, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
