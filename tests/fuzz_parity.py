"""Randomised parity hunt on the GPU box: random tree shapes and numberings x random handle options x random batches
and entry points, distances (bit for bit) and MRCA ids against the CPU oracle.  Test infrastructure (it calls the
oracle): tests/test_gpu_fuzz.py runs a short, seeded session; by hand:
  python tests/fuzz_parity.py [seconds] [seed] [big]      (big: 2^14 .. 2^20 leaves, mostly deep shapes)"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from suchtree_amd import _capi, synth
from conftest import oracle_both
from test_tables_emulated import _general_tree

OPTS = {"tile_sort": (0, 1), "ladder_scalar": (0, 1), "ladder_sums": (0, 1), "ladder_dynamic": (0, 1), "batch_probe": (0, 1), "cherries": (0, 1),
        "ladder_min_pairs": (0, 131072), "tree_rmq": (0, 1), "mrca_ranks": (0, 1), "rec_a4": (0, 1), "walk_ladder": (0, 1),
        "prefer_walk_sorted": (0, 1), "walk_crown": (0, 1), "walk_sort": (0, 1), "lineage_lens": (0, 1),
        "lineage_sums": (0, 1), "small_batch_path": (0, 1), "wire24": (0, 1), "wire48": (0, 1)}


def run(budget=240.0, seed=None, big=False):
    """One session; returns True when every batch matched (a mismatch is printed with everything needed to repeat it)."""
    seed = int(seed) if seed else int(time.time())
    rng = np.random.default_rng(seed)
    print("seed", seed, flush=True)
    t_end = time.time() + budget
    cases = checks = 0
    while time.time() < t_end:
        kind = rng.integers(0, 10)
        if big:
            kind = max(int(kind), 2)
            parent, dist = synth.skewed_tree(rng, int(2 ** rng.uniform(14, 20)), float(rng.choice([0.0, 0.5, 0.7, 0.8, 0.85, 0.9, 0.93])))
        elif kind == 0:
            parent, dist = synth.balanced_tree(int(rng.integers(1, 17)))
        elif kind == 1:
            parent, dist = _general_tree(rng, int(rng.integers(2, 30000)), int(rng.integers(1, 9)))
        else:
            leaves = int(2 ** rng.uniform(1, 17.5))
            parent, dist = synth.skewed_tree(rng, leaves, float(rng.choice([0.0, 0.2, 0.5, 0.7, 0.8, 0.9, 0.97, 0.995])))
        n = len(parent)
        permuted = kind >= 8
        if permuted:      # ids that are not in-order positions
            new_id = rng.permutation(n)
            p2 = np.empty(n, np.int32); d2 = np.empty(n, np.float32)
            p2[new_id] = np.where(parent >= 0, new_id[np.maximum(parent, 0)], -1); d2[new_id] = dist
            parent, dist = p2, d2
        if rng.integers(0, 6) == 0:      # special lengths: zeros, negatives, tiny, huge
            k = rng.integers(0, n, max(1, n // 10))
            dist = dist.copy()
            dist[k] = rng.choice(np.array([0.0, -0.5, 1e-30, 3e37, 2.220446e-16], np.float32), len(k))
            dist[parent < 0] = -1.0
        budget_mb = int(rng.choice([0, 0, 0, 1, 4, 16, 64]))
        strategy = str(rng.choice(["auto", "auto", "canopy", "walk"]))
        try:
            dev = _capi.DeviceTree(parent, dist, strategy=strategy, table_mb=budget_mb or None)
        except Exception as e:      # noqa: BLE001 -- e.g. canopy refused on this shape: say so and go on
            print("   create refused:", n, strategy, budget_mb, str(e)[:80], flush=True)
            continue
        info = dev.info()
        cases += 1
        for _ in range(int(rng.integers(1, 5))):
            chosen = {}
            for k in rng.choice(list(OPTS), int(rng.integers(0, 6)), replace=False):
                chosen[str(k)] = int(rng.choice(OPTS[str(k)]))
                try:
                    dev.set_option(str(k), chosen[str(k)])
                except Exception as e:      # noqa: BLE001
                    chosen[str(k)] = "refused"
            m = int(2 ** rng.uniform(0, 19.5)) if not big else int(2 ** rng.uniform(12, 21))
            m = max(1, min(m, int(4e10 / max(1, info["depth"]) ** 2)))      # (the oracle's MRCA is O(depth^2) per pair)
            pairs = rng.integers(0, n, (m, 2))
            mode = rng.integers(0, 4)
            if mode == 1 and n > 64:
                a = rng.integers(0, n - 40, m); pairs = np.stack([a, a + rng.integers(0, 40, m)], 1)
            elif mode == 2:
                pairs[:, 1] = pairs[:, 0]
            elif mode == 3:
                pairs[: m // 2, 0] = int(np.flatnonzero(parent < 0)[0])
            want_dist, want_mrca = bool(rng.integers(0, 4)), bool(rng.integers(0, 3))
            if not (want_dist or want_mrca):
                want_dist = True
            d, mm = dev.distances_host(pairs, want_dist, want_mrca)
            wd, wm = oracle_both(parent, dist, pairs)
            ok = (d is None or np.array_equal(d.view(np.uint64), wd.view(np.uint64)) or
                  np.array_equal(np.nan_to_num(d, nan=-7.0).view(np.uint64), np.nan_to_num(wd, nan=-7.0).view(np.uint64))) and \
                 (mm is None or np.array_equal(mm, wm))
            checks += 1
            if not ok:
                bad_d = None if d is None else np.flatnonzero(d.view(np.uint64) != wd.view(np.uint64))[:5]
                bad_m = None if mm is None else np.flatnonzero(mm != wm)[:5]
                print("MISMATCH seed", seed, "case", cases, "n", n, "kind", int(kind), "permuted", permuted, "strategy", strategy,
                      "budget", budget_mb, "opts", chosen, "pairs", m, "mode", int(mode), "info",
                      {k: info[k] for k in ("depth", "canopy_nodes", "record_bytes", "strategy", "dropped_tables")},
                      "bad dist at", bad_d, "bad mrca at", bad_m, flush=True)
                if bad_d is not None and len(bad_d):
                    i = int(bad_d[0]); print("   pair", pairs[i], "got", d[i], "want", wd[i])
                if bad_m is not None and len(bad_m):
                    i = int(bad_m[0]); print("   pair", pairs[i], "got", mm[i], "want", wm[i])
                return False
        # the generators and other entry points of the ABI on the same handle (whatever options the last batch left)
        def same(got, want):
            return np.array_equal(np.nan_to_num(got, nan=-7.0).view(np.uint64), np.nan_to_num(want, nan=-7.0).view(np.uint64))
        def fail(what, extra=""):
            print("MISMATCH seed", seed, "case", cases, what, "n", n, "kind", int(kind), "permuted", permuted, "strategy", strategy,
                  "budget", budget_mb, "opts", chosen, extra, flush=True)
            raise AssertionError("fuzz mismatch (see the line above)")
        what = rng.integers(0, 5) if info["depth"] <= 2000 else 5      # (beyond that the oracle takes minutes per batch)
        if what == 5:
            pass
        elif what == 0:      # triangle: pair k = i(i-1)/2 + j -> (ids[j], ids[i])
            ids = rng.integers(0, n, int(rng.integers(2, 500)))
            i, j = np.tril_indices(len(ids), -1)
            wd, wm = oracle_both(parent, dist, np.stack((ids[j], ids[i]), 1))
            total = len(i)
            k0 = int(rng.integers(0, total)); cnt = int(rng.integers(1, total - k0 + 1))
            d, mm = dev.triangle_host(ids, k0, cnt, True, True)
            if not (same(d, wd[k0:k0 + cnt]) and np.array_equal(mm, wm[k0:k0 + cnt])): fail("triangle", (len(ids), k0, cnt))
        elif what == 1:      # grid, rectangular and symmetric
            rows = rng.integers(0, n, int(rng.integers(1, 300))); cols = rng.integers(0, n, int(rng.integers(1, 300)))
            sym = bool(rng.integers(0, 2))
            if sym: cols = rows
            r, c = np.divmod(np.arange(len(rows) * len(cols)), len(cols))
            a, b = rows[r], cols[c]
            if sym:
                a, b = np.where(r <= c, rows[r], rows[c]), np.where(r <= c, rows[c], rows[r])
            wd, wm = oracle_both(parent, dist, np.stack((a, b), 1))
            d, mm = dev.grid_host(rows, cols, sym, 0, None, True, True)
            if not (same(d, wd) and np.array_equal(mm, wm)): fail("grid", (len(rows), len(cols), sym))
        elif what == 2:      # strides and int32 ids
            m = int(rng.integers(1, 50000))
            base = rng.integers(0, n, (m, 6))
            view = base[:, 1::3] if rng.integers(0, 2) else np.asfortranarray(base[:, :2])
            wd, wm = oracle_both(parent, dist, np.ascontiguousarray(view))
            d, mm = dev.distances_host(view, True, True)
            if not (same(d, wd) and np.array_equal(mm, wm)): fail("strided", view.strides)
            d, mm = dev.distances_host(np.ascontiguousarray(view).astype(np.int32), True, True)
            if not (same(d, wd) and np.array_equal(mm, wm)): fail("int32 ids")
        elif what == 3:      # quartet topologies
            from oracle.oracle import OracleTree
            q = rng.integers(0, n, (int(rng.integers(1, 20000)), 4))
            want = OracleTree(parent, dist).quartets(q)
            got = dev.quartets_host(q)
            if not np.array_equal(got, want): fail("quartets")
        else:      # k nearest: the k smallest distances, ascending
            qs = rng.integers(0, n, int(rng.integers(1, 40))); cs = rng.integers(0, n, int(rng.integers(1, 400)))
            k = int(rng.integers(1, min(len(cs), 16) + 1))
            index, dd = dev.knn_host(qs, cs, k)
            r, c = np.divmod(np.arange(len(qs) * len(cs)), len(cs))
            wd, _ = oracle_both(parent, dist, np.stack((qs[r], cs[c]), 1))
            wd = wd.reshape(len(qs), len(cs))
            for row in range(len(qs)):
                want = np.sort(wd[row], kind="stable")[:k]
                if not (same(dd[row], want) and same(wd[row][index[row]], want)): fail("knn", (row, k, dd[row], want))
        checks += 1
        dev.close()
    print("fuzz: %d trees, %d batches, no mismatch" % (cases, checks), flush=True)
    return True


if __name__ == "__main__":
    os.environ.setdefault("SUCHTREE_AMD_TUNE_CACHE", "0")      # (a session by hand leaves no records behind; read per tree creation)
    ok = run(float(sys.argv[1]) if len(sys.argv) > 1 else 240.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
             len(sys.argv) > 3 and sys.argv[3] == "big")
    sys.exit(0 if ok else 1)
