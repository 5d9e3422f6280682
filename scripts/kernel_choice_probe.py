"""Kernel families on deep / long-record trees (GPU box): what the handle chose when it was created (host_tune.h)
and every candidate forced by options, 2e7 random leaf pairs device-resident.
  python scripts/kernel_choice_probe.py ml | nj | s70 | s80 | s85 | <leaves>:<skew>"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from suchtree_amd import _capi, synth
from test_gpu_parity import _random_shape_tree
which = sys.argv[1]
if which in ("ml", "nj"):
    z = np.load("tests/golden/%s_tree.npz" % which); p, d = z["parent"], z["distance"]
elif which == "s70": p, d = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.7)
elif which == "s80": p, d = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.8)
elif which == "s85": p, d = _random_shape_tree(np.random.default_rng(5), 1_000_000, 0.85)
elif ":" in which: p, d = _random_shape_tree(np.random.default_rng(5), int(which.split(":")[0]), float(which.split(":")[1]))
t0 = time.perf_counter(); tree = _capi.DeviceTree(p, d); t_create = time.perf_counter() - t0
i = tree.info()
print(which, "tune=%s create %.3f s" % (os.environ.get("SUCHTREE_AMD_AUTOTUNE", "1"), t_create), {k: i[k] for k in ("depth", "canopy_nodes", "understory_max", "record_bytes", "lineage_entries", "big_batch_kernel", "tuned")}, flush=True)
leaves = np.flatnonzero(np.bincount(p[p >= 0], minlength=len(p)) == 0)
n = 20_000_000
g = torch.Generator(device="cuda").manual_seed(1)
li = torch.from_numpy(leaves.astype(np.int64)).cuda()
pairs = li[torch.randint(0, len(leaves), (n, 2), generator=g, device="cuda")]
out_d = torch.empty(n, dtype=torch.float64, device="cuda")
out_m = torch.empty(n, dtype=torch.int32, device="cuda")
def t(label):
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tree.distances_device(pairs.data_ptr(), n, out_d.data_ptr(), out_m.data_ptr()); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("   %-34s %.2f ms  %.3e pairs/s  checksum %.6f %d" % (label, min(ts), n / min(ts) * 1e3, float(out_d.sum()), int(out_m.long().sum())), flush=True)
t("default")
tree.set_option("prefer_walk_sorted", 0)
tree.set_option("ladder_scalar", 0); tree.set_option("tile_sort", 1); t("tile-sorted canopy kernel")
tree.set_option("tile_sort", 0); tree.set_option("ladder_scalar", 0); t("predicated kernel (ilp)")
tree.set_option("ladder_scalar", 1); tree.set_option("ladder_min_pairs", 0); t("scalar ladder kernel"); tree.set_option("ladder_dynamic", 0); t("scalar ladder kernel, static deal"); tree.set_option("ladder_dynamic", 1); tree.set_option("ladder_scalar", 0)
tree.set_strategy("walk"); t("walk family")
tree.close()
