"""Host path by NUMA node (round 6; runs ON THE GPU BOX): where the CPU passes and the link side of st_distances_host
stand when the calling process -- its copy pool, its numpy arrays (first touch) and the pinned staging slots it allocates
-- is confined to the CPUs of ONE NUMA node, for every node of the host, against the unconfined default.

    python scripts/host_numa_probe.py            # parent: topology, then one child per binding
    python scripts/host_numa_probe.py --child    # one measurement (inherits the parent's affinity)

Per binding: whole calls (both outputs / distances only, reused arrays), the pack + unpack passes alone (handle option
measure = 4), the GPU / link side alone (measure = 2), and the pipeline's own phase trace of one call (measure = 1).
"""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def topology():
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
    gpus = {}
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(d + "/vendor").read().strip() == "0x1002":
                gpus[os.path.basename(os.path.dirname(d))] = int(open(d + "/numa_node").read())
        except OSError:
            pass
    return nodes, gpus


def child():
    from suchtree_amd import _capi, synth
    from suchtree_amd._capi import MeasureOnly
    n = 50_000_000
    parent, dist = synth.balanced_tree(20)
    tree = _capi.DeviceTree(parent, dist, device=0)
    pairs = synth.random_leaf_pairs(1 << 20, n, seed=3)
    h_d, h_m = np.empty(n), np.empty(n, dtype=np.int32)
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)

    def best(fn, reps=3):
        t = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t = min(t, time.perf_counter() - t0)
        return n / t

    def switched(bits):
        tree.set_option("measure", bits)
        try:
            def call():
                try:
                    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
                except MeasureOnly:
                    pass
            return best(call)
        finally:
            tree.set_option("measure", 0)

    out = {"cpus": len(os.sched_getaffinity(0)),
           "both_outputs": best(lambda: tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)),
           "distances_only": best(lambda: tree.distances_host(pairs, True, False, out_dist=h_d)),
           "cpu_passes_alone": switched(4), "link_side_alone": switched(2)}
    sys.stderr.flush()
    tree.set_option("measure", 1)      # one traced call: the host thread's time by phase, on stderr
    tree.distances_host(pairs, True, True, out_dist=h_d, out_mrca=h_m)
    tree.set_option("measure", 0)
    print(json.dumps(out), flush=True)
    tree.close()


def main():
    if "--child" in sys.argv:
        return child()
    nodes, gpus = topology()
    print("# host: %d CPUs, NUMA nodes %s" % (os.cpu_count(), {k: "%d CPUs (%d-%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items()}))
    print("# GPUs by DRM card -> NUMA node: %s" % gpus)
    allowed = sorted(os.sched_getaffinity(0))
    print("# this process may run on %d CPUs" % len(allowed))
    bindings = [("unconfined", allowed)] + [("node %d" % k, [c for c in v if c in allowed]) for k, v in nodes.items() if len(nodes) > 1]
    for name, cpus in bindings:
        if not cpus:
            print("== %s: no allowed CPU" % name)
            continue
        def bind(cpus=cpus):
            os.sched_setaffinity(0, cpus)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], preexec_fn=bind, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        trace = [l for l in p.stderr.splitlines() if l.startswith("[pipe]")]
        print("== %s (%d CPUs)" % (name, len(cpus)))
        if line:
            d = json.loads(line[-1])
            print("   both outputs %.3e  distances only %.3e  CPU passes alone %.3e  link side alone %.3e pairs/s" %
                  (d["both_outputs"], d["distances_only"], d["cpu_passes_alone"], d["link_side_alone"]))
        else:
            print("   failed: rc %d %s" % (p.returncode, p.stderr[-400:]))
        for l in trace[-1:]:
            print("   " + l)


if __name__ == "__main__":
    main()
