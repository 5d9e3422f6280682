"""bench_legs.py's host-side arithmetic (no GPU): which committed PMC summary may speak for a launch, the shape of a leg's
roofline block with and without one, and the per-leg CPU baseline on the oracle."""
import json
import os

import numpy as np

import bench
import bench_legs
from conftest import ROOT
from suchtree_amd import synth


class _Be:
    no_microbench = True      # (the in-process sector microbenchmark needs a GPU)
    local_rank = 0


LADDER = "void st::k_canopy_ladder<31, st::SrcContig, %s>(st::CanopyParams, st::SrcContig, long long, st::DistSink, st::MrcaSink, st::Fault*, unsigned long long*, int const*)"


def test_a_summary_speaks_only_for_its_own_kernel_form_and_batch():
    t = {"hbm_bytes_per_launch": 7e8, "pairs_per_launch": 1e7, "kernel_full_name": LADDER % "true"}
    assert bench_legs.traffic_matches(t, "canopy_ladder", 10_000_000, ladder_sums=1)
    assert not bench_legs.traffic_matches(t, "canopy_ladder", 10_000_000, ladder_sums=0)      # the other form of the kernel
    assert not bench_legs.traffic_matches(t, "canopy_ladder", 20_000_000, ladder_sums=1)      # another batch size
    assert not bench_legs.traffic_matches(t, "canopy_ilp", 10_000_000)                        # another kernel
    assert not bench_legs.traffic_matches(t, "walk_sorted", 10_000_000)
    assert not bench_legs.traffic_matches(None, "canopy_ladder", 10_000_000)
    w = dict(t, kernel_full_name="void st::k_walk_sorted<4, true, st::SrcContig>(st::WalkParams, ...)")
    assert bench_legs.traffic_matches(w, "walk_sorted", 10_000_000) and bench_legs.traffic_matches(w, "walk", 10_000_000)


def test_leg_roofline_with_and_without_counters(tmp_path, monkeypatch):
    fake = {"hbm_bytes_per_launch": 690e6, "pairs_per_launch": 1e7, "kernel_full_name": LADDER % "false", "tag": "nj_r99",
            "kernels": {LADDER % "false": {"calls": 4, "avg_ns": 500000.0}},
            "counters_mean_per_launch": {"TCC_EA0_RDREQ_sum": 7.65e6, "TCP_TCC_READ_REQ_sum": 5.67e7}}
    monkeypatch.setattr(bench_legs, "load_traffic", lambda tag: (fake, "profiles/traffic_%s_r99.json" % tag))
    r = bench_legs.leg_roofline(_Be(), "nj", "canopy_ladder", 2.0e10, 0.5, 10_000_000, 983.5, 1 << 30, ladder_sums=0)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic"] == 690e6 and abs(r["traffic_bytes_per_pair"] - 69.0) < 1e-9
    assert abs(r["achieved"] - 69.0 * 2.0e10 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and r["frac"] < 1
    assert abs(r["rocprof"]["frac"] - 690e6 / 500e-6 / 1e9 / 8000.0) < 1e-12
    assert r["algorithmic"]["exceeds_peak"] and r["algorithmic"]["frac_of_hbm_peak"] > 2 and r["algorithmic"]["why"]
    assert abs(r["request_rate"]["fabric_reads_per_pair"] - 0.765) < 1e-9 and abs(r["l2_request_rate"]["l2_reads_per_pair"] - 5.67) < 1e-9
    # the handle runs the joint form, the summary is of the climbing form: the block says so instead of borrowing its bytes
    r = bench_legs.leg_roofline(_Be(), "nj", "canopy_ladder", 2.4e10, 0.42, 10_000_000, 983.5, 1 << 30, ladder_sums=1)
    assert r["traffic"] is None and "no committed counter pass" in r["achieved_is"] and r["frac"] == r["algorithmic"]["frac_of_hbm_peak"]
    assert "lineage sums" in r["kernel"]


def test_leg_cpu_baseline_on_the_oracle():
    from oracle.oracle import OracleTree
    parent, dist = synth.balanced_tree(10)
    pairs = synth.random_leaf_pairs(1 << 10, 60_000, seed=5)
    O = OracleTree(parent, dist)
    cpu, d = bench_legs.leg_cpu_baseline(O, pairs, "a test batch", seconds=0.2)
    assert cpu["kind"] in ("port", "reference") and cpu["unit"] == "pairs/s" and cpu["value"] > 0 and cpu["single_thread_value"] > 0
    assert cpu["cores"] == len(os.sched_getaffinity(0)) and "a test batch" in cpu["sample"]
    assert 0 < len(d) <= len(pairs) and np.array_equal(d.view(np.int64), O.distances(pairs[:len(d)]).view(np.int64))


def test_headline_counters_speak_for_the_default_configuration_only():
    args = bench.parse([])
    info = {"strategy": "canopy", "canopy_nodes": 16383, "record_bytes": 64}
    t = {"hbm_bytes_per_launch": 1.2e10, "pairs_per_launch": 1e8, "kernel_full_name": "void st::k_canopy_ilp<7, 1, st::SrcContig, true>(...)",
         "config": {"levels": 20, "canopy_nodes": 16383, "record_bytes": 64}}
    assert bench.traffic_speaks_for(t, args, info, 1) == (True, None)
    ok, why = bench.traffic_speaks_for(t, bench.parse(["--levels", "18"]), dict(info, canopy_nodes=16383), 1)
    assert not ok and "levels" in why
    ok, why = bench.traffic_speaks_for(t, args, dict(info, strategy="walk"), 1)
    assert not ok and "walk" in why
    ok, why = bench.traffic_speaks_for(dict(t, config=None), bench.parse(["--levels", "18"]), info, 1)      # a summary of an earlier round
    assert not ok
    assert bench.traffic_speaks_for(None, args, info, 1)[0] is False
    # the newest committed summary is of the default configuration (what bench.py's default run reports against)
    committed, name = bench.latest_traffic()
    assert committed is not None and bench.traffic_speaks_for(committed, args, info, 1)[0], name
    foot = bench.gather_footprint_bytes({"strategy": "canopy", "a_side_bytes": 4, "b_table_bytes_per_leaf": 16, "record_bytes": 64, "n_nodes": 1}, 1 << 20)
    assert foot == 20 << 20
