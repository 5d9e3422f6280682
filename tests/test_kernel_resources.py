"""Register budgets of the kernels whose occupancy depends on them (CPU: hipcc's resource-usage remarks of the
gfx950 code objects).  A 1024-lane workgroup needs <= 128 VGPRs to launch at all and <= 64 for two of them per CU;
a cold path inlined into a hot kernel has pushed one over that line before (round 4: the pipelined chain sum in
the shared-portal path took k_canopy_ladder<15> from 51 to 77 VGPRs and ml.tree from 2.85e10 to 1.92e10 pairs/s)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "suchtree_amd", "csrc")


def _resources(unit, tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                          "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                          "-o", str(tmp_path / "unit.o"), os.path.join(CSRC, unit)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m and name:
            res.setdefault(name, {})["vgpr"] = int(m.group(1))
        m = re.search(r"TotalSGPRs: (\d+)", line)
        if m and name:
            res.setdefault(name, {})["sgpr"] = int(m.group(1))
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name:
            assert int(m.group(1)) == 0, (name, "spills")
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_register_budgets_of_the_canopy_kernels(tmp_path):
    v = _resources("launch_canopy.hip", tmp_path)

    def of(fragment):
        hits = {k: n for k, n in v.items() if fragment in k}
        assert hits, fragment
        return hits

    # two 1024-lane workgroups per CU (canopy / ladder images of at most 80 KiB) are 8 waves per SIMD: at most 64
    # VGPRs and at most 80 SGPRs (the hardware admits floor(800 / (sgprs rounded up to 16 + 16)) waves per SIMD)
    for frag in ("k_canopy_ilpILi7ELi1E", "k_canopy_ilpILi3ELi1E", "k_canopy_ilpILi1ELi1E", "k_canopy_ladderILi15E"):
        for k, n in of(frag).items():
            assert n["vgpr"] <= 64 and n["sgpr"] <= 80, (k, n)
    # every kernel of the family is launched with 1024 lanes: at most 128 VGPRs
    for k, n in v.items():
        if "k_canopy" in k or "k_mrca_ranks" in k:
            assert n["vgpr"] <= 128, (k, n)
