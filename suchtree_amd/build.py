"""Build libsuchtree_hip.so in-tree with hipcc for gfx950.

    python -m suchtree_amd.build [--force]

The library is git-ignored (history stays source-only) but travels to the GPU
box with the repo snapshot, so it must be built before a gpurun call.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsuchtree_hip.so")
# Four HIP translation units (each kernel family with its launch functions + the C ABI and host side) and two
# host-only C++ files: compiled to objects in parallel, then linked.
HIP_SOURCES = [os.path.join(CSRC, f) for f in ("suchtree_hip.hip", "launch_walk.hip", "launch_canopy.hip",
                                               "launch_canopy_sorted.hip")]
CPP_SOURCES = [os.path.join(CSRC, "tree_prep.cpp"), os.path.join(CSRC, "newick_parse.cpp")]
SOURCES = HIP_SOURCES + CPP_SOURCES
# (source, extra flags, object name): launch_canopy_sorted.hip holds the slowest instantiations and is compiled
# in three parts, two pair sources each
UNITS = [(src, [], os.path.basename(src) + ".o") for src in SOURCES if not src.endswith("launch_canopy_sorted.hip")]
UNITS = [(os.path.join(CSRC, "launch_canopy_sorted.hip"), ["-DST_SORTED_PART=%d" % k], "launch_canopy_sorted.%d.o" % k)
         for k in range(3)] + UNITS
OBJ_DIR = os.path.join(HERE, "build")
MICRO_LIB = os.path.join(HERE, "libst_microbench.so")      # measurement helpers for bench.py, not the product
MICRO_SRC = os.path.join(CSRC, "microbench.hip")
HEADERS = [os.path.join(CSRC, h) for h in (
    "tree_prep.h", "pair_math.h", "host_pipe.h", "host_copy.h", "device_common.h", "launch_geometry.h", "st_tree.h",
    "launch_policy.h", "launch_decl.h", "launch_canopy_sorted.h", "kernels_walk.h", "kernels_canopy.h",
    "kernels_canopy_sorted.h", "kernels_misc.h", "host_tree.h", "host_launch.h", "host_path.h", "host_upload.h")]
HEADERS.append(os.path.join(HERE, "..", "include", "suchtree_hip.h"))

FLAGS = [
    "--offload-arch=gfx950",
    "-O3", "-std=c++17", "-fPIC",
    # float32 adds must stay single, ordered adds (reference accumulates in C float)
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-result",
]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libsuchtree_hip.so")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build_microbench(force=False, verbose=False):
    """libst_microbench.so: the random-sector / stream-copy ceilings bench.py measures in-process."""
    if not force and os.path.exists(MICRO_LIB) and os.path.getmtime(MICRO_LIB) >= os.path.getmtime(MICRO_SRC):
        return MICRO_LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", MICRO_LIB, MICRO_SRC,
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return MICRO_LIB


NAMES_SRC = os.path.join(CSRC, "names_ext.c")


def names_ext_path():
    import sysconfig
    return os.path.join(HERE, "_names" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_names_ext(force=False, verbose=False):
    """_names extension (CPython C API, host only): the name -> id loop of distances_by_name."""
    import sysconfig
    lib = names_ext_path()
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(NAMES_SRC):
        return lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for the _names extension")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I", sysconfig.get_paths()["include"], "-o", lib, NAMES_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False, extra=()):
    """Compile every translation unit (in parallel) and link libsuchtree_hip.so."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    include = ["-I", os.path.join(HERE, "..", "include")]

    def compile_one(unit):
        src, unit_flags, name = unit
        obj = os.path.join(OBJ_DIR, name)
        cmd = [hipcc()] + FLAGS + list(extra) + unit_flags + include + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), max(1, os.cpu_count() or 2))) as pool:
        objs = list(pool.map(compile_one, UNITS))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save")])
    print(LIB)
