"""Build libsuchtree_hip.so in-tree with hipcc for gfx950.

    python -m suchtree_amd.build [--force]

The library is git-ignored (history stays source-only) but travels to the GPU
box with the repo snapshot, so it must be built before a gpurun call.
"""
import glob as _glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsuchtree_hip.so")
# Four HIP translation units (each kernel family with its launch functions + the C ABI and host side; launch_canopy.hip in three
# parts) and two host-only C++ files: compiled to objects in parallel, then linked.
HIP_SOURCES = [os.path.join(CSRC, f) for f in ("suchtree_hip.hip", "launch_walk.hip", "launch_canopy.hip",
                                               "launch_canopy_sorted.hip")]
CPP_SOURCES = [os.path.join(CSRC, "tree_prep.cpp"), os.path.join(CSRC, "newick_parse.cpp")]
SOURCES = HIP_SOURCES + CPP_SOURCES
# (source, extra flags, object name): launch_canopy.hip holds the slowest instantiations (the scalar ladder kernel's two forms, the
# predicated kernel's long-chain forms) and is compiled in three parts, two pair sources each
UNITS = [(src, [], os.path.basename(src) + ".o") for src in SOURCES if not src.endswith("launch_canopy.hip")]
UNITS = [(os.path.join(CSRC, "launch_canopy.hip"), ["-DST_CANOPY_PART=%d" % k], "launch_canopy.%d.o" % k)
         for k in range(3)] + UNITS
OBJ_DIR = os.path.join(HERE, "build")
MICRO_LIB = os.path.join(HERE, "libst_microbench.so")      # measurement helpers for bench.py, not the product
MICRO_SRC = os.path.join(CSRC, "microbench.hip")
# every header under csrc/ (a new one cannot be forgotten) + the public C ABI header
HEADERS = sorted(_glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "suchtree_hip.h")]
LOCK = os.path.join(HERE, ".build.lock")

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


class _BuildLock:
    """One builder at a time across processes (`bench.py --gpus N` starts N ranks that all load the library; a
    fresh copy of the repo has none): an exclusive flock on a file beside the library.  Whoever gets the lock
    second finds the library fresh and does nothing."""

    def __enter__(self):
        import fcntl
        self.f = open(LOCK, "a+")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def _tmp_name(path):
    return "%s.tmp.%d" % (path, os.getpid())


def _link_into_place(cmd_head, out, cmd_tail, verbose):
    """Link to a temporary name, then rename over the target: a concurrent dlopen sees the old file or the
    new one, never a half-written one."""
    tmp = _tmp_name(out)
    cmd = cmd_head + ["-o", tmp] + cmd_tail
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return out


def build_microbench(force=False, verbose=False):
    """libst_microbench.so: the random-sector / stream-copy ceilings bench.py measures in-process."""
    def fresh():
        return os.path.exists(MICRO_LIB) and os.path.getmtime(MICRO_LIB) >= os.path.getmtime(MICRO_SRC)
    if not force and fresh():
        return MICRO_LIB
    with _BuildLock():
        if not force and fresh():
            return MICRO_LIB
        return _link_into_place([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"], MICRO_LIB,
                                [MICRO_SRC, "-Wl,-rpath,/opt/rocm/lib"], verbose)


NAMES_SRC = os.path.join(CSRC, "names_ext.c")


def names_ext_path():
    import sysconfig
    return os.path.join(HERE, "_names" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_names_ext(force=False, verbose=False):
    """_names extension (CPython C API, host only): the name -> id loop of distances_by_name."""
    import sysconfig
    lib = names_ext_path()

    def fresh():
        return os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(NAMES_SRC)
    if not force and fresh():
        return lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for the _names extension")
    with _BuildLock():
        if not force and fresh():
            return lib
        return _link_into_place([cc, "-O2", "-fPIC", "-shared", "-Wall", "-I", sysconfig.get_paths()["include"]], lib,
                                [NAMES_SRC], verbose)


def build(force=False, verbose=False, extra=()):
    """Compile every translation unit (in parallel) and link libsuchtree_hip.so."""
    if not force and not stale():
        return LIB
    with _BuildLock():
        if not force and not stale():      # (another process built it while this one waited for the lock)
            return LIB
        return _build_locked(verbose, extra)


def _build_locked(verbose, extra):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    include = ["-I", os.path.join(HERE, "..", "include")]

    def compile_one(unit):
        src, unit_flags, name = unit
        obj = os.path.join(OBJ_DIR, name)
        tmp = _tmp_name(obj)
        cmd = [hipcc()] + FLAGS + list(extra) + unit_flags + include + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), max(1, os.cpu_count() or 2))) as pool:
        objs = list(pool.map(compile_one, UNITS))
    return _link_into_place([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"], LIB,
                            objs + ["-Wl,-rpath,/opt/rocm/lib", "-lpthread"], verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save")])
    print(LIB)
