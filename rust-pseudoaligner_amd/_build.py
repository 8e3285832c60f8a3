"""Build recipe of the product library (hipcc, gfx950), in-tree so that the .so travels with the repository snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"

PRODUCT_SO = PKG / "libpseudoaligner_amd.so"

HOST_SOURCES = ["host_index.cpp", "dbg_build.cpp", "device_flatten.cpp", "synth.cpp", "fastq.cpp", "record_stream.cpp", "host_batch.cpp"]
HIP_SOURCES = ["kernels.hip", "map_pool.hip", "device_index.hip", "collective.hip", "barcode_counts.hip", "index_build.hip", "index_fill.hip", "count_sort.hip", "resolve.hip", "render.hip", "fastq_scan.hip", "compact.hip"]


# flags of single sources. map_pool.hip: without LLVM's machine-sinking pass the mapping kernel is 1 % faster at config 3 (7.608 -> 7.536 ms, three interleaved
# pairs on one box), 0.2 % at config 5, 0.5-1.7 % at config3r (profiles/r06_compiler_flags_ab.txt; eight other scheduler / if-conversion flags: within noise or slower)
EXTRA_FLAGS = {"map_pool.hip": ["-mllvm", "-disable-machine-sink"]}


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), proc.stdout, proc.stderr))
    return proc


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the product library cannot be built")


OBJ_DIR = PKG / "_build"   # per-source objects (git-ignored): only what changed is recompiled


def _deps_of(src: Path):
    """the source plus every header it can see (all of csrc/*.hpp and the public header: coarse, always safe)"""
    return [src] + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + [ROOT / "include" / "pseudoaligner_amd.h", Path(__file__)]


def build_product(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: HIP kernels + C ABI + host runtime -> libpseudoaligner_amd.so (one object per source,
    compiled in parallel, then linked)"""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [CSRC / s for s in HOST_SOURCES + HIP_SOURCES]
    OBJ_DIR.mkdir(exist_ok=True)
    hipcc = hipcc_path()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function"]
    jobs = []
    for s in srcs:
        obj = OBJ_DIR / (s.name + ".o")
        if force or _stale(obj, _deps_of(s)):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(s.name, []) + ["-x", "hip", "-c", str(s), "-o", str(obj)])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    objs = [OBJ_DIR / (s.name + ".o") for s in srcs]
    if jobs or _stale(PRODUCT_SO, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [str(o) for o in objs] + ["-ldl", "-lz", "-o", str(PRODUCT_SO)])
    return PRODUCT_SO


MAXBINS_SO = OBJ_DIR / "libpseudoaligner_amd_maxbins2.so"


def build_maxbins_variant(force: bool = False) -> Path:
    """A TEST build of the product: count_sort.hip compiled with -DPA_MAX_BINS=2, everything else the product's own objects. A class-count table
    of more than 65 536 slots is then "beyond MAX_BINS" and takes the plain-atomics path of count_sort.hip for batches of ANY size — the path
    that in the product only tables beyond 8.4 M classes take for large batches (tests/test_gpu_scale.py loads it through PA_PRODUCT_SO)."""
    build_product()
    hipcc = hipcc_path()
    src = CSRC / "count_sort.hip"
    obj = OBJ_DIR / "count_sort.maxbins2.o"
    if force or _stale(obj, _deps_of(src)):
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-DPA_MAX_BINS=2", "-x", "hip", "-c", str(src), "-o", str(obj)])
    objs = [OBJ_DIR / (s + ".o") for s in HOST_SOURCES + HIP_SOURCES if s != "count_sort.hip"] + [obj]
    if force or _stale(MAXBINS_SO, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [str(o) for o in objs] + ["-ldl", "-lz", "-o", str(MAXBINS_SO)])
    return MAXBINS_SO


ABI_CHECK_SRC = ROOT / "integration" / "c" / "abi_check.c"
ABI_CHECK_BIN = ROOT / "integration" / "c" / "abi_check"


def build_abi_check(force: bool = False) -> Path:
    """gcc -Wall -Werror: the plain-C client that calls every entry point of include/pseudoaligner_amd.h (type check of the
    header from C; the tests run it)"""
    if force or _stale(ABI_CHECK_BIN, [ABI_CHECK_SRC, ROOT / "include" / "pseudoaligner_amd.h", PRODUCT_SO]):
        _run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(ABI_CHECK_SRC), "-L", str(PKG),
              "-lpseudoaligner_amd", "-Wl,-rpath," + str(PKG), "-Wl,-rpath,$ORIGIN/../../rust-pseudoaligner_amd", "-o", str(ABI_CHECK_BIN)])
    return ABI_CHECK_BIN


if __name__ == "__main__":
    import sys
    print(build_product("--force" in sys.argv))
