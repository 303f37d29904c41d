"""Build recipe for libmisplat.so (hipcc, gfx950 only, in-tree).

``python -m collab_splats_amd.build`` or ``build()``.  The library is written next to this
file so it travels with the source tree; it is git-ignored (history stays source-only).
hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libmisplat.so")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-I", INCLUDE,
          "-Wall", "-Wno-unused-function", "-fno-fast-math"]
# project.hip: fixed IEEE operation order (no FMA contraction) so the integer binning stage is
# reproducible bit for bit against an independent fp32 implementation of the same order.
SOURCES = {
    "project.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    "bucket.hip": [],
    "raster.hip": [],
    # blend.hip: the compositing loops are written for the issue slots they cost -- packed (v_pk_*) where two pixels run, plain
    # where one does; the SLP vectoriser re-packs the one-pixel bodies across unrelated values (103 packed + 83 moves in the
    # backward's trip loop, 154 registers; without it 37 + 17 and 115)
    "blend.hip": ["-fno-slp-vectorize"],
    "epilogue.hip": [],
    "ssim.hip": [],
    "optim.hip": [],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(INCLUDE, "misplat.h"), os.path.join(CSRC, "sh_eval.h"), os.path.join(CSRC, "internal.h")]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s, __file__] + headers):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    try:                                              # the source revision travels with the library (the GPU box has no .git)
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "collab_splats_amd", "include", "bench.py"]) != 0
        with open(os.path.join(HERE, "_build_rev.txt"), "w") as f:
            f.write(rev + ("+" if dirty else "") + "\n")
    except Exception:
        pass
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
