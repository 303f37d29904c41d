"""Build a variant of libmisplat.so with extra -D flags on chosen source files (A/B experiments on the GPU box):
    python scripts/build_variant.py <name> <file.hip>[,<file2.hip>] -DFOO=1 [-DBAR=2 ...]
-> collab_splats_amd/_exp/libmisplat_<name>.so   (use with MISPLAT_LIB=... ; scripts/ab_lib.sh)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import build as B

name, files, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build()
exp = os.path.join(B.HERE, "_exp")
os.makedirs(exp, exist_ok=True)
objs = []
for src, extra in B.SOURCES.items():
    o = os.path.join(B.OBJ, src.replace(".hip", ".o"))
    if src in files:
        o = os.path.join(exp, f"{name}_{src.replace('.hip', '.o')}")
        subprocess.check_call([B._hipcc()] + B.COMMON + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", o])
    objs.append(o)
out = os.path.join(exp, f"libmisplat_{name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
print(out)
