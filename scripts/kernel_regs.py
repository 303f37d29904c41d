"""Registers / scratch of the kernels of one source file, from `hipcc -S` (design tooling).

    python scripts/kernel_regs.py blend.hip [substring]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-ffp-contract=off"] if src == "project.hip" else (["-fno-slp-vectorize"] if src == "blend.hip" else [])
out = os.path.join(tempfile.gettempdir(), src.replace(".hip", ".s"))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-fno-fast-math",
                *extra, *[a for a in sys.argv[3:]], "-S", "--cuda-device-only", os.path.join(ROOT, "collab_splats_amd", "csrc", src), "-o", out],
               check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
filt = "c++filt"
for b in txt.split(".amdhsa_kernel ")[1:]:
    name = b.split("\n")[0]
    d = dict(re.findall(r"\.amdhsa_(\w+)\s+(\S+)", b))
    dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    dn = dn.split("(")[0][:100]
    if flt in dn:
        print(f"{dn:100s} vgpr {d.get('next_free_vgpr'):>4s} sgpr {d.get('next_free_sgpr'):>4s} scratch {d.get('private_segment_fixed_size'):>5s}")
print(out)
