"""The trip loop of a compositing kernel from `hipcc -S` output: instruction counts by class and the text (design tooling).

    python scripts/hot_loop.py /tmp/blend.s blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E [marker] [--dump]

The loop is found as the innermost loop (a backward branch to a label) that contains the marker (default v_exp_f32;
row_half_mirror for the backward: its butterfly).
"""
import re, sys
from collections import Counter
path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", l)][0]
end = [i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i]][0]
label = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        label[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in label and label[m.group(1)] < i:
        loops.append((label[m.group(1)], i))
marker = next((a for a in sys.argv[3:] if not a.startswith("--")), "v_exp_f32")
exps = [i for i in range(start, end) if marker in lines[i]]
cands = [(a, b) for a, b in loops if any(a <= e <= b for e in exps)]
a, b = min(cands, key=lambda ab: ab[1] - ab[0])
c = Counter()
for l in lines[a:b + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or re.match(r"^\.?\w+:", t):
        continue
    op = t.split()[0]
    for pre, name in (("v_pk_", "v_pk"), ("scratch_", "SCRATCH"), ("ds_", "ds"), ("global_", "global"), ("v_cmp", "v_cmp"), ("v_cndmask", "v_cndmask"),
                      ("v_exp", "trans"), ("v_rcp", "trans"), ("v_mov", "v_mov"), ("v_readfirstlane", "v_readfirstlane"), ("v_permlane", "v_permlane"),
                      ("s_cbranch", "s_branch"), ("s_branch", "s_branch"), ("s_waitcnt", "s_waitcnt"), ("s_nop", "s_nop"), ("s_", "salu")):
        if op.startswith(pre):
            c[name] += 1
            break
    else:
        c["v_dpp" if "dpp" in t else ("v_plain" if op.startswith("v_") else op)] += 1
print(f"{needle}: loop lines {a - start}..{b - start} ({b - a + 1} lines)")
print("  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
if "--dump" in sys.argv:
    print("\n".join(lines[a:b + 1]))
