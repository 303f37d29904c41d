"""Instruction-class table of a kernel's hottest loop from `hipcc -S` output (gfx950).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-fast-math -S --cuda-device-only -o /tmp/blend.s \
        collab_splats_amd/csrc/blend.hip
    python scripts/isa_table.py /tmp/blend.s 'blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E' [--dump]

Finds every backward branch of the kernel (a loop), takes the INNERMOST loop with the most vector instructions (the
per-(band, Gaussian) trip of the compositing kernels) and counts its instructions by class.  Blocks behind a
wave-uniform branch inside the loop (s_cbranch_*) are listed separately: they do not run on every trip.
"""
import re
import sys
from collections import Counter


def kernel_lines(path, needle):
    out, on = [], False
    for ln in open(path):
        if not on:
            if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", ln):
                on = True
            continue
        if ".end_amdhsa_kernel" in ln or re.match(r"^\s*\.section", ln):
            break
        out.append(ln.rstrip("\n"))
    return out


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk_* (packed fp32, 2 issue slots)"
    if "dpp" in op:
        return "v_*_dpp"
    if op.startswith("v_permlane"):
        return "v_permlane*_swap"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "transcendental (v_exp/v_rcp)"
    if op.startswith("v_cmp"):
        return "v_cmp*"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_fma", "v_fmac", "v_mul", "v_add", "v_sub", "v_mad", "v_max", "v_min")):
        return "plain fp32/int VALU (fma/mul/add/max/min)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "v_mov / lane reads"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith("ds_"):
        return "LDS (ds_*)"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "s_waitcnt / s_nop"
    if op.startswith("s_"):
        return "SALU / branch"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = kernel_lines(path, needle)
    if not lines:
        sys.exit(f"kernel {needle} not found")
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))

    def is_inst(ln):
        s = ln.strip()
        return bool(s) and not s.startswith((";", ".", "//")) and not re.match(r"^\.?\w+:", s)

    def valu(a, b):
        return sum(1 for ln in lines[a:b + 1] if is_inst(ln) and ln.strip().startswith("v_"))

    inner = [(a, b) for a, b in loops if not any((c > a or (c == a and d < b)) and d <= b and (c, d) != (a, b) and c >= a for c, d in loops)]
    a, b = max(inner, key=lambda ab: valu(*ab))
    print(f"kernel {needle}: {len(lines)} lines, {len(loops)} loops; hottest innermost loop = lines {a}..{b}")
    # split the loop body into the straight path and the blocks skipped by forward s_cbranch inside the loop
    skipped = set()
    for i in range(a, b):
        m = re.match(r"^\s*s_cbranch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in label_at and i < label_at[m.group(1)] <= b:
            skipped.update(range(i + 1, label_at[m.group(1)]))
    always, cond = Counter(), Counter()
    for i in range(a, b + 1):
        ln = lines[i]
        if not is_inst(ln):
            continue
        op = ln.split()[0]
        (cond if i in skipped else always)[classify(op)] += 1
    tot_a = sum(v for k, v in always.items() if k.startswith(("v_", "plain", "trans", "other VALU")))
    tot_c = sum(v for k, v in cond.items() if k.startswith(("v_", "plain", "trans", "other VALU")))
    print(f"{'class':48s} {'every trip':>10s} {'behind a wave-uniform branch':>30s}")
    for k in sorted(set(always) | set(cond), key=lambda k: -(always[k] + cond[k])):
        print(f"{k:48s} {always[k]:10d} {cond[k]:30d}")
    print(f"{'VALU total':48s} {tot_a:10d} {tot_c:30d}")
    if dump:
        for i in range(a, b + 1):
            print(("  ? " if i in skipped else "    ") + lines[i])


if __name__ == "__main__":
    main()
