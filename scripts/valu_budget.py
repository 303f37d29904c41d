"""Class-weighted vector-issue budget of the compositing kernels.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-fast-math -S --cuda-device-only -o /tmp/blend.s \
        collab_splats_amd/csrc/blend.hip
    python scripts/valu_budget.py /tmp/blend.s profiles/pmc_traffic.json > profiles/r04_valu_budget.json

`valu_issue_frac` used to multiply every vector instruction by 4 cycles.  The hardware guide's nominal rate for plain fp32 on a
SIMD-32 datapath is 2 cycles per wave-64 instruction; the micro-benchmark of this repository (scripts/ubench/valu_rates.hip,
4 waves per SIMD, 2.4 GHz) measured per CLASS: plain 2.5, DPP add 4.5, packed fp32 4.9, compare 4.0 (compare + select 8.1 the
pair), transcendental 8.5, permlane swap 7.5.  This script

  * takes the instruction-class counts of the hottest loop of each kernel from `hipcc -S` (scripts/isa_table.py),
  * prices a trip: sum(count x cost) for the part every trip runs and for the part behind the "some pixel contributes"
    branch (taken with probability `contrib`, measured: 8 staged entries in 9),
  * turns the PMC instruction count of a launch (profiles/pmc_traffic.json: SQ_INSTS_VALU, GRBM_GUI_ACTIVE) into the share of
    SIMD issue cycles under four rules: flat 4 cycles (the old figure), the guide's nominal 2, the class-weighted mean of the
    loop, and -- a MEASUREMENT rather than a model -- SQ_ACTIVE_INST_VALU (quad-cycles the vector ALU was busy).
"""
import json
import re
import sys
from collections import Counter

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import isa_table as T                                             # noqa: E402

COST = {                                                          # cycles per wave instruction (scripts/ubench/valu_rates.hip)
    "plain fp32/int VALU (fma/mul/add/max/min)": 2.5,
    "v_pk_* (packed fp32, 2 issue slots)": 4.9,
    "v_*_dpp": 4.5,
    "v_permlane*_swap": 7.5,
    "transcendental (v_exp/v_rcp)": 8.5,
    "v_cmp*": 4.0,
    "v_cndmask": 4.1,
    "v_mov / lane reads": 2.5,
    "other VALU": 2.5,
}
# (forward: the trip loop is taken from the plain-record instantiation -- the on-demand-colour one has the same trip loop, but
# its innermost loop with the most vector instructions is the SH evaluation walk, which the "hottest loop" rule would pick)
KERNELS = {"blend_fwd": "blend_fwd_kernelILi4ELi2ELi0ELb0E", "blend_bwd": "blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E"}
CONTRIB = 8.0 / 9.0                                               # staged entries some pixel takes
N_SIMD = 1024


def loop_classes(path, needle):
    lines = T.kernel_lines(path, needle)
    if not lines:
        raise SystemExit(f"kernel {needle} not found in {path}")
    label_at = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
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
    skipped = set()
    for i in range(a, b):
        m = re.match(r"^\s*s_cbranch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in label_at and i < label_at[m.group(1)] <= b:
            skipped.update(range(i + 1, label_at[m.group(1)]))
    always, cond = Counter(), Counter()
    for i in range(a, b + 1):
        if is_inst(lines[i]):
            cls = T.classify(lines[i].split()[0])
            if cls in COST:
                (cond if i in skipped else always)[cls] += 1
    return always, cond


def main():
    asm, pmc_path = sys.argv[1], sys.argv[2]
    pmc = json.load(open(pmc_path))
    out = {"_note": __doc__.split("This script")[0].strip().split("\n\n", 1)[1].replace("\n", " "),
           "cost_cycles_per_instruction": COST, "contributing_trip_fraction": round(CONTRIB, 4), "kernels": {}}
    for name, needle in KERNELS.items():
        always, cond = loop_classes(asm, needle)
        n_a, n_c = sum(always.values()), sum(cond.values())
        cyc_a = sum(COST[k] * v for k, v in always.items())
        cyc_c = sum(COST[k] * v for k, v in cond.items())
        mean_cost = (cyc_a + CONTRIB * cyc_c) / (n_a + CONTRIB * n_c)
        per_class = {k: {"every_trip": always[k], "contributing_trip": cond[k], "cycles": round(COST[k] * (always[k] + cond[k]), 1)}
                     for k in sorted(set(always) | set(cond), key=lambda k: -COST[k] * (always[k] + cond[k]))}
        entry = {"instructions_every_trip": n_a, "instructions_contributing_trip": n_c,
                 "cycles_every_trip": round(cyc_a, 1), "cycles_contributing_trip": round(cyc_c, 1),
                 "mean_cycles_per_instruction": round(mean_cost, 3), "by_class": per_class}
        c = pmc.get("_counters", {}).get(name)
        if c and "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            simd_cycles = N_SIMD * c["GRBM_GUI_ACTIVE"] / 8.0
            insts = c["SQ_INSTS_VALU"]
            entry["issue_frac"] = {
                "flat_4_cycles (the old valu_issue_frac)": round(insts * 4.0 / simd_cycles, 4),
                "guide_nominal_2_cycles": round(insts * 2.0 / simd_cycles, 4),
                "class_weighted (sum count x cost of the loop)": round(insts * mean_cost / simd_cycles, 4),
                "measured_busy (SQ_ACTIVE_INST_VALU x 4)": (round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4)
                                                            if "SQ_ACTIVE_INST_VALU" in c else None),
            }
            entry["pmc_git_rev"] = pmc.get("git_rev")
        out["kernels"][name] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
