"""Build profiles/pmc_traffic.json from a scripts/pmc_run.sh output directory (PMC passes of bench.py):
HBM bytes per launch of the compositing kernels and their VALU issue fraction.

    python scripts/pmc_traffic.py gpurun_out/<tag>/pmc <git rev> > profiles/pmc_traffic.json
"""
import collections, csv, glob, json, sys
root, rev = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "unknown")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("blend_fwd", "blend_bwd"):
            if key in r["Kernel_Name"]:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "per launch, rocprofv3 PMC in separate passes (scripts/pmc_run.sh): traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 "
                "(FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM': gfx950 reports 1/2 of wide coalesced reads; WRITE_SIZE "
                "includes the 64-byte gradient atomics); valu_issue_frac = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * "
                "GRBM_GUI_ACTIVE / 8): the share of SIMD issue cycles spent on vector ALU instructions (v_pk_* and "
                "transcendentals cost more than 4, so this is a lower bound). Workload: bench.py default.",
       "git_rev": rev, "valu_issue_frac": {}}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        out[k] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        out["valu_issue_frac"][k] = round(m["SQ_INSTS_VALU"] * 4.0 / (1024 * m["GRBM_GUI_ACTIVE"] / 8.0), 4)
    out.setdefault("_counters", {})[k] = {c: float(f"{v:.6g}") for c, v in sorted(m.items())}
print(json.dumps(out, indent=1))
