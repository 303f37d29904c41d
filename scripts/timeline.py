"""Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: kernels of one steady-state step with their start offset,
duration and the idle gap before each (where does a small scene's step time go?).

    python scripts/timeline.py <kernel_trace.csv> [marker-kernel-substring]
"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "project_pack_fwd"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 2:]                                   # steady state: second half of the run
if not steps:
    sys.exit("no steps found")
spans = [int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps]
period = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(steps[:-1], steps[1:])]
busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) for s in steps]
print(f"steps analysed {len(steps)}: period median {statistics.median(period)/1e3:.1f} us, first-to-last-kernel span "
      f"{statistics.median(spans)/1e3:.1f} us, kernel busy {statistics.median(busy)/1e3:.1f} us, kernels/step {statistics.median(len(s) for s in steps)}")
mid = sorted(zip(period, steps[:-1]), key=lambda t: t[0])[len(period) // 2][1]
t0 = int(mid[0]["Start_Timestamp"]); prev = t0
for r in mid:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"  +{(a - t0)/1e3:7.1f} us  gap {(a - prev)/1e3:6.1f}  dur {(b - a)/1e3:6.1f}  {name}")
    prev = b
