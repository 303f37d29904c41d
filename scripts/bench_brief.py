"""One-screen summary of a bench.py JSON line (file path or stdin)."""
import json
import sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print("headline", d["value"], "Msplats/s", d["ms_per_step"], "ms/step, device median", d["device_ms_median"])
for k, v in d.get("variants", {}).items():
    print("  ", k, v)
c = d["config"]
print("path", c.get("path"), c.get("graph_cache"))
r = d["roofline"]
print("roof", r["kernel"], r["kernel_ms"], "frac", r["frac"], "traversed", r.get("traversed_entries"), r.get("achieved_traversed"),
      "copy", r.get("copy_roof_variants_GBs"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"].get("value"), "parity", d.get("grad_max_rel_err", {}).get("value"), d.get("grad_max_rel_err", {}).get("on"))
print("isects per view", c.get("n_isects_per_view"))
if "allreduce_ms" in d:
    print("allreduce_ms", d["allreduce_ms"])
