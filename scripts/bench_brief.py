"""One-screen summary of a bench.py JSON line (file path or stdin)."""
import json
import sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print("headline", d["value"], "Msplats/s", d["ms_per_step"], "ms/step, device median", d["device_ms_median"], "| fixed view",
      d.get("value_fixed_view"), d.get("ms_fixed_view"))
for k, v in d.get("variants", {}).items():
    print("  ", k, v)
c = d["config"]
print("path", c.get("path"), "graph hit rate", c.get("graph_hit_rate"))
r = d["roofline"]
print("roof", r["kernel"], r["kernel_ms"], "frac", r["frac"], r.get("bytes_basis"), "nominal", r.get("frac_nominal"), "step", r.get("step_frac"),
      r.get("step_frac_nominal"), "units", r.get("units"), "copy roof", r.get("copy_roof_GBs"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"].get("value"), "parity", d.get("grad_max_rel_err", {}).get("value"), d.get("grad_max_rel_err", {}).get("on"))
if "allreduce_ms" in d:
    print("allreduce_ms", d["allreduce_ms"])
