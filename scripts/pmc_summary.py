"""Summarise rocprofv3 --pmc CSVs: per kernel, per counter, mean over dispatches."""
import csv, glob, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("blend_fwd", "blend_bwd", "slab_reduce", "color_sh_kernel<true", "color_sh_kernel<false", "project_pack_fwd", "project_pack_bwd", "tile_emit", "tile_offsets", "tile_sort_reg", "tile_sort_rest", "tile_count", "gauss_bwd_sparse", "bucket_count", "bucket_rows", "bucket_tile_count", "bucket_tile_fill", "bucket_tile_scan", "unit_order", "rocprim"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in out.items():
    print(k, {c: f"{v:.4g}" for c, v in sorted(d.items())})
json.dump(out, open(root + "/summary.json", "w"), indent=1)
