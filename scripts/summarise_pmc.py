"""Per-kernel means of rocprofv3 --pmc passes (one or more output directories) -> one CSV under profiles/.
usage: python scripts/summarise_pmc.py <out_csv> <dir> [<dir> ...]      (only kernels of this library are kept)"""
import collections, csv, glob, os, sys
out, dirs = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "moma" not in k:
                continue
            k = k.replace("void moma::(anonymous namespace)::", "moma::").replace("moma::(anonymous namespace)::", "moma::").split("(")[0]
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""),
                       r.get("LDS_Block_Size", ""))
with open(out, "w") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches", "grid", "workgroup", "vgpr", "agpr", "lds_bytes"])
    for k in sorted(res):
        for c in sorted(res[k]):
            v = res[k][c]
            w.writerow([k, c, "%.1f" % (sum(v) / len(v)), len(v), *meta[k]])
print("wrote", out)
