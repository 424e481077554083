"""Raw per-launch rows of one kernel from rocprofv3 --pmc passes (one counter per directory) -> one CSV under profiles/.
usage: python scripts/pmc_rows.py <out_csv> <kernel substring> <dir> [<dir> ...]"""
import csv, glob, os, sys
out, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = []
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        n = 0
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                rows.append((r["Counter_Name"], n, r["Dispatch_Id"], float(r["Counter_Value"]), r.get("Grid_Size", "")))
                n += 1
with open(out, "w") as o:
    w = csv.writer(o)
    w.writerow(["counter", "launch", "dispatch_id", "value_KB", "grid", "note"])
    for c, n, disp, v, g in rows:
        w.writerow([c, n, disp, "%.1f" % v, g, "bytes = 2*1024*value (gfx950 FETCH_SIZE correction)" if c == "FETCH_SIZE" else "bytes = 1024*value"])
print("wrote", out, len(rows), "rows")
