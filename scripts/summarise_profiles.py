"""Condense rocprofv3 output directories into the small summaries committed under profiles/.
usage: python scripts/summarise_profiles.py <round_tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]"""
import collections
import csv
import glob
import os
import sys

tag, stats_dir = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(n):
    return n.replace("void moma::(anonymous namespace)::", "moma::").replace("moma::(anonymous namespace)::", "moma::")[:140]


f = sorted(glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv")))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in rows:
        if "moma::" in r["Name"] or float(r["Percentage"]) >= 0.5:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3),
                        "%.2f" % (float(r["MaxNs"]) / 1e3), "%.2f" % float(r["Percentage"])])
    w.writerow(["TOTAL (all kernels)", "", "%.3f" % (tot / 1e6), "", "", "", "100"])
print("wrote", os.path.join(out_dir, f"{tag}_kernel_stats.csv"))

if len(sys.argv) >= 5:
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[3:5]:
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "moma" in r["Kernel_Name"]:
                    res[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out_dir, f"{tag}_k2_hbm_traffic.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches", "note"])
        for k, cs in sorted(res.items()):
            for c, v in sorted(cs.items()):
                note = ""
                if c == "FETCH_SIZE":
                    note = "KiB as reported; gfx950 reports 1/2 of wide coalesced reads -> bytes = 2*1024*value"
                if c == "WRITE_SIZE":
                    note = "KiB; bytes = 1024*value"
                w.writerow([k, c, "%.1f" % (sum(v) / len(v)), len(v), note])
    print("wrote", os.path.join(out_dir, f"{tag}_k2_hbm_traffic.csv"))
