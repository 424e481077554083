"""Per-launch medians of the K1 fast-path kernels from a rocprofv3 kernel trace, grouped by launch sequence."""
import csv, glob, collections, statistics, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
      for r in rows if 'k1_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
names = [('F' if 'core_fwd' in k[0] else 'B' if 'core_bwd' in k[0] else 'G') + str(k[2]) for k in ks]
starts = [i for i in range(len(ks) - 1) if names[i + 1].startswith('F')]
sigs = collections.defaultdict(list)
for a, b in zip(starts, starts[1:] + [len(ks)]):
    sigs[tuple(names[a:b])].append([k[1] for k in ks[a:b]])
for sig, v in sigs.items():
    med = [statistics.median(c) for c in zip(*v)]
    print(len(v), sig, ["%.2f" % m for m in med], "sum %.1f" % sum(med))
