"""Per-kernel means of rocprofv3 --pmc counter passes (counter_collection.csv files under <prefix>*): one CSV row
per (kernel, grid size) with the mean of every counter over its dispatches.  usage: pmc_summary.py <dir prefix>"""
import csv, glob, sys, collections
prefix = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")[:60]
        key = (name, row.get("Grid_Size", ""))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "dispatches"] + counters)
for (name, grid), v in sorted(acc.items()):
    n = max(len(x) for x in v.values())
    w.writerow([name, grid, n] + [("%.6g" % (sum(v[c]) / len(v[c]))) if c in v else "" for c in counters])
