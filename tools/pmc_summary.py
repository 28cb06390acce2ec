"""Per-kernel summary of rocprofv3 --pmc counter passes (counter_collection.csv files under <prefix>*): one CSV row per
(kernel, grid size) with, for every counter, the MEAN PER DISPATCH over its dispatches EXCLUDING the first batch's (the target's warm-up batch:
a cold dispatch once doubled a committed mean), and `<counter>_min` / `<counter>_max` over the same dispatches so that an
outlier shows.  usage: pmc_summary.py <dir prefix> [--keep-first | --drop N]"""
import csv, glob, sys, collections
prefix = sys.argv[1]
keep_first = "--keep-first" in sys.argv
drop = int(sys.argv[sys.argv.index("--drop") + 1]) if "--drop" in sys.argv else 1
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")[:60]
        key = (name, row.get("Grid_Size", ""))
        acc[key][row["Counter_Name"]].append((int(row.get("Dispatch_Id", "0") or 0), float(row["Counter_Value"])))
counters = sorted({c for v in acc.values() for c in v})
# batches the target ran = dispatches of K1's count pass (one per batch); a kernel launched k times per batch (k6_anchor's rounds)
# has k times as many dispatches: launches_per_batch, and the warm-up batch's k dispatches are dropped together
n_batches = max([max(len(lst) for lst in v.values()) for (name, _), v in acc.items() if "k1_roi_count" in name] or [1])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "dispatches", "launches_per_batch"] + counters + [c + s for c in counters for s in ("_min", "_max")])
for (name, grid), v in sorted(acc.items()):
    vals = {}
    per_batch = max(1, round(max(len(lst) for lst in v.values()) / n_batches))
    for c, lst in v.items():
        lst = [x for _, x in sorted(lst)]
        vals[c] = lst[drop * per_batch:] if (len(lst) > drop * per_batch and not keep_first) else lst
    n = max(len(x) for x in vals.values())
    w.writerow([name, grid, n, per_batch] + [("%.6g" % (sum(vals[c]) / len(vals[c]))) if c in vals else "" for c in counters] +
               [("%.6g" % f(vals[c])) if c in vals else "" for c in counters for f in (min, max)])
