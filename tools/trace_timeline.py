"""Steady-state timeline of a rocprofv3 --kernel-trace CSV of the pipelined bench: per queue, the kernels of the last
steps with start / end relative to the window, and how many kernels run concurrently over time.
usage: python tools/trace_timeline.py <kernel_trace.csv> [window_ms=6]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"])) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(win * 1e6) - int(3e6)      # a window a few ms before the end (still inside the timed steps)
t1 = t0 + int(win * 1e6)
sel = [e for e in ev if e[0] >= t0 and e[0] < t1]
short = lambda n: n.replace("ilcc::", "").split("(")[0].replace("void ", "")[:22]
for q in sorted({e[3] for e in sel}):
    print("queue", q)
    for s, e, n, _, gx, gy in [x for x in sel if x[3] == q]:
        print("   %8.3f .. %8.3f  (%7.3f ms)  %-22s grid %d" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, short(n), gx * max(gy, 1)))
# concurrency profile
pts = []
for s, e, n, q, gx, gy in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cur, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
print("kernels running concurrently (share of the window):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# time with a K6 full pass running
full = [(s, e) for s, e, n, q, gx, gy in sel if "k6_grid_cost" in n and gx * max(gy, 1) > 1000000]
print("K6 full passes in window: %d, total %.3f ms of %.1f ms" % (len(full), sum(e - s for s, e in full) / 1e6, win))
