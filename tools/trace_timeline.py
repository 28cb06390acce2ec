"""Steady-state timeline of a rocprofv3 --kernel-trace CSV of the pipelined bench: per queue, the kernels of the last
steps with start / end relative to the window, and how many kernels run concurrently over time.
usage: python tools/trace_timeline.py <kernel_trace.csv> [window_ms=6] [window_end_before_trace_end_ms=3] [summary]
`summary`: no per-kernel listing; per kernel name the count / mean / max duration inside the window, the share of the window in which
two K6 full passes overlap (they are chained: must be 0), the queues the launches landed on, the busy union and -- an occupancy
proxy -- the time-weighted sum of workgroups of all running kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"])) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
back = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
summary = len(sys.argv) > 4 and sys.argv[4] == "summary"
t0 = t_end - int(win * 1e6) - int(back * 1e6)      # a window a few ms before the end (still inside the timed steps)
t1 = t0 + int(win * 1e6)
sel = [e for e in ev if e[0] >= t0 and e[0] < t1]
short = lambda n: n.replace("ilcc::", "").split("(")[0].replace("void ", "")[:22]
for q in sorted({e[3] for e in sel}) if not summary else []:
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

if summary:
    import statistics
    by = collections.defaultdict(list)
    for s_, e_, n, q, gx, gy in sel:
        key = short(n) + ("/full" if "k6_grid_cost" in n and gx * max(gy, 1) > 1000000 else "")
        by[key].append((e_ - s_) / 1e3)
    print("window %.1f ms ending %.1f ms before the end of the trace; queues used: %s" % (win, back, sorted({e[3] for e in sel})))
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("  %-28s n %3d  mean %8.1f us  max %8.1f us  total %8.3f ms" % (k, len(v), statistics.mean(v), max(v), sum(v) / 1e3))
    # overlap of full passes with each other
    fs = sorted(full)
    ov = sum(max(0, min(fs[i][1], fs[j][1]) - max(fs[i][0], fs[j][0])) for i in range(len(fs)) for j in range(i + 1, len(fs)))
    print("  two K6 full passes overlapping: %.3f ms; start-to-start period of full passes: %s us" % (
        ov / 1e6, [round((fs[i + 1][0] - fs[i][0]) / 1e3) for i in range(len(fs) - 1)]))
    busy = sum(v for k, v in hist.items() if k > 0) / tot
    print("  busy union %.3f of the window; mean kernels running %.2f" % (busy, sum(k * v for k, v in hist.items()) / tot))
    # per queue: how many distinct kernels (streams map onto hardware queues)
    for q in sorted({e[3] for e in sel}):
        qs = [x for x in sel if x[3] == q]
        gaps = [(qs[i + 1][0] - qs[i][1]) / 1e3 for i in range(len(qs) - 1)]
        print("  queue %d: %d launches, busy %.3f ms, gaps between consecutive launches: median %.1f us, max %.1f us" % (
            q, len(qs), sum(e_ - s_ for s_, e_, *_ in qs) / 1e6, statistics.median(gaps) if gaps else 0.0, max(gaps) if gaps else 0.0))
