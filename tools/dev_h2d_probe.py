"""Where the H2D-inclusive leg loses its 6-9 % of the link: the same submit / wait pipeline (pinned host batches, four in flight) with
(a) clicks far from every point (every kernel returns early: the copies alone, through the library), (b) the GRID mode, (c) the
reference mode, (d) raw hipMemcpyAsync on one stream; GB/s each.  usage: dev_h2d_probe.py [steps=24]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
F, NP, NB = 1024, 28800, 4
import multiprocessing as mp
def gen(a):
    return synth.make_batch(a[1], seed=a[0])[:2]
with mp.get_context("fork").Pool(16) as pool:
    parts = pool.map(gen, [(0xC0FFEE + lo, 32) for lo in range(0, NB * F, 32)])
clouds = np.concatenate([p[0] for p in parts]).reshape(NB, F, NP, 4); clicks = np.concatenate([p[1] for p in parts]).reshape(NB, F, 3)
pinned = [torch.from_numpy(clouds[b]).pin_memory() for b in range(NB)]
pk = [torch.from_numpy(clicks[b]).pin_memory() for b in range(NB)]
far = [torch.full((F, 3), 1000.0).pin_memory() for b in range(NB)]
nbytes = pinned[0].numel() * 4
def leg(name, solver, ck, depth=4):
    p = N.default_params(); p.solver = solver
    e = LidarCornersBatch(F, NP, p); e.set_result_mode(N.RESULTS_COMPACT); e.reserve(1792, 2560)
    def run(n):
        infl = []
        for s in range(n):
            infl.append(e.submit_host(pinned[s % NB].data_ptr(), F, NP, ck[s % NB].data_ptr()))
            if len(infl) == depth: e.wait_compact(infl.pop(0))
        while infl: e.wait_compact(infl.pop(0))
    run(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-46s %.2f GB/s  %.3f ms per step  %.0f frames/s" % (name, nbytes * steps / dt / 1e9, 1e3 * dt / steps, F * steps / dt), flush=True)
    e.close()
bufs = [torch.empty_like(pinned[0], device="cuda") for _ in range(4)]
cs = torch.cuda.Stream()
def raw(n):
    with torch.cuda.stream(cs):
        for b in range(n): bufs[b % 4].copy_(pinned[b % NB], non_blocking=True)
        cs.synchronize()
raw(4); t0 = time.perf_counter(); raw(steps); dt = time.perf_counter() - t0
print("%-46s %.2f GB/s" % ("raw copies, one stream", nbytes * steps / dt / 1e9), flush=True)
leg("library, no work (clicks far away)", N.SOLVER_GRID, far)
leg("library, GRID", N.SOLVER_GRID, pk)
leg("library, REFERENCE_LOCAL", N.SOLVER_REFERENCE_LOCAL, pk)
leg("library, REFERENCE_LOCAL, 2 in flight", N.SOLVER_REFERENCE_LOCAL, pk, depth=2)
leg("library, GRID, 2 in flight", N.SOLVER_GRID, pk, depth=2)
raw(4); t0 = time.perf_counter(); raw(steps); dt = time.perf_counter() - t0
print("%-46s %.2f GB/s" % ("raw copies again", nbytes * steps / dt / 1e9), flush=True)

# copy cadence from the library's HIP-event timeline (GRID, four in flight): when each batch's input copy was done, when its last
# kernel ended -- a copy-bound pipeline shows copy completions one copy time apart
import ctypes as C
p = N.default_params()
e = LidarCornersBatch(F, NP, p); e.set_result_mode(N.RESULTS_COMPACT); e.reserve(1792, 2560)
lib = N.lib()
infl = []
def run(n):
    for s in range(n):
        infl.append(e.submit_host(pinned[s % NB].data_ptr(), F, NP, pk[s % NB].data_ptr()))
        if len(infl) == 4: e.wait_compact(infl.pop(0))
    while infl: e.wait_compact(infl.pop(0))
run(8)
lib.ilcc_debug_timeline_enable(e._h, 1)
run(16)
rows = (C.c_double * (15 * 64))()
n = lib.ilcc_debug_timeline_fetch(e._h, rows, 64)
T = np.array(rows[:15 * n]).reshape(n, 15)
T = T[np.argsort(T[:, 1])]
print("batch: copy done (ms), delta to previous copy done, kernels K1 start -> end, batch end - copy done")
for i in range(1, n):
    print("  %2d  %8.3f  +%6.3f   life after copy %.3f ms" % (i, T[i, 1], T[i, 1] - T[i - 1, 1], T[i, 14] - T[i, 1]))
e.close()
