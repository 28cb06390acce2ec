"""The real timeline of the pipelined bench at a given depth, WITHOUT a profiler (rocprofv3's kernel trace changes the very
thing in question: under it the depth cliff disappears).  Uses ilcc_debug_timeline_* (HIP-event times of every batch relative
to one reference event) and the host's own clock around submit / wait.
usage: [ILCC_HIP_LIB=build/ab/libilcc_hip_s8.so] python tools/dev_depth_timeline.py DEPTH [steps=40] [out.json] [config=2|5]"""
import ctypes as C, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
CONFIG = int(sys.argv[4]) if len(sys.argv) > 4 else 2
F, n_points = (1024, 28800) if CONFIG == 2 else (64, 131072)
clouds, clicks, gts = bench.generate(CONFIG, F, 0xC0FFEE, 16)
import torch
from lidar_camera_calibration_amd import LidarCornersBatch
from lidar_camera_calibration_amd import _native as N
dev = torch.device("cuda", 0)
d_cloud = torch.from_numpy(clouds).to(dev)
d_click = torch.from_numpy(clicks).to(dev)
params = N.default_params()
if CONFIG == 5:
    params.board_w, params.board_h, params.grid_length = 9, 12, 0.10
    params.n_th = params.n_ty = params.n_tz = 129
    params.th_min, params.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    params.ty_min = params.tz_min = -0.10
    params.ty_step = params.tz_step = 0.10 / 64
est = LidarCornersBatch(F, n_points, params, device=0)
est.set_result_mode(N.RESULTS_COMPACT)
est.reserve(6000, 20000) if CONFIG == 5 else est.reserve(1792, 2560)
L = est._lib


def run(n, log=None):
    inflight = []
    for s in range(n):
        t0 = time.perf_counter()
        tk = est.submit_device(d_cloud.data_ptr(), F, n_points, d_click.data_ptr())
        t1 = time.perf_counter()
        inflight.append((tk, t0, t1))
        if len(inflight) == DEPTH:
            tk, a, b = inflight.pop(0)
            w0 = time.perf_counter()
            est.wait_compact(tk)
            if log is not None:
                log.append((a, b, w0, time.perf_counter()))
    while inflight:
        tk, a, b = inflight.pop(0)
        w0 = time.perf_counter()
        est.wait_compact(tk)
        if log is not None:
            log.append((a, b, w0, time.perf_counter()))


run(60)
torch.cuda.synchronize()
assert L.ilcc_debug_timeline_enable(est._h, 1) == 0
host = []
t_ref = time.perf_counter()
run(steps, host)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_ref
rows = np.zeros((steps + 8, N.TIMELINE_COLS))
n = L.ilcc_debug_timeline_fetch(est._h, rows.ctypes.data_as(C.POINTER(C.c_double)), len(rows))
rows = rows[:n]
host = 1e3 * (np.array(host) - t_ref)
cols = ["start", "k1count", "k1", "k2", "k3", "k45", "k5w", "seed", "refine", "anchor", "prepass", "full_start", "full_end", "end"]
T = rows[:, 1:]
mid = slice(8, n - 8)   # steady state
period = float(np.median(np.diff(T[mid, 0])))
out = {
    "depth": DEPTH, "slots_in_library": int(rows[:, 0].max()) + 1, "frames_per_s": F * steps / t_all, "ms_per_step_host": 1e3 * t_all / steps,
    "period_ms_start_to_start": period,
    "period_ms_full_pass_start_to_start": float(np.median(np.diff(T[mid, 11]))),
    "batch_life_ms_first_to_last_kernel": float(np.median(T[mid, 13] - T[mid, 0])),
    "batches_resident_on_gpu": float(np.median(T[mid, 13] - T[mid, 0])) / period,
    "span_ms": {c: float(np.median(T[mid, i + 1] - T[mid, i])) for i, c in enumerate(cols[1:])},
    "chain_wait_ms": float(np.median(T[mid, 11] - T[mid, 10])),
    "full_pass_ms": float(np.median(T[mid, 12] - T[mid, 11])),
    "chain_idle_ms_between_full_passes": float(np.median(T[mid, 11][1:] - T[mid, 12][:-1])),
    "full_pass_waited_for_its_own_front_end_share": float(np.mean((T[mid, 11] - T[mid, 10]) < 0.02)),
    "host_ms": {"submit": float(np.median(host[:, 1] - host[:, 0])), "wait_call": float(np.median(host[:, 3] - host[:, 2])),
                "submit_to_wait_returned": float(np.median(host[:, 3] - host[:, 0]))},
    "what": "HIP-event times per batch (median over the steady-state batches); span_ms[c] = time from the previous event to c on the "
            "batch's own stream (full_start: the wait for the previous batch's full pass); chain idle = gap between one full pass's "
            "end and the next one's start",
}
print(json.dumps(out))
if len(sys.argv) > 3 and sys.argv[3] != "-":
    json.dump(out, open(sys.argv[3], "w"), indent=1)
# the last few batches, absolute
for r in rows[-6:]:
    print("slot %d: " % r[0] + " ".join("%s %.3f" % (c, v - rows[-6, 1]) for c, v in zip(cols, r[1:])))
