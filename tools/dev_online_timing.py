"""The online caller (get_chessboard_by_point on un-cropped clouds) synchronous and with 1 / 2 / 4 calls in flight: ms per call, host time per
submit / wait, the H2D copy of a call alone, stage times.  EXTRA=1: beside another live handle.  (through gpurun from the repo root)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lidar_camera_calibration_amd import LidarCornersBatch
from lidar_camera_calibration_amd import _native as N
clouds, clicks, gts = bench.generate(2, 128, 0xC0FFEE, 16)
F, n_points = 128, 28800
p = N.default_params(); p.gray_rate = 2.4
if os.environ.get('EXTRA'):
    big = LidarCornersBatch(1024, n_points, p, device=0); big.set_result_mode(N.RESULTS_COMPACT); big.reserve(1792, 2560)
    dd = torch.from_numpy(np.tile(clouds[:128].reshape(128, n_points, 4), (8, 1, 1))).cuda(); dk = torch.from_numpy(np.tile(clicks[:128].reshape(128, 3), (8, 1))).cuda()
    for _ in range(6): big.wait(big.submit_device(dd.data_ptr(), 1024, n_points, dk.data_ptr()), want_results=True)
est = LidarCornersBatch(F, n_points, p, device=0)
pts = np.ascontiguousarray(gts[:F].mean(axis=1), dtype=np.float32)
c = np.ascontiguousarray(clouds[:F].reshape(F, n_points, 4))
pc = [torch.from_numpy(c).pin_memory() for _ in range(4)]
pp = [torch.from_numpy(pts).pin_memory() for _ in range(4)]
for _ in range(3): est.chessboard_by_point(c, pts)
def run(n, depth):
    infl = []; ts = []; tw = []
    t0 = time.perf_counter()
    for k in range(n):
        if len(infl) == depth:
            a = time.perf_counter(); est.wait_chessboard_by_point(infl.pop(0)); tw.append(time.perf_counter() - a)
        a = time.perf_counter(); infl.append(est.submit_chessboard_by_point(pc[k % 4].data_ptr(), F, n_points, pp[k % 4].data_ptr())); ts.append(time.perf_counter() - a)
    while infl:
        a = time.perf_counter(); est.wait_chessboard_by_point(infl.pop(0)); tw.append(time.perf_counter() - a)
    dt = time.perf_counter() - t0
    print("depth", depth, "ms/call %.3f" % (1e3 * dt / n), "submit median %.3f ms" % (1e3 * np.median(ts)), "wait median %.3f ms" % (1e3 * np.median(tw)))
for d in (1, 2, 4):
    run(8, d); run(40, d)
# plain H2D of the same bytes
x = torch.empty_like(pc[0], device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(20): x.copy_(pc[k % 4], non_blocking=True)
torch.cuda.synchronize(); print("H2D alone ms per 59 MB: %.3f" % (1e3 * (time.perf_counter() - t0) / 20))
tm = est.timing(); print({k: round(getattr(tm, k), 3) for k in ("roi_crop", "cluster", "ransac_plane", "plane_frame_hist", "total")})
