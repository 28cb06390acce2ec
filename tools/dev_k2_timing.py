"""Per-phase cycle counters of K2 (device printf of frame 0) against a -DILCC_K2_TIMING build (tools/build_variant.sh k2t -DILCC_K2_TIMING):
    ILCC_HIP_LIB=build/ab/libilcc_hip_k2t.so python tools/dev_k2_timing.py [online]
`online`: the un-cropped clouds of get_chessboard_by_point (the hashed-cell path); default: ROI-cropped VLP-16 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
board, lidar = synth.Board(), synth.vlp16()
F = 128
clouds, clicks, gts, _ = synth.make_batch(F, lidar, board, seed=0xC0FFEE)
est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
if len(sys.argv) > 1 and sys.argv[1] == "online":
    est.reserve(2048, lidar.n_points)
    pts = np.ascontiguousarray(gts.mean(axis=1), dtype=np.float32)
    for _ in range(3):
        t0 = time.perf_counter()
        r = est.chessboard_by_point(clouds, pts)
        dt = time.perf_counter() - t0
    print("online: %.3f ms per call, cluster stage %.3f ms, found %d, second-tier frames %d of %d (3 calls)"
          % (1e3 * dt, est.timing().cluster, sum(1 for x in r if x.status == 0), est.timing().online_second_tier_frames, 3 * F))
    print("found_board", sum(x.found_board for x in r), "statuses", sorted(set((x.status, sum(1 for y in r if y.status == x.status)) for x in r)))
else:
    d_c, d_k = torch.from_numpy(clouds).cuda(), torch.from_numpy(clicks).cuda()
    for _ in range(3):
        r = est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
    print("n_roi f0", r[0].n_roi, "timing", est.timing().cluster)
