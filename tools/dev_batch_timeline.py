"""One 128-frame batch ALONE on the chip (synchronous calls): median HIP-event stage times, the un-contended
counterpart of bench.py's stage_ms_last_batch_overlapped.  usage: python tools/dev_batch_timeline.py [solver 1|0] [reps] [frames=128]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
solver = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
d_c = torch.from_numpy(clouds).cuda(); d_k = torch.from_numpy(clicks).cuda()
p = N.default_params(); p.solver = solver
est = LidarCornersBatch(F, 28800, p)
est.reserve(1792, 2560)
rows = []
for r in range(reps):
    est.extract_device(d_c.data_ptr(), F, 28800, d_k.data_ptr())
    t = est.timing()
    rows.append([t.roi_crop, t.cluster, t.ransac_plane, t.plane_frame_hist, t.grid_cost, t.refine_corners, t.total])
m = np.median(np.array(rows[5:]), axis=0)
print(json.dumps(dict(zip(("roi_crop", "cluster", "ransac_plane", "plane_frame_hist", "grid_cost", "refine_corners", "total"), [round(float(v), 4) for v in m]))))
