import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
board, lidar = synth.Board(), synth.vlp16()
F = 128
clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=0xC0FFEE)
est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
d_c, d_k = torch.from_numpy(clouds).cuda(), torch.from_numpy(clicks).cuda()
for _ in range(3):
    r = est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
print("n_roi f0", r[0].n_roi, "timing", est.timing().cluster)
