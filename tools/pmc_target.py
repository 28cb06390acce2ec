"""Minimal target for rocprofv3 --pmc passes: three synchronous 128-frame config-2 batches (GRID mode), one batch
alone on the chip each time -- per-dispatch counters of every kernel of the path without bench.py's other legs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
F = 128
clouds, clicks, _, _ = synth.make_batch(F, seed=0xC0FFEE)
d_c = torch.from_numpy(clouds).cuda(); d_k = torch.from_numpy(clicks).cuda()
est = LidarCornersBatch(F, 28800, N.default_params())
for _ in range(3):
    est.extract_device(d_c.data_ptr(), F, 28800, d_k.data_ptr())
est.close()
