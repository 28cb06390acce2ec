import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_camera_calibration_amd import synth, LidarCornersBatch, _native as N
F = 128
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
dev = torch.device("cuda", 0)
dc = torch.from_numpy(clouds).to(dev); dk = torch.from_numpy(clicks).to(dev)
torch.cuda.synchronize()
for prune in (0, 1, 2, 5):
    p = N.default_params(); p.grid_prune = prune
    e = LidarCornersBatch(F, 28800, p)
    for _ in range(3): e.extract_device(dc.data_ptr(), F, 28800, dk.data_ptr())
    e.reset_timing()
    for _ in range(10): e.extract_device(dc.data_ptr(), F, 28800, dk.data_ptr())
    t = e.timing()
    print("prune", prune, "noseed", os.environ.get("ILCC_K6_NOSEED"), "k6 ms %.3f" % (t.grid_cost_ms_sum / t.grid_cost_launches),
          "executed frac %.4f" % (t.grid_cost_evals_sum / t.grid_cost_evals_nominal_sum))
    e.close()
