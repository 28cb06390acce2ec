import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
F=256
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
e = LidarCornersBatch(F, 28800, N.default_params()); e.reserve(2048,2560)
r = e.extract(clouds, clicks)
ok=[x for x in r if x.status in (0,11)]
t=np.array([x.grid_ties for x in ok]); it=np.array([x.iters_a for x in ok]); hp=np.array([x.iters_b for x in ok])
print('ties mean %.2f median %d p90 %d max %d; rounds mean %.2f median %d p90 %d max %d; hops mean %.3f' % (t.mean(), np.median(t), np.percentile(t,90), t.max(), it.mean(), np.median(it), np.percentile(it,90), it.max(), hp.mean()))
