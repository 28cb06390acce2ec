"""Light target for rocprofv3 --pmc / --kernel-trace passes: synchronous batches (GRID mode), one batch alone on the chip each
time -- per-dispatch counters of every kernel of the path without bench.py's other legs.  The FIRST batch (config 5: the first
two) is a warm-up that tools/pmc_summary.py drops: round 3's config-5 file averaged a first dispatch in that issued 4.3 x the K6
instructions -- its handle was reserved for 4500 labelled points and these frames hold up to ~5.4 k, so the frames above the
reserved capacity walked their points through L2 until the handle had grown.
usage: pmc_target.py [frames_per_batch=512] [config=2|5] [solver=1|0]     (the batch sizes bench.py runs: 1024 / 128;
solver 0 = ILCC_SOLVER_REFERENCE_LOCAL: K7a instead of K6 + K7r)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
config = int(sys.argv[2]) if len(sys.argv) > 2 else 2
F = int(sys.argv[1]) if len(sys.argv) > 1 else (1024 if config == 2 else 128)
params = N.default_params()
params.solver = int(sys.argv[3]) if len(sys.argv) > 3 else N.SOLVER_GRID
if config == 5:      # BASELINE configs[4], as bench.py --config 5 sets it up
    lidar = synth.hdl64()
    clouds, clicks, _, _ = synth.make_batch(F, lidar, synth.Board(9, 12, 0.10), seed=0xC0FFEE, range_m=(2.0, 3.0),
                                            yaw_deg=25.0, pitch_deg=15.0, roll_deg=30.0)
    params.board_w, params.board_h, params.grid_length = 9, 12, 0.10
    params.n_th = params.n_ty = params.n_tz = 129
    params.th_min, params.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    params.ty_min = params.tz_min = -0.10
    params.ty_step = params.tz_step = 0.10 / 64
    n = lidar.n_points
else:
    clouds, clicks, _, _ = synth.make_batch(F, seed=0xC0FFEE)
    n = 28800
d_c = torch.from_numpy(clouds).cuda(); d_k = torch.from_numpy(clicks).cuda()
est = LidarCornersBatch(F, n, params)
# every dispatch takes the steady-state kernels: the capacities are reserved exactly as bench.py reserves them (config 5:
# 6000 labelled points = 72 KB per K6 workgroup, two per CU like the bench's; reserving the maximum, 8192 = 96 KB, would
# profile a one-workgroup-per-CU full pass the bench never runs).  The first batch is still a warm-up (pmc_summary.py drops it).
est.reserve(6400, 20000) if config == 5 else est.reserve(1792, 2560)
for _ in range(4):
    est.extract_device(d_c.data_ptr(), F, n, d_k.data_ptr())
t = est.timing()
print("pmc_target config %d, %d frames: grid_cost %.4f ms, total %.4f ms" % (config, F, t.grid_cost, t.total))
est.close()
