"""Per-wavefront cycle budget of K6's full pass (s_memtime build): where a workgroup's life goes.
Build the instrumented library first (tools/build_variant.sh k6timing -DILCC_K6_TIMING) and run on the GPU box:
    ILCC_HIP_LIB=build/ab/libilcc_hip_k6timing.so python tools/dev_k6_timing.py [frames] [config=2|5]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N

import numpy as np
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
config = int(sys.argv[2]) if len(sys.argv) > 2 else 2
params = N.default_params()
if config == 5:      # BASELINE configs[4], as bench.py --config 5 sets it up
    board, lidar = synth.Board(9, 12, 0.10), synth.hdl64()
    clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=0xC0FFEE, range_m=(2.0, 3.0), yaw_deg=25.0, pitch_deg=15.0, roll_deg=30.0)
    params.board_w, params.board_h, params.grid_length = 9, 12, 0.10
    params.n_th = params.n_ty = params.n_tz = 129
    params.th_min, params.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    params.ty_min = params.tz_min = -0.10
    params.ty_step = params.tz_step = 0.10 / 64
else:
    board, lidar = synth.Board(), synth.vlp16()
    clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=0xC0FFEE)
est = LidarCornersBatch(F, lidar.n_points, params, device=0)
d_c, d_k = torch.from_numpy(clouds).cuda(), torch.from_numpy(clicks).cuda()
lib = N.lib()
prof = (C.c_ulonglong * 16)()
for _ in range(3):
    est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
lib.ilcc_debug_k6_profile(prof, 1)
reps = 1   # one record per wavefront of the LAST launch
est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
lib.ilcc_debug_k6_profile(prof, 1)
p = [float(x) for x in prof]
waves, tot = p[0], p[1]
print("k6 stage (HIP events, instrumented build): %.4f ms" % est.timing().grid_cost)
print("wavefronts per launch %.0f, mean life %.0f cycles, labelled points per frame %.0f (interior class %.0f)"
      % (waves / reps, tot / waves, p[10] / waves, p[11] / waves))
print("share of wavefront cycles: staging %.1f %%, rejected tiles %.1f %%, surviving tiles %.1f %%, epilogue %.1f %%, other %.1f %%"
      % (100 * p[2] / tot, 100 * p[4] / tot, 100 * p[6] / tot, 100 * p[9] / tot, 100 * (tot - p[2] - p[4] - p[6] - p[9]) / tot))
print("tiles per wavefront: %.1f rejected at the first test (%.0f cycles each), %.2f survive it (%.0f cycles and %.0f walk positions each; "
      "%.3f run to the end)" % (p[3] / waves, p[4] / max(p[3], 1), p[5] / waves, p[6] / max(p[5], 1), p[7] / max(p[5], 1), p[8] / waves))
if p[12]:
    print("bound tests that let a tile walk on (after the first): %.1f per wavefront; candidates still alive at them: mean %.2f of 16; "
          "<= 4 alive at %.1f %%, <= 2 alive at %.1f %% of these tests" % (p[12] / waves, p[13] / p[12], 100 * p[14] / p[12], 100 * p[15] / p[12]))
