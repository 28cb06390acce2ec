"""BASELINE config 5 ("roofline run"): dense 64-ring cloud (131 072 points), 11 x 8-corner board @ 0.10 m,
fine grid 129 x 129 x 129 candidates x 2 phases.  Prints one JSON line; not the bench contract."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
board = synth.Board(9, 12, 0.10)
lidar = synth.hdl64()
rng = np.random.default_rng(5)
clouds, clicks, gts = [], [], []
for f in range(F):
    pose = synth.random_pose(rng, range_m=(2.2, 3.0), yaw_deg=20, pitch_deg=15, roll_deg=25)
    clouds.append(synth.make_frame(lidar, board, pose, 500 + f)); clicks.append(synth.make_click(pose, 500 + f))
    gts.append(synth.true_corners(pose, board))
clouds, clicks = np.stack(clouds), np.stack(clicks)
p = N.default_params()
p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
g = 0.10
p.n_th = p.n_ty = p.n_tz = 129
p.th_min, p.th_step = -np.radians(16.0), np.radians(0.25)
p.ty_min = p.tz_min = -g
p.ty_step = p.tz_step = g / 64
est = LidarCornersBatch(F, lidar.n_points, p, device=0)
d_c, d_k = torch.from_numpy(clouds).cuda(), torch.from_numpy(clicks).cuda()
for _ in range(2):
    res = est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
est.reset_timing()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    res = est.extract_device(d_c.data_ptr(), F, lidar.n_points, d_k.data_ptr())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
tm = est.timing()
ok = [f for f in range(F) if res[f].status == 0]
err = [synth.corner_error(res[f].corners_array(), gts[f], board) for f in ok]
L = max(1, tm.grid_cost_launches)
print(json.dumps({"config": "config 5: 64 rings x 2048 azimuths, 11x8 corners @0.10 m, grid 129^3 x 2", "frames": F, "steps": steps,
                  "frames_per_s": F * steps / dt, "ms_per_step_synchronous": 1e3 * dt / steps,
                  "k6_ms_per_step": tm.grid_cost_ms_sum / L, "labelled_points_mean": float(np.mean([res[f].n_black + res[f].n_white for f in ok])),
                  "evals_nominal_per_step": tm.grid_cost_evals_nominal_sum / L, "evals_executed_per_step": tm.grid_cost_evals_sum / L,
                  "executed_fraction": tm.grid_cost_evals_sum / max(1, tm.grid_cost_evals_nominal_sum),
                  "frames_ok": len(ok), "median_corner_error_mm": 1e3 * float(np.median(err)) if err else None,
                  "max_corner_error_mm": 1e3 * max(err) if err else None,
                  "algorithmic_GBps": (16 * lidar.n_points + 12 * 88 + 64) * F * steps / dt / 1e9}))
