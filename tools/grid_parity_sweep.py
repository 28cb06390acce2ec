"""GRID-mode parity sweep: HIP path vs the CPU oracle's exhaustive grid search on N seeded frames
(argmin index, corners).  The oracle side runs on all host threads (ctypes releases the GIL).
    python tools/grid_parity_sweep.py N SEED [grid|grid5|local] [OUT.json]
OUT.json (default gpurun_out/parity_sweep_<mode>_<N>_<seed>.json): the counts below as one JSON object -- the committed copies
under profiles/ are the evidence for the argmin parity, which is empirical by construction."""
import concurrent.futures as cf
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
from oracle import binding as ob

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xBEEF
mode = sys.argv[3] if len(sys.argv) > 3 else "grid"   # "grid", "grid5" (BASELINE config 5's frames and grid) or "local" (the reference trajectory)
def config5(p):   # BASELINE configs[4]: 11 x 8 corners @0.10 m, the 129 x 129 x 129 x 2 grid
    p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
    p.n_th = p.n_ty = p.n_tz = 129
    p.th_min, p.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    p.ty_min = p.tz_min = -0.10
    p.ty_step = p.tz_step = 0.10 / 64
    return p


gp = N.default_params()
if mode == "grid5":      # "grid5": the grid mode on BASELINE config 5's dense 64-ring frames
    board, lidar = synth.Board(9, 12, 0.10), synth.hdl64()
    clouds, clicks, gts, _ = synth.make_batch(F, lidar, board, seed=seed, range_m=(2.0, 3.0), yaw_deg=25.0, pitch_deg=15.0, roll_deg=30.0)
    config5(gp)
else:
    board, lidar = synth.Board(), synth.vlp16()
    clouds, clicks, gts, _ = synth.make_batch(F, lidar, board, seed=seed)
if mode == "local":
    gp.solver = N.SOLVER_REFERENCE_LOCAL
est = LidarCornersBatch(F, lidar.n_points, gp, device=0)
res = est.extract(clouds, clicks)
p = ob.default_params()
p.solver = ob.SOLVER_GRID if mode in ("grid", "grid5") else ob.SOLVER_REFERENCE_LOCAL
if mode == "grid5":
    config5(p)
p.phase_mode = 2
p.accum_float = 0
t0 = time.perf_counter()
with cf.ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as ex:
    ref = list(ex.map(lambda f: ob.extract(clouds[f], clicks[f], p), range(F)))
print("oracle: %.1f s" % (time.perf_counter() - t0))
same_status = sum(int(res[f].status == ref[f].status) for f in range(F))
both = [f for f in range(F) if res[f].status in (0, 11) and ref[f].status in (0, 11)]
same_idx = sum(int(res[f].grid_index == ref[f].grid_index) for f in both)
# GRID mode refines on integer sums: theta_t, costs and margin must be the oracle's bit for bit
same_theta = sum(int(tuple(res[f].theta_t) == tuple(ref[f].theta_t) and res[f].sel_cost == ref[f].sel_cost
                     and res[f].basin_margin == ref[f].basin_margin) for f in both) if mode in ("grid", "grid5") else -1
flagged = sum(int(res[f].status == 11) for f in range(F))
overflow = sum(int(res[f].flags & N.FLAG_TIE_OVERFLOW != 0) for f in range(F))
same_conf = sum(int((res[f].cells_hit, res[f].n_oob, res[f].flags & ~N.FLAGS_FP32_ONLY) == (ref[f].cells_hit, ref[f].n_oob, ref[f].flags))
                for f in both)
low_cov = sum(int(res[f].flags & N.FLAG_LOW_COVERAGE != 0) for f in both)
dev = [float(np.abs(res[f].corners_array() - ob.result_corners(ref[f])).max()) for f in both]
it = sum(int(res[f].iters_a == ref[f].iters_a and res[f].iters_b == ref[f].iters_b) for f in both)
print("mode", mode)
print("frames %d  status agree %d  both with corners %d  grid argmin identical %d  rounds/hops (iterations) identical %d  "
      "theta_t+cost+margin bit-identical %d  flagged ambiguous %d  tie-list overflows %d  max corner deviation %.3g m  "
      "frames above 1e-6 m: %d  cells_hit/n_oob/flags identical %d  flagged low coverage %d"
      % (F, same_status, len(both), same_idx, it, same_theta, flagged, overflow, max(dev) if dev else 0.0, sum(d > 1e-6 for d in dev),
         same_conf, low_cov))
import json
import subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(root, "gpurun_out", "parity_sweep_%s_%d_%#x.json" % (mode, F, seed))
summary = {
    "tool": "tools/grid_parity_sweep.py %d %#x %s" % (F, seed, mode), "mode": mode, "frames": F, "seed": "%#x" % seed,
    "sensor": "hdl64 131072 pts, 11x8 board @0.10 m, 129^3 x 2 grid" if mode == "grid5" else "vlp16 28800 pts, 7x5 board @0.15 m, 61x40x40 x 2 grid",
    "status_agree": same_status, "frames_with_corners_on_both_sides": len(both), "grid_argmin_identical": same_idx,
    "rounds_hops_identical": it, "theta_t_cost_margin_bit_identical": same_theta, "cells_hit_n_oob_flags_identical": same_conf,
    "tie_list_overflows": overflow, "flagged_ambiguous": flagged, "flagged_low_coverage": low_cov,
    "max_corner_deviation_m": max(dev) if dev else 0.0, "frames_above_1e-6_m": int(sum(d > 1e-6 for d in dev)),
    "mismatching_frames": [int(f) for f in both if res[f].grid_index != ref[f].grid_index][:64],
    "library": os.path.basename(N.LIB_PATH),
}
try:
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    print("wrote", out_path)
except OSError as e:
    print("could not write", out_path, e)
for f in both:
    if res[f].grid_index != ref[f].grid_index:
        print("  frame", f, "gpu", res[f].grid_index, res[f].grid_cost, "oracle", ref[f].grid_index, ref[f].grid_cost)
