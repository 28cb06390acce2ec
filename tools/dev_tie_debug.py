import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
from oracle import binding as ob
clouds, clicks, _, _ = synth.make_batch(1, seed=0xBEEF + 125)
op = ob.default_params(); op.solver = ob.SOLVER_GRID; op.accum_float = 0
ref = ob.extract(clouds[0], clicks[0], op)
print("oracle", ref.grid_index, ref.grid_cost, ref.n_black, ref.n_white)
for prune in (1, 0):
    p = N.default_params(); p.grid_prune = prune
    e = LidarCornersBatch(1, clouds.shape[1], p)
    r = e.extract(clouds, clicks)[0]
    print("gpu prune", prune, r.grid_index, r.grid_cost, r.n_black, r.n_white, r.status)
    yz, lab = e.fetch_labelled(0)
    bi, bc, vol = e.grid_cost(yz, lab, True, True)
    print("  volume argmin", bi, bc, "vol[33157]", vol[33157], "vol[34756]", vol[34756], "n within 2e-5 of min:", int((vol <= vol.min() * (1 + 2e-5)).sum()))
    y = yz[:, 0].copy(); z = yz[:, 1].copy()
    out = ob.grid_search(y, z, lab.astype(np.int8), op, 1, want_volume=True)
    ov = out[-1]
    print("  oracle search on the GPU's labelled points: argmin", out[0], "cost[33157] %.12g cost[34756] %.12g" % (ov[33157], ov[34756]))
    e.close()
