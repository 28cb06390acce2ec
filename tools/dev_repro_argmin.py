"""Repro driver for one frame of a parity sweep whose grid argmin differed from the oracle's: the frame alone and inside batches of
several sizes, on a fresh / a reserved handle, with pruning on and off.  usage: dev_repro_argmin.py SEED FRAME [lib...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
from oracle import binding as ob
seed, fr = int(sys.argv[1], 0), int(sys.argv[2])
def frames(lo, n):
    c, k, _, _ = synth.make_batch(n, seed=seed + lo)
    return c, k
c1, k1 = frames(fr, 1)
p = ob.default_params(); p.solver = ob.SOLVER_GRID; p.phase_mode = 2; p.accum_float = 0
o = ob.extract(c1[0], k1[0], p)
print("oracle: status", o.status, "grid_index", o.grid_index, "grid_cost", o.grid_cost, "n_lab", o.n_black + o.n_white, "theta_t", list(o.theta_t))
def run(tag, n_before, n_total, reserve, prune=1):
    lo = fr - n_before
    c, k = frames(lo, n_total)
    gp = N.default_params(); gp.grid_prune = prune
    e = LidarCornersBatch(n_total, 28800, gp)
    if reserve: e.reserve(*reserve)
    for rep in range(2):
        r = e.extract(c, k)[n_before]
        print("%-44s call %d: grid_index %d grid_cost %.9g %s theta_t==oracle %s" % (tag, rep, r.grid_index, r.grid_cost, "OK " if r.grid_index == o.grid_index else "DIFF", tuple(r.theta_t) == tuple(o.theta_t)))
    e.close()
run("alone, fresh handle", 0, 1, None)
run("alone, reserved 1792", 0, 1, (1792, 2560))
run("alone, fresh, no pruning", 0, 1, None, prune=0)
run("64-frame batch, fresh", 10, 64, None)
run("64-frame batch, reserved", 10, 64, (1792, 2560))
run("600-frame batch (k6_locate), fresh", 82, 600, None)
run("600-frame batch (k6_locate), reserved 1792", 82, 600, (1792, 2560))
run("600-frame batch (k6_locate), reserved 2560", 82, 600, (2560, 2560))
