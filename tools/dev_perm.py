import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_camera_calibration_amd import synth, LidarCornersBatch, _native as N
F=128
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
e = LidarCornersBatch(F, 28800, N.default_params())
def rec(res):
    return [(r.status, r.n_roi, r.n_cluster, r.n_plane, r.n_black, r.n_white, r.grid_index, round(r.grid_cost,6), tuple(r.theta_t), r.iters_a, r.iters_b, tuple(np.ctypeslib.as_array(r.corners)[:6])) for r in res]
a = rec(e.extract(clouds, clicks))
perm = np.random.default_rng(0).permutation(F)
b = rec(e.extract(clouds[perm], clicks[perm]))
c = rec(e.extract(clouds, clicks))
nd=0
for i in range(F):
    if b[i] != a[perm[i]]:
        nd+=1
        if nd<8: print('DIFF frame', perm[i], 'at slot', i, '\n ', a[perm[i]], '\n ', b[i])
print('n diff perm', nd, 'n diff repeat', sum(x!=y for x,y in zip(a,c)))
sub = rec(e.extract(clouds[5:9], clicks[5:9]))
print('sub diff', sum(x!=y for x,y in zip(sub,a[5:9])))
