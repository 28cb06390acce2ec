import json, os, sys, multiprocessing as mp
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def job(args):
    from lidar_camera_calibration_amd import synth
    from oracle import binding as ob
    f, seed0, rng_lo, rng_hi = args
    board = synth.Board(); s = seed0 + f
    prng = np.random.Generator(np.random.Philox(key=(s ^ 0x905E) & 0xFFFFFFFFFFFFFFFF))
    pose = synth.random_pose(prng, range_m=(rng_lo, rng_hi)) if rng_lo > 0 else synth.random_pose(prng)
    cloud = synth.make_frame(synth.vlp16(), board, pose, s)
    click = synth.make_click(pose, s)
    p = ob.default_params(); p.solver = ob.SOLVER_GRID
    r, cb, pc = ob.extract(cloud, click, p, want_clouds=True)
    if r.status not in (0, 11): return dict(f=f, status=int(r.status), range=float(np.linalg.norm(pose.centre)), err=None)
    err = synth.corner_error(ob.result_corners(r), synth.true_corners(pose, board), board)
    th, ty, tz = r.theta_t
    gz0, gz1 = r.gray_zone
    I = pc[:, 3]; lab = np.where(I < gz0, 0, np.where(I > gz1, 1, -1))
    m = lab >= 0
    y, z = pc[m, 1].astype(np.float64), pc[m, 2].astype(np.float64)
    g, W, H = p.grid_length, p.board_w, p.board_h
    yy = np.cos(th) * y - np.sin(th) * z + ty; zz = np.sin(th) * y + np.cos(th) * z + tz
    i = (yy + W * g / 2) / g; j = (zz + H * g / 2) / g
    inb = (i > 0) & (i < W) & (j > 0) & (j < H)
    ci = np.floor(i[inb]).astype(int); cj = np.floor(j[inb]).astype(int)
    cols = len(set(ci.tolist())); rows = len(set(cj.tolist()))
    cells = len(set(zip(ci.tolist(), cj.tolist())))
    n_oob = int((~inb).sum())
    # range of the board
    rng_m = float(np.linalg.norm(pose.centre))
    return dict(f=f, status=int(r.status), err=1e3 * err, n_lab=int(m.sum()), cols=cols, rows=rows, cells=cells, n_oob=n_oob,
                cost=float(r.sel_cost), cpp=float(r.sel_cost) / max(1, int(m.sum())), margin=float(r.basin_margin), range=rng_m,
                n_plane=int(r.n_plane), rounds=int(r.iters_a), hops=int(r.iters_b))

if __name__ == '__main__':
    # usage: confidence_study.py N SEED OUT.json [range_lo range_hi]   (default range: the bench's 2-3.5 m)
    n = int(sys.argv[1]); seed0 = int(sys.argv[2], 0)
    lo, hi = (float(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (0.0, 0.0)
    with mp.get_context('fork').Pool(os.cpu_count() or 8) as pool:
        allrows = pool.map(job, [(f, seed0, lo, hi) for f in range(n)], chunksize=8)
    failed = [x for x in allrows if x['err'] is None]
    rows = [x for x in allrows if x['err'] is not None]
    json.dump(allrows, open(sys.argv[3], 'w'))
    print('no corners:', len(failed), 'by status', {s: sum(1 for x in failed if x['status'] == s) for s in sorted({x['status'] for x in failed})})
    bad = [x for x in rows if x['status'] == 0 and x['err'] > 20]
    print(len(rows), 'frames; OK & >20mm:', len(bad))
    for x in sorted(bad, key=lambda x: -x['err']): print(x)
