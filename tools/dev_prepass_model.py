"""CPU model of K6's box pre-pass (numpy + the oracle), written before the kernel: for N synthetic frames, the share of a
frame's 61 x 100 tiles that (a) the 8-point first block of the walk and (b) a box lower bound on 8 / 16 / 32 / 64 rim points can
reject against the frame's true minimum.  The rim order here is random, the kernel's is the golden-ratio walk.
usage: python tools/dev_prepass_model.py [frames=6]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as O
from lidar_camera_calibration_amd import synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
p = O.default_params()
g = p.grid_length; W, H = p.board_w, p.board_h
print('g', g, 'W,H', W, H, 'n', p.n_th, p.n_ty, p.n_tz, 'ty step', p.ty_step, 'in squares', p.ty_step / g, 'delta', p.huber_delta)
delta = np.float64(p.huber_delta)
def T(r):
    q = np.minimum(r, delta); return q * (r - 0.5 * q)
tot = {}
for f in range(F):
    res, cb, pc = O.extract(clouds[f], clicks[f], p, want_clouds=True)
    if res.status != 0: continue
    inten = pc[:, 3]; gz = res.gray_zone
    keep = (inten < gz[0]) | (inten > gz[1])
    y = pc[keep, 1].astype(np.float64); z = pc[keep, 2].astype(np.float64); white = inten[keep] > gz[1]
    M = len(y)
    flat, bound, _ = O.grid_search(y, z, white.astype(np.int8), p, True)
    lim2 = 0.5 * bound * (1 + 2e-5)
    ay = ((p.ty_min + np.arange(p.n_ty) * p.ty_step) + W * g / 2) / g
    az = ((p.tz_min + np.arange(p.n_tz) * p.tz_step) + H * g / 2) / g
    rng = np.random.default_rng(f)
    stat = np.zeros(8)
    for k in range(p.n_th):
        th = p.th_min + k * p.th_step
        c, s = np.cos(th), np.sin(th)
        pi = (c * y - s * z) / g; pj = (s * y + c * z) / g
        # classes
        def u(v, a, Wh): return np.abs(v + a - Wh) - Wh
        Wh, Hh = W / 2, H / 2
        umax = np.maximum.reduce([u(pi, ay[0], Wh), u(pi, ay[-1], Wh), u(pj, az[0], Hh), u(pj, az[-1], Hh)])
        border = umax >= 0
        ac, zc = ay[np.argmin(np.abs(ay - Wh))], az[np.argmin(np.abs(az - Hh))]
        rimm = border & (np.maximum(u(pi, ac, Wh), u(pj, zc, Hh)) > -0.3)
        idx = np.flatnonzero(rimm); rng.shuffle(idx)
        for npre_i, npre in enumerate((8, 16, 32, 64)):
            sel = idx[:npre]
            a, b = pi[sel], pj[sel]
            # per candidate partial (half sums, min over phase) on the first 8 rim points -> current first test
            if npre == 8:
                I = a[None, None, :] + ay[:, None, None]; J = b[None, None, :] + az[None, :, None]
                fi, fj = np.floor(I), np.floor(J)
                Rin = (0.5 - np.abs(I - fi - 0.5)) + (0.5 - np.abs(J - fj - 0.5))
                ui, uj = np.abs(I - Wh) - Wh, np.abs(J - Hh) - Hh
                oob = np.maximum(ui, uj) >= 0
                R = np.where(oob, np.abs(ui) + np.abs(uj), Rin)
                odd = ((fi + fj + white[sel][None, None, :]) % 2) == 1
                w0 = np.where(oob, 0.5, np.where(odd, 0.5, 0.0)); w1 = np.where(oob, 0.5, np.where(odd, 0.0, 0.5))
                Tt = T(R)
                part = np.minimum((Tt * w0).sum(-1), (Tt * w1).sum(-1))      # [n_ty, n_tz]
                dead = part > lim2
                tiles_dead = dead.reshape(p.n_ty // 4, 4, p.n_tz // 4, 4).all(axis=(1, 3))
                stat[0] += tiles_dead.size; stat[1] += tiles_dead.sum()
            # box bound per tile
            lo_a, hi_a = ay[0::4], ay[3::4]; lo_z, hi_z = az[0::4], az[3::4]
            ui_lo = u(a[None, :], lo_a[:, None], Wh); ui_hi = u(a[None, :], hi_a[:, None], Wh)     # [tiles_a, pts]
            uj_lo = u(b[None, :], lo_z[:, None], Hh); uj_hi = u(b[None, :], hi_z[:, None], Hh)
            def med0(x, yv): return np.where(x * yv <= 0, 0.0, np.where(np.abs(x) < np.abs(yv), x, yv))
            mi, mj = med0(ui_lo, ui_hi), med0(uj_lo, uj_hi)
            min_i, min_j = np.minimum(ui_lo, ui_hi), np.minimum(uj_lo, uj_hi)
            cert = np.maximum(min_i[:, None, :], min_j[None, :, :]) >= 0
            r = np.abs(mi)[:, None, :] + np.abs(mj)[None, :, :]
            LB = (np.where(cert, T(r), 0.0) * 0.5).sum(-1)
            box_dead = LB > lim2
            stat[2 + npre_i] += box_dead.sum()
            if npre == 8: stat[6] += (box_dead & ~tiles_dead).sum()
    print(f, 'M', M, 'bound', round(bound, 4), 'tiles', int(stat[0]), 'first-test dead %.3f' % (stat[1] / stat[0]),
          'box dead with 8/16/32/64 rim points: ' + ' '.join('%.3f' % (stat[2 + i] / stat[0]) for i in range(4)))
