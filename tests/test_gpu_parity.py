"""GPU parity tests (pytest -m gpu, on the MI355X box): every call goes through the C-ABI of
libilcc_hip.so and is compared with the CPU oracle on the same seeded inputs.

Tolerances: index/byte/integer work (ROI, cluster, plane inliers, labels, histogram-derived gray zone, grid
argmin index, and in ILCC_SOLVER_GRID the whole refinement -- fixed-point costs, lattice coordinates, theta_t,
rounds, hops, margin) must be IDENTICAL; float stages 1e-6 m; the reference-trajectory solver runs in double on
both sides and must agree to 1e-6 (theta_t) -- corners to 1e-5 m (GRID: 1e-6 m, only cosf/sinf of the final
roll may differ in the last bit), far inside the 1e-3 m bar of BASELINE.json.
fp32 grid cost vs the fixed-point oracle: 2e-5 relative + 2e-6 absolute.
"""
import ctypes as C
import os

import numpy as np
import pytest

from lidar_camera_calibration_amd import LidarCornersBatch, LidarCornersEst, synth
from lidar_camera_calibration_amd import _native as N

pytestmark = pytest.mark.gpu

BOARD = synth.Board()


@pytest.fixture(scope="module")
def frames():
    c1, k1, g1, _ = synth.make_batch(6, fixture_poses=True)
    c2, k2, g2, _ = synth.make_batch(10, seed=4242, range_m=(2.0, 3.0))
    return np.concatenate([c1, c2]), np.concatenate([k1, k2]), np.concatenate([g1, g2])


@pytest.fixture(scope="module")
def est():
    e = LidarCornersBatch(64, 28800, N.default_params())
    yield e
    e.close()


def _oparams(ob, solver):
    op = ob.default_params()
    op.solver = solver
    return op


def _set_solver(est, solver, **kw):
    p = N.default_params()
    p.solver = solver
    for k, v in kw.items():
        setattr(p, k, v)
    est.set_params(p)
    return p


@pytest.mark.parametrize("solver", [N.SOLVER_GRID, N.SOLVER_REFERENCE_LOCAL])
def test_every_stage_matches_the_oracle(ob, est, frames, solver):
    clouds, clicks, gts = frames
    _set_solver(est, solver)
    res = est.extract(clouds, clicks)
    op = _oparams(ob, solver)
    worst = 0.0
    for f in range(len(clicks)):
        r = res[f]
        o, ocb, opc = ob.extract(clouds[f], clicks[f], op, want_clouds=True)
        assert r.status == o.status, f
        assert (r.n_roi, r.n_cluster, r.n_plane) == (o.n_roi, o.n_cluster, o.n_plane), f
        # a1 ROI: same points, same order
        roi_idx = ob.roi_crop(clouds[f], clicks[f], op)
        assert np.array_equal(est.fetch_cloud(f, N.CLOUD_ROI), clouds[f][roi_idx])
        if o.status not in (N.OK, N.AMBIGUOUS):
            continue
        # a2 cluster / a3 plane inliers: identical clouds
        clu_idx, _ = ob.cluster(clouds[f][roi_idx], clicks[f], op)
        assert np.array_equal(est.fetch_cloud(f, N.CLOUD_CLUSTER), clouds[f][roi_idx][clu_idx])
        assert np.array_equal(est.fetch_cloud(f, N.CLOUD_CHESSBOARD), ocb)
        # a4 plane frame
        assert np.abs(np.array(r.pca) - np.array(o.pca)).max() < 1e-6
        assert np.abs(est.fetch_cloud(f, N.CLOUD_PCA) - opc).max() < 1e-6
        # a5 gray zone + labels
        assert np.allclose(r.gray_zone, o.gray_zone, rtol=1e-12)
        assert (r.n_black, r.n_gray, r.n_white) == (o.n_black, o.n_gray, o.n_white)
        yz, lab = est.fetch_labelled(f)
        inten = opc[:, 3].astype(np.float64)
        keep = (inten < o.gray_zone[0]) | (inten > o.gray_zone[1])
        assert np.array_equal(yz, opc[keep][:, 1:3]) and np.array_equal(lab, (inten[keep] > o.gray_zone[1]))
        # a6-a8 search + solve, a9 corners
        assert r.phase == o.phase
        assert (r.iters_a, r.iters_b) == (o.iters_a, o.iters_b)
        dev = np.abs(r.corners_array() - ob.result_corners(o)).max()
        worst = max(worst, dev)
        # the confidence signal (squares holding points / points off the board under the final pose) and its flag
        assert (r.cells_hit, r.n_oob) == (o.cells_hit, o.n_oob), (f, r.cells_hit, o.cells_hit, r.n_oob, o.n_oob)
        assert (r.flags & ~N.FLAGS_FP32_ONLY) == o.flags, (f, r.flags, o.flags)
        if solver == N.SOLVER_GRID:
            assert r.grid_index == o.grid_index, f
            assert r.grid_cost == pytest.approx(o.grid_cost, rel=2e-5, abs=2e-6)
            # the refinement works on integer sums: nothing about it may differ
            assert tuple(r.theta_t) == tuple(o.theta_t), (f, list(r.theta_t), list(o.theta_t))
            assert (r.cost_a, r.cost_b, r.sel_cost, r.basin_margin) == (o.cost_a, o.cost_b, o.sel_cost, o.basin_margin)
            assert r.sel_cost <= o.grid_cost          # monotone: never above the grid argmin's cost
            assert (r.flags & (N.FLAG_TIE_OVERFLOW | N.FLAG_REFINE_CAPPED)) == 0
            assert dev < 1e-6, (f, dev)
        else:
            assert np.allclose(r.theta_t, o.theta_t, atol=1e-6), (f, list(r.theta_t), list(o.theta_t))
            assert r.cost_a == pytest.approx(o.cost_a, rel=1e-9, abs=1e-12)
            assert r.cost_b == pytest.approx(o.cost_b, rel=1e-9, abs=1e-12)
            assert dev < 1e-5, (f, dev)             # BASELINE bar: 1e-3 m
        assert r.n_corners == 35
    assert worst < 1e-5


def test_bundled_pose_corners_config3(ob, est, golden_dir):
    """BASELINE config 3 on the VLP-16 / 10 mm-noise frames: six synthetic frames whose true corners are the rows of
    pointgrey_lidar_{1..6}.txt.  GPU == CPU oracle (the config's second criterion) is asserted; the distance to the
    bundled files on THIS sensor model is limited by what 16 rings and 10 mm of range noise resolve and is only
    reported and sanity-bounded here -- the <= 1e-3 m criterion is test_config3_dense_low_noise... below."""
    clouds, clicks, _, _ = synth.make_batch(6, fixture_poses=True)
    for solver in (N.SOLVER_GRID, N.SOLVER_REFERENCE_LOCAL):
        _set_solver(est, solver)
        res = est.extract(clouds, clicks)
        op = _oparams(ob, solver)
        errs = []
        for n in range(6):
            fix = np.loadtxt(os.path.join(golden_dir, "pointgrey_lidar_%d.txt" % (n + 1)))
            got = res[n].corners_array()
            o = ob.extract(clouds[n], clicks[n], op)
            assert res[n].status == o.status == N.OK
            assert np.abs(got - ob.result_corners(o)).max() < 1e-5      # config 3: |GPU - CPU oracle| <= 1e-3 m
            errs.append(synth.corner_error(got, fix, BOARD))
        print("config 3, VLP-16 frames, solver %d: |GPU - bundled file| = %s mm" % (solver, np.round(1e3 * np.array(errs), 2)))
        # what this sensor model resolves (profiles/r03_noise_floor_study.json: 2.1 mm median with NO range noise -- the
        # 15 mm beam footprint greys the square edges out -- 2.7 mm at 10 mm): measured 2.4-6.9 mm (grid), 1.9-5.3 mm (reference mode)
        assert max(errs) <= 0.008, errs


def _dense_low_noise_fixture_frames():
    """The six fixture poses seen by a dense, quiet sensor: 128 rings over +-12 deg (0.19 deg apart), 1800 azimuth
    steps of 0.2 deg, 1 mm range noise, 3 mm beam footprint -- the regime in which the method itself, not the
    sensor model, sets the error (CPU oracle on these frames: 0.22-0.70 mm from the bundled files in both modes)."""
    lidar = synth.Lidar(np.linspace(-12.0, 12.0, 128), 1800)
    board = synth.Board()
    clouds, clicks = [], []
    for n in range(6):
        pose = synth.pose_from_fixture(n)
        clouds.append(synth.make_frame(lidar, board, pose, 0xD0 + n, sigma_r=0.001, footprint=0.003))
        clicks.append(synth.make_click(pose, 0xD0 + n))
    return lidar, np.stack(clouds), np.stack(clicks)


def test_config3_dense_low_noise_lands_within_1mm_of_the_bundled_files(ob, golden_dir):
    """BASELINE config 3's pass criterion as written: max |GPU - fixture| <= 1e-3 m and
    max |GPU - CPU oracle| <= 1e-3 m, both solver modes, on the dense low-noise variant of the six poses."""
    lidar, clouds, clicks = _dense_low_noise_fixture_frames()
    e = LidarCornersBatch(6, lidar.n_points, N.default_params())
    for solver in (N.SOLVER_GRID, N.SOLVER_REFERENCE_LOCAL):
        _set_solver(e, solver)
        res = e.extract(clouds, clicks)
        op = _oparams(ob, solver)
        for n in range(6):
            fix = np.loadtxt(os.path.join(golden_dir, "pointgrey_lidar_%d.txt" % (n + 1)))
            got = res[n].corners_array()
            o = ob.extract(clouds[n], clicks[n], op)
            assert res[n].status == o.status == N.OK, (solver, n, res[n].status, o.status)
            assert np.abs(got - ob.result_corners(o)).max() < 1e-5
            err = synth.corner_error(got, fix, BOARD)
            assert err <= 1e-3, (solver, n, err)
    e.close()


def _rand_points(rng, m):
    yz = np.stack([rng.uniform(-0.7, 0.7, m), rng.uniform(-0.9, 0.9, m)], 1).astype(np.float32)
    lab = rng.integers(0, 2, m).astype(np.uint8)
    return yz, lab


def _walk_stride(m):
    """csrc/ilcc_internal.h walk_stride(): ~0.618 M, odd, coprime to M"""
    import math
    if m <= 2:
        return 1
    s = int(np.float32(m) * np.float32(0.6180339)) | 1
    while math.gcd(s, m) != 1:
        s += 2
    return 1 if s >= m else s


def test_walk_layout_is_a_class_partition_of_the_golden_walk(est, frames):
    """K5w (the grid search's point order, written once per frame): the walk layout must be a permutation of the labelled
    points; every part keeps golden-ratio walk order; interior-class points are in the board under EVERY rotation and
    translation of the grid (accumulate_interior's precondition -- the property the exactness of the fast term rests on);
    of the other points at most a tenth never leave it (the class is decided by a conservative bound)."""
    clouds, clicks, _ = frames
    p = _set_solver(est, N.SOLVER_GRID)
    res = est.extract(clouds, clicks)
    g, W, H = p.grid_length, p.board_w, p.board_h
    th = p.th_min + np.arange(p.n_th) * p.th_step
    ay = np.array([(p.ty_min + a * p.ty_step + W * g / 2) / g for a in (0, p.n_ty - 1)])
    az = np.array([(p.tz_min + b * p.tz_step + H * g / 2) / g for b in (0, p.n_tz - 1)])
    checked = 0
    for f in range(len(clouds)):
        if res[f].status not in (0, 11):
            continue
        yz, lab = est.fetch_labelled(f)
        wyz, wlab, n_in, n_rim = est.fetch_walk(f)
        M = len(yz)
        assert len(wyz) == M and 0 <= n_in <= M and n_in + n_rim <= M
        # permutation (points may repeat: compare as sorted records)
        rec = lambda a, l: np.sort(np.rec.fromarrays([a[:, 0], a[:, 1], l]), order=["f0", "f1", "f2"])
        assert np.array_equal(rec(yz, lab), rec(wyz, wlab))
        # every part in walk order: map each layout entry back to its walk slot and require increasing slots per part
        S = _walk_stride(M)
        slot_of = {}
        for sl in range(M):
            i = (sl * S) % M
            slot_of.setdefault((yz[i, 0].tobytes(), yz[i, 1].tobytes(), int(lab[i])), []).append(sl)
        for lo, hi in ((0, n_in), (n_in, n_in + n_rim), (n_in + n_rim, M)):
            last = -1
            for k in range(lo, hi):
                cands = slot_of[(wyz[k, 0].tobytes(), wyz[k, 1].tobytes(), int(wlab[k]))]
                nxt = [c for c in cands if c > last]
                assert nxt, "part [%d, %d) of frame %d leaves walk order at %d" % (lo, hi, f, k)
                last = min(nxt)
        # class property, in double (the kernel decides in fp32: allow a rounding margin on the border side only)
        y, z = wyz[:, 0].astype(np.float64), wyz[:, 1].astype(np.float64)
        worst = np.full(M, -np.inf)
        for t in th:
            pi, pj = (np.cos(t) * y - np.sin(t) * z) / g, (np.sin(t) * y + np.cos(t) * z) / g
            for a in ay:
                worst = np.maximum(worst, np.abs(pi + a - W / 2) - W / 2)
            for b in az:
                worst = np.maximum(worst, np.abs(pj + b - H / 2) - H / 2)
        assert (worst[:n_in] < 0).all(), "an interior-class point can leave the board"
        # (the class is decided by a bound on |i|, |j| over the rotations: a few per cent of the points that never leave
        # the board stay in the border class, which is always safe -- but not many)
        assert (worst[n_in:] < -1e-3).sum() <= 0.1 * M + 8
        checked += 1
    assert checked >= 12


@pytest.mark.parametrize("m", [0, 1, 63, 64, 65, 1000, 2500])
@pytest.mark.parametrize("use_oob", [1, 0])
def test_grid_cost_volume_matches_oracle(ob, est, m, use_oob):
    """K6 alone on arbitrary labelled points (incl. empty / ragged wavefront tails): the whole
    cost volume and the selected candidate vs the fp64 oracle."""
    rng = np.random.default_rng(100 + m)
    p = _set_solver(est, N.SOLVER_GRID, n_th=7, n_ty=9, n_tz=10, th_min=-0.12, th_step=0.04,
                    ty_min=-0.12, ty_step=0.03, tz_min=-0.15, tz_step=0.03)
    yz, lab = _rand_points(rng, m)
    bi, bc, vol = est.grid_cost(yz, lab, use_oob, want_volume=True)
    op = ob.default_params()
    for k in ("n_th", "n_ty", "n_tz", "th_min", "th_step", "ty_min", "ty_step", "tz_min", "tz_step"):
        setattr(op, k, getattr(p, k))
    oflat, oc, ovol = ob.grid_search(yz[:, 0], yz[:, 1], lab.astype(np.int8), op, use_oob, want_volume=True)
    assert vol.shape == ovol.shape == (7 * 9 * 10 * 2,)
    err = np.abs(vol - ovol)
    tol = 2e-5 * np.abs(ovol) + 2e-6
    bad = err > tol
    # fp32 vs fp64 can put a point on the other side of the (discontinuous) outer board edge: allow
    # isolated candidates to differ by one point's worth of cost
    assert bad.mean() <= (0.002 if use_oob else 0.0), (bad.sum(), err.max())
    assert err[bad].max(initial=0.0) < 0.5
    # selection: same candidate, or an exact-tie / rounding-level near-tie in the oracle's own volume
    assert ovol[bi] <= oc + 2e-5 * abs(oc) + 2e-6
    if m == 0:
        assert bi == oflat and bc == 0.0


def test_grid_cost_more_points_than_lds_stage(ob, est):
    """M' above the LDS staging capacity takes the global-memory path: same numbers."""
    rng = np.random.default_rng(5)
    p = _set_solver(est, N.SOLVER_GRID, n_th=3, n_ty=4, n_tz=5)
    yz, lab = _rand_points(rng, 20000)
    bi, bc, vol = est.grid_cost(yz, lab, 1, want_volume=True)
    op = ob.default_params()
    op.n_th, op.n_ty, op.n_tz = 3, 4, 5
    oflat, oc, ovol = ob.grid_search(yz[:, 0], yz[:, 1], lab.astype(np.int8), op, 1, want_volume=True)
    assert np.abs(vol - ovol).max() < 2e-5 * ovol.max() + 0.6      # border flips: <= 1 point's cost
    assert np.median(np.abs(vol - ovol) / ovol) < 1e-5
    assert ovol[bi] <= oc * (1 + 1e-4)


@pytest.mark.parametrize("tlw,oob", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_local_solver_matches_oracle(ob, est, frames, tlw, oob):
    """K7a (Ceres-style trust region) alone vs orc_get_theta_t on the labelled points of real frames."""
    clouds, clicks, _ = frames
    _set_solver(est, N.SOLVER_REFERENCE_LOCAL)
    est.extract(clouds[:4], clicks[:4])
    op = ob.default_params()
    labelled = [est.fetch_labelled(f) for f in range(4)]     # the solver entry reuses frame 0's buffers
    for yz, lab in labelled:
        pts = np.concatenate([np.zeros((len(yz), 1), np.float32), yz,
                              np.where(lab[:, None] == 1, 200.0, 0.0).astype(np.float32)], 1)
        t, c, it = est.get_theta_t(yz, lab, tlw, oob)
        to, co, ito = ob.get_theta_t(pts, [50.0, 150.0], op, tlw, oob)
        assert it == ito
        assert np.allclose(t, to, atol=1e-7), (t, to)
        assert c == pytest.approx(co, rel=1e-9, abs=1e-12)
    t, c, it = est.get_theta_t(np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), 0, 1, (0.1, 0.0, 0.0))
    assert it == 0 and c == 0.0 and t[0] == 0.1


def test_reference_solver_one_wavefront_per_solve_equals_the_workgroup_layout(ob):
    """K7a has two layouts: batches above 256 frames give every (frame, phase) solve ONE wavefront (four solves per
    workgroup, no barrier after staging), smaller batches a whole 256-thread workgroup.  Both add the terms in the same
    order: a frame's record is bit-identical whichever batch it came in, and both walk the oracle's iterations."""
    F = 272
    clouds, clicks, _, _ = synth.make_batch(F, seed=0xBEEF)
    p = N.default_params()
    p.solver = N.SOLVER_REFERENCE_LOCAL
    e = LidarCornersBatch(F, 28800, p)
    big = e.extract(clouds, clicks)
    big = [(r.status, r.iters_a, r.iters_b, tuple(r.theta_t), r.cost_a, r.cost_b, r.sel_cost, r.phase, r.corners_array().copy()) for r in big]
    small = e.extract(clouds[100:124], clicks[100:124])
    n_ok = 0
    for k, r in enumerate(small):
        b = big[100 + k]
        assert (r.status, r.iters_a, r.iters_b, tuple(r.theta_t), r.cost_a, r.cost_b, r.sel_cost, r.phase) == b[:8], k
        assert np.array_equal(r.corners_array(), b[8])
        n_ok += r.status == N.OK
    assert n_ok >= 20
    op = _oparams(ob, N.SOLVER_REFERENCE_LOCAL)
    for f in list(range(0, 8)) + [271]:
        o = ob.extract(clouds[f], clicks[f], op)
        b = big[f]
        assert b[0] == o.status
        if o.status != N.OK:
            continue
        assert (b[1], b[2], b[7]) == (o.iters_a, o.iters_b, o.phase), f
        assert np.allclose(b[3], o.theta_t, atol=1e-7)
        assert b[4] == pytest.approx(o.cost_a, rel=1e-9, abs=1e-12) and b[5] == pytest.approx(o.cost_b, rel=1e-9, abs=1e-12)
        assert np.abs(b[8] - ob.result_corners(o)).max() < 1e-5
    e.close()


def test_pattern_refine_kernel_matches_the_oracle_exactly(ob, est, frames):
    """K7r alone (ilcc_pattern_refine) vs orc_pattern_refine on the labelled points of real frames, from the grid
    argmin and from displaced starts (incl. one square off with the colours swapped -> a basin hop): lattice
    coordinates, phase, both fixed-point costs, rounds and hops must be IDENTICAL -- integer sums leave no room."""
    clouds, clicks, _ = frames
    p = _set_solver(est, N.SOLVER_GRID)
    res = est.extract(clouds[:6], clicks[:6])
    labelled = [est.fetch_labelled(f) for f in range(6)]
    op = ob.default_params()
    rng = np.random.default_rng(3)
    n_hops = 0
    for f, (yz, lab) in enumerate(labelled):
        cell = res[f].grid_index >> 1
        k, a, b = cell // (p.n_ty * p.n_tz), (cell // p.n_tz) % p.n_ty, cell % p.n_tz
        starts = [([16 * k, 16 * a, 16 * b], res[f].grid_index & 1),
                  ([16 * k + 5, 16 * a + 320, 16 * b - 3], (res[f].grid_index & 1) ^ 1),       # one square along y
                  ([int(rng.integers(0, 960)), int(rng.integers(0, 640)), int(rng.integers(0, 640))], int(rng.integers(0, 2))),
                  ([-400, 100, 700], 0)]                                                        # theta beyond the grid
        for lat0, ph0 in starts:
            got = est.pattern_refine(yz, lab, lat0, ph0)
            want = ob.pattern_refine(yz[:, 0], yz[:, 1], lab.astype(np.int8), op, lat0, ph0)
            assert list(got[0]) == list(want[0]) and got[1:] == want[1:], (f, lat0, got, want)
            n_hops += got[5]
    assert n_hops >= 6                                     # the displaced starts did hop
    # refinement switched off: the start is kept, the basin check still runs
    p0 = _set_solver(est, N.SOLVER_GRID, refine_div=0)
    op0 = ob.default_params()
    op0.refine_div = 0
    yz, lab = labelled[0]
    got = est.pattern_refine(yz, lab, [30, 20, 20], 0)
    want = ob.pattern_refine(yz[:, 0], yz[:, 1], lab.astype(np.int8), op0, [30, 20, 20], 0)
    assert list(got[0]) == [30, 20, 20] == list(want[0]) and got[1:] == want[1:] and got[4] == 0
    # no points at all: every cost is 0 -> margin 0
    _set_solver(est, N.SOLVER_GRID)
    got = est.pattern_refine(np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), [480, 320, 320], 0)
    assert list(got[0]) == [480, 320, 320] and got[2] == 0 and got[3] == 0


def test_ambiguous_frames_are_flagged_not_silently_returned(ob):
    """A board seen only through its middle rows cannot be placed along the long axis: a shift by one square with the
    colours swapped costs exactly the same.  GRID mode returns ILCC_AMBIGUOUS (margin 0) with the corners still in the
    record; the host mirror rejects such a scan unless told otherwise; ambiguity_eps <= 0 switches the flag off."""
    board = synth.Board()
    pose = synth.pose_from_fixture(0)
    cloud = synth.make_frame(synth.vlp16(), board, pose, 0xA1)
    click = synth.make_click(pose, 0xA1)
    # keep only returns within +-0.28 m of the board centre along its long axis (fixture pose 1: long axis ~ -y)
    along = (cloud[:, :3] - pose.centre.astype(np.float32)) @ pose.v.astype(np.float32)
    on_board = np.linalg.norm(cloud[:, :3] - pose.centre.astype(np.float32), axis=1) < 1.0
    cut = np.ascontiguousarray(cloud[~on_board | (np.abs(along) < 0.28)])
    p = N.default_params()
    e = LidarCornersBatch(1, len(cut), p)
    r = e.extract(cut[None], click[None])[0]
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    o = ob.extract(cut, click, op)
    assert r.status == o.status == N.AMBIGUOUS, (r.status, o.status)
    assert r.basin_margin == o.basin_margin and r.basin_margin < p.ambiguity_eps
    assert r.n_corners == 35 and tuple(r.theta_t) == tuple(o.theta_t)
    assert len(e.fetch_cloud(0, N.CLOUD_OPTIM)) == r.n_plane
    p.ambiguity_eps = 0.0
    e.set_params(p)
    r0 = e.extract(cut[None], click[None])[0]
    assert r0.status == N.OK and r0.flags == N.FLAG_LOW_COVERAGE      # (the middle-rows-only board is under-sampled too)
    p.refine_max_rounds = 2          # the pattern search is cut off long before its stride reaches the lattice: flagged
    e.set_params(p)
    r2 = e.extract(cut[None], click[None])[0]
    op.refine_max_rounds, op.ambiguity_eps = 2, 0.0
    o2 = ob.extract(cut, click, op)
    assert r2.flags & N.FLAG_REFINE_CAPPED and r2.iters_a == o2.iters_a and tuple(r2.theta_t) == tuple(o2.theta_t)
    assert r2.sel_cost >= r0.sel_cost
    e.close()
    m = LidarCornersEst(max_points_per_frame=len(cut))
    m.setROI(cut, click)
    assert m.EuclideanCluster() is True
    m.PCA()
    corners = []
    assert m.get_corners(corners) is False and corners == []
    m.accept_ambiguous = True            # the basin signal alone: the middle-rows-only board is ALSO under-sampled,
    assert m.get_corners(corners) is False and corners == []    # and the coverage gate has its own switch
    m.accept_low_coverage = True
    assert m.get_corners(corners) is True and len(corners) == 35
    m.close()


def test_low_coverage_flag_and_accept_rule(ob):
    """The second confidence signal: cells_hit / n_oob equal the oracle's; a board whose far
    end is cut off by the ROI leaves squares empty -> ILCC_FLAG_LOW_COVERAGE, and the class mirror's get_corners
    rejects the scan unless accept_low_coverage is set (its own switch: accept_ambiguous does not open this gate);
    min_cell_coverage <= 0 switches the flag off."""
    board = synth.Board()
    pose = synth.pose_from_fixture(0)
    cloud = synth.make_frame(synth.vlp16(), board, pose, 0xC0FFEE)
    click = synth.make_click(pose, 0xC0FFEE)
    # full board: every square sampled
    p = N.default_params()
    e = LidarCornersBatch(1, len(cloud), p)
    r = e.extract(cloud[None], click[None])[0]
    o = ob.extract(cloud, click, _oparams(ob, N.SOLVER_GRID))
    assert r.status == N.OK and (r.cells_hit, r.n_oob, r.flags) == (o.cells_hit, o.n_oob, o.flags)
    assert r.cells_hit >= 46 and not (r.flags & N.FLAG_LOW_COVERAGE)
    # the same frame with the upper third of the board's points removed before the call (an occluded board)
    c0 = pose.centre
    up = np.array(pose.v if abs(pose.v[2]) > abs(pose.u[2]) else pose.u)
    up = up * np.sign(up[2])
    keep = ((cloud[:, :3] - c0) @ up < 0.18) | (np.linalg.norm(cloud[:, :3] - c0, axis=1) > 1.0)
    cut = np.ascontiguousarray(cloud[keep])
    e2 = LidarCornersBatch(1, len(cut), p)
    r2 = e2.extract(cut[None], click[None])[0]
    o2 = ob.extract(cut, click, _oparams(ob, N.SOLVER_GRID))
    assert r2.status == o2.status and (r2.cells_hit, r2.n_oob) == (o2.cells_hit, o2.n_oob)
    assert (r2.flags & ~N.FLAGS_FP32_ONLY) == o2.flags
    if r2.status in (N.OK, N.AMBIGUOUS):
        assert r2.cells_hit < 0.9 * 48 and (r2.flags & N.FLAG_LOW_COVERAGE)
        m = LidarCornersEst(device=0, max_points_per_frame=len(cut))
        m.setROI(cut, click)
        assert m.EuclideanCluster()
        m.PCA()
        got = []
        assert m.get_corners(got) is False and got == []
        if r2.status == N.OK:
            m.accept_ambiguous = True        # the other signal's switch does not accept an under-sampled scan
            assert m.get_corners(got) is False and got == []
            m.accept_ambiguous = False
        else:
            m.accept_ambiguous = True
        m.accept_low_coverage = True
        got = []
        assert m.get_corners(got) is True and len(got) == 35
        m.close()
        p.min_cell_coverage = 0.0
        e2.set_params(p)
        r3 = e2.extract(cut[None], click[None])[0]
        assert not (r3.flags & N.FLAG_LOW_COVERAGE) and r3.cells_hit == r2.cells_hit
    e.close()
    e2.close()


def test_bad_frames_do_not_abort_the_batch(ob, est, frames):
    clouds, clicks, _ = frames
    _set_solver(est, N.SOLVER_GRID)
    n = clouds.shape[1]
    batch = np.stack([clouds[0], clouds[1], clouds[2], clouds[3]])
    ck = clicks[:4].copy()
    ck[1] = [50.0, 50.0, 50.0]                        # click far from everything: empty ROI
    batch[2, :, :3] += np.float32(1000.0)             # board gone from the box, only sparse far points
    batch[3, :, 3] = 42.0                             # constant intensity: histogram degenerate
    res = est.extract(batch, ck)
    assert res[0].status == N.OK and res[0].n_corners == 35
    assert res[1].status == N.NO_ROI_POINTS and res[1].n_corners == 0
    assert res[2].status in (N.NO_ROI_POINTS, N.NO_CLUSTER)
    assert res[3].status == N.DEGENERATE_HIST
    op = _oparams(ob, N.SOLVER_GRID)
    for f in range(4):
        assert ob.extract(batch[f], ck[f], op).status == res[f].status
    assert len(est.fetch_cloud(1, N.CLOUD_ROI)) == 0
    # the good frame is unaffected by its neighbours
    alone = est.extract(batch[:1], ck[:1])[0]
    assert np.array_equal(alone.corners_array(), res[0].corners_array())


def test_ragged_batch_and_single_frame_entry(ob, frames):
    clouds, clicks, _ = frames
    p = N.default_params()
    e = LidarCornersBatch(4, 28800, p)
    lens = [28800, 20000, 28800 - 7]
    flat = np.concatenate([clouds[i][:lens[i]] for i in range(3)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    res = e.extract(flat, clicks[:3], offsets=off)
    op = _oparams(ob, N.SOLVER_GRID)
    for i in range(3):
        o = ob.extract(clouds[i][:lens[i]], clicks[i], op)
        assert res[i].status == o.status and res[i].n_points == lens[i]
        if o.status in (N.OK, N.AMBIGUOUS):
            assert np.abs(res[i].corners_array() - ob.result_corners(o)).max() < 1e-6
    e.close()
    # the reference's call sequence through the host mirror
    m = LidarCornersEst(max_points_per_frame=28800)
    m.register_viewer()
    m.setROI(clouds[0], clicks[0])
    assert m.EuclideanCluster() is True
    m.PCA()
    corners = []
    assert m.get_corners(corners) is True and len(corners) == 35
    assert m.m_cloud_corners.shape == (35, 4) and np.all(m.m_cloud_corners[:, 3] == 50.0)
    assert m.m_cloud_optim.shape == m.m_cloud_PCA.shape == m.m_cloud_chessboard.shape
    assert np.abs(m.m_cloud_optim[:, 0] - m.m_cloud_PCA[:, 0]).max() == 0.0       # roll about x only
    assert np.array_equal(np.array(corners, dtype=np.float32), res[0].corners_array())   # same frame, same path
    m.close()


def test_branch_and_bound_does_not_change_the_result(ob, frames):
    """grid_prune=1 (seeding pass + early exit of beaten candidate tiles) must return exactly what the
    cut-free exhaustive pass returns: same argmin index, same corners; and both equal the oracle's argmin."""
    clouds, clicks, _ = frames
    out = {}
    for prune in (0, 1):
        p = N.default_params()
        p.grid_prune = prune
        e = LidarCornersBatch(16, 28800, p)
        res = e.extract(clouds, clicks)
        out[prune] = [(r.status, r.grid_index, r.corners_array()) for r in res]
        tm = e.timing()
        if prune:
            assert tm.grid_cost_evals_sum < 0.6 * tm.grid_cost_evals_nominal_sum       # it does cut work
        else:
            assert tm.grid_cost_evals_sum >= tm.grid_cost_evals_nominal_sum
        e.close()
    for a, b in zip(out[0], out[1]):
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[2], b[2])
    # K6 alone on arbitrary points, pruned entry vs the oracle's exhaustive argmin
    rng = np.random.default_rng(77)
    p = N.default_params()
    e = LidarCornersBatch(1, 28800, p)
    op = ob.default_params()
    op.n_th, op.n_ty, op.n_tz = 9, 12, 12
    for k in ("n_th", "n_ty", "n_tz"):
        setattr(p, k, getattr(op, k))
    e.set_params(p)
    for m in (200, 1500):
        yz, lab = _rand_points(rng, m)
        bi, bc, _ = e.grid_cost(yz, lab, 1, want_volume=False)            # branch-and-bound variant
        oflat, oc, ovol = ob.grid_search(yz[:, 0], yz[:, 1], lab.astype(np.int8), op, 1, want_volume=True)
        assert ovol[bi] <= oc * (1 + 2e-5) + 2e-6 and bc == pytest.approx(ovol[bi], rel=2e-5, abs=2e-6)
    e.close()


@pytest.mark.parametrize("grid", [dict(n_th=21, th_step=1.5 * np.pi / 180, n_ty=37, n_tz=39, t_step=0.0081),   # partial tiles on both axes
                                  dict(n_th=13, th_step=2.5 * np.pi / 180, n_ty=8, n_tz=8, t_step=0.046),      # 3 steps >= 0.9 squares: no box pre-pass
                                  dict(n_th=31, th_step=1.0 * np.pi / 180, n_ty=80, n_tz=76, t_step=0.004)])   # narrow boxes, 20 x 19 tiles
def test_box_prepass_on_other_grids_matches_the_oracle(ob, frames, grid):
    """The full pass's box pre-pass (a lower bound per 4 x 4 tile from the out-of-board cost at the tile's extreme
    translations) must never drop the argmin or a near tie: GRID-mode results on grids with partial tiles, with boxes too
    wide for the pre-pass (the host switches it off) and with many narrow tiles are the oracle's, bit for bit."""
    clouds, clicks, _ = frames
    p, op = N.default_params(), ob.default_params()
    op.solver = ob.SOLVER_GRID
    for q in (p, op):
        q.n_th, q.n_ty, q.n_tz = grid["n_th"], grid["n_ty"], grid["n_tz"]
        q.th_step = grid["th_step"]
        q.th_min = -0.5 * (grid["n_th"] - 1) * grid["th_step"]
        q.ty_step = q.tz_step = grid["t_step"]
        q.ty_min = -0.5 * (grid["n_ty"] - 1) * grid["t_step"]
        q.tz_min = -0.5 * (grid["n_tz"] - 1) * grid["t_step"]
    e = LidarCornersBatch(16, 28800, p)
    res = e.extract(clouds, clicks)
    tm = e.timing()
    wide = 3.0 * grid["t_step"] >= 0.9 * p.grid_length
    assert (tm.grid_cost_box_evals_sum == 0) == wide
    n = 0
    for f in range(len(clouds)):
        ref = ob.extract(clouds[f], clicks[f], op)
        assert res[f].status == ref.status
        if ref.status in (0, 11):
            assert res[f].grid_index == ref.grid_index
            assert tuple(res[f].theta_t) == tuple(ref.theta_t) and res[f].sel_cost == ref.sel_cost
            n += 1
    assert n >= 12
    e.close()


def test_common_prepass_leaves_out_points_far_from_the_rotation_centre(ob):
    """k6_triple_prepass bounds five thetas at once from the extremes of a point's five rotated images; a point whose images lie
    more than half a square apart (further than ~2 m from the plane frame's origin on the default grid) is left out of that
    bound.  A 5 m wide 'board' -- a wall patch with a chequered intensity -- puts a third of the labelled points there: grid
    argmin, theta_t, costs and corners must still be the oracle's."""
    rng = np.random.default_rng(9)
    yy, zz = np.meshgrid(np.arange(-2.5, 2.5, 0.04), np.arange(-1.0, 1.0, 0.04), indexing="ij")
    y, z = yy.ravel() + rng.normal(0, 0.002, yy.size), zz.ravel() + rng.normal(0, 0.002, yy.size)
    th = 0.07                                                      # the pattern is turned against the wall's principal axes
    u, v = np.cos(th) * y - np.sin(th) * z + 0.031, np.sin(th) * y + np.cos(th) * z - 0.022
    white = ((np.floor(u / 0.15) + np.floor(v / 0.15)) % 2) == 0
    inten = np.where(white, 100.0, 12.0) + rng.normal(0, 4.0, y.size)
    cloud = np.stack([np.full(y.size, 3.0) + rng.normal(0, 0.003, y.size), y, z, inten], 1).astype(np.float32)
    click = np.array([3.0, 0.0, 0.0], np.float32)
    p = N.default_params()
    p.roi_half[0], p.roi_half[1], p.roi_half[2] = 1.0, 3.0, 2.0
    e = LidarCornersBatch(1, len(cloud), p)
    e.reserve(8192, 8192)
    r = e.extract(cloud[None], click[None])[0]
    tm = e.timing()
    e.close()
    op = _oparams(ob, N.SOLVER_GRID)
    op.roi_half[0], op.roi_half[1], op.roi_half[2] = 1.0, 3.0, 2.0
    o = ob.extract(cloud, click, op)
    assert 4096 < r.n_black + r.n_white <= 8192                     # staged in LDS: the pre-passes run
    assert tm.grid_cost_box_evals_sum > 0
    assert (r.status, r.n_roi, r.n_cluster, r.n_plane) == (o.status, o.n_roi, o.n_cluster, o.n_plane)
    assert r.status in (N.OK, N.AMBIGUOUS)
    assert r.grid_index == o.grid_index and tuple(r.theta_t) == tuple(o.theta_t)
    assert (r.iters_a, r.iters_b, r.sel_cost, r.basin_margin) == (o.iters_a, o.iters_b, o.sel_cost, o.basin_margin)
    assert np.abs(r.corners_array() - ob.result_corners(o)).max() < 1e-6


def test_async_submit_wait_matches_synchronous_calls(frames):
    """Four batches in flight (submit/wait) return exactly what four synchronous calls return."""
    import torch
    clouds, clicks, _ = frames
    dev = torch.device("cuda", 0)
    e = LidarCornersBatch(8, 28800, N.default_params())
    sets = [(0, 5), (5, 9), (9, 13), (13, 16)]
    sync = [[r.corners_array() for r in e.extract(clouds[a:b], clicks[a:b])] for a, b in sets]
    d = [(torch.from_numpy(clouds[a:b].copy()).to(dev), torch.from_numpy(clicks[a:b].copy()).to(dev)) for a, b in sets]
    torch.cuda.synchronize()
    tickets = [e.submit_device(dc.data_ptr(), len(dk), 28800, dk.data_ptr()) for dc, dk in d]
    with pytest.raises(Exception):
        e.submit_device(d[0][0].data_ptr(), len(d[0][1]), 28800, d[0][1].data_ptr())   # all slots taken
    for (a, b), t, want in zip(sets, tickets, sync):
        got = [r.corners_array() for r in e.wait(t)]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    assert len(e.fetch_cloud(0, N.CLOUD_CHESSBOARD)) > 100          # last completed batch is inspectable
    # host-input submit (ilcc_submit_batch): the H2D copy rides on the batch's stream; same results again
    pinned = [(torch.from_numpy(clouds[a:b].copy()).pin_memory(), torch.from_numpy(clicks[a:b].copy()).pin_memory()) for a, b in sets]
    tickets = [e.submit_host(pc.data_ptr(), len(pk), 28800, pk.data_ptr()) for pc, pk in pinned]
    for t, want in zip(tickets, sync):
        got = [r.corners_array() for r in e.wait(t)]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    # pageable host memory works too (the runtime stages the copy; slower, same results)
    pageable = [(np.ascontiguousarray(clouds[a:b]), np.ascontiguousarray(clicks[a:b])) for a, b in sets[:2]]
    tickets = [e.submit_host(pc.ctypes.data, len(pk), 28800, pk.ctypes.data) for pc, pk in pageable]
    for t, want in zip(tickets, sync[:2]):
        got = [r.corners_array() for r in e.wait(t)]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    e.close()


def test_full_size_batch_properties():
    """BASELINE config 4's per-GPU shard (128 x 28 800 points): determinism, batch-composition
    independence and permutation equivariance -- size-independent properties, no oracle needed."""
    F = 128
    clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
    e = LidarCornersBatch(F, 28800, N.default_params())
    r1 = np.array([np.ctypeslib.as_array(r.corners)[:105].copy() for r in e.extract(clouds, clicks)])
    full = e.extract(clouds, clicks)
    s1 = [r.status for r in full]
    low = [bool(r.flags & N.FLAG_LOW_COVERAGE) for r in full]
    r2 = np.array([np.ctypeslib.as_array(r.corners)[:105].copy() for r in e.extract(clouds, clicks)])
    assert np.array_equal(r1, r2)                                   # bitwise repeatable
    perm = np.random.default_rng(0).permutation(F)
    r3 = np.array([np.ctypeslib.as_array(r.corners)[:105].copy() for r in e.extract(clouds[perm], clicks[perm])])
    assert np.array_equal(r3, r1[perm])                             # frames are independent
    sub = np.array([np.ctypeslib.as_array(r.corners)[:105].copy() for r in e.extract(clouds[5:9], clicks[5:9])][:4])
    assert np.array_equal(sub, r1[5:9])
    ok = [f for f in range(F) if s1[f] == 0]
    amb = [f for f in range(F) if s1[f] == N.AMBIGUOUS]
    assert len(ok) + len(amb) >= 0.95 * F                           # corners delivered (OK or flagged ambiguous)
    err = np.array([synth.corner_error(r1[f].reshape(35, 3), gts[f], BOARD) for f in ok])
    assert np.median(err) < 0.004                                   # 2.7-2.9 mm: the floor of this sensor model (profiles/r03_noise_floor_study.json)
    # the recommended accept rule (include/ilcc_hip.h): status OK and the coverage flag clear -- no accepted frame is a
    # square (150 mm) off, and the rule keeps >= 90 % of the batch
    acc = [f for f in ok if not low[f]]
    err_acc = np.array([synth.corner_error(r1[f].reshape(35, 3), gts[f], BOARD) for f in acc])
    # (measured on these 128 frames: 119 accepted, worst accepted frame 16.7 mm, p99 of the OK frames 9.1 mm)
    assert len(acc) >= 0.92 * F and err_acc.max() < 0.02, (len(acc), err_acc.max())
    assert np.percentile(err, 99) < 0.010 or np.sort(err)[-3] < 0.010, np.sort(err)[-4:]   # all but the one or two one-square slips
    # every result is an exact planar 0.15 m lattice (what the consumer relies on)
    for f in ok[:16]:
        g = r1[f].reshape(5, 7, 3)
        assert np.allclose(np.linalg.norm(np.diff(g, axis=1), axis=-1), 0.15, atol=1e-5)
        assert np.allclose(np.linalg.norm(np.diff(g, axis=0), axis=-1), 0.15, atol=1e-5)
    e.close()


def config5_params(p):
    """BASELINE config 5: 11 x 8 corners @0.10 m, FINE grid ty, tz in [-g, g] step g/64 (129 x 129),
    theta in [-16, 16] deg step 0.25 deg (129): 2 146 689 candidates x 2 phases."""
    p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
    p.n_th, p.n_ty, p.n_tz = 129, 129, 129
    p.th_min, p.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    p.ty_min = p.tz_min = -0.10
    p.ty_step = p.tz_step = 0.10 / 64
    return p


def test_config5_dense_64_ring_cloud_fine_grid(ob):
    """BASELINE config 5 as specified: 64 rings x 2048 azimuths = 131 072 points, 11x8-corner board @0.10 m, the
    129 x 129 x 129 x 2 grid.  Grid argmin index, refinement and corners vs the oracle (whose exhaustive search
    uses an exact integer bound, so it finishes in seconds)."""
    board = synth.Board(9, 12, 0.10)
    e = LidarCornersBatch(4, 131072, config5_params(N.default_params()))
    op = config5_params(ob.default_params())
    op.solver = ob.SOLVER_GRID
    clouds, clicks, poses = [], [], []
    for k in range(4):
        rng = np.random.default_rng(11 + k)
        pose = synth.random_pose(rng, range_m=(2.2, 2.6), yaw_deg=15, pitch_deg=10, roll_deg=25)
        clouds.append(synth.make_frame(synth.hdl64(), board, pose, 77 + k))
        clicks.append(synth.make_click(pose, 77 + k))
        poses.append(pose)
    res = e.extract(np.stack(clouds), np.stack(clicks))
    tm = e.timing()
    assert tm.grid_cost_evals_nominal_sum >= 4 * 3000 * 129 ** 3        # the fine grid really ran
    for k in range(4):
        r = res[k]
        o = ob.extract(clouds[k], clicks[k], op)
        assert r.status == o.status == N.OK and r.n_corners == 88
        assert (r.n_roi, r.n_cluster, r.n_plane) == (o.n_roi, o.n_cluster, o.n_plane)
        assert r.grid_index == o.grid_index
        assert tuple(r.theta_t) == tuple(o.theta_t) and (r.iters_a, r.iters_b) == (o.iters_a, o.iters_b)
        assert np.abs(r.corners_array() - ob.result_corners(o)).max() < 1e-6
        assert synth.corner_error(r.corners_array(), synth.true_corners(poses[k], board), board) < 0.01
    e.close()


def test_large_roi_takes_the_global_memory_paths(ob):
    """A 12 x 12 x 12 m ROI over a 64-ring cloud: > 16 384 ROI points in a bounding grid far larger than K2's cell bitmap (the
    hashed-block cell table in global memory takes the frame) and more labelled
    points than a fresh handle's K6 / K7 LDS stage: these variants must give the oracle's result."""
    board = synth.Board(9, 12, 0.10)
    rng = np.random.default_rng(21)
    pose = synth.random_pose(rng, range_m=(2.0, 2.3), yaw_deg=10, pitch_deg=8, roll_deg=20)
    cloud = synth.make_frame(synth.hdl64(), board, pose, 5)
    click = synth.make_click(pose, 5)
    p = N.default_params()
    p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
    p.roi_half[0], p.roi_half[1], p.roi_half[2] = 6.0, 6.0, 6.0      # keeps ~ every near point
    p.n_th, p.n_ty, p.n_tz = 9, 8, 8
    p.th_min, p.th_step = -0.04, 0.01
    p.ty_min = p.tz_min = -0.04
    p.ty_step = p.tz_step = 0.01
    e = LidarCornersBatch(1, 131072, p)
    r = e.extract(cloud[None], click[None])[0]
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    for k in ("board_w", "board_h", "grid_length", "n_th", "n_ty", "n_tz", "th_min", "th_step", "ty_min",
              "ty_step", "tz_min", "tz_step"):
        setattr(op, k, getattr(p, k))
    for a in range(3):
        op.roi_half[a] = 6.0
    o = ob.extract(cloud, click, op)
    assert r.n_roi > 16384, r.n_roi
    assert (r.status, r.n_roi, r.n_cluster, r.n_plane) == (o.status, o.n_roi, o.n_cluster, o.n_plane)
    assert r.status in (N.OK, N.AMBIGUOUS) and r.grid_index == o.grid_index
    assert tuple(r.theta_t) == tuple(o.theta_t)
    assert np.abs(r.corners_array() - ob.result_corners(o)).max() < 1e-6
    e.close()


def test_large_roi_one_workgroup_and_multi_workgroup_clustering_agree(ob):
    """K2's three ways through a dense ROI (~12 k points, ~1 300 occupied cells): a fresh handle's cell arrays hold 512 cells,
    so its first call clusters these frames with the point-level spatial hash (its own workgroup: the multi-workgroup
    kernels are not armed); its second call, the capacity grown, takes the cell-level path with the sorted points in
    HBM; a reserved handle takes that path at once -- identical clusters, planes and corners (and the oracle's: previous test)."""
    board = synth.Board(9, 12, 0.10)
    rng = np.random.default_rng(33)
    clouds, clicks = [], []
    for k in range(3):
        pose = synth.random_pose(rng, range_m=(2.0, 2.6), yaw_deg=10, pitch_deg=8, roll_deg=20)
        clouds.append(synth.make_frame(synth.hdl64(), board, pose, 40 + k))
        clicks.append(synth.make_click(pose, 40 + k))
    clouds, clicks = np.stack(clouds), np.stack(clicks)
    p = N.default_params()
    p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
    p.n_th, p.n_ty, p.n_tz = 9, 8, 8
    p.th_min, p.th_step = -0.04, 0.01
    p.ty_min = p.tz_min = -0.04
    p.ty_step = p.tz_step = 0.01
    cold = LidarCornersBatch(3, 131072, p)
    r_cold = cold.extract(clouds, clicks)            # more occupied cells than a fresh handle's 512: point-level hash, unarmed
    r_warm = cold.extract(clouds, clicks)            # capacity grown: components on cells, sorted points in HBM
    res = LidarCornersBatch(3, 131072, p)
    res.reserve(4500, 25000)
    r_res = res.extract(clouds, clicks)              # cell capacity reserved up front (25 000 ROI points also arm the multi-workgroup kernels: unused here)
    assert min(r.n_roi for r in r_cold) > 4096
    for a, b, c in zip(r_cold, r_warm, r_res):
        for r in (b, c):
            assert (a.status, a.n_roi, a.n_cluster, a.n_plane, a.grid_index) == (r.status, r.n_roi, r.n_cluster, r.n_plane, r.grid_index)
            assert tuple(a.theta_t) == tuple(r.theta_t) and np.array_equal(a.corners_array(), r.corners_array())
    for f in range(3):
        assert np.array_equal(cold.fetch_cloud(f, N.CLOUD_CLUSTER), res.fetch_cloud(f, N.CLOUD_CLUSTER))
    cold.close()
    res.close()


def test_reserved_handle_runs_its_first_batch_like_a_warmed_one(frames):
    """ilcc_reserve: the staging capacities no longer depend on what the handle has seen.  A frame
    with more labelled points than a fresh handle's 1024 takes K6's LDS walk on the FIRST call of a reserved handle,
    exactly like on a warmed one: same results, executed evaluations within the run-to-run spread of the branch and bound."""
    board = synth.Board()
    rng = np.random.default_rng(77)
    clouds, clicks = [], []
    for k in range(4):
        pose = synth.random_pose(rng, range_m=(2.0, 2.2), yaw_deg=5, pitch_deg=5, roll_deg=10)   # close: > 1024 labelled points
        clouds.append(synth.make_frame(synth.vlp16(), board, pose, 700 + k))
        clicks.append(synth.make_click(pose, 700 + k))
    clouds, clicks = np.stack(clouds), np.stack(clicks)
    warm = LidarCornersBatch(4, 28800, N.default_params())
    r0 = warm.extract(clouds, clicks)
    assert max(r.n_black + r.n_white for r in r0) > 1024
    cold_evals = int(warm.timing().grid_cost_evals_sum)      # fresh handle: anchor + full pass walked the points through L2
    warm.reset_timing()
    r1 = warm.extract(clouds, clicks)
    t1 = warm.timing()
    res = LidarCornersBatch(4, 28800, N.default_params())
    res.reserve(2048, 2560)
    r2 = res.extract(clouds, clicks)
    t2 = res.timing()
    # same kernels, same walk: executed evaluations agree within the run-to-run spread of the branch and bound (the L2 walk
    # of the fresh handle has one class of points and another order: its count is not comparable, only reported)
    assert abs(int(t1.grid_cost_evals_sum) - int(t2.grid_cost_evals_sum)) <= 0.10 * t1.grid_cost_evals_sum, (cold_evals, t1.grid_cost_evals_sum, t2.grid_cost_evals_sum)
    print("executed K6 evaluations: fresh handle %d, warmed %d, reserved (first call) %d" % (cold_evals, t1.grid_cost_evals_sum, t2.grid_cost_evals_sum))
    for a, b, c in zip(r0, r1, r2):
        assert (a.status, a.grid_index) == (b.status, b.grid_index) == (c.status, c.grid_index)
        assert tuple(a.theta_t) == tuple(b.theta_t) == tuple(c.theta_t)
        assert np.array_equal(b.corners_array(), c.corners_array())
        # round 6: the fresh handle's full pass stages the first 1024 walk positions and reads the rest of the same walk through L2
        # (grid_cost_body<OVERFLOW>): the same points in the same order -- the fp32 cost of the argmin is the same float
        assert a.grid_cost == b.grid_cost == c.grid_cost
    assert abs(cold_evals - int(t1.grid_cost_evals_sum)) <= 0.25 * t1.grid_cost_evals_sum, (cold_evals, t1.grid_cost_evals_sum)
    warm.close()
    res.close()


def test_reserved_handle_first_batch_config5():
    """The same on BASELINE config 5's dense frames (a handle reserved for 4500 labelled points issued
    4.3 x the K6 instructions on its first batch -- frames above the reserved capacity walk their points through L2).
    Reserved for what these frames hold, the first call executes what a warmed handle's call executes."""
    board = synth.Board(9, 12, 0.10)
    clouds, clicks, _, _ = synth.make_batch(4, synth.hdl64(), board, seed=0xC0FFEE, range_m=(2.0, 3.0), yaw_deg=25.0,
                                            pitch_deg=15.0, roll_deg=30.0)
    warm = LidarCornersBatch(4, 131072, config5_params(N.default_params()))
    r0 = warm.extract(clouds, clicks)
    m_max = max(r.n_black + r.n_white for r in r0)
    roi_max = max(r.n_roi for r in r0)
    cold = int(warm.timing().grid_cost_evals_sum)
    warm.reset_timing()
    r1 = warm.extract(clouds, clicks)
    t1 = warm.timing()
    res = LidarCornersBatch(4, 131072, config5_params(N.default_params()))
    res.reserve(m_max, roi_max)
    r2 = res.extract(clouds, clicks)
    t2 = res.timing()
    print("config 5: labelled points <= %d, ROI points <= %d; executed K6 evaluations: fresh %d, warmed %d, reserved (first call) %d"
          % (m_max, roi_max, cold, t1.grid_cost_evals_sum, t2.grid_cost_evals_sum))
    assert abs(int(t1.grid_cost_evals_sum) - int(t2.grid_cost_evals_sum)) <= 0.10 * t1.grid_cost_evals_sum
    for a, b, c in zip(r0, r1, r2):
        assert (a.status, a.grid_index) == (b.status, b.grid_index) == (c.status, c.grid_index)
        assert tuple(a.theta_t) == tuple(b.theta_t) == tuple(c.theta_t)
        assert np.array_equal(b.corners_array(), c.corners_array())
    warm.close()
    res.close()


def test_frames_above_the_staging_capacity_config5():
    """Round 6: a frame with more labelled points than the K6 full pass stages per workgroup keeps its box pre-passes -- the first
    `grid_lds_points` walk positions in LDS, the rest of the same walk read through L2 (grid_cost_body<OVERFLOW>) -- instead of
    falling to the LDS-free body.  BASELINE config 5's dense frames on a handle reserved for FEWER labelled points than they hold
    vs a handle that stages them whole: the same walk, the same sums -- argmin, its fp32 cost, theta_t, corners identical; and the
    executed evaluations stay of the same order (the LDS-free body executes ~100 x as many)."""
    board = synth.Board(9, 12, 0.10)
    clouds, clicks, _, _ = synth.make_batch(4, synth.hdl64(), board, seed=0xC0FFEE, range_m=(2.0, 3.0), yaw_deg=25.0,
                                            pitch_deg=15.0, roll_deg=30.0)
    whole = LidarCornersBatch(4, 131072, config5_params(N.default_params()))
    whole.reserve(6400, 20000)
    r_whole = whole.extract(clouds, clicks)
    m = [r.n_black + r.n_white for r in r_whole]
    assert min(m) > 3200 and max(m) <= 6400, m
    # every frame alone on a handle whose staging is the largest multiple of 256 BELOW its labelled count: above the capacity, with
    # the interior class (~ half of the points) and the pre-pass's sample (a quarter) inside the staged prefix
    for f, a in enumerate(r_whole):
        cap = ((m[f] - 1) // 256) * 256
        small = LidarCornersBatch(1, 131072, config5_params(N.default_params()))
        small.reserve(cap, 20000)
        one = LidarCornersBatch(1, 131072, config5_params(N.default_params()))
        one.reserve(6400, 20000)
        b = small.extract(clouds[f:f + 1], clicks[f:f + 1])[0]
        w = one.extract(clouds[f:f + 1], clicks[f:f + 1])[0]
        ev_small, ev_one = int(small.timing().grid_cost_evals_sum), int(one.timing().grid_cost_evals_sum)
        for r in (b, w):
            assert (a.status, a.grid_index, a.grid_cost) == (r.status, r.grid_index, r.grid_cost), (f, m[f], cap)
            assert tuple(a.theta_t) == tuple(r.theta_t) and (a.iters_a, a.iters_b, a.basin_margin, a.flags) == (r.iters_a, r.iters_b, r.basin_margin, r.flags)
            assert np.array_equal(a.corners_array(), r.corners_array())
        # (a handle below the frame's count also sends the SEED launches of a small batch through the LDS-free body -- another sample,
        # another bound: the full pass may execute several times less or somewhat more; what must not happen is the ~100 x of a full
        # pass without box pre-passes)
        assert ev_small <= 3 * ev_one, (f, ev_small, ev_one)
        small.close()
        one.close()
    whole.close()


def test_cluster_size_gates_and_non_finite_points(ob, frames):
    """EuclideanClusterExtraction's [min, max] size gate: when the click's component is inadmissible the
    largest admissible one is taken (plane_index 0); NaN/inf points never reach the clustering."""
    clouds, clicks, _ = frames
    cloud = clouds[0].copy()
    cloud[::97, 0] = np.nan
    cloud[5::131, 2] = np.inf
    for cmin, cmax in ((100, 25000), (100, 1500), (2500, 25000)):
        p = N.default_params()
        p.cluster_min, p.cluster_max = cmin, cmax
        e = LidarCornersBatch(1, 28800, p)
        r = e.extract(cloud[None], clicks[:1])[0]
        op = ob.default_params()
        op.solver = ob.SOLVER_GRID
        op.cluster_min, op.cluster_max = cmin, cmax
        o = ob.extract(cloud, clicks[0], op)
        assert (r.status, r.n_roi, r.n_cluster, r.n_plane) == (o.status, o.n_roi, o.n_cluster, o.n_plane), (cmin, cmax)
        roi = e.fetch_cloud(0, N.CLOUD_ROI)
        assert np.isfinite(roi[:, :3]).all()
        if o.status in (N.OK, N.AMBIGUOUS):
            assert np.abs(r.corners_array() - ob.result_corners(o)).max() < 1e-6
        e.close()


def test_cluster_threshold_geometry_matches_the_oracle(ob):
    """K2 builds its components on cells of side 0.57 tol (a cell is a clique; neighbours are looked for within +-2 cells per
    axis): geometry made to sit on those thresholds.  Clumps of coincident points whose mutual distances straddle the 0.12 m
    tolerance by 1e-5 m (along an axis, along face and space diagonals, across the 2-cell reach), points ON cell borders of the
    bounding grid, a chain that only holds together through near-threshold links -- 24 ragged frames, each compared point for
    point with the oracle's BFS clustering of the same ROI cloud."""
    rng = np.random.default_rng(2024)
    tol = 0.12
    frames, clicks = [], []
    dirs = [np.array(d, np.float64) / np.linalg.norm(d) for d in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1),
                                                                    (1, -1, 1), (2, 1, 0), (1, 2, 2), (-1, 1, 2), (3, 1, 2))]
    for t in range(24):
        base = np.array([2.5, 0.0, 0.0]) + rng.uniform(-0.2, 0.2, 3)
        pts = []
        # a chain of clumps: link k is just inside (even k) or just outside (odd k) the tolerance
        at = base.copy()
        for k in range(10):
            n = int(rng.integers(25, 60))
            pts.append(np.repeat(at[None], n, 0) + (rng.uniform(-1e-4, 1e-4, (n, 3)) if k % 3 == 0 else 0.0))
            d = dirs[(t + k) % len(dirs)]
            at = at + d * (tol + (1e-5 if k % 2 else -1e-5) + 2e-4 * (k % 3 == 0))   # (the jittered clumps need a little more room)
        # points on the cell borders of a 0.0684 m lattice anchored at the cloud's minimum, and a dense sheet
        lat = base + np.array([-0.6, -0.6, -0.3]) + 0.57 * tol * np.stack(np.meshgrid(np.arange(9), np.arange(9), np.arange(3), indexing="ij"), -1).reshape(-1, 3)
        pts.append(lat)
        sheet = base + np.array([0.3, 0.0, 0.0]) + np.stack([np.zeros(400), rng.uniform(-0.4, 0.4, 400), rng.uniform(-0.3, 0.3, 400)], 1)
        pts.append(sheet)
        xyz = np.concatenate(pts).astype(np.float32)
        cloud = np.concatenate([xyz, np.full((len(xyz), 1), 40.0, np.float32)], 1)
        frames.append(cloud[rng.permutation(len(cloud))])
        clicks.append((base if t % 2 else base + np.array([0.3, 0.0, 0.0])).astype(np.float32))
    off = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).astype(np.uint64)
    p = N.default_params()
    p.cluster_min = 20
    e = LidarCornersBatch(len(frames), int(off[-1]), p)
    res = e.extract(np.concatenate(frames), np.stack(clicks), offsets=off)
    op = ob.default_params()
    op.cluster_min = 20
    sizes = set()
    for f, cloud in enumerate(frames):
        roi = cloud[ob.roi_crop(cloud, clicks[f], op)]
        idx, _ = ob.cluster(roi, clicks[f], op)
        got = e.fetch_cloud(f, N.CLOUD_CLUSTER)
        assert res[f].n_roi == len(roi) and res[f].n_cluster == len(idx), (f, res[f].n_cluster, len(idx))
        assert np.array_equal(got, roi[idx]), f
        sizes.add(len(idx))
    assert len(sizes) > 3            # the near-threshold links really produce different partitions
    e.close()


def test_online_caller_get_chessboard_by_point(ob):
    """get_chessboard_by_point (no ROI crop, tolerance 0.10, plane >= 500 points) + gray-zone
    colouring on whole 28 800-point clouds: the two-tier online path (a verified window around the click on K2's LDS cell
    grid, the whole cloud through the hashed-block k2h_* chain for the frames the window cannot vouch for) vs the oracle's BFS
    clustering."""
    clouds, _, _, poses = synth.make_batch(6, fixture_poses=True, seed=900)
    pts = np.stack([(p.centre + [0.04, -0.05, 0.03]) for p in poses]).astype(np.float32)
    pts[4] = [-3.0, 2.0, 5.0]                            # predicted centre off the board: whatever surface is nearest wins
    pts[3] = [0.3, 0.0, -0.9]                            # near the sparse ground rings close to the sensor
    clouds = clouds.copy()
    clouds[5, ::5, :3] = np.nan                          # a sparser, partly non-finite cloud
    p = N.default_params()
    assert p.online_cluster_tol == 0.10 and p.cluster_tol == 0.12   # LidarCornersEst.cpp:80 vs :131 -- the default IS the reference's
    p.gray_rate = 2.4                                    # launch/lidar_chessboard_online.launch:14
    e = LidarCornersBatch(6, 28800, p)
    res = e.chessboard_by_point(clouds, pts)
    op = ob.default_params()
    op.cluster_tol, op.gray_rate = 0.10, 2.4
    n_ok = 0
    for f in range(6):
        o, ocb, ocl = ob.chessboard_by_point(clouds[f], pts[f], op)
        r = res[f]
        assert r.status == o.status, (f, r.status, o.status)
        assert (r.n_roi, r.n_cluster, r.n_plane, r.found_board) == (o.n_roi, o.n_cluster, o.n_plane, o.phase), f
        if o.status in (0, N.BOARD_NOT_FOUND) and o.n_plane >= 3:
            assert np.array_equal(e.fetch_cloud(f, N.CLOUD_CHESSBOARD), ocb)
            assert np.allclose(r.gray_zone, o.gray_zone, rtol=1e-12)
            assert np.array_equal(e.fetch_classes(f), ocl)
            assert (r.n_black, r.n_gray, r.n_white) == (o.n_black, o.n_gray, o.n_white)
        n_ok += int(r.status == 0)
    assert n_ok >= 3
    e.close()
    m = LidarCornersEst(max_points_per_frame=28800, params=p)
    ok, out = m.get_chessboard_by_point(clouds[0], pts[0])
    assert ok and len(out) == res[0].n_plane
    rgb = m.color_by_gray_zone()
    assert rgb.shape == (len(out), 3) and set(map(tuple, np.unique(rgb, axis=0))) <= {(10, 10, 10), (255, 0, 0), (255, 255, 255)}
    ok2, out2 = m.get_chessboard_by_point(clouds[4], pts[4])
    assert ok2 == (res[4].status == 0) and len(out2) == res[4].n_plane
    # the front-half record of get_chessboard_by_point is never mistaken for a finished extraction: the
    # reference's call sequence afterwards runs the whole path
    assert np.allclose(m.get_gray_zone(), res[4].gray_zone if ok2 else m.get_gray_zone())
    m.setROI(clouds[0], pts[0])
    assert m.EuclideanCluster() is True
    m.PCA()
    corners = []
    got = m.get_corners(corners)
    assert m.result.n_corners == 35 and (len(corners) == 35) == got
    m.close()


def test_online_caller_in_flight_equals_the_synchronous_call():
    """ilcc_submit_chessboard_by_point / ilcc_wait_chessboard_by_point (up to four calls in flight, H2D on each call's own
    stream) return what ilcc_chessboard_by_point_batch returns for the same clouds -- every field of the records, the plane
    clouds and the classes of the call last waited for -- also when the calls differ (three different batches in flight)."""
    import torch
    batches = []
    for k in range(3):
        clouds, _, _, poses = synth.make_batch(5, seed=1200 + k)
        pts = np.stack([(p.centre + [0.03, -0.04, 0.02]) for p in poses]).astype(np.float32)
        if k == 1:
            pts[2] = [-3.0, 2.0, 5.0]                     # nothing in the window: second tier
        batches.append((np.ascontiguousarray(clouds), pts))
    p = N.default_params()
    p.gray_rate = 2.4
    e = LidarCornersBatch(5, 28800, p)
    want = []
    for clouds, pts in batches:
        res = e.chessboard_by_point(clouds, pts)
        want.append(([bytes(r) for r in res], [e.fetch_cloud(f, N.CLOUD_CHESSBOARD) for f in range(5)], [e.fetch_classes(f) for f in range(5)]))
    pinned = [(torch.from_numpy(c).pin_memory(), torch.from_numpy(q).pin_memory()) for c, q in batches]
    tickets = [e.submit_chessboard_by_point(c.data_ptr(), 5, 28800, q.data_ptr()) for c, q in pinned]
    for k, t in enumerate(tickets):
        res = e.wait_chessboard_by_point(t)
        assert [bytes(r) for r in res] == want[k][0], k
        for f in range(5):
            assert np.array_equal(e.fetch_cloud(f, N.CLOUD_CHESSBOARD), want[k][1][f]), (k, f)
            assert np.array_equal(e.fetch_classes(f), want[k][2][f]), (k, f)
    assert sum(1 for r in res if r.status == 0) >= 3
    e.close()


def _patch(rng, centre, u, v, half_u, half_v, pitch, jitter=0.002):
    """points of a planar patch: a lattice of `pitch` metres spanned by the unit vectors u, v, a little noise along the normal"""
    a = np.arange(-half_u, half_u + 1e-9, pitch)
    b = np.arange(-half_v, half_v + 1e-9, pitch)
    A, B = np.meshgrid(a, b, indexing="ij")
    n = np.cross(u, v)
    pts = centre + A.reshape(-1, 1) * u + B.reshape(-1, 1) * v + rng.normal(0, jitter, (A.size, 1)) * n
    return np.concatenate([pts, rng.uniform(5, 90, (len(pts), 1))], axis=1).astype(np.float32)


def test_a_ticket_is_waited_for_by_its_own_call_family():
    """tickets carry their kind.  A ticket of ilcc_submit_chessboard_by_point refused by ilcc_wait / ilcc_wait_compact
    (they would skip the by-point epilogue), a ticket of ilcc_submit_batch refused by ilcc_wait_chessboard_by_point (it would run that
    epilogue on staging no copy has filled); the refused batch stays in flight and its own wait still delivers it."""
    from lidar_camera_calibration_amd import IlccError
    clouds, clicks, gts, _ = synth.make_batch(4, seed=77)
    pts = np.ascontiguousarray(gts.mean(axis=1), dtype=np.float32)
    e = LidarCornersBatch(4, 28800, N.default_params())
    want = [(r.status, r.n_plane) for r in e.chessboard_by_point(clouds, pts)]
    c, k = np.ascontiguousarray(clouds), np.ascontiguousarray(pts)
    t = e.submit_chessboard_by_point(c.ctypes.data, 4, 28800, k.ctypes.data)
    with pytest.raises(IlccError) as ei:
        e.wait(t)
    assert ei.value.status == N.BAD_ARGUMENT and "ilcc_wait_chessboard_by_point" in str(ei.value)
    with pytest.raises(IlccError):
        e.wait_compact(t)
    assert [(r.status, r.n_plane) for r in e.wait_chessboard_by_point(t)] == want
    ck = np.ascontiguousarray(clicks)
    ref = [(r.status, r.n_plane, r.n_roi) for r in e.extract(clouds, clicks)]
    t = e.submit_host(c.ctypes.data, 4, 28800, ck.ctypes.data)
    with pytest.raises(IlccError) as ei:
        e.wait_chessboard_by_point(t)
    assert ei.value.status == N.BAD_ARGUMENT and "ilcc_submit_batch" in str(ei.value)
    assert [(r.status, r.n_plane, r.n_roi) for r in e.wait(t)] == ref
    e.close()


def test_online_caller_two_tiers_agree_with_the_oracle_on_the_tier_boundary(ob):
    """The online caller answers from a +-1.25 m window around the click when it can PROVE the window's answer is the whole
    cloud's, and reruns the frame on the whole cloud otherwise (k2_cluster.hip, fine_cluster_frame's first-tier checks).  Scenes
    built to sit on each side of each check: a compact patch (tier 1); the same patch with a strip that reaches to 1.10 m /
    1.16 m from the click (inside / outside the 'no member within tol of a face' margin of 1.149 m); a wall wider than the
    window; a clump of 60 points at the click (inadmissible: the reference falls back to the LARGEST component of the whole
    cloud, here a far wall the window never sees); a click with nothing within the window; an empty cloud.  Every frame must
    give the oracle's counts, plane cloud and classes, and exactly the frames that need it take the second tier."""
    rng = np.random.default_rng(5150)
    ex, ey, ez = np.eye(3)
    click = np.array([3.0, 0.2, 0.1])
    far_wall = _patch(rng, np.array([9.0, -4.0, 0.5]), ey, ez, 1.5, 1.0, 0.04)          # ~ 3 900 points, > 2 m from every window
    compact = _patch(rng, click + [0.02, 0, 0], ey, ez, 0.45, 0.35, 0.025)              # ~ 1 070 points
    def strip(reach):    # a 6 cm wide strip from the patch's edge out to `reach` metres from the click along +y
        return _patch(rng, click + [0.02, (0.45 + reach) / 2, 0.0], ey, ez, (reach - 0.45) / 2, 0.03, 0.025)
    scenes = [
        (np.concatenate([compact, far_wall]), click, False),
        (np.concatenate([compact, strip(1.10), far_wall]), click, False),
        (np.concatenate([compact, strip(1.16), far_wall]), click, True),
        (np.concatenate([compact, strip(1.60), far_wall]), click, True),
        (np.concatenate([_patch(rng, click + [0.02, 0, 0], ey, ez, 2.0, 0.6, 0.03), far_wall]), click, True),   # wider than the window
        (np.concatenate([_patch(rng, click, ey, ez, 0.05, 0.04, 0.012), far_wall]), click, True),                 # 60-odd points at the click
        (np.concatenate([compact, far_wall]), click + [0.0, 2.0, 1.5], True),                                     # nothing in the window
        (np.full((64, 4), np.nan, dtype=np.float32), click, True),                                               # no finite point at all
    ]
    n_max = max(len(c) for c, _, _ in scenes)
    clouds = np.full((len(scenes), n_max, 4), np.nan, dtype=np.float32)     # ragged frames padded with non-finite points (dropped by K1)
    for k, (c, _, _) in enumerate(scenes):
        clouds[k, :len(c)] = c[rng.permutation(len(c))]
    pts = np.stack([q for _, q, _ in scenes]).astype(np.float32)
    p = N.default_params()
    p.gray_rate = 2.4
    e = LidarCornersBatch(len(scenes), n_max, p)
    e.reset_timing()
    res = e.chessboard_by_point(clouds, pts)
    second = e.timing().online_second_tier_frames
    op = ob.default_params()
    op.cluster_tol, op.gray_rate = 0.10, 2.4
    for f in range(len(scenes)):
        o, ocb, ocl = ob.chessboard_by_point(clouds[f], pts[f], op)
        r = res[f]
        assert r.status == o.status, (f, r.status, o.status)
        assert (r.n_roi, r.n_cluster, r.n_plane, r.found_board) == (o.n_roi, o.n_cluster, o.n_plane, o.phase), f
        if o.status in (0, N.BOARD_NOT_FOUND) and o.n_plane >= 3:
            assert np.array_equal(e.fetch_cloud(f, N.CLOUD_CHESSBOARD), ocb), f
            assert np.array_equal(e.fetch_classes(f), ocl), f
    assert res[0].found_board and res[1].found_board and res[3].found_board and not res[5].found_board
    assert res[5].n_cluster > 3000                       # the far wall: the largest admissible component of the WHOLE cloud
    assert second == sum(1 for _, _, t2 in scenes if t2), second
    e.close()


def test_device_records_equal_host_packing():
    """ilcc_wait_records_device (K9: records packed on the GPU for the RCCL gather) == sharding.pack_records
    of the same results, including failed frames (zero corners) -- SURVEY.md 8e."""
    import torch
    from lidar_camera_calibration_amd import LidarCornersBatch, synth
    from lidar_camera_calibration_amd import _native as N
    from lidar_camera_calibration_amd.sharding import pack_records, record_floats
    board, lidar = synth.Board(), synth.vlp16()
    F = 12
    clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=321)
    clicks[3] = (50.0, 50.0, 50.0)          # nothing in the ROI
    clouds[7, :, 3] = 40.0                  # flat intensity -> degenerate histogram
    est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
    d_clouds = torch.from_numpy(clouds).cuda()
    d_clicks = torch.from_numpy(clicks).cuda()
    d_rec = torch.full((F, record_floats(board.n_corners)), -3.0, dtype=torch.float32, device="cuda")
    t = est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr())
    res = est.wait(t, d_rec.data_ptr(), board.n_corners, tag_base=4096)
    want = pack_records(res, F, board.n_corners, tag_base=4096)
    got = d_rec.cpu().numpy()
    assert res[3].status != 0 and res[7].status != 0 and res[0].status == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    from lidar_camera_calibration_amd.sharding import verify_records
    verify_records(got, 4096 + np.arange(F))
    est.close()


def test_compact_result_mode_ships_the_gather_records_only():
    """ILCC_RESULTS_COMPACT (ABI 4, SURVEY.md 8d: 12 * n_corners + 64 result bytes per frame): the batch's stream packs the
    K9 records and only those come back with the batch -- bit-identical to sharding.pack_records of the full records,
    which stay in HBM (ilcc_fetch_results) and equal a FULL-mode run of the same frames; both waits work in both modes."""
    import torch
    from lidar_camera_calibration_amd.sharding import pack_records, record_floats, verify_records
    board, lidar = synth.Board(), synth.vlp16()
    F = 12
    clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=321)
    clicks[3] = (50.0, 50.0, 50.0)          # nothing in the ROI
    clouds[7, :, 3] = 40.0                  # flat intensity -> degenerate histogram
    est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
    d_clouds, d_clicks = torch.from_numpy(clouds).cuda(), torch.from_numpy(clicks).cuda()
    torch.cuda.synchronize()
    full = est.wait(est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr()))
    want = pack_records(full, F, board.n_corners, tag_base=0)
    assert want[0, 19] == full[0].n_roi > 0                       # header slot 19 carries n_roi (the handle sizes K2 from it)
    late = est.wait_compact(est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr()))   # FULL mode: packed inside the wait
    assert np.array_equal(late.view(np.uint32), want.view(np.uint32))
    est.set_result_mode(N.RESULTS_COMPACT)
    t = est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr())
    with pytest.raises(Exception):
        est.set_result_mode(N.RESULTS_FULL)                       # not with a batch in flight
    rec = est.wait_compact(t)
    assert rec.shape == (F, record_floats(board.n_corners)) and rec.nbytes == F * 500
    assert np.array_equal(rec.view(np.uint32), want.view(np.uint32))
    verify_records(rec, np.arange(F))
    back = est.fetch_results(0, F)                                # the full records of that batch, read from HBM
    for a, b in zip(full, back):
        assert (a.status, a.n_roi, a.n_plane, a.grid_index, a.flags) == (b.status, b.n_roi, b.n_plane, b.grid_index, b.flags)
        assert tuple(a.theta_t) == tuple(b.theta_t) and np.array_equal(a.corners_array(), b.corners_array())
    assert len(est.fetch_cloud(0, N.CLOUD_CHESSBOARD)) == full[0].n_plane      # the fetch entries fill the records in themselves
    res = est.wait(est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr()))   # the full wait still works (copy inside)
    assert all(np.array_equal(a.corners_array(), b.corners_array()) and a.status == b.status for a, b in zip(full, res))
    est.close()


def test_near_tie_is_ordered_like_the_fp64_oracle(ob):
    """Seed 0xBEEF + 125 is a frame whose two best grid candidates (different translation basins, 137 mm apart in
    the corners) cost 0.121899381 and 0.121899392: fp32 sums cannot order them.  K6 lists the near ties of the
    bound, K7r re-orders them on exact fixed-point sums -- the argmin must be the oracle's (found by
    tools/grid_parity_sweep.py)."""
    from lidar_camera_calibration_amd import LidarCornersBatch, synth
    from lidar_camera_calibration_amd import _native as N
    clouds, clicks, _, _ = synth.make_batch(1, seed=0xBEEF + 125)
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    op.accum_float = 0
    ref = ob.extract(clouds[0], clicks[0], op)
    for prune in (1, 0):
        p = N.default_params()
        p.grid_prune = prune
        e = LidarCornersBatch(1, clouds.shape[1], p)
        r = e.extract(clouds, clicks)[0]
        e.close()
        assert r.status == ref.status and ref.status in (N.OK, N.AMBIGUOUS)
        assert r.grid_index == ref.grid_index and r.grid_index in (33157, 34756)   # the two tied basins
        assert r.grid_ties >= 2 and (r.flags & N.FLAG_TIE_OVERFLOW) == 0 and (r.flags & ~N.FLAGS_FP32_ONLY) == ref.flags
        assert tuple(r.theta_t) == tuple(ref.theta_t)
        assert np.abs(r.corners_array() - ob.result_corners(ref)).max() < 1e-6


def test_points_on_cell_borders_of_the_grid_argmin(ob):
    """Adversarial input for the fp32 grid pass.  K6 ranks the grid on fp32 sums; a labelled point
    within fp32 rounding of a cell border under a candidate may be put into the OTHER cell there, and its term then differs from
    the exact (fp64) one by a whole residual.  Here 48 points are placed ON cell borders (to the float32 of their coordinates:
    a few 1e-7 square) of the grid argmin and of its grid neighbours, on real frames' labelled points, and the pipeline's solver
    (ilcc_grid_solve) is run beside the oracle's exhaustive exact search:
      * ILCC_FLAG_BORDER_RISK fires (K7r recomputes K6's fp32 coordinates for the 27-neighbourhood of the argmin);
      * with it the refinement drops its first-round shortcut, so everything downstream of the grid argmin is EXACT: lattice
        point, phase, both fixed-point costs, rounds and hops == the oracle's pattern search from the same start;
      * the grid argmin itself either equals the oracle's exact one, or the flag is set (the documented limit: the argmin of an
        fp32 ranking, measure-zero on real data).  Without the gate the shortcut skipped a round the oracle moves in."""
    from lidar_camera_calibration_amd import LidarCornersBatch, synth
    from lidar_camera_calibration_amd import _native as N
    F = 6
    clouds, clicks, _, _ = synth.make_batch(F, seed=0xB0BDE4)
    p = N.default_params()
    est = LidarCornersBatch(F, clouds.shape[1], p)
    res = est.extract(clouds, clicks)
    sets = [est.fetch_labelled(f) for f in range(F) if res[f].status in (N.OK, N.AMBIGUOUS)]
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    g, W, H, div = p.grid_length, p.board_w, p.board_h, p.refine_div
    rng = np.random.default_rng(5)
    n_flag = n_same_argmin = 0
    for yz, lab in sets:
        flat0, _, _ = ob.grid_search(yz[:, 0], yz[:, 1], lab.astype(np.int8), op, 1)
        cell = flat0 >> 1
        k0, a0, b0 = cell // (p.n_ty * p.n_tz), (cell // p.n_tz) % p.n_ty, cell % p.n_tz
        extra = []
        for _ in range(48):
            k = int(np.clip(k0 + rng.integers(-1, 2), 0, p.n_th - 1))
            th = p.th_min + k * p.th_step
            if rng.random() < 0.5:      # on an i-border of candidate (k, a): ((cos y - sin z) + ty + W g / 2) / g = n
                a = int(np.clip(a0 + rng.integers(-1, 2), 0, p.n_ty - 1))
                ty, n = p.ty_min + a * p.ty_step, int(rng.integers(1, W))
                z = rng.uniform(-0.4, 0.4) * H * g
                y = ((n * g - W * g / 2.0 - ty) + np.sin(th) * z) / np.cos(th)
            else:                       # on a j-border of candidate (k, b): ((sin y + cos z) + tz + H g / 2) / g = n
                b = int(np.clip(b0 + rng.integers(-1, 2), 0, p.n_tz - 1))
                tz, n = p.tz_min + b * p.tz_step, int(rng.integers(1, H))
                y = rng.uniform(-0.4, 0.4) * W * g
                z = ((n * g - H * g / 2.0 - tz) - np.sin(th) * y) / np.cos(th)
            extra.append((y, z))
        yz2 = np.concatenate([yz, np.array(extra, dtype=np.float32)])
        lab2 = np.concatenate([lab, rng.integers(0, 2, len(extra)).astype(np.uint8)])
        got = est.grid_solve(yz2, lab2)
        flat, _, _ = ob.grid_search(yz2[:, 0], yz2[:, 1], lab2.astype(np.int8), op, 1)
        # the refinement is exact from the solver's OWN grid argmin, whatever the fp32 ranking did
        c2 = got["grid_index"] >> 1
        start = np.array([c2 // (p.n_ty * p.n_tz), (c2 // p.n_tz) % p.n_ty, c2 % p.n_tz], dtype=np.int32) * div
        q, ph, cq, aq, rounds, hops = ob.pattern_refine(yz2[:, 0], yz2[:, 1], lab2.astype(np.int8), op, start, got["grid_index"] & 1)
        assert (tuple(got["lat"]), got["phase"], got["cost_q"], got["alt_cost_q"], got["rounds"], got["hops"]) == \
               (tuple(q), ph, cq, aq, rounds, hops), (got, q, ph, cq, aq, rounds, hops)
        near = abs(c2 // (p.n_ty * p.n_tz) - k0) <= 1 and abs((c2 // p.n_tz) % p.n_ty - a0) <= 1 and abs(c2 % p.n_tz - b0) <= 1
        if near and c2 == cell:
            assert got["flags"] & N.FLAG_BORDER_RISK, got      # 48 points within ~3e-7 square of borders of this very neighbourhood
        assert got["grid_index"] == flat or (got["flags"] & N.FLAG_BORDER_RISK), (got, flat)
        n_flag += int(bool(got["flags"] & N.FLAG_BORDER_RISK))
        n_same_argmin += int(got["grid_index"] == flat)
    assert len(sets) >= 4 and n_flag >= len(sets) - 1
    print("border-adversarial frames: %d, flagged %d, grid argmin == exact argmin on %d" % (len(sets), n_flag, n_same_argmin))
    est.close()


def test_fused_locate_equals_the_three_launches():
    """Batches of >= 512 frames locate the grid minimum with ONE launch (k6_locate: seed, refinement and anchor in one workgroup
    per frame), smaller ones with three k6_grid_cost launches over (theta, frame) grids.  Same candidates on the same points in
    the same order: every field of the result must be identical, and so must the executed-evaluation-independent outputs
    (grid argmin, cost, margin, refinement trajectory, corners).  64 distinct frames, tiled 8 x into one 512-frame batch."""
    from lidar_camera_calibration_amd import LidarCornersBatch, synth
    from lidar_camera_calibration_amd import _native as N
    clouds, clicks, _, _ = synth.make_batch(64, seed=0xF05ED)
    small = LidarCornersBatch(64, clouds.shape[1], N.default_params())
    small.reserve(2048, 4096)
    r_small = small.extract(clouds, clicks)
    big = LidarCornersBatch(512, clouds.shape[1], N.default_params())
    big.reserve(2048, 4096)
    r_big = big.extract(np.tile(clouds, (8, 1, 1)), np.tile(clicks, (8, 1)))
    n_ok = 0
    for f in range(512):
        a, b = r_small[f % 64], r_big[f]
        assert (a.status, a.flags, a.n_roi, a.n_cluster, a.n_plane, a.grid_index) == (b.status, b.flags, b.n_roi, b.n_cluster, b.n_plane, b.grid_index), f
        if a.status in (N.OK, N.AMBIGUOUS):
            # (grid_ties is not compared: how many candidates were LISTED as near ties depends on how far the frame's bound had come
            # down when each completed -- a superset of the true near ties either way, re-ranked exactly by K7r)
            assert (a.grid_cost, a.basin_margin, a.sel_cost, a.iters_a, a.iters_b, a.cells_hit, a.n_oob) == \
                   (b.grid_cost, b.basin_margin, b.sel_cost, b.iters_a, b.iters_b, b.cells_hit, b.n_oob), f
            assert tuple(a.theta_t) == tuple(b.theta_t) and np.array_equal(a.corners_array(), b.corners_array()), f
            n_ok += 1
    assert n_ok >= 8 * 55
    small.close()
    big.close()


def test_sparse_wide_roi_takes_the_hashed_block_path(ob):
    """<= 4096 ROI points whose bounding grid is far larger than K2's cell bitmap (a 12 x 12 x 6 m ROI over a thinned cloud:
    ~2.7 M cells of 0.068 m against 96 k bits): the frame takes the cell-level components with the cells in a hash table of
    4 x 4 x 4-cell blocks in global memory (the frame's own workgroup: hashed_cluster_frame).
    Stage counts and the cluster cloud must equal the oracle's."""
    board = synth.Board()
    pose = synth.pose_from_fixture(1)
    cloud = synth.make_frame(synth.vlp16(), board, pose, 11)
    click = synth.make_click(pose, 11)
    # keep the neighbourhood of the board dense (so that the click still lands in a >= 100-point cluster) and every
    # 12th point elsewhere: ~3 k points spread over a 12 x 12 x 6 m box
    near = np.linalg.norm(cloud[:, :3] - click[None, :], axis=1) < 0.9
    keep = near | (np.arange(len(cloud)) % 12 == 0)
    thin = np.ascontiguousarray(cloud[keep])
    p = N.default_params()
    p.roi_half[0], p.roi_half[1], p.roi_half[2] = 6.0, 6.0, 3.0
    e = LidarCornersBatch(1, len(thin), p)
    r = e.extract(thin[None], click[None])[0]
    got = e.fetch_cloud(0, N.CLOUD_CLUSTER)
    e.close()
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    op.roi_half[0], op.roi_half[1], op.roi_half[2] = 6.0, 6.0, 3.0
    o, chess, pca = ob.extract(thin, click, op, want_clouds=True)
    assert 256 < r.n_roi <= 4096, r.n_roi
    ext = thin[:, :3].max(0) - thin[:, :3].min(0)
    assert np.prod(np.floor(np.minimum(ext, [12, 12, 6]) / (0.57 * 0.12)) + 5) > 96 * 1024      # the cell bitmap does not hold it
    assert (r.status, r.n_roi, r.n_cluster, r.n_plane) == (o.status, o.n_roi, o.n_cluster, o.n_plane)
    assert r.n_cluster >= 100 and len(got) == r.n_cluster
    if r.status in (N.OK, N.AMBIGUOUS):
        assert np.abs(r.corners_array() - ob.result_corners(o)).max() < 1e-6


def test_tiny_wide_frames_do_not_write_into_their_neighbours(ob):
    """the hashed-block path keeps its block arrays (>= 1024 words each) in the frame's own slices of the handle's
    buffers, 4 words per INPUT point -- a frame of fewer than 256 input points spread over a bounding grid larger than K2's
    cell bitmap used to write up to 1024 words past its slice, into the next frame's tables (past the allocation for the last
    frame).  Such frames take the point-level search now.  A ragged batch alternating tiny wide frames with full frames: every
    frame's stage counts and cluster cloud equal the oracle's, run twice (the corruption was a race)."""
    board = synth.Board()
    rng = np.random.default_rng(5)
    clouds, clicks = [], []
    for k in range(7):   # (tiny, full, tiny, full, tiny, full, tiny: the last frame's tables would end past the allocation)
        pose = synth.pose_from_fixture(k % 6)
        cloud = synth.make_frame(synth.vlp16(), board, pose, 31 + k)
        click = synth.make_click(pose, 31 + k)
        if k % 2 == 0:   # ~200 input points: ~150 on the board (a >= 100-point cluster), the rest spread over 12 x 12 x 6 m
            d = np.linalg.norm(cloud[:, :3] - click[None, :], axis=1)
            near = np.flatnonzero(d < 0.45)[:150]
            far = rng.choice(np.flatnonzero(d > 1.5), 60, replace=False)
            cloud = np.ascontiguousarray(cloud[np.sort(np.concatenate([near, far]))])
            assert len(cloud) < 256
        clouds.append(cloud)
        clicks.append(click)
    p = N.default_params()
    p.roi_half[0], p.roi_half[1], p.roi_half[2] = 6.0, 6.0, 3.0
    op = ob.default_params()
    op.solver = ob.SOLVER_GRID
    op.roi_half[0], op.roi_half[1], op.roi_half[2] = 6.0, 6.0, 3.0
    offsets = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.uint64)
    flat = np.ascontiguousarray(np.concatenate(clouds))
    e = LidarCornersBatch(len(clouds), max(len(c) for c in clouds), p)
    want = [ob.extract(clouds[f], clicks[f], op, want_clouds=True) for f in range(len(clouds))]
    for _ in range(2):
        res = e.extract(flat, np.stack(clicks), offsets)
        for f, r in enumerate(res):
            o = want[f][0]
            assert (r.status, r.n_roi, r.n_cluster, r.n_plane) == (o.status, o.n_roi, o.n_cluster, o.n_plane), f
            if r.n_cluster:
                roi_idx = ob.roi_crop(clouds[f], clicks[f], op)
                clu_idx, _ = ob.cluster(clouds[f][roi_idx], clicks[f], op)
                assert np.array_equal(e.fetch_cloud(f, N.CLOUD_CLUSTER), clouds[f][roi_idx][clu_idx]), f
    e.close()


def test_parameter_validation_and_second_handle():
    """params_ok rejects what the kernels cannot run (a (ty, tz) table pair beyond the LDS the grid kernel may
    ask for used to fail at launch and return stale data with ILCC_OK); a second handle in the same process works."""
    from lidar_camera_calibration_amd import IlccError
    e = LidarCornersBatch(2, 28800, N.default_params())
    for field, value in (("refine_div", 3), ("refine_div", 128), ("refine_max_rounds", -1), ("online_cluster_tol", 0.0),
                         ("n_ty", 5000), ("grid_prune", 2), ("board_w", 9), ("min_cell_coverage", 1.5),
                         ("ty_step", 10.0)):    # (ty_step = 10 m: one board square rounds to 0 refinement-lattice steps)
        p = N.default_params()
        setattr(p, field, value)
        with pytest.raises(IlccError):   # (n_ty = 5000: an axis is limited to 4096 candidates, so that n_ty + n_tz always fits the LDS
            # the grid kernel is allowed to ask for)
            e.set_params(p)
    # the failed ilcc_set_params calls left the handle on its previous parameter set, tables included
    # the K7r diagnostic entry refuses a theta lattice coordinate outside its cos/sin table
    yz = np.zeros((8, 2), np.float32)
    lab = np.zeros(8, np.uint8)
    lib = N.lib()
    for bad_th in (-10 ** 6, 10 ** 6):
        lat = (C.c_int32 * 3)(bad_th, 0, 0)
        ph = C.c_int32(0)
        st = lib.ilcc_pattern_refine(e._h, N.fptr(yz), lab.ctypes.data_as(C.POINTER(C.c_uint8)), 8, lat, C.byref(ph), None, None, None, None)
        assert st == N.BAD_ARGUMENT
    e2 = LidarCornersBatch(2, 28800, N.default_params())       # second handle, same device
    clouds, clicks, _, _ = synth.make_batch(2, fixture_poses=True)
    a = [r.corners_array() for r in e.extract(clouds, clicks)]
    b = [r.corners_array() for r in e2.extract(clouds, clicks)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    e.close()
    e2.close()
