"""ctypes binding of oracle/libilcc_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package (lidar_camera_calibration_amd must fail loudly without its HIP library).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libilcc_oracle.so")
MAX_CORNERS = 256


class Params(C.Structure):
    _fields_ = [
        ("roi_half", C.c_double * 3),
        ("cluster_tol", C.c_double),
        ("cluster_min", C.c_int32),
        ("cluster_max", C.c_int32),
        ("ransac_thresh", C.c_double),
        ("ransac_hyp", C.c_int32),
        ("ransac_seed", C.c_uint32),
        ("hist_bins", C.c_int32),
        ("gray_rate", C.c_double),
        ("huber_delta", C.c_double),
        ("grid_length", C.c_double),
        ("board_w", C.c_int32),
        ("board_h", C.c_int32),
        ("solver", C.c_int32),
        ("accum_float", C.c_int32),
        ("phase_mode", C.c_int32),
        ("n_th", C.c_int32), ("n_ty", C.c_int32), ("n_tz", C.c_int32),
        ("th_min", C.c_double), ("th_step", C.c_double),
        ("ty_min", C.c_double), ("ty_step", C.c_double),
        ("tz_min", C.c_double), ("tz_step", C.c_double),
        ("refine_div", C.c_int32), ("refine_max_rounds", C.c_int32),
        ("refine_th_margin", C.c_int32), ("refine_pad_", C.c_int32),
        ("ambiguity_eps", C.c_double),
        ("min_cell_coverage", C.c_double),
        ("ransac_probability", C.c_double),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("n_roi", C.c_int32), ("n_cluster", C.c_int32), ("n_plane", C.c_int32),
        ("n_black", C.c_int32), ("n_gray", C.c_int32), ("n_white", C.c_int32),
        ("n_corners", C.c_int32),
        ("phase", C.c_int32),
        ("iters_a", C.c_int32), ("iters_b", C.c_int32),
        ("grid_index", C.c_int32),
        ("gray_zone", C.c_double * 2),
        ("theta_t", C.c_double * 3),
        ("cost_a", C.c_double), ("cost_b", C.c_double),
        ("sel_cost", C.c_double),
        ("grid_cost", C.c_double),
        ("basin_margin", C.c_double),
        ("flags", C.c_int32), ("cells_hit", C.c_int32), ("n_oob", C.c_int32), ("pad_", C.c_int32),
        ("pca", C.c_float * 16),
        ("corners", C.c_float * (MAX_CORNERS * 3)),
    ]


SOLVER_REFERENCE_LOCAL = 0
SOLVER_GRID = 1
OK, AMBIGUOUS = 0, 11
FLAG_REFINE_CAPPED, FLAG_LOW_COVERAGE = 2, 4
COST_Q_ONE = float(1 << 40)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "ilcc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        dp = C.POINTER(C.c_double)
        bp = C.POINTER(C.c_int8)
        pp = C.POINTER(Params)
        L.orc_default_params.argtypes = [pp]
        L.orc_roi_crop.argtypes = [fp, C.c_int32, fp, pp, ip]
        L.orc_roi_crop.restype = C.c_int32
        L.orc_cluster.argtypes = [fp, C.c_int32, fp, pp, ip, ip]
        L.orc_cluster.restype = C.c_int32
        L.orc_ransac_plane.argtypes = [fp, C.c_int32, pp, ip, fp]
        L.orc_ransac_plane.restype = C.c_int32
        L.orc_plane_frame.argtypes = [fp, C.c_int32, pp, fp, fp]
        L.orc_plane_frame.restype = C.c_int32
        L.orc_gray_zone.argtypes = [fp, C.c_int32, pp, dp, dp]
        L.orc_gray_zone.restype = C.c_int32
        L.orc_residual.argtypes = [dp, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                   C.c_int32, C.c_int32, C.c_int32, dp]
        L.orc_residual.restype = C.c_double
        L.orc_cost.argtypes = [dp, fp, fp, bp, C.c_int32, pp, C.c_int32, C.c_int32]
        L.orc_cost.restype = C.c_double
        L.orc_get_theta_t.argtypes = [fp, C.c_int32, dp, pp, C.c_int32, C.c_int32, dp, dp]
        L.orc_get_theta_t.restype = C.c_int32
        L.orc_grid_search.argtypes = [fp, fp, bp, C.c_int32, pp, C.c_int32, dp, dp]
        L.orc_grid_search.restype = C.c_int32
        L.orc_cost_q.argtypes = [dp, fp, fp, bp, C.c_int32, pp, C.c_int32, C.c_int32]
        L.orc_cost_q.restype = C.c_int64
        L.orc_pattern_refine.argtypes = [fp, fp, bp, C.c_int32, pp, ip, ip, C.POINTER(C.c_int64), ip, ip]
        L.orc_pattern_refine.restype = C.c_int64
        L.orc_lattice_point.argtypes = [pp, ip, dp]
        L.orc_lattice_point.restype = None
        L.orc_corners.argtypes = [fp, dp, pp, fp]
        L.orc_corners.restype = C.c_int32
        L.orc_extract.argtypes = [fp, C.c_int32, fp, pp, C.POINTER(Result), fp, fp]
        L.orc_extract.restype = C.c_int32
        L.orc_chessboard_by_point.argtypes = [fp, C.c_int32, fp, pp, C.c_int32, C.POINTER(Result), fp,
                                              C.POINTER(C.c_uint8)]
        L.orc_chessboard_by_point.restype = C.c_int32
        L.orc_format_float.argtypes = [C.c_float, C.c_char_p, C.c_int32]
        L.orc_format_float.restype = C.c_int32
        L.orc_solve_pose_3d2d.argtypes = [dp, dp, C.c_int32, dp, dp, dp, dp]
        L.orc_solve_pose_3d2d.restype = C.c_int32
        L.orc_hsv_to_rgb.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
        L.orc_hsv_to_rgb.restype = None
        L.orc_project_intensity.argtypes = [fp, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_project_intensity.restype = C.c_int32
        L.orc_colourise.argtypes = [fp, C.c_int32, C.c_void_p, C.c_double, C.POINTER(C.c_uint8), C.c_uint32, fp]
        L.orc_colourise.restype = C.c_int32
        _lib = L
    return _lib


def default_params() -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    return p


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def roi_crop(xyzi, click, p):
    x, xp = _f(xyzi)
    c, cp = _f(click)
    idx = np.empty(len(x), dtype=np.int32)
    m = lib().orc_roi_crop(xp, len(x), cp, C.byref(p), idx.ctypes.data_as(C.POINTER(C.c_int32)))
    return idx[:m].copy()


def cluster(roi, click, p):
    x, xp = _f(roi)
    c, cp = _f(click)
    idx = np.empty(max(len(x), 1), dtype=np.int32)
    lab = np.empty(max(len(x), 1), dtype=np.int32)
    m = lib().orc_cluster(xp, len(x), cp, C.byref(p), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                          lab.ctypes.data_as(C.POINTER(C.c_int32)))
    return idx[:m].copy(), lab[:len(x)].copy()


def ransac_plane(pts, p):
    x, xp = _f(pts)
    idx = np.empty(max(len(x), 1), dtype=np.int32)
    pl = np.zeros(4, dtype=np.float32)
    m = lib().orc_ransac_plane(xp, len(x), C.byref(p), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                               pl.ctypes.data_as(C.POINTER(C.c_float)))
    return idx[:m].copy(), pl


def plane_frame(pts, p):
    x, xp = _f(pts)
    pca = np.zeros(16, dtype=np.float32)
    out = np.zeros((len(x), 4), dtype=np.float32)
    st = lib().orc_plane_frame(xp, len(x), C.byref(p), pca.ctypes.data_as(C.POINTER(C.c_float)),
                               out.ctypes.data_as(C.POINTER(C.c_float)))
    return st, pca.reshape(4, 4), out


def gray_zone(intensity, p):
    x, xp = _f(intensity)
    rl = np.zeros(2)
    gz = np.zeros(2)
    st = lib().orc_gray_zone(xp, len(x), C.byref(p), rl.ctypes.data_as(C.POINTER(C.c_double)),
                             gz.ctypes.data_as(C.POINTER(C.c_double)))
    return st, rl, gz


def residual(theta_t, y, z, w, h, g, tlw, laser_white, use_oob, want_jac=False):
    t, tp = _d(theta_t)
    jac = np.zeros(3)
    r = lib().orc_residual(tp, float(y), float(z), int(w), int(h), float(g), int(tlw),
                           int(laser_white), int(use_oob),
                           jac.ctypes.data_as(C.POINTER(C.c_double)) if want_jac else None)
    return (r, jac) if want_jac else r


def cost(theta_t, y, z, label, p, tlw, use_oob):
    t, tp = _d(theta_t)
    yy, yp = _f(y)
    zz, zp = _f(z)
    lab = np.ascontiguousarray(label, dtype=np.int8)
    return lib().orc_cost(tp, yp, zp, lab.ctypes.data_as(C.POINTER(C.c_int8)), len(yy),
                          C.byref(p), int(tlw), int(use_oob))


def cost_q(theta_t, y, z, label, p, tlw, use_oob) -> int:
    """fixed-point cost of ORC_SOLVER_GRID (units of 2^-40)"""
    t, tp = _d(theta_t)
    yy, yp = _f(y)
    zz, zp = _f(z)
    lab = np.ascontiguousarray(label, dtype=np.int8)
    return int(lib().orc_cost_q(tp, yp, zp, lab.ctypes.data_as(C.POINTER(C.c_int8)), len(yy),
                                C.byref(p), int(tlw), int(use_oob)))


def lattice_point(p, lat):
    q = np.ascontiguousarray(lat, dtype=np.int32)
    out = np.zeros(3)
    lib().orc_lattice_point(C.byref(p), q.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def pattern_refine(y, z, label, p, lat, phase):
    """-> (lat[3], phase, cost_q, alt_cost_q, rounds, hops)"""
    yy, yp = _f(y)
    zz, zp = _f(z)
    lab = np.ascontiguousarray(label, dtype=np.int8)
    q = np.array(lat, dtype=np.int32)
    ph, rounds, hops = C.c_int32(int(phase)), C.c_int32(0), C.c_int32(0)
    alt = C.c_int64(0)
    c = lib().orc_pattern_refine(yp, zp, lab.ctypes.data_as(C.POINTER(C.c_int8)), len(yy), C.byref(p),
                                 q.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ph), C.byref(alt),
                                 C.byref(rounds), C.byref(hops))
    return q, ph.value, int(c), int(alt.value), rounds.value, hops.value


def get_theta_t(pts_pca, gz, p, tlw, use_oob, theta_t0=(0.0, 0.0, 0.0)):
    x, xp = _f(pts_pca)
    g, gp = _d(gz)
    t = np.array(theta_t0, dtype=np.float64)
    c = C.c_double(0)
    it = lib().orc_get_theta_t(xp, len(x), gp, C.byref(p), int(tlw), int(use_oob),
                               t.ctypes.data_as(C.POINTER(C.c_double)), C.byref(c))
    return t, c.value, it


def grid_search(y, z, label, p, use_oob, want_volume=False):
    yy, yp = _f(y)
    zz, zp = _f(z)
    lab = np.ascontiguousarray(label, dtype=np.int8)
    bc = C.c_double(0)
    vol = np.zeros(p.n_th * p.n_ty * p.n_tz * 2) if want_volume else None
    flat = lib().orc_grid_search(yp, zp, lab.ctypes.data_as(C.POINTER(C.c_int8)), len(yy),
                                 C.byref(p), int(use_oob), C.byref(bc),
                                 vol.ctypes.data_as(C.POINTER(C.c_double)) if want_volume else None)
    return flat, bc.value, vol


def corners(pca, theta_t, p):
    m, mp = _f(np.asarray(pca).reshape(-1))
    t, tp = _d(theta_t)
    out = np.zeros(MAX_CORNERS * 3, dtype=np.float32)
    n = lib().orc_corners(mp, tp, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:3 * n].reshape(n, 3).copy()


def extract(xyzi, click, p, want_clouds=False):
    x, xp = _f(xyzi)
    c, cp = _f(click)
    res = Result()
    cb = pc = None
    cbp = pcp = None
    if want_clouds:
        cb = np.zeros((len(x), 4), dtype=np.float32)
        pc = np.zeros((len(x), 4), dtype=np.float32)
        cbp = cb.ctypes.data_as(C.POINTER(C.c_float))
        pcp = pc.ctypes.data_as(C.POINTER(C.c_float))
    lib().orc_extract(xp, len(x), cp, C.byref(p), C.byref(res), cbp, pcp)
    if want_clouds:
        return res, cb[:res.n_plane].copy(), pc[:res.n_plane].copy()
    return res


def chessboard_by_point(xyzi, point, p, min_plane=500):
    """get_chessboard_by_point + colouring classes -> (Result, plane cloud, classes)"""
    x, xp = _f(xyzi)
    c, cp = _f(point)
    res = Result()
    cb = np.zeros((len(x), 4), dtype=np.float32)
    cl = np.zeros(len(x), dtype=np.uint8)
    lib().orc_chessboard_by_point(xp, len(x), cp, C.byref(p), int(min_plane), C.byref(res),
                                  cb.ctypes.data_as(C.POINTER(C.c_float)), cl.ctypes.data_as(C.POINTER(C.c_uint8)))
    return res, cb[:res.n_plane].copy(), cl[:res.n_plane].copy()


def result_corners(res: Result) -> np.ndarray:
    return np.ctypeslib.as_array(res.corners)[:3 * res.n_corners].reshape(-1, 3).copy()


def format_float(v: float) -> str:
    buf = C.create_string_buffer(64)
    lib().orc_format_float(C.c_float(v), buf, 64)
    return buf.value.decode()


def solve_pose_3d2d(pts3d, pts2d, camera, r0=(0.0, 0.0, 0.0), t0=(0.0, 0.0, 0.0)):
    """(fx, cx, fy, cy) camera; returns (r, t, final_cost, iterations)."""
    p3 = np.ascontiguousarray(pts3d, dtype=np.float64)
    p2 = np.ascontiguousarray(pts2d, dtype=np.float64)
    cam = np.asarray(camera, dtype=np.float64)
    r = np.array(r0, dtype=np.float64)
    t = np.array(t0, dtype=np.float64)
    fc = C.c_double(0)
    it = lib().orc_solve_pose_3d2d(_d(p3)[1], _d(p2)[1], len(p3), _d(cam)[1], _d(r)[1], _d(t)[1], C.byref(fc))
    return r, t, fc.value, it


HIT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("pad", "u1"), ("index", "<u4")])


def hsv_to_rgb(h, s=100, v=100):
    out = (C.c_uint8 * 3)()
    lib().orc_hsv_to_rgb(int(h), int(s), int(v), out)
    return tuple(out)


def project_intensity(xyzi, cam, dis=50.0, lo=0.0, hi=60.0):
    """cam: any ctypes struct laid out like orc_camera_model (the product's CameraModel is)."""
    a, ap = _f(xyzi)
    hits = np.zeros(len(a), HIT_DTYPE)
    m = lib().orc_project_intensity(ap, len(a), C.addressof(cam), dis, lo, hi, hits.ctypes.data)
    return hits[:m]


def colourise(xyzi, cam, image_bgr, dis=50.0):
    a, ap = _f(xyzi)
    img = np.ascontiguousarray(image_bgr, dtype=np.uint8)
    out = np.zeros((len(a), 4), np.float32)
    m = lib().orc_colourise(ap, len(a), C.addressof(cam), dis, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.strides[0],
                            out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:m]
