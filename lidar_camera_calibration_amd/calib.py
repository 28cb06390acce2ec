"""Consumer closure: corner files -> LiDAR->camera extrinsic.

ctypes mirror of ``include/ilcc_calib.h`` (``libilcc_calib.so``), the host-only C++ restatement of
the offline half of the reference's ``calib_lidar_cam`` node
(/root/reference/ilcc2/test/calib_lidar_cam.cpp:72-165, src/Optimization.cpp:13-91,
src/ImageCornersEst.cpp:213-306,430-488).  No GPU work: a 210 x 6 dense solve.
"""
import ctypes as C
import os

import numpy as np

from . import _native  # loads libilcc_hip.so first (libilcc_calib links against it)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)


def lib():
    global _LIB
    if _LIB is None:
        _native.lib()
        path = os.path.join(_HERE, "libilcc_calib.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.ilcc_read_cam_corners.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, _dp]
        L.ilcc_read_cam_corners.restype = C.c_int32
        L.ilcc_lidar2cam_axis_roughly.argtypes = [C.c_char_p, _dp]
        L.ilcc_lidar2cam_axis_roughly.restype = C.c_int32
        L.ilcc_check_order_lidar.argtypes = [_dp, C.c_int32, C.c_int32]
        L.ilcc_check_order_lidar.restype = None
        L.ilcc_check_order_cam.argtypes = [_dp, C.c_int32, C.c_int32]
        L.ilcc_check_order_cam.restype = None
        L.ilcc_solve_pose_3d2d.argtypes = [_dp, _dp, C.c_int32, _dp, _dp, _dp, _dp]
        L.ilcc_solve_pose_3d2d.restype = C.c_int32
        L.ilcc_extrinsic_write.argtypes = [C.c_char_p, _dp]
        L.ilcc_extrinsic_write.restype = C.c_int32
        L.ilcc_extrinsic_read.argtypes = [C.c_char_p, _dp]
        L.ilcc_extrinsic_read.restype = C.c_int32
        L.ilcc_calib_lidar_cam.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]
        L.ilcc_calib_lidar_cam.restype = C.c_int32
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def read_cam_corners(filename, num, board=(7, 5)):
    out = np.zeros((num, 2))
    n = lib().ilcc_read_cam_corners(os.fsencode(filename), num, board[0], board[1], _p(out))
    if n < 0:
        raise FileNotFoundError(filename)
    return out[:n]


def lidar2cam_axis_roughly(camera_name):
    T = np.zeros((4, 4))
    lib().ilcc_lidar2cam_axis_roughly(camera_name.encode(), _p(T))
    return T


def check_order_lidar(xyz, board=(7, 5)):
    a = np.ascontiguousarray(xyz, dtype=np.float64).copy()
    lib().ilcc_check_order_lidar(_p(a), board[0], board[1])
    return a


def check_order_cam(xy, board=(7, 5)):
    a = np.ascontiguousarray(xy, dtype=np.float64).copy()
    lib().ilcc_check_order_cam(_p(a), board[0], board[1])
    return a


def solve_pose_3d2d(pts3d, pts2d, camera, r0=(0, 0, 0), t0=(0, 0, 0)):
    """camera = (fx, cx, fy, cy).  Returns (r, t, final_cost, iterations)."""
    p3 = np.ascontiguousarray(pts3d, dtype=np.float64)
    p2 = np.ascontiguousarray(pts2d, dtype=np.float64)
    cam = np.asarray(camera, dtype=np.float64)
    r = np.array(r0, dtype=np.float64)
    t = np.array(t0, dtype=np.float64)
    cost = C.c_double(0)
    it = lib().ilcc_solve_pose_3d2d(_p(p3), _p(p2), len(p3), _p(cam), _p(r), _p(t), C.byref(cost))
    if it < 0:
        raise ValueError("solve_pose_3d2d: bad arguments")
    return r, t, cost.value, it


def extrinsic_write(filename, T):
    a = np.ascontiguousarray(T, dtype=np.float64)
    if lib().ilcc_extrinsic_write(os.fsencode(filename), _p(a)) != 0:
        raise OSError(filename)


def extrinsic_read(filename):
    T = np.zeros((4, 4))
    if lib().ilcc_extrinsic_read(os.fsencode(filename), _p(T)) != 0:
        raise OSError(filename)
    return T


def calib_lidar_cam(process_data_dir, camera_name, bag_num, camera, board=(7, 5)):
    """The whole offline node.  Returns (T_lidar2cam 4x4, mean reprojection error in px)."""
    T = np.zeros((4, 4))
    cam = np.asarray(camera, dtype=np.float64)
    err = C.c_double(0)
    rc = lib().ilcc_calib_lidar_cam(os.fsencode(process_data_dir), camera_name.encode(), bag_num, board[0], board[1],
                                    _p(cam), _p(T), C.byref(err))
    if rc != 0:
        raise RuntimeError(f"calib_lidar_cam failed ({rc})")
    return T, err.value
