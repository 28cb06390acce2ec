"""LiDAR -> image projection after calibration: ctypes mirror of
``include/ilcc_project.h`` (K8).  pcd2image's per-point colour and rgblidar's XYZRGB cloud
(/root/reference/ilcc2/test/pcd2image.cpp:40-82, test/rgblidar.cpp:45-78)."""
import ctypes as C

import numpy as np

from . import _native

PROJECT_EXPORTS = ["ilcc_project_intensity_device", "ilcc_colourise_device"]

HIT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("pad", "u1"), ("index", "<u4")])


class CameraModel(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3),
                ("fx", C.c_double), ("cx", C.c_double), ("fy", C.c_double), ("cy", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32)]

    @classmethod
    def from_extrinsic(cls, T_lidar2cam, camera, image_size):
        """T (4x4), camera = (fx, cx, fy, cy), image_size = (width, height) -- ImageCornersEst::setRt + camK."""
        T = np.asarray(T_lidar2cam, dtype=np.float64)
        m = cls()
        m.R[:] = T[:3, :3].reshape(-1)
        m.t[:] = T[:3, 3]
        m.fx, m.cx, m.fy, m.cy = [float(v) for v in camera]
        m.width, m.height = int(image_size[0]), int(image_size[1])
        return m


_ready = False


def _lib():
    global _ready
    L = _native.lib()
    if not _ready:
        cp = C.POINTER(CameraModel)
        L.ilcc_project_intensity_device.argtypes = [C.c_void_p, C.c_uint32, cp, C.c_double, C.c_double, C.c_double,
                                                    C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        L.ilcc_project_intensity_device.restype = C.c_int32
        L.ilcc_colourise_device.argtypes = [C.c_void_p, C.c_uint32, cp, C.c_double, C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.POINTER(C.c_uint32), C.c_void_p]
        L.ilcc_colourise_device.restype = C.c_int32
        _ready = True
    return L


def _check(st):
    if st != _native.OK:
        raise RuntimeError("%s: %s" % (_native.strerror(st), _native.lib().ilcc_last_error(None).decode()))


def project_intensity_device(d_xyzi_ptr, n_points, cam, d_hits_ptr, distance_valid=50.0, inten_low=0.0, inten_high=60.0,
                             stream=0) -> int:
    """Returns the number of hit records written at d_hits_ptr (16 B each, HIT_DTYPE)."""
    n = C.c_uint32(0)
    _check(_lib().ilcc_project_intensity_device(C.c_void_p(d_xyzi_ptr), n_points, C.byref(cam), distance_valid, inten_low,
                                                inten_high, C.c_void_p(d_hits_ptr), C.byref(n), C.c_void_p(stream)))
    return n.value


def colourise_device(d_xyzi_ptr, n_points, cam, d_image_ptr, image_step, d_out_ptr, distance_valid=50.0, stream=0) -> int:
    """Returns the number of XYZRGB records (16 B each) written at d_out_ptr."""
    n = C.c_uint32(0)
    _check(_lib().ilcc_colourise_device(C.c_void_p(d_xyzi_ptr), n_points, C.byref(cam), distance_valid,
                                        C.c_void_p(d_image_ptr), image_step, C.c_void_p(d_out_ptr), C.byref(n),
                                        C.c_void_p(stream)))
    return n.value
