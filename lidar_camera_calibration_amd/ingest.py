"""rosbag / PointCloud2 ingestion without ROS: ctypes mirror of
``include/ilcc_ingest.h``.  What /root/reference/ilcc2/test/get_lidar_corners.cpp:136-164 does with
rosbag::View + pcl::fromROSMsg; the field gather (K0) runs on the GPU."""
import ctypes as C
import os

import numpy as np

from . import _native

FIELD_ABSENT = 0xFFFFFFFF
POINTCLOUD2_MD5 = "1158d486dd51d683ce2f1be655c3c181"

INGEST_EXPORTS = ["ilcc_bag_first_message", "ilcc_pointcloud2_parse", "ilcc_pointcloud2_unpack_device",
                  "ilcc_bag_first_cloud"]


class Layout(C.Structure):
    _fields_ = [
        ("height", C.c_uint32), ("width", C.c_uint32), ("point_step", C.c_uint32), ("row_step", C.c_uint32),
        ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32), ("off_intensity", C.c_uint32),
        ("is_bigendian", C.c_uint32), ("is_dense", C.c_uint32),
        ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("seq", C.c_uint32), ("n_fields", C.c_uint32),
        ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64),
        ("frame_id", C.c_char * 64),
    ]

    @property
    def n_points(self):
        return self.height * self.width


_ready = False


def _lib():
    global _ready
    L = _native.lib()
    if not _ready:
        u8p = C.POINTER(C.c_uint8)
        L.ilcc_bag_first_message.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.ilcc_bag_first_message.restype = C.c_int32
        L.ilcc_pointcloud2_parse.argtypes = [u8p, C.c_uint64, C.POINTER(Layout)]
        L.ilcc_pointcloud2_parse.restype = C.c_int32
        L.ilcc_pointcloud2_unpack_device.argtypes = [C.c_void_p, C.POINTER(Layout), C.c_void_p, C.c_void_p]
        L.ilcc_pointcloud2_unpack_device.restype = C.c_int32
        L.ilcc_bag_first_cloud.argtypes = [C.c_int32, C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.c_uint32,
                                           C.POINTER(C.c_uint32)]
        L.ilcc_bag_first_cloud.restype = C.c_int32
        _ready = True
    return L


class IngestError(RuntimeError):
    def __init__(self, status):
        self.status = status
        super().__init__("%s: %s" % (_native.strerror(status), _native.lib().ilcc_last_error(None).decode()))


def bag_first_message(bag_path, topic, md5sum=None) -> bytes:
    """First message (bag time order) on `topic` that would instantiate as the given type."""
    L = _lib()
    n = C.c_uint64(0)
    md5 = md5sum.encode() if md5sum else None
    st = L.ilcc_bag_first_message(os.fsencode(bag_path), topic.encode(), md5, None, 0, C.byref(n))
    if st not in (_native.OK, _native.CAPACITY):
        raise IngestError(st)
    buf = (C.c_uint8 * max(1, n.value))()
    st = L.ilcc_bag_first_message(os.fsencode(bag_path), topic.encode(), md5, buf, n.value, C.byref(n))
    if st != _native.OK:
        raise IngestError(st)
    return bytes(buf[:n.value])


def parse_pointcloud2(msg: bytes) -> Layout:
    lay = Layout()
    arr = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg if msg else b"\0")
    st = _lib().ilcc_pointcloud2_parse(arr, len(msg), C.byref(lay))
    if st != _native.OK:
        raise IngestError(st)
    return lay


def unpack_device(d_data_ptr: int, layout: Layout, d_xyzi_ptr: int, stream: int = 0):
    """K0 on the given hipStream_t (0 = default stream); asynchronous."""
    st = _lib().ilcc_pointcloud2_unpack_device(C.c_void_p(d_data_ptr), C.byref(layout), C.c_void_p(d_xyzi_ptr),
                                               C.c_void_p(stream))
    if st != _native.OK:
        raise IngestError(st)


def bag_first_cloud(bag_path, topic="/velodyne_points", device=0) -> np.ndarray:
    """(n, 4) float32 XYZI of the first PointCloud2 on `topic` -- get_lidar_corners.cpp:136-164."""
    L = _lib()
    n = C.c_uint32(0)
    st = L.ilcc_bag_first_cloud(device, os.fsencode(bag_path), topic.encode(), None, 0, C.byref(n))
    if st not in (_native.OK, _native.CAPACITY):
        raise IngestError(st)
    out = np.zeros((n.value, 4), dtype=np.float32)
    if n.value:
        st = L.ilcc_bag_first_cloud(device, os.fsencode(bag_path), topic.encode(),
                                    out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n))
        if st != _native.OK:
            raise IngestError(st)
    return out
