"""Minimal ROS bag format 2.0 WRITER for the tests (the reference's bags are stripped from the
repository): bag header, chunks (none / bz2 / lz4 via the system liblz4), connection records, message
data, index data, then the index section (connections + chunk infos).  Written from the published
format description; plays the role rosbag::Bag::write plays for the reference's recordings."""
import bz2
import ctypes as C
import struct

import numpy as np

POINTCLOUD2_MD5 = "1158d486dd51d683ce2f1be655c3c181"
FLOAT32, UINT16, UINT8 = 7, 4, 2


def _field(name: str, value: bytes) -> bytes:
    body = name.encode() + b"=" + value
    return struct.pack("<I", len(body)) + body


def _record(header_fields, data: bytes) -> bytes:
    h = b"".join(_field(k, v) for k, v in header_fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _time(sec, nsec):
    return struct.pack("<II", sec, nsec)


def lz4_frame(data: bytes) -> bytes:
    L = C.CDLL("liblz4.so.1")
    L.LZ4F_compressFrameBound.restype = C.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    L.LZ4F_compressFrame.restype = C.c_size_t
    L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    cap = L.LZ4F_compressFrameBound(len(data), None)
    out = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(out, cap, data, len(data), None)
    return out.raw[:n]


def pointcloud2(points: np.ndarray, fields, point_step, *, seq=0, stamp=(0, 0), frame_id="velodyne", height=1,
                row_pad=0, is_bigendian=0, is_dense=1) -> bytes:
    """Serialise sensor_msgs/PointCloud2.  `points`: structured array or raw (n, point_step) uint8;
    `fields`: [(name, offset, datatype, count)]."""
    raw = np.ascontiguousarray(points).view(np.uint8).reshape(-1, point_step)
    n = raw.shape[0]
    assert n % height == 0
    width = n // height
    row_step = width * point_step + row_pad
    rows = raw.reshape(height, width * point_step)
    if row_pad:
        rows = np.concatenate([rows, np.full((height, row_pad), 0xAB, np.uint8)], 1)
    data = rows.tobytes()
    out = struct.pack("<III", seq, stamp[0], stamp[1])
    out += struct.pack("<I", len(frame_id)) + frame_id.encode()
    out += struct.pack("<II", height, width)
    out += struct.pack("<I", len(fields))
    for name, offset, datatype, count in fields:
        out += struct.pack("<I", len(name)) + name.encode() + struct.pack("<IBI", offset, datatype, count)
    out += struct.pack("<B", is_bigendian) + struct.pack("<II", point_step, row_step)
    out += struct.pack("<I", len(data)) + data + struct.pack("<B", is_dense)
    return out


class BagWriter:
    """messages: list of (topic, type, md5, (sec, nsec), payload); chunks: list of message-index lists."""

    def __init__(self, path, compression="none"):
        self.path, self.compression = path, compression
        self.conns = {}       # (topic, type, md5) -> id
        self.chunks = []      # list of [(conn, time, payload)]

    def conn(self, topic, typ, md5):
        key = (topic, typ, md5)
        if key not in self.conns:
            self.conns[key] = len(self.conns)
        return self.conns[key]

    def add_chunk(self, msgs):
        self.chunks.append([(self.conn(t, ty, md5), tm, payload) for (t, ty, md5, tm, payload) in msgs])

    def _conn_record(self, key, cid):
        topic, typ, md5 = key
        data = _field("topic", topic.encode()) + _field("type", typ.encode()) + _field("md5sum", md5.encode()) + \
            _field("message_definition", b"# stripped\n")
        return _record([("op", b"\x07"), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], data)

    def write(self, indexed=True):
        by_id = {v: k for k, v in self.conns.items()}
        body, infos = b"", []
        pos = 13 + 4096
        for msgs in self.chunks:
            inner, seen, index = b"", set(), {}
            for cid, tm, payload in msgs:
                if cid not in seen:
                    seen.add(cid)
                    inner += self._conn_record(by_id[cid], cid)
                index.setdefault(cid, []).append((tm, len(inner)))
                inner += _record([("op", b"\x02"), ("conn", struct.pack("<I", cid)), ("time", _time(*tm))], payload)
            comp = {"none": lambda b: b, "bz2": bz2.compress, "lz4": lz4_frame}[self.compression](inner)
            rec = _record([("op", b"\x05"), ("compression", self.compression.encode()),
                           ("size", struct.pack("<I", len(inner)))], comp)
            for cid, ents in index.items():
                d = b"".join(_time(*tm) + struct.pack("<I", off) for tm, off in ents)
                rec += _record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", cid)),
                                ("count", struct.pack("<I", len(ents)))], d)
            times = [tm for _, tm, _ in msgs]
            infos.append((pos, min(times), max(times), {cid: len(e) for cid, e in index.items()}))
            body += rec
            pos += len(rec)
        index_pos = pos
        tail = b"".join(self._conn_record(by_id[cid], cid) for cid in sorted(by_id))
        for cpos, t0, t1, counts in infos:
            d = b"".join(struct.pack("<II", cid, n) for cid, n in counts.items())
            tail += _record([("op", b"\x06"), ("ver", struct.pack("<I", 1)), ("chunk_pos", struct.pack("<Q", cpos)),
                             ("start_time", _time(*t0)), ("end_time", _time(*t1)),
                             ("count", struct.pack("<I", len(counts)))], d)
        hf = [("op", b"\x03"), ("index_pos", struct.pack("<Q", index_pos if indexed else 0)),
              ("conn_count", struct.pack("<I", len(by_id))), ("chunk_count", struct.pack("<I", len(infos)))]
        h = b"".join(_field(k, v) for k, v in hf)
        pad = 4096 - 8 - len(h)
        header = struct.pack("<I", len(h)) + h + struct.pack("<I", pad) + b" " * pad
        assert len(header) == 4096
        with open(self.path, "wb") as f:
            f.write(b"#ROSBAG V2.0\n" + header + body + tail)


def velodyne_points(xyzi: np.ndarray) -> tuple:
    """The velodyne_pointcloud driver's record: x y z (0,4,8) pad intensity (16) ring u16 (20), point_step 32."""
    dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"],
                   "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
    a = np.zeros(len(xyzi), dt)
    a["x"], a["y"], a["z"], a["intensity"] = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2], xyzi[:, 3]
    a["ring"] = np.arange(len(xyzi)) % 16
    fields = [("x", 0, FLOAT32, 1), ("y", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1), ("intensity", 16, FLOAT32, 1),
              ("ring", 20, UINT16, 1)]
    return a, fields, 32
