"""Host-side mirror of the reference's ``LidarCornersEst`` for the corner-extraction path.

Same method names, argument meaning and error behaviour as
``/root/reference/ilcc2/include/ilcc2/LidarCornersEst.h:12-84`` as far as the node
``ilcc2/test/get_lidar_corners.cpp:112-201`` uses them, with the two PCL viewers / key presses
replaced by automatic acceptance.  All arithmetic runs in libilcc_hip.so on the MI355X; this file
is plumbing (numpy in, numpy out) over the C-ABI and never falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _native as N


class IlccError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = int(status)
        msg = N.strerror(status)
        super().__init__(f"{msg}: {detail}" if detail else msg)


class LidarCornersEst:
    """``LidarCornersEst`` over the HIP C-ABI.

    The per-frame members of the reference (``m_cloud_ROI``, ``m_cloud_chessboard``,
    ``m_cloud_PCA``, ``m_cloud_optim``, ``m_cloud_corners``) are exposed as numpy arrays
    (n x 4 float32: x, y, z, intensity) after ``get_corners``.
    """

    def __init__(self, device: int = -1, max_frames: int = 1, max_points_per_frame: int = 150_000,
                 params: Optional[N.Params] = None):
        self._lib = N.lib()
        self.params = params if params is not None else N.default_params()
        self._device = device
        self._max_frames = int(max_frames)
        self._max_points = int(max_frames) * int(max_points_per_frame)
        self._h = None
        self._cloud = None
        self._click = None
        self._result: Optional[N.Result] = None
        # The automatic stand-ins for the operator who would press 'r' at the viewer, one switch per signal:
        # ILCC_AMBIGUOUS scans (a basin one square away costs about the same) -> get_corners() returns False unless
        # accept_ambiguous; ILCC_OK scans flagged ILCC_FLAG_LOW_COVERAGE (fewer than params.min_cell_coverage of the squares
        # hold a labelled point: a far board on a 16-ring sensor) -> False unless accept_low_coverage.  The flag stays in
        # `result.flags` either way.
        self.accept_ambiguous = False
        self.accept_low_coverage = False
        self.m_click_point = None
        self.m_cloud_ROI = self.m_cloud_chessboard = self.m_cloud_PCA = None
        self.m_cloud_optim = self.m_cloud_corners = None

    # -- lifetime ---------------------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            h = self._lib.ilcc_create(self._device, C.byref(self.params), self._max_frames, self._max_points)
            if not h:
                raise IlccError(N.HIP_ERROR, self._lib.ilcc_last_error(None).decode())
            self._h = h
        return self._h

    def close(self):
        if self._h is not None:
            self._lib.ilcc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != N.OK:
            raise IlccError(st, self._lib.ilcc_last_error(self._h).decode() if self._h else "")

    def set_params(self, params: N.Params):
        self.params = params
        if self._h is not None:
            self._check(self._lib.ilcc_set_params(self._h, C.byref(self.params)))

    # -- the reference's surface ------------------------------------------------------------
    def register_viewer(self):
        """LidarCornersEst.h:26-37 opens two PCLVisualizer windows; there is no display on a GPU
        node, candidates are accepted automatically (keys 'o' and 'k')."""

    def set_chessboard_param(self, cam_yaml: str) -> bool:
        """LidarCornersEst.cpp:20-46: False (and a message) when the yaml cannot be opened."""
        st = self._lib.ilcc_set_chessboard_param(C.byref(self.params), cam_yaml.encode())
        if st != N.OK:
            print("can not open " + cam_yaml)
            return False
        if self._h is not None:
            self._check(self._lib.ilcc_set_params(self._h, C.byref(self.params)))
        print("grid_length:", self.params.grid_length)
        print("grid_in_x:", self.params.board_w)
        print("grid_in_y:", self.params.board_h)
        return True

    def setROI(self, cloud: np.ndarray, point: Sequence[float]):
        """LidarCornersEst.cpp:48-70.  ``cloud``: n x 4 float32 XYZI, ``point``: the rviz click."""
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        if cloud.ndim != 2 or cloud.shape[1] != 4:
            raise ValueError("cloud must be n x 4 (x, y, z, intensity)")
        self._cloud = cloud
        self._click = np.ascontiguousarray(point, dtype=np.float32).reshape(3)
        self.m_click_point = self._click.copy()
        self._result = None

    def _run(self):
        if self._result is None:
            if self._cloud is None:
                raise RuntimeError("setROI must be called first")
            res = N.Result()
            st = self._lib.ilcc_extract(self._handle(), N.fptr(self._cloud), len(self._cloud),
                                        N.fptr(self._click), C.byref(res))
            self._check(st)
            self._result = res
        return self._result

    def _fetch(self, which: int) -> np.ndarray:
        n = self._lib.ilcc_fetch_cloud(self._h, 0, which, None, 0)
        if n < 0:
            raise IlccError(-n)
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        self._lib.ilcc_fetch_cloud(self._h, 0, which, N.fptr(out), n)
        return out[:n]

    def get_chessboard_by_point(self, incloud: np.ndarray, point: Sequence[float], min_plane_points: int = 500):
        """LidarCornersEst.cpp:72-115 as the online node calls it (lidar_chessboard_online.cpp:91):
        whole-cloud clustering with ``params.cluster_tol`` (the reference hard-codes 0.10 there),
        the cluster around ``point``, its plane.  Returns ``(ok, outcloud)`` like the reference's
        ``bool`` + ``outcloud`` reference argument."""
        cloud = np.ascontiguousarray(incloud, dtype=np.float32)
        pt = np.ascontiguousarray(point, dtype=np.float32).reshape(3)
        off = np.array([0, len(cloud)], dtype=np.uint64)
        res = (N.Result * 1)()
        st = self._lib.ilcc_chessboard_by_point_batch(self._handle(), N.fptr(cloud),
                                                      off.ctypes.data_as(C.POINTER(C.c_uint64)), 1, N.fptr(pt),
                                                      int(min_plane_points), res)
        self._check(st)
        self._front = res[0]      # gray zone / classes of THIS call; never taken for a get_corners() result
        self._result = None       # a later EuclideanCluster()/get_corners() re-runs the whole path on setROI's cloud
        out = self._fetch(N.CLOUD_CHESSBOARD) if res[0].status in (N.OK, N.BOARD_NOT_FOUND) else np.zeros((0, 4), np.float32)
        return res[0].status == N.OK, out

    def get_gray_zone(self, cloud=None, rate: Optional[float] = None) -> np.ndarray:
        """LidarCornersEst.cpp:303-328 on the plane cloud of the last call (computed on the device
        with ``params.gray_rate``)."""
        src = self._result if self._result is not None else getattr(self, "_front", None)
        if src is None:
            raise RuntimeError("no plane cloud yet: call get_chessboard_by_point or PCA first")
        return np.array(src.gray_zone)

    def color_by_gray_zone(self) -> np.ndarray:
        """LidarCornersEst.cpp:452-499: n x 3 uint8 RGB of the last plane cloud (black 10,10,10 /
        gray 255,0,0 / white 255,255,255), from the device-side classification."""
        n = self._lib.ilcc_fetch_classes(self._h, 0, None, 0)
        if n < 0:
            raise IlccError(-n)
        cl = np.zeros(max(n, 1), dtype=np.uint8)
        self._lib.ilcc_fetch_classes(self._h, 0, cl.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        lut = np.array([[10, 10, 10], [255, 0, 0], [255, 255, 255]], dtype=np.uint8)
        return lut[cl[:n]]

    def EuclideanCluster(self) -> bool:
        """LidarCornersEst.cpp:124-186.  False when no board plane is found (the reference returns
        False on key 'r' and throws when there is no cluster at all)."""
        res = self._run()
        self.m_cloud_ROI = self._fetch(N.CLOUD_ROI)
        if res.status in (N.NO_ROI_POINTS, N.NO_CLUSTER, N.NO_PLANE):
            return False
        self.m_cloud_chessboard = self._fetch(N.CLOUD_CHESSBOARD)
        print("chessboard plane size:", len(self.m_cloud_chessboard))
        return True

    def PCA(self):
        """LidarCornersEst.cpp:366-372 (transformbyPCA + get_gray_zone(…, 2.5))."""
        res = self._run()
        self.m_cloud_PCA = self._fetch(N.CLOUD_PCA)
        self.pca_matrix = np.array(res.pca, dtype=np.float32).reshape(4, 4)
        self.m_gray_zone = np.array(res.gray_zone)

    def get_corners(self, corners: list) -> bool:
        """LidarCornersEst.cpp:374-450: fills ``corners`` with (x, y, z) triples, False if rejected."""
        res = self._run()
        if res.status == N.AMBIGUOUS and not self.accept_ambiguous:
            print("reject this scan (ambiguous: basin margin %.3g)" % res.basin_margin)
            return False
        if res.status not in (N.OK, N.AMBIGUOUS):
            print("reject this scan")
            return False
        if (res.flags & N.FLAG_LOW_COVERAGE) and not self.accept_low_coverage:
            print("reject this scan (pattern under-sampled: %d of %d squares hold points)"
                  % (res.cells_hit, self.params.board_w * self.params.board_h))
            return False
        c = res.corners_array()
        self.m_cloud_optim = self._fetch(N.CLOUD_OPTIM)
        self.m_cloud_corners = np.concatenate([c, np.full((len(c), 1), 50.0, np.float32)], 1)  # :531
        corners.extend(map(tuple, c.astype(np.float64)))
        return True

    @property
    def result(self) -> Optional[N.Result]:
        return self._result


class LidarCornersBatch:
    """Batched entry: many independent frames per call (inputs on host or already in HBM)."""

    def __init__(self, max_frames: int, max_points_per_frame: int, params: Optional[N.Params] = None,
                 device: int = -1):
        self._lib = N.lib()
        self.params = params if params is not None else N.default_params()
        self._h = self._lib.ilcc_create(device, C.byref(self.params), int(max_frames),
                                        int(max_frames) * int(max_points_per_frame))
        if not self._h:
            raise IlccError(N.HIP_ERROR, self._lib.ilcc_last_error(None).decode())

    def close(self):
        if self._h:
            self._lib.ilcc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return self._lib.ilcc_last_error(self._h).decode()

    def set_params(self, params: N.Params):
        self.params = params
        st = self._lib.ilcc_set_params(self._h, C.byref(params))
        if st != N.OK:
            raise IlccError(st, self._err())

    def set_result_mode(self, mode: int):
        """ILCC_RESULTS_FULL (default) / ILCC_RESULTS_COMPACT: which records ride back with a submitted batch."""
        st = self._lib.ilcc_set_result_mode(self._h, int(mode))
        if st != N.OK:
            raise IlccError(st, self._err())

    def wait_compact(self, ticket) -> np.ndarray:
        """The batch's compact records: [n_frames, RECORD_HEADER + 3 * board corners] float32 (layout:
        ``sharding.pack_records``; tag = frame index within the batch)."""
        t, n_frames = ticket
        # the record width was fixed when the batch was submitted: ask the library, not self.params (a mutable struct)
        width = int(self._lib.ilcc_record_floats(self._h, t))
        if width == 0:
            raise IlccError(N.BAD_ARGUMENT, "wait_compact: no batch in flight under this ticket")
        rec = np.empty((n_frames, width), dtype=np.float32)
        st = self._lib.ilcc_wait_compact(self._h, t, N.fptr(rec), rec.size)
        if st != N.OK:
            raise IlccError(st, self._err())
        return rec

    def fetch_results(self, first: int, n: int):
        """Full records [first, first + n) of the last completed batch (read from HBM)."""
        res = (N.Result * n)()
        st = self._lib.ilcc_fetch_results(self._h, int(first), int(n), res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def reserve(self, labelled_points_per_frame: int, roi_points_per_frame: int):
        """ilcc_reserve: size the on-chip staging (grid search: labelled points; clustering: ROI points) up front, so the
        first batch takes the same kernels as a warmed handle.  Results never depend on it."""
        st = self._lib.ilcc_reserve(self._h, int(labelled_points_per_frame), int(roi_points_per_frame))
        if st != N.OK:
            raise IlccError(st, self._err())

    def extract(self, clouds: np.ndarray, clicks: np.ndarray, offsets: Optional[np.ndarray] = None):
        """clouds: [F,N,4] (or [total,4] with ``offsets`` [F+1]); clicks: [F,3].  Returns Result array."""
        clouds = np.ascontiguousarray(clouds, dtype=np.float32)
        clicks = np.ascontiguousarray(clicks, dtype=np.float32).reshape(-1, 3)
        f = len(clicks)
        if offsets is None:
            n = clouds.shape[-2]
            offsets = np.arange(f + 1, dtype=np.uint64) * np.uint64(n)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = (N.Result * f)()               # fresh records per call (the caller owns them)
        st = self._lib.ilcc_extract_batch(self._h, N.fptr(clouds), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          f, N.fptr(clicks), res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def extract_device(self, d_xyzi_ptr: int, n_frames: int, n_points: int, d_clicks_ptr: int):
        """Inputs resident in HBM: raw device pointers (e.g. ``tensor.data_ptr()``), fixed N per frame.
        The producer's stream must be synchronised before the call."""
        offsets = np.arange(n_frames + 1, dtype=np.uint64) * np.uint64(n_points)
        res = (N.Result * n_frames)()
        st = self._lib.ilcc_extract_batch_device(self._h, C.c_void_p(d_xyzi_ptr),
                                                 offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n_frames,
                                                 C.c_void_p(d_clicks_ptr), res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def submit_device(self, d_xyzi_ptr: int, n_frames: int, n_points: int, d_clicks_ptr: int):
        """Asynchronous ``extract_device``: returns a ticket; up to 4 batches in flight per handle."""
        offsets = np.arange(n_frames + 1, dtype=np.uint64) * np.uint64(n_points)
        ticket = C.c_int32(-1)
        st = self._lib.ilcc_submit_batch_device(self._h, C.c_void_p(d_xyzi_ptr),
                                                offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n_frames,
                                                C.c_void_p(d_clicks_ptr), C.byref(ticket))
        if st != N.OK:
            raise IlccError(st, self._err())
        return ticket.value, n_frames

    def submit_host(self, xyzi_ptr: int, n_frames: int, n_points: int, clicks_ptr: int):
        """Asynchronous extraction from HOST buffers (raw addresses, e.g. of pinned torch tensors / numpy arrays that
        stay alive until ``wait``): the H2D copy rides on the batch's own stream and overlaps with the other batches
        in flight."""
        offsets = np.arange(n_frames + 1, dtype=np.uint64) * np.uint64(n_points)
        ticket = C.c_int32(-1)
        st = self._lib.ilcc_submit_batch(self._h, C.c_void_p(xyzi_ptr), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         n_frames, C.c_void_p(clicks_ptr), C.byref(ticket))
        if st != N.OK:
            raise IlccError(st, self._err())
        return ticket.value, n_frames

    def wait(self, ticket, d_records_ptr: Optional[int] = None, n_corners: int = 0, tag_base: int = 0,
             want_results: bool = True):
        """Results of a submitted batch.  With ``d_records_ptr`` (device memory, n_frames x (RECORD_HEADER +
        3*n_corners) float32) the fixed-size gather records are also packed on the GPU
        (``ilcc_wait_records_device``); record f carries the tag ``tag_base + f``.  ``want_results=False`` (only with
        ``d_records_ptr``): nothing but the device-side records is produced."""
        t, n_frames = ticket
        res = (N.Result * n_frames)() if (want_results or d_records_ptr is None) else None
        if d_records_ptr is not None:
            st = self._lib.ilcc_wait_records_device(self._h, t, res, C.c_void_p(d_records_ptr), n_corners,
                                                    int(tag_base) & 0xFFFFFFFF)
        else:
            st = self._lib.ilcc_wait(self._h, t, res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def fetch_cloud(self, frame: int, which: int) -> np.ndarray:
        n = self._lib.ilcc_fetch_cloud(self._h, frame, which, None, 0)
        if n < 0:
            raise IlccError(-n)
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        self._lib.ilcc_fetch_cloud(self._h, frame, which, N.fptr(out), n)
        return out[:n]

    def chessboard_by_point(self, clouds: np.ndarray, points: np.ndarray, min_plane_points: int = 500,
                            offsets: Optional[np.ndarray] = None):
        """Batched ``get_chessboard_by_point`` (host buffers)."""
        clouds = np.ascontiguousarray(clouds, dtype=np.float32)
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        f = len(points)
        if offsets is None:
            offsets = np.arange(f + 1, dtype=np.uint64) * np.uint64(clouds.shape[-2])
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = (N.Result * f)()
        st = self._lib.ilcc_chessboard_by_point_batch(self._h, N.fptr(clouds),
                                                      offsets.ctypes.data_as(C.POINTER(C.c_uint64)), f,
                                                      N.fptr(points), int(min_plane_points), res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def submit_chessboard_by_point(self, xyzi_ptr: int, n_frames: int, n_points: int, points_ptr: int):
        """Asynchronous ``chessboard_by_point`` from HOST buffers (raw addresses of pinned arrays that stay alive until
        ``wait_chessboard_by_point``): up to 4 calls in flight, the copy of one overlapping the kernels of the others."""
        offsets = np.arange(n_frames + 1, dtype=np.uint64) * np.uint64(n_points)
        ticket = C.c_int32(-1)
        st = self._lib.ilcc_submit_chessboard_by_point(self._h, C.c_void_p(xyzi_ptr), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       n_frames, C.c_void_p(points_ptr), C.byref(ticket))
        if st != N.OK:
            raise IlccError(st, self._err())
        return ticket.value, n_frames

    def wait_chessboard_by_point(self, ticket, min_plane_points: int = 500):
        t, n_frames = ticket
        res = (N.Result * n_frames)()
        st = self._lib.ilcc_wait_chessboard_by_point(self._h, t, int(min_plane_points), res)
        if st != N.OK:
            raise IlccError(st, self._err())
        return res

    def fetch_classes(self, frame: int) -> np.ndarray:
        n = self._lib.ilcc_fetch_classes(self._h, frame, None, 0)
        if n < 0:
            raise IlccError(-n)
        cl = np.zeros(max(n, 1), dtype=np.uint8)
        self._lib.ilcc_fetch_classes(self._h, frame, cl.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return cl[:n]

    def fetch_labelled(self, frame: int):
        n = self._lib.ilcc_fetch_labelled(self._h, frame, None, None, 0)
        if n < 0:
            raise IlccError(-n)
        yz = np.zeros((max(n, 1), 2), dtype=np.float32)
        lab = np.zeros(max(n, 1), dtype=np.uint8)
        self._lib.ilcc_fetch_labelled(self._h, frame, N.fptr(yz), lab.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        return yz[:n], lab[:n]

    def fetch_walk(self, frame: int):
        """The frame's labelled points in the grid search's walk layout: (yz, label, n_interior, n_rim)."""
        counts = (C.c_uint32 * 2)()
        n = self._lib.ilcc_fetch_walk(self._h, frame, None, None, 0, counts)
        if n < 0:
            raise IlccError(-n)
        yz = np.zeros((max(n, 1), 2), dtype=np.float32)
        lab = np.zeros(max(n, 1), dtype=np.uint8)
        self._lib.ilcc_fetch_walk(self._h, frame, N.fptr(yz), lab.ctypes.data_as(C.POINTER(C.c_uint8)), n, counts)
        return yz[:n], lab[:n], int(counts[0]), int(counts[1])

    def grid_cost(self, yz: np.ndarray, label: np.ndarray, use_oob: bool = True, want_volume: bool = False):
        yz = np.ascontiguousarray(yz, dtype=np.float32).reshape(-1, 2)
        label = np.ascontiguousarray(label, dtype=np.uint8)
        p = self.params
        vol = np.zeros(p.n_th * p.n_ty * p.n_tz * 2, dtype=np.float32) if want_volume else None
        bi, bc = C.c_int32(-1), C.c_float(0)
        st = self._lib.ilcc_grid_cost(self._h, N.fptr(yz), label.ctypes.data_as(C.POINTER(C.c_uint8)), len(label),
                                      int(use_oob), N.fptr(vol) if want_volume else None, C.byref(bi), C.byref(bc))
        if st != N.OK:
            raise IlccError(st, self._err())
        return bi.value, bc.value, vol

    def pattern_refine(self, yz: np.ndarray, label: np.ndarray, lat, phase: int):
        """The GRID-mode refinement kernel alone -> (lat[3], phase, cost_q, alt_cost_q, rounds, hops)."""
        yz = np.ascontiguousarray(yz, dtype=np.float32).reshape(-1, 2)
        label = np.ascontiguousarray(label, dtype=np.uint8)
        q = np.array(lat, dtype=np.int32)
        ph, rounds, hops = C.c_int32(int(phase)), C.c_int32(0), C.c_int32(0)
        cq, aq = C.c_int64(0), C.c_int64(0)
        st = self._lib.ilcc_pattern_refine(self._h, N.fptr(yz), label.ctypes.data_as(C.POINTER(C.c_uint8)), len(label),
                                           q.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ph), C.byref(cq),
                                           C.byref(aq), C.byref(rounds), C.byref(hops))
        if st != N.OK:
            raise IlccError(st, self._err())
        return q, ph.value, cq.value, aq.value, rounds.value, hops.value

    def grid_solve(self, yz: np.ndarray, label: np.ndarray) -> dict:
        """The pipeline's GRID solver (locate launches, common pre-pass, full pass with near ties, refinement with its first-round
        shortcut) on caller-supplied labelled points -> dict(grid_index, grid_cost, lat, phase, cost_q, alt_cost_q, rounds, hops,
        flags, ties).  Diagnostic entry (adversarial inputs for the fp32 ranking)."""
        yz = np.ascontiguousarray(yz, dtype=np.float32).reshape(-1, 2)
        label = np.ascontiguousarray(label, dtype=np.uint8)
        q = np.zeros(3, dtype=np.int32)
        gi, ph, rounds, hops, flags, ties = (C.c_int32(0) for _ in range(6))
        gc = C.c_float(0)
        cq, aq = C.c_int64(0), C.c_int64(0)
        st = self._lib.ilcc_grid_solve(self._h, N.fptr(yz), label.ctypes.data_as(C.POINTER(C.c_uint8)), len(label), C.byref(gi),
                                       C.byref(gc), q.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ph), C.byref(cq), C.byref(aq),
                                       C.byref(rounds), C.byref(hops), C.byref(flags), C.byref(ties))
        if st != N.OK:
            raise IlccError(st, self._err())
        return dict(grid_index=gi.value, grid_cost=gc.value, lat=q, phase=ph.value, cost_q=cq.value, alt_cost_q=aq.value,
                    rounds=rounds.value, hops=hops.value, flags=flags.value, ties=ties.value)

    def get_theta_t(self, yz: np.ndarray, label: np.ndarray, topleft_white: bool, use_oob: bool,
                    theta_t0=(0.0, 0.0, 0.0)):
        yz = np.ascontiguousarray(yz, dtype=np.float32).reshape(-1, 2)
        label = np.ascontiguousarray(label, dtype=np.uint8)
        t = np.array(theta_t0, dtype=np.float64)
        cost, it = C.c_double(0), C.c_int32(0)
        st = self._lib.ilcc_get_theta_t(self._h, N.fptr(yz), label.ctypes.data_as(C.POINTER(C.c_uint8)), len(label),
                                        int(topleft_white), int(use_oob), t.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(cost), C.byref(it))
        if st != N.OK:
            raise IlccError(st, self._err())
        return t, cost.value, it.value

    def timing(self) -> N.Timing:
        t = N.Timing()
        self._lib.ilcc_get_timing(self._h, C.byref(t))
        return t

    def reset_timing(self):
        self._lib.ilcc_reset_timing(self._h)


def save_corners2txt(corners_xyz: np.ndarray, filename: str):
    """get_lidar_corners.cpp:27-36 through the C-ABI writer (identical number formatting)."""
    c = np.ascontiguousarray(corners_xyz, dtype=np.float32).reshape(-1, 3)
    st = N.lib().ilcc_save_corners2txt(N.fptr(c), len(c), filename.encode())
    if st != N.OK:
        raise IlccError(st, filename)


def read_lidar_corners(filename: str, num: int) -> np.ndarray:
    """ImageCornersEst::read_lidar_corners (ImageCornersEst.cpp:281-299)."""
    out = np.zeros((num, 3), dtype=np.float64)
    n = N.lib().ilcc_read_lidar_corners(filename.encode(), num, out.ctypes.data_as(C.POINTER(C.c_double)))
    if n < 0:
        raise IlccError(-n, filename)
    return out[:n]
