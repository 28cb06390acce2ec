"""ctypes binding of libilcc_hip.so (the C-ABI declared in include/ilcc_hip.h).

There is deliberately no fallback: if the HIP library is missing, or no HIP device is present,
loading / ``create`` raises.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILCC_HIP_LIB selects another build of the same library (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get("ILCC_HIP_LIB") or os.path.join(_HERE, "libilcc_hip.so")
MAX_CORNERS = 256

OK, NO_ROI_POINTS, NO_CLUSTER, NO_PLANE, DEGENERATE_HIST, TOO_FEW_POINTS, BAD_ARGUMENT, CAPACITY, \
    HIP_ERROR, IO_ERROR, BOARD_NOT_FOUND, AMBIGUOUS = range(12)
FLAG_TIE_OVERFLOW, FLAG_REFINE_CAPPED, FLAG_LOW_COVERAGE, FLAG_BORDER_RISK = 1, 2, 4, 8
FLAGS_FP32_ONLY = FLAG_TIE_OVERFLOW | FLAG_BORDER_RISK   # flags about the fp32 grid pass: the (exact) oracle has no counterpart
RECORD_HEADER = 20
TIMELINE_COLS = 15
COST_Q_ONE = float(1 << 40)
SOLVER_REFERENCE_LOCAL, SOLVER_GRID = 0, 1
CLOUD_ROI, CLOUD_CLUSTER, CLOUD_CHESSBOARD, CLOUD_PCA, CLOUD_OPTIM = range(5)

# every symbol include/ilcc_hip.h declares
EXPORTS = [
    "ilcc_abi_version", "ilcc_strerror", "ilcc_last_error", "ilcc_default_params",
    "ilcc_set_chessboard_param", "ilcc_create", "ilcc_destroy", "ilcc_set_params", "ilcc_reserve", "ilcc_extract",
    "ilcc_extract_batch", "ilcc_extract_batch_device", "ilcc_submit_batch_device", "ilcc_submit_batch", "ilcc_wait", "ilcc_wait_records_device", "ilcc_fetch_cloud", "ilcc_fetch_labelled", "ilcc_fetch_walk", "ilcc_chessboard_by_point_batch", "ilcc_submit_chessboard_by_point", "ilcc_wait_chessboard_by_point", "ilcc_fetch_classes",
    "ilcc_grid_cost", "ilcc_grid_solve", "ilcc_pattern_refine", "ilcc_get_theta_t", "ilcc_get_timing", "ilcc_reset_timing",
    "ilcc_save_corners2txt", "ilcc_read_lidar_corners",
    "ilcc_set_result_mode", "ilcc_wait_compact", "ilcc_record_floats", "ilcc_fetch_results",
    "ilcc_debug_timeline_enable", "ilcc_debug_timeline_fetch",
]
ABI_VERSION = 5            # the layout of Params / Result / Timing below is ILCC_ABI_VERSION 5 of include/ilcc_hip.h
RESULTS_FULL, RESULTS_COMPACT = 0, 1


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
        ("phase_mode", C.c_int32),
        ("max_iterations", C.c_int32),
        ("grid_prune", C.c_int32),
        ("n_th", C.c_int32), ("n_ty", C.c_int32), ("n_tz", C.c_int32),
        ("th_min", C.c_double), ("th_step", C.c_double),
        ("ty_min", C.c_double), ("ty_step", C.c_double),
        ("tz_min", C.c_double), ("tz_step", C.c_double),
        ("refine_div", C.c_int32), ("refine_max_rounds", C.c_int32),
        ("refine_th_margin", C.c_int32), ("reserved0", C.c_int32),
        ("ambiguity_eps", C.c_double),
        ("online_cluster_tol", C.c_double),
        ("min_cell_coverage", C.c_double),
        ("ransac_probability", C.c_double),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("n_points", C.c_int32),
        ("n_roi", C.c_int32), ("n_cluster", C.c_int32), ("n_plane", C.c_int32),
        ("n_black", C.c_int32), ("n_gray", C.c_int32), ("n_white", C.c_int32),
        ("n_corners", C.c_int32),
        ("phase", C.c_int32),
        ("iters_a", C.c_int32), ("iters_b", C.c_int32),
        ("grid_index", C.c_int32),
        ("found_board", C.c_int32),
        ("grid_cost", C.c_float),
        ("plane", C.c_float * 4),
        ("pca", C.c_float * 16),
        ("gray_zone", C.c_double * 2),
        ("theta_t", C.c_double * 3),
        ("cost_a", C.c_double), ("cost_b", C.c_double),
        ("sel_cost", C.c_double),
        ("basin_margin", C.c_double),
        ("flags", C.c_int32),
        ("grid_ties", C.c_int32),
        ("cells_hit", C.c_int32),
        ("n_oob", C.c_int32),
        ("corners", C.c_float * (MAX_CORNERS * 3)),
    ]

    def corners_array(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.corners)[:3 * self.n_corners].reshape(-1, 3).copy()


class Timing(C.Structure):
    _fields_ = [
        ("roi_crop", C.c_float), ("cluster", C.c_float), ("ransac_plane", C.c_float),
        ("plane_frame_hist", C.c_float), ("grid_cost", C.c_float), ("refine_corners", C.c_float),
        ("total", C.c_float),
        ("grid_cost_launches", C.c_uint32),
        ("grid_cost_ms_sum", C.c_double),
        ("grid_cost_evals_sum", C.c_uint64),
        ("grid_cost_evals_nominal_sum", C.c_uint64),
        ("grid_cost_evals_interior_sum", C.c_uint64),
        ("grid_cost_box_evals_sum", C.c_uint64),
        ("grid_cost_kernel_ms_sum", C.c_double),
        ("grid_cost_full_ms_sum", C.c_double),
        ("walk_order_ms_sum", C.c_double),
        ("grid_cost_prepass_ms_sum", C.c_double),
        ("grid_cost_locate_ms_sum", C.c_double),
        ("batches", C.c_uint64),
        ("stage_ms_sum", C.c_double * 7),
        ("roi_count_ms_sum", C.c_double),
        ("online_second_tier_frames", C.c_uint64),
    ]


_lib = None


def lib():
    """Load libilcc_hip.so; raises OSError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, fp = C.c_void_p, C.POINTER(C.c_float)
        pp, rp = C.POINTER(Params), C.POINTER(Result)
        L.ilcc_abi_version.restype = C.c_int32
        got = L.ilcc_abi_version()
        if got != ABI_VERSION:
            # a stale build would load silently and every Result after frame 0 would be read at the wrong stride
            raise OSError(f"{LIB_PATH} implements ABI {got}, this package declares ABI {ABI_VERSION}: rebuild it "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
        L.ilcc_strerror.argtypes = [C.c_int32]
        L.ilcc_strerror.restype = C.c_char_p
        L.ilcc_last_error.argtypes = [vp]
        L.ilcc_last_error.restype = C.c_char_p
        L.ilcc_default_params.argtypes = [pp]
        L.ilcc_set_chessboard_param.argtypes = [pp, C.c_char_p]
        L.ilcc_set_chessboard_param.restype = C.c_int32
        L.ilcc_create.argtypes = [C.c_int32, pp, C.c_uint32, C.c_uint64]
        L.ilcc_create.restype = vp
        L.ilcc_destroy.argtypes = [vp]
        L.ilcc_set_params.argtypes = [vp, pp]
        L.ilcc_set_params.restype = C.c_int32
        L.ilcc_reserve.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.ilcc_reserve.restype = C.c_int32
        L.ilcc_extract.argtypes = [vp, fp, C.c_uint32, fp, rp]
        L.ilcc_extract.restype = C.c_int32
        L.ilcc_extract_batch.argtypes = [vp, fp, C.POINTER(C.c_uint64), C.c_uint32, fp, rp]
        L.ilcc_extract_batch.restype = C.c_int32
        L.ilcc_extract_batch_device.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_uint32, vp, rp]
        L.ilcc_extract_batch_device.restype = C.c_int32
        L.ilcc_submit_batch_device.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_uint32, vp, C.POINTER(C.c_int32)]
        L.ilcc_submit_batch_device.restype = C.c_int32
        L.ilcc_submit_batch.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_uint32, vp, C.POINTER(C.c_int32)]
        L.ilcc_submit_batch.restype = C.c_int32
        L.ilcc_wait.argtypes = [vp, C.c_int32, rp]
        L.ilcc_wait.restype = C.c_int32
        L.ilcc_wait_records_device.argtypes = [vp, C.c_int32, rp, vp, C.c_uint32, C.c_uint32]
        L.ilcc_wait_records_device.restype = C.c_int32
        L.ilcc_chessboard_by_point_batch.argtypes = [vp, fp, C.POINTER(C.c_uint64), C.c_uint32, fp, C.c_int32, rp]
        L.ilcc_chessboard_by_point_batch.restype = C.c_int32
        L.ilcc_submit_chessboard_by_point.argtypes = [vp, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.POINTER(C.c_int32)]
        L.ilcc_submit_chessboard_by_point.restype = C.c_int32
        L.ilcc_wait_chessboard_by_point.argtypes = [vp, C.c_int32, C.c_int32, rp]
        L.ilcc_wait_chessboard_by_point.restype = C.c_int32
        L.ilcc_fetch_classes.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64]
        L.ilcc_fetch_classes.restype = C.c_int64
        L.ilcc_fetch_cloud.argtypes = [vp, C.c_uint32, C.c_int32, fp, C.c_uint64]
        L.ilcc_fetch_cloud.restype = C.c_int64
        L.ilcc_fetch_labelled.argtypes = [vp, C.c_uint32, fp, C.POINTER(C.c_uint8), C.c_uint64]
        L.ilcc_fetch_labelled.restype = C.c_int64
        L.ilcc_fetch_walk.argtypes = [vp, C.c_uint32, fp, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint32)]
        L.ilcc_fetch_walk.restype = C.c_int64
        L.ilcc_grid_cost.argtypes = [vp, fp, C.POINTER(C.c_uint8), C.c_uint32, C.c_int32, fp,
                                     C.POINTER(C.c_int32), fp]
        L.ilcc_grid_cost.restype = C.c_int32
        L.ilcc_grid_solve.argtypes = [vp, fp, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_int32), fp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ilcc_grid_solve.restype = C.c_int32
        L.ilcc_pattern_refine.argtypes = [vp, fp, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ilcc_pattern_refine.restype = C.c_int32
        L.ilcc_get_theta_t.argtypes = [vp, fp, C.POINTER(C.c_uint8), C.c_uint32, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.ilcc_get_theta_t.restype = C.c_int32
        L.ilcc_get_timing.argtypes = [vp, C.POINTER(Timing)]
        L.ilcc_reset_timing.argtypes = [vp]
        L.ilcc_save_corners2txt.argtypes = [fp, C.c_uint32, C.c_char_p]
        L.ilcc_save_corners2txt.restype = C.c_int32
        L.ilcc_read_lidar_corners.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_double)]
        L.ilcc_read_lidar_corners.restype = C.c_int32
        L.ilcc_set_result_mode.argtypes = [vp, C.c_int32]
        L.ilcc_set_result_mode.restype = C.c_int32
        L.ilcc_wait_compact.argtypes = [vp, C.c_int32, fp, C.c_uint64]
        L.ilcc_wait_compact.restype = C.c_int32
        L.ilcc_record_floats.argtypes = [vp, C.c_int32]
        L.ilcc_record_floats.restype = C.c_uint32
        L.ilcc_debug_timeline_enable.argtypes = [vp, C.c_int32]
        L.ilcc_debug_timeline_enable.restype = C.c_int32
        L.ilcc_debug_timeline_fetch.argtypes = [vp, C.POINTER(C.c_double), C.c_uint32]
        L.ilcc_debug_timeline_fetch.restype = C.c_int32
        L.ilcc_fetch_results.argtypes = [vp, C.c_uint32, C.c_uint32, rp]
        L.ilcc_fetch_results.restype = C.c_int32
        _lib = L
    return _lib


def default_params() -> Params:
    p = Params()
    lib().ilcc_default_params(C.byref(p))
    return p


def strerror(status: int) -> str:
    return lib().ilcc_strerror(int(status)).decode()


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))
