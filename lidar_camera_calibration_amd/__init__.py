"""MI355X-native LiDAR chessboard-corner extraction (ilcc2's get_lidar_corners path).

Product code only: HIP kernels + C-ABI in ``csrc/`` (libilcc_hip.so), the host mirror of the
reference's ``LidarCornersEst`` and the synthetic-cloud generator.  Never imports ``oracle/``.
"""
from . import synth  # noqa: F401
from .lidar_corners_est import (IlccError, LidarCornersBatch, LidarCornersEst, read_lidar_corners,  # noqa: F401
                                save_corners2txt)
