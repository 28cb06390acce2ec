"""MI355X-native LiDAR chessboard-corner extraction (ilcc2's get_lidar_corners path).

Product code only: HIP kernels + C-ABI in ``csrc/`` (libilcc_hip.so), the host mirror of the
reference's ``LidarCornersEst`` and the synthetic-cloud generator.  Never imports ``oracle/``.
"""
from . import synth  # noqa: F401
from .lidar_corners_est import (IlccError, LidarCornersBatch, LidarCornersEst, read_lidar_corners,  # noqa: F401
                                save_corners2txt)

# LidarCornersBatch keeps up to four batches in flight, each on its own HIP stream.  The HIP runtime maps streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.  This only takes
# effect when the runtime has not been initialised yet (import this package -- or set the variable -- before torch).
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
