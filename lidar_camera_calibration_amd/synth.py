"""Seeded synthetic spinning-LiDAR frames with a chessboard.

The reference's input bags (``bag/2018-12-03-{1..6}.bag``) are not in the tree
(``/root/reference/.MISSING_LARGE_BLOBS``), so every workload in BASELINE.json is fed from this
generator.  A frame is what ``pcl::fromROSMsg`` hands to ``LidarCornersEst::setROI``
(``ilcc2/test/get_lidar_corners.cpp:163-183``): N packed float32 ``x y z intensity`` records in
sensor firing order (azimuth-major, ring-minor).

Scene: chessboard (W x H squares of side g, one border ring of squares around the inner corner
lattice) + ground plane + back wall + a far shell for rays that hit nothing (N stays constant).
Range noise N(0, sigma_r); intensity white N(100,8) / black N(12,4) / other N(40,15), clipped to
[0,255]; a Gaussian beam footprint (sigma 1.5 cm) blends black and white across square edges.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
from scipy.special import erf

# board poses recovered from ilcc2/process_data/pointgrey_lidar_{1..6}.txt:
# centre, e_outer (short axis step), e_inner (long axis step)
FIXTURE_POSES = [
    ((2.062885, 0.160476, 0.001947), (0.008092, -0.000394, -0.149781), (-0.014037, -0.149341, -0.000365)),
    ((2.409136, -0.193880, 0.079921), (0.002968, -0.006210, -0.149842), (-0.078429, -0.127808, 0.003744)),
    ((2.409740, 0.856759, 0.047132), (-0.011674, -0.001709, 0.149535), (-0.083748, 0.124339, -0.005117)),
    ((2.551976, 0.391551, 0.006192), (-0.028303, -0.050572, 0.138352), (-0.071489, 0.127896, 0.032125)),
    ((2.257671, 0.762138, 0.025796), (0.060627, 0.042522, 0.130446), (0.044725, -0.140949, 0.025159)),
    ((2.828328, -0.127461, 0.105297), (-0.061594, 0.011997, 0.136244), (0.025606, 0.147791, -0.001437)),
]


@dataclasses.dataclass
class Lidar:
    elevations_deg: np.ndarray
    n_azimuth: int

    @property
    def n_points(self) -> int:
        return len(self.elevations_deg) * self.n_azimuth


def vlp16() -> Lidar:
    """16 rings -15..+15 deg step 2, 1800 azimuth steps of 0.2 deg -> N = 28 800."""
    return Lidar(np.arange(-15.0, 15.01, 2.0), 1800)


def hdl64() -> Lidar:
    """64 rings -25..+15 deg uniform, 2048 azimuth steps -> N = 131 072 (config 5)."""
    return Lidar(np.linspace(-25.0, 15.0, 64), 2048)


@dataclasses.dataclass
class Board:
    """W x H squares with W <= H (LidarCornersEst.cpp:31-39), side g."""
    w: int = 6
    h: int = 8
    g: float = 0.15

    @property
    def n_corners(self) -> int:
        return (self.w - 1) * (self.h - 1)


@dataclasses.dataclass
class Pose:
    centre: np.ndarray   # board centre in the LiDAR frame
    u: np.ndarray        # unit vector along the W-square (short) side = file's outer loop
    v: np.ndarray        # unit vector along the H-square (long) side = file's inner loop
    topleft_white: bool = True

    @property
    def normal(self) -> np.ndarray:
        return np.cross(self.u, self.v)


def pose_from_fixture(idx: int) -> Pose:
    c, eo, ei = FIXTURE_POSES[idx]
    u = np.asarray(eo, dtype=np.float64)
    v = np.asarray(ei, dtype=np.float64)
    u = u / np.linalg.norm(u)
    v = v - u * float(u @ v)
    v = v / np.linalg.norm(v)
    return Pose(np.asarray(c, dtype=np.float64), u, v, True)


def random_pose(rng: np.random.Generator, range_m=(2.0, 3.5), yaw_deg=35.0, pitch_deg=25.0,
                roll_deg=30.0, bearing_deg=25.0) -> Pose:
    """Config-4 pose sampling: range U(2,3.5) m, yaw +-35, pitch +-25, roll +-30 deg."""
    r = rng.uniform(*range_m)
    bearing = math.radians(rng.uniform(-bearing_deg, bearing_deg))
    centre = np.array([r * math.cos(bearing), r * math.sin(bearing), rng.uniform(-0.05, 0.12)])
    # nominal frame facing the sensor: normal = -radial, long axis horizontal, short axis up
    radial = np.array([math.cos(bearing), math.sin(bearing), 0.0])
    up = np.array([0.0, 0.0, 1.0])
    side = np.cross(up, radial)
    yaw, pitch, roll = (math.radians(rng.uniform(-a, a)) for a in (yaw_deg, pitch_deg, roll_deg))

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + math.sin(ang) * k + (1 - math.cos(ang)) * (k @ k)

    rmat = rot(up, yaw) @ rot(side, pitch) @ rot(radial, roll)
    u = rmat @ up
    v = rmat @ side
    return Pose(centre, u, v, bool(rng.integers(0, 2)))


def true_corners(pose: Pose, board: Board) -> np.ndarray:
    """(W-1)*(H-1) x 3, outer loop over the short axis, inner over the long axis
    (LidarCornersEst.cpp:513-534)."""
    out = []
    for i in range(1, board.w):
        for j in range(1, board.h):
            out.append(pose.centre + (i - board.w / 2.0) * board.g * pose.u
                       + (j - board.h / 2.0) * board.g * pose.v)
    return np.asarray(out)


def _blurred_square_wave(s: np.ndarray, n_sq: int, g: float, sigma: float) -> np.ndarray:
    """+-1 square wave (+1 on the first square of the board) blurred by a 1-D Gaussian."""
    edges = (np.arange(-3, n_sq + 4) - n_sq / 2.0) * g
    out = np.full_like(s, 1.0)            # value left of the first listed edge (k = -3)
    sign = -1.0
    denom = math.sqrt(2.0) * max(sigma, 1e-9)
    for e in edges:
        out = out + sign * (1.0 + erf((s - e) / denom))
        sign = -sign
    return out


def make_frame(lidar: Lidar, board: Board, pose: Pose, seed: int, sigma_r: float = 0.01,
               footprint: float = 0.015, ground_z: float = -1.0, wall_behind: float = 1.5,
               far: float = 50.0) -> np.ndarray:
    """Returns an (N,4) float32 array x,y,z,intensity."""
    rng = np.random.Generator(np.random.Philox(key=int(seed) & 0xFFFFFFFFFFFFFFFF))
    el = np.radians(lidar.elevations_deg)
    az = np.arange(lidar.n_azimuth) * (2.0 * math.pi / lidar.n_azimuth) - math.pi
    azg, elg = np.meshgrid(az, el, indexing="ij")          # azimuth-major firing order
    d = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], -1).reshape(-1, 3)

    n = pose.normal
    t_best = np.full(d.shape[0], far)
    kind = np.zeros(d.shape[0], dtype=np.int8)             # 0 far, 1 board, 2 ground, 3 wall

    # board
    dn = d @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (pose.centre @ n) / dn
    hit = d * t[:, None] - pose.centre
    su, sv = hit @ pose.u, hit @ pose.v
    ok = (t > 0) & np.isfinite(t) & (np.abs(su) <= board.w * board.g / 2) & (np.abs(sv) <= board.h * board.g / 2)
    ok &= t < t_best
    t_best = np.where(ok, t, t_best)
    kind = np.where(ok, 1, kind)
    # ground
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ground_z / d[:, 2]
    ok = (t > 0) & np.isfinite(t) & (t < t_best)
    t_best = np.where(ok, t, t_best)
    kind = np.where(ok, 2, kind)
    # back wall (vertical, facing the sensor, wall_behind metres past the board centre)
    hdir = np.array([pose.centre[0], pose.centre[1], 0.0])
    dist = np.linalg.norm(hdir)
    hdir = hdir / dist
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (dist + wall_behind) / (d @ hdir)
    ok = (t > 0) & np.isfinite(t) & (t < t_best)
    t_best = np.where(ok, t, t_best)
    kind = np.where(ok, 3, kind)

    rngn = rng.standard_normal((d.shape[0], 2))
    t_noisy = t_best + sigma_r * rngn[:, 0]
    xyz = d * t_noisy[:, None]

    # intensity
    inten = 40.0 + 15.0 * rngn[:, 1]
    on = kind == 1
    if on.any():
        hit = d[on] * t_best[on, None] - pose.centre
        su, sv = hit @ pose.u, hit @ pose.v
        wave = _blurred_square_wave(su, board.w, board.g, footprint) * \
            _blurred_square_wave(sv, board.h, board.g, footprint)
        sgn = 1.0 if pose.topleft_white else -1.0
        wf = np.clip(0.5 * (1.0 + sgn * wave), 0.0, 1.0)
        mean = wf * 100.0 + (1 - wf) * 12.0
        std = np.sqrt(wf * 8.0 ** 2 + (1 - wf) * 4.0 ** 2)
        inten[on] = mean + std * rngn[on, 1]
    inten = np.clip(inten, 0.0, 255.0)
    return np.concatenate([xyz, inten[:, None]], 1).astype(np.float32)


def make_click(pose: Pose, seed: int) -> np.ndarray:
    """rviz click stand-in: board centre + U(-0.1,0.1)^3."""
    rng = np.random.Generator(np.random.Philox(key=(int(seed) ^ 0xC11C) & 0xFFFFFFFFFFFFFFFF))
    return (pose.centre + rng.uniform(-0.1, 0.1, 3)).astype(np.float32)


def make_batch(n_frames: int, lidar: Lidar | None = None, board: Board | None = None,
               seed: int = 0xC0FFEE, fixture_poses: bool = False, **pose_kw):
    """n_frames frames -> (clouds [F,N,4] f32, clicks [F,3] f32, gt corners [F,C,3] f64, poses)."""
    lidar = lidar or vlp16()
    board = board or Board()
    clouds, clicks, gts, poses = [], [], [], []
    for f in range(n_frames):
        s = seed + f
        if fixture_poses:
            pose = pose_from_fixture(f % len(FIXTURE_POSES))
        else:
            prng = np.random.Generator(np.random.Philox(key=(s ^ 0x905E) & 0xFFFFFFFFFFFFFFFF))
            pose = random_pose(prng, **pose_kw)
        clouds.append(make_frame(lidar, board, pose, s))
        clicks.append(make_click(pose, s))
        gts.append(true_corners(pose, board))
        poses.append(pose)
    return np.stack(clouds), np.stack(clicks), np.stack(gts), poses


def corner_error(est: np.ndarray, gt: np.ndarray, board: Board) -> float:
    """max |est - gt| (m) over corners, minimised over the four index symmetries of the lattice
    (the eigenvector signs of the plane frame are arbitrary; the consumer
    re-orders with check_order_lidar, ilcc2/src/ImageCornersEst.cpp:461-488)."""
    a, b = board.w - 1, board.h - 1
    e = np.asarray(est, dtype=np.float64).reshape(a, b, 3)
    g = np.asarray(gt, dtype=np.float64).reshape(a, b, 3)
    best = math.inf
    for fo in (False, True):
        for fi in (False, True):
            x = e[::-1] if fo else e
            x = x[:, ::-1] if fi else x
            best = min(best, float(np.linalg.norm(x - g, axis=-1).max()))
    return best
