"""Frame sharding across the GPUs of one node and the single gather of per-frame corner records.

Frames (bag x click) are independent -- the reference loops bags independently and overwrites every
member of LidarCornersEst per frame (``ilcc2/test/get_lidar_corners.cpp:130-211``) -- so ranks own
contiguous blocks of frames, run the whole path locally, and rank 0 collects fixed-size result
records with ONE collective (``torch.distributed.gather``: RCCL on GPUs, gloo in the CPU tests).
There is no other exchange step on this path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

HEADER_FLOATS = 16
MAX_CORNERS = 256
RECORD_FLOATS = HEADER_FLOATS + 3 * MAX_CORNERS   # upper bound; pack with n_corners to shrink


def record_floats(n_corners: int) -> int:
    return HEADER_FLOATS + 3 * int(n_corners)


def shard_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of ceil(F/G) frames per rank (the last ranks may own fewer)."""
    per = -(-n_frames // world)
    lo = min(n_frames, rank * per)
    return lo, min(n_frames, lo + per)


_RESULT_DTYPE = None


def _as_struct_array(results, n_frames: int):
    """ctypes array of ilcc_result -> numpy structured view (no copy); None for other sequences."""
    try:
        import ctypes
        from . import _native as N
        global _RESULT_DTYPE
        if isinstance(results, ctypes.Array) and results._type_ is N.Result:
            if _RESULT_DTYPE is None:
                _RESULT_DTYPE = np.dtype(N.Result)
            return np.frombuffer(results, dtype=_RESULT_DTYPE, count=n_frames)
    except Exception:
        pass
    return None


def pack_records(results: Sequence, n_frames: int, n_corners: Optional[int] = None) -> np.ndarray:
    """ilcc_result-like records -> [n_frames, record_floats] float32.

    header: status, n_corners, phase, grid_index, iters_a, iters_b, cost_a, cost_b, sel_cost,
    theta, ty, tz, n_plane, n_black, n_white, 0 ; then corners x y z."""
    if n_corners is None:
        n_corners = max([int(results[f].n_corners) for f in range(n_frames)] + [0])
    out = np.zeros((n_frames, record_floats(n_corners)), dtype=np.float32)
    sa = _as_struct_array(results, n_frames)
    if sa is not None:                      # vectorised: one numpy pass over the whole batch
        for col, name in enumerate(("status", "n_corners", "phase", "grid_index", "iters_a", "iters_b", "cost_a",
                                    "cost_b", "sel_cost")):
            out[:, col] = sa[name]
        out[:, 9:12] = sa["theta_t"]
        out[:, 12] = sa["n_plane"]
        out[:, 13] = sa["n_black"]
        out[:, 14] = sa["n_white"]
        out[:, HEADER_FLOATS:] = sa["corners"][:, :3 * n_corners]
        ok = np.arange(3 * n_corners)[None, :] < 3 * np.minimum(sa["n_corners"], n_corners)[:, None]
        out[:, HEADER_FLOATS:] *= ok
        return out
    for f in range(n_frames):
        r = results[f]
        out[f, :15] = (r.status, r.n_corners, r.phase, r.grid_index, r.iters_a, r.iters_b, r.cost_a, r.cost_b,
                       r.sel_cost, r.theta_t[0], r.theta_t[1], r.theta_t[2], r.n_plane, r.n_black, r.n_white)
        k = min(int(r.n_corners), n_corners)
        if k > 0:
            out[f, HEADER_FLOATS:HEADER_FLOATS + 3 * k] = np.ctypeslib.as_array(r.corners)[:3 * k]
    return out


def unpack_corners(records: np.ndarray) -> List[np.ndarray]:
    out = []
    for rec in np.asarray(records):
        k = int(rec[1])
        out.append(np.asarray(rec[HEADER_FLOATS:HEADER_FLOATS + 3 * k], dtype=np.float32).reshape(k, 3))
    return out


def gather_records(local, world: int, rank: int, dst: int = 0, async_op: bool = False, force_collective: bool = False):
    """One ``torch.distributed.gather`` of this rank's [F_local, R] record tensor to ``dst``.
    Every rank must pass the same shape (pad the last shard).  Returns [world*F_local, R] on dst
    (None elsewhere); with ``async_op`` returns ``(work, bufs)`` so the caller can overlap the
    collective with the next batch and ``work.wait()`` later."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:   # (force_collective: run the 1-rank collective anyway, a bench test hook)
        return (None, [local]) if async_op else local
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    work = dist.gather(local, gather_list=bufs, dst=dst, async_op=async_op)
    if async_op:
        return work, bufs
    return torch.cat(bufs, 0) if rank == dst else None


def run_sharded(extract_fn, clouds: np.ndarray, clicks: np.ndarray, world: int, rank: int, n_corners: int,
                device=None):
    """Shard [F,N,4] frames over ranks, run ``extract_fn(clouds, clicks) -> results`` on the local
    block, gather records on rank 0.  ``extract_fn`` is the HIP-backed LidarCornersBatch.extract in
    production; the CPU tests inject their own producer to exercise the sharding/gather logic."""
    import torch
    f_total = len(clicks)
    per = -(-f_total // world)
    lo, hi = shard_range(f_total, world, rank)
    rec = np.zeros((per, record_floats(n_corners)), dtype=np.float32)
    rec[:, 0] = -1.0                                   # padding marker
    if hi > lo:
        res = extract_fn(clouds[lo:hi], clicks[lo:hi])
        rec[:hi - lo] = pack_records(res, hi - lo, n_corners)
    t = torch.from_numpy(rec)
    if device is not None:
        t = t.to(device)
    g = gather_records(t, world, rank)
    if g is None:
        return None
    g = g.cpu().numpy()
    return g[g[:, 0] >= 0][:f_total]
