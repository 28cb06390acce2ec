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

HEADER_FLOATS = 20   # ILCC_RECORD_HEADER
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


def record_check(corner_block: np.ndarray, tags: np.ndarray) -> np.ndarray:
    """The 24-bit fold K9 stores in header slot 17: xor over the corner floats' bits times (2*position+1), xor
    tag * 0x9E3779B1, folded.  corner_block: [F, 3*n_corners] float32, tags: [F]."""
    bits = np.ascontiguousarray(corner_block, dtype=np.float32).view(np.uint32).astype(np.uint64)
    mult = (2 * np.arange(bits.shape[1], dtype=np.uint64) + 1)[None, :]
    x = np.bitwise_xor.reduce((bits * mult) & np.uint64(0xFFFFFFFF), axis=1) if bits.shape[1] else np.zeros(len(bits), np.uint64)
    t = x ^ ((np.asarray(tags, dtype=np.uint64) & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF))
    return ((t ^ (t >> np.uint64(24))) & np.uint64(0xFFFFFF)).astype(np.uint32)


def verify_records(records: np.ndarray, expect_tags: np.ndarray) -> None:
    """What rank 0 does with the gathered block: every record must carry the tag of the frame that belongs at its
    position (the right rank's records in the right place) and a check word that matches its contents."""
    rec = np.asarray(records, dtype=np.float32)
    tags = rec[:, 16].astype(np.int64)
    want = np.asarray(expect_tags, dtype=np.int64) & 0xFFFFFF
    if not np.array_equal(tags, want):
        bad = int(np.nonzero(tags != want)[0][0])
        raise AssertionError("gathered record %d carries tag %d, expected %d" % (bad, tags[bad], want[bad]))
    chk = record_check(rec[:, HEADER_FLOATS:], tags)
    if not np.array_equal(chk, rec[:, 17].astype(np.uint32)):
        bad = int(np.nonzero(chk != rec[:, 17].astype(np.uint32))[0][0])
        raise AssertionError("gathered record %d fails its content check" % bad)


def pack_records(results: Sequence, n_frames: int, n_corners: Optional[int] = None, tag_base: int = 0) -> np.ndarray:
    """ilcc_result-like records -> [n_frames, record_floats] float32.

    header: status, n_corners, phase, grid_index, iters_a, iters_b, cost_a, cost_b, sel_cost,
    theta, ty, tz, n_plane, n_black, n_white, basin_margin, tag, check, flags, n_roi ; then corners x y z
    (bit-identical to the device-side K9 ``pack_records``)."""
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
        out[:, 15] = sa["basin_margin"]
        out[:, 18] = sa["flags"]
        out[:, 19] = sa["n_roi"]
        out[:, HEADER_FLOATS:] = sa["corners"][:, :3 * n_corners]
        ok = np.arange(3 * n_corners)[None, :] < 3 * np.maximum(sa["n_corners"], 0)[:, None]
        out[:, HEADER_FLOATS:] = np.where(ok, out[:, HEADER_FLOATS:], np.float32(0))
        tags = (int(tag_base) + np.arange(n_frames, dtype=np.int64)) & 0xFFFFFF
        out[:, 16] = tags
        out[:, 17] = record_check(out[:, HEADER_FLOATS:], tags)
        return out
    for f in range(n_frames):
        r = results[f]
        out[f, :15] = (r.status, r.n_corners, r.phase, r.grid_index, r.iters_a, r.iters_b, r.cost_a, r.cost_b,
                       r.sel_cost, r.theta_t[0], r.theta_t[1], r.theta_t[2], r.n_plane, r.n_black, r.n_white)
        out[f, 15] = getattr(r, "basin_margin", 0.0)
        out[f, 18] = getattr(r, "flags", 0)
        out[f, 19] = getattr(r, "n_roi", 0)
        k = min(int(r.n_corners), n_corners)
        if k > 0:
            out[f, HEADER_FLOATS:HEADER_FLOATS + 3 * k] = np.ctypeslib.as_array(r.corners)[:3 * k]
    tags = (int(tag_base) + np.arange(n_frames, dtype=np.int64)) & 0xFFFFFF
    out[:, 16] = tags
    out[:, 17] = record_check(out[:, HEADER_FLOATS:], tags)
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
        rec[:hi - lo] = pack_records(res, hi - lo, n_corners, tag_base=lo)   # tag = global frame index
    t = torch.from_numpy(rec)
    if device is not None:
        t = t.to(device)
    g = gather_records(t, world, rank)
    if g is None:
        return None
    g = g.cpu().numpy()
    g = g[g[:, 0] >= 0][:f_total]
    verify_records(g, np.arange(len(g)))               # every frame's record, intact, at its own position
    return g


# ---------------------------------------------------------------- host-side placement of a rank (8 ranks share one host)
def parse_cpulist(text: str) -> List[int]:
    """`0-63,128-191` (sysfs cpulist) -> [0, ..., 63, 128, ..., 191]."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_numa_node(pci_address: str, sysfs: str = "/sys") -> int:
    """NUMA node of the PCI function `dddd:bb:dd.f` (-1: unknown / the platform does not say)."""
    import os
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_address.lower(), "numa_node")) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return -1


def pin_rank_to_gpu_numa(pci_address: Optional[str], sysfs: str = "/sys", apply: bool = True) -> dict:
    """Pin THIS process to the cores of the NUMA node its GPU hangs off, so that the rank's pinned input buffers (allocated
    after this call: first touch) and its submit thread sit next to the GPU's PCIe root -- at N = 8 the ranks of one node
    share the host's memory bandwidth and a buffer on the far socket crosses the inter-socket link on every H2D copy.
    Returns {"pci", "numa_node", "cpus" (count), "pinned"}; does nothing (pinned False) when the node is unknown."""
    import os
    info = {"pci": pci_address, "numa_node": -1, "cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0,
            "pinned": False}
    if not pci_address:
        return info
    node = gpu_numa_node(pci_address, sysfs)
    info["numa_node"] = node
    if node < 0:
        return info
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            cpus = parse_cpulist(fh.read())
    except OSError:
        return info
    if hasattr(os, "sched_getaffinity"):
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed]
    if not cpus:
        return info
    info["cpus"] = len(cpus)
    if apply and hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, cpus)
        info["pinned"] = True
    return info
