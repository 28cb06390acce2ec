#!/usr/bin/env python3
"""bench.py -- chessboard-corner-extraction frames/s on synthetic VLP-16 clouds (BASELINE.json).

A "step" is one pass of the whole hot path (K1 crop -> K2 cluster -> K3 RANSAC plane -> K4/K5 plane frame + gray
zone -> K6 exhaustive (theta,ty,tz) x phase grid cost -> K7r monotone refinement + basin check -> K7b corners) over
configs[3]'s 1024 synthetic frames PER GPU, fed as ONE batch of 1024 frames (configs[1] frames: 16 rings x
1800 azimuths = 28 800 XYZI points, 7x5-corner board @0.15 m, one random board pose per frame) through the
library's submit/wait pipeline (up to 4 batches = 4 steps in flight).  The batch is 472 MB of input per GPU --
more than the 256 MB Infinity Cache -- so K1 reads HBM, not cache.  (Rounds 3 and 4 until the common pre-pass: 2 batches of
512; since K6's share fell, larger batches pay: 855 k frames/s at 2 x 512, 925 k at 1 x 1024.)  Inputs are resident in HBM when the timed region
starts (the bench contract: `value` is never a PCIe-inclusive rate); the same pipeline with every batch starting in
pinned HOST memory is timed right after, on every rank, and reported at top level as `value_h2d_inclusive` with the
link's own rate beside it (`link_GBps_achieved`, `link_bound_frames_per_s`, `link_frac`) -- SURVEY.md 8(d)'s metric as
written counts that copy, and at one GPU config 2 is PCIe-bound by it.  Per-frame result records come back to the host
in the library's compact form (ILCC_RESULTS_COMPACT: 500 B per frame) and, for N > 1, one step's 1024 records per rank
are gathered to rank 0 with ONE RCCL gather per step (frames are independent: no other collective); rank 0 verifies
tags and content checks of what arrived.
Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W`.

Warm-up: W steps, then blocks of 5 steps until the median step time of a block is within 2 % of the previous block's and
0.3 s have passed (clock ramp and pipeline fill are not steady state); exactly K steps are then timed between barriers.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- the dominant kernel (k6_grid_cost: five launches per batch -- seed, refinement, anchor, common pre-pass,
                  full pass -- each bracketed by HIP events on the batch's own stream).  VALU-bound (points live in LDS,
                  no MFMA): `bound: "valu"`, achieved = (executed point-candidate evaluations x VALU instructions each)
                  / k6_ms_alone, the launches' summed duration with ONE batch in flight (a short leg right after the timed
                  region: the only exclusive durations; a rocprofv3 trace of `--in-flight 1` adds up to the same).  The
                  timed region's own event spans (four batches sharing the chip) are `k6_ms_pipelined` / `frac_pipelined`;
                  `frac_rocprof_*` price the same work on the committed rocprofv3 kernel stats.  All scalars: the
                  algorithmic HBM figure BASELINE.json asks for (`hbm_frac`: 16 N + 12 corners + 64 bytes per frame over
                  the whole step; `k1_hbm_frac`: K1's count pass alone), SURVEY.md 8(d)'s H2D-inclusive rate and the
                  link's (`h2d_inclusive_frames_per_s`, `link_frac`).
  cpu_baseline -- the CPU oracle's reference-faithful path (crop, cluster, RANSAC, PCA, histogram, two-pass
                  Ceres-style local solve for both colour phases), one host thread, median of 5 runs on a bounded
                  sample of the same frames.  A port (the reference cannot be built here: PCL/Eigen/Ceres/ROS
                  absent) and only a reported baseline.
  grid_vs_reference_path_mm -- how far the headline mode's corners (ILCC_SOLVER_GRID) are from the
                  reference-trajectory mode's (ILCC_SOLVER_REFERENCE_LOCAL, which matches the CPU port to < 1e-6 m)
                  on the same frames, and that mode's own pipelined frames/s.
`--config 5` runs BASELINE config 5 instead (64-ring 131 072-point clouds, 11x8 board @0.10 m, 129^3 x 2 grid).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# One HIP stream per batch in flight (4) + torch's streams: with the runtime's default of 4 hardware queues two
# of them would share a queue and serialise.  Must be set before the HIP runtime initialises (i.e. before torch);
# libilcc_hip.so does the same when it is loaded first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-instructions/s
# (= the 157.3 TFLOP/s fp32 vector peak of MI355X_MICROARCH.md when every instruction is an FMA)
VALU_ISSUE_PEAK_T = 78.6
# VALU instructions of ONE (point, candidate) evaluation of k6_grid_cost (both colour phases), counted by
# tools/k6_isa_count.sh in the gfx950 assembly of the term functions (probe kernels in csrc/k6_grid_cost.hip, the library's
# compile flags): border-class point (out-of-board logic included) / interior-class point (it cannot leave the board under
# any translation of the grid).  Bound tests, tile prologues, address arithmetic and staging are NOT credited.  The
# committed output of the script is profiles/r06_k6_isa_count.json (with the opcode histogram); tests/test_host_logic.py::test_k6_credit_matches_the_isa
# re-runs the script and fails when these constants, that file and the current source disagree.
K6_VALU_OPS_BORDER = 29.0
K6_VALU_OPS_INTERIOR = 15.0
# ... and of ONE (point, 16-candidate tile) evaluation of the full pass's box pre-pass (box_term: a lower bound for the whole tile)
K6_VALU_OPS_BOX = 24.0
# PMC passes of the K6 stage at the batch sizes this bench runs (tools/gpu_pmc.sh -> profiles/): per-launch counters of one
# batch alone on the chip.  roofline.traffic and roofline.issued_vs_credited are computed from these files at run time.
PMC_FILES = {(2, 1024): ("profiles/r06_pmc_cfg2_1024f.csv", "profiles/r05_pmc_cfg2_1024f.csv", "profiles/r04_pmc_cfg2_1024f.csv"),
             (2, 512): ("profiles/r04_pmc_cfg2_512f.csv", "profiles/r03_pmc_cfg2_512f.csv"),
             (5, 128): ("profiles/r06_pmc_cfg5_128f.csv", "profiles/r05_pmc_cfg5_128f.csv"),
             (5, 64): ("profiles/r04_pmc_cfg5_64f.csv",)}   # (round 3's config-5 file averaged a cold first dispatch in: not used)
# rocprofv3 --kernel-trace --stats of this bench's own command (tools/gpu_profile.sh): pipelined (4 batches in flight) and
# --in-flight 1 (one batch alone on the chip).  roofline.rocprof recomputes `frac` from their per-kernel averages.
KSTATS_FILES = {2: (("profiles/r06_kernel_stats_bench_20_5.csv", "profiles/r05_kernel_stats_bench_20_5.csv"),
                    ("profiles/r06_kernel_stats_bench_inflight1.csv", "profiles/r05_kernel_stats_bench_inflight1.csv")),
                5: (("profiles/r06_kernel_stats_config5.csv", "profiles/r05_kernel_stats_config5.csv"),
                    ("profiles/r06_kernel_stats_config5_inflight1.csv", "profiles/r05_kernel_stats_config5_inflight1.csv"))}
# the term's opcode histogram priced per issue class (tools/k6_issue_ceiling.py): roofline.mix_ceiling
ISSUE_CEILING_FILE = "profiles/r06_k6_issue_ceiling.json"


def k6_pmc(config, frames_per_batch):
    """Counters of the K6 launches (locate: k6_locate, or seed + refinement + k6_anchor rounds; common pre-pass; full pass) of ONE
    batch from the committed PMC summary: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts
    half the bytes of wide reads on gfx950; separate passes), issued VALU wavefront-instructions, busy cycles.  A summary row is a
    mean per dispatch; `launches_per_batch` (round 5) says how many dispatches of it a batch makes.  None when no file matches."""
    import csv
    path = next((q for q in PMC_FILES.get((config, frames_per_batch), ()) if os.path.exists(os.path.join(ROOT, q))), None)
    if not path:
        return None
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, path))) if any(k in r["kernel"] for k in
            ("k6_grid_cost", "k6_triple_prepass", "k6_group_prepass", "k6_locate", "k6_anchor"))]
    if not 3 <= len(rows) <= 6:
        return None
    f = lambda r, k: float(r[k]) if r.get(k) not in (None, "") else 0.0
    per = lambda r: f(r, "launches_per_batch") or 1.0
    # the full pass = the k6_grid_cost instance that issues most (config 5's common pre-pass issues MORE than the full pass: round 5's
    # issue_utilisation_full_pass_alone of that config was the pre-pass's)
    full = max([r for r in rows if "k6_grid_cost" in r["kernel"]] or rows, key=lambda r: f(r, "SQ_INSTS_VALU"))
    return {"file": path,
            "traffic_bytes": int(sum(per(r) * (2.0 * f(r, "FETCH_SIZE") + f(r, "WRITE_SIZE")) * 1024.0 for r in rows)),
            "valu_wave_instr": sum(per(r) * f(r, "SQ_INSTS_VALU") for r in rows),
            "full_pass": {"valu_wave_instr": f(full, "SQ_INSTS_VALU"), "gui_active_cycles_per_xcd": f(full, "GRBM_GUI_ACTIVE") / 8.0,
                          "valu_busy_quad_cycles": f(full, "SQ_ACTIVE_INST_VALU")}}


def k6_rocprof(config, credited_lane_instr_per_batch):
    """roofline.frac recomputed from the committed rocprofv3 kernel-stats CSVs of this bench's command (tools/gpu_profile.sh): the
    summed duration of every K6 kernel (k6_locate / k6_grid_cost / k6_anchor / k6_group_prepass) divided by the batches the run made
    (= calls of k5w_walk_order, one per batch), pipelined and with one batch alone on the chip."""
    import csv
    out = {}
    for tag, paths in zip(("pipelined", "in_flight_1"), KSTATS_FILES.get(config, ())):
        path = next((q for q in paths if os.path.exists(os.path.join(ROOT, q))), None)
        if not path:
            continue
        full = os.path.join(ROOT, path)
        rows = list(csv.DictReader(open(full)))
        k6 = [r for r in rows if any(k in r.get("Name", "") for k in ("k6_grid_cost", "k6_locate", "k6_anchor", "k6_group_prepass", "k6_triple_prepass"))]
        batches = next((int(r["Calls"]) for r in rows if "k5w_walk_order" in r.get("Name", "")), 0)
        if not k6 or not batches:
            continue
        ms = sum(float(r["TotalDurationNs"]) for r in k6) / batches / 1e6
        rate = credited_lane_instr_per_batch / (ms * 1e-3) / 1e12
        out[tag] = {"file": path, "batches": batches, "k6_ms_per_batch": ms, "achieved_this_run": rate, "frac_this_run": rate / VALU_ISSUE_PEAK_T,
                    "kernels_us_per_batch": {r["Name"].replace("ilcc::", "").replace("void ", "")[:40]: round(float(r["TotalDurationNs"]) / batches / 1e3, 1) for r in k6},
                    "what": "THIS run's credited work over the committed profile's per-batch K6 time"}
    return out or None


def k6_mix_ceiling(evals_border, evals_interior, box_evals):
    """The issue ceiling of THIS run's credited instruction mix: every credited instruction priced at its class's issue cost (full
    rate: one wave64 instruction per SIMD every 2 cycles = 78.6 T lane-instr/s chip-wide; floor / fract / min / max / med3 / v_cmp /
    v_cndmask / packed ops: half of that) -- tools/k6_issue_ceiling.py's histogram of the term's gfx950 assembly.  Returns
    (ceiling in T lane-instr/s, the same priced at the rates the micro-benchmark measured, file) or None."""
    path = os.path.join(ROOT, ISSUE_CEILING_FILE)
    if not os.path.exists(path):
        return None
    c = json.load(open(path))["classes"]
    n = {"border": evals_border, "interior": evals_interior, "box": box_evals}
    instr = sum(n[k] * c[k]["valu_instr"] for k in n)
    if instr <= 0:
        return None
    units = sum(n[k] * c[k]["issue_units_nominal"] for k in n)
    t_meas = sum(n[k] * c[k]["valu_instr"] / c[k]["mix_ceiling_measured_T"] for k in n)
    return VALU_ISSUE_PEAK_T * instr / units, instr / t_meas, ISSUE_CEILING_FILE


def _gen_chunk(args):
    from lidar_camera_calibration_amd import synth
    kind, n, seed = args
    if kind == 5:
        return synth.make_batch(n, synth.hdl64(), synth.Board(9, 12, 0.10), seed=seed,
                                range_m=(2.0, 3.0), yaw_deg=25.0, pitch_deg=15.0, roll_deg=30.0)[:3]
    return synth.make_batch(n, seed=seed)[:3]


def generate(kind, n_frames, seed, workers):
    """n_frames seeded synthetic frames (frame f always comes from seed + f, however the work is split)."""
    chunk = max(1, min(32, n_frames // max(1, workers)))
    jobs = [(kind, min(chunk, n_frames - lo), seed + lo) for lo in range(0, n_frames, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.map(_gen_chunk, jobs)
    else:
        parts = [_gen_chunk(j) for j in jobs]
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))


def ensure_world(args):
    """`--gpus N` means N ranks, one per device -- or the run fails.  Under a launcher (WORLD_SIZE set: the driver's
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) the two numbers must agree.  A plain
    `python bench.py --gpus N` with N > 1 replaces itself with that launcher command (rendezvous on 127.0.0.1), so the
    flag can never silently run one rank and report n_gpus 1.  N ranks need N devices; ILCC_BENCH_SINGLE_DEVICE (the 1-GPU
    test hook that puts every rank on device 0) lifts that."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    single = bool(os.environ.get("ILCC_BENCH_SINGLE_DEVICE"))
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to report a line whose "
                             "n_gpus is not what was asked for" % (args.gpus, ws))
        return
    if args.gpus == 1:
        return
    import socket
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and not single:
        raise SystemExit("bench.py: --gpus %d needs %d HIP devices, this node shows %d (one rank per device; "
                         "ILCC_BENCH_SINGLE_DEVICE=1 is the 1-GPU test hook)" % (args.gpus, args.gpus, ndev))
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: re-executing as `%s`" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="2: BASELINE configs[1] frames (the headline); 5: configs[4], the dense-cloud fine-grid run")
    ap.add_argument("--frames-per-batch", type=int, default=0, help="default 1024 (config 2) / 128 (config 5)")
    ap.add_argument("--batches-per-step", type=int, default=0, help="batches per step and GPU: default 1")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="distinct batches resident per GPU that the steps rotate through (default 4: step s submits batch s mod 4, "
                         "no step repeats its predecessor's data); at least --batches-per-step")
    ap.add_argument("--no-config5-leg", action="store_true", help="skip the short config-5 leg of the default run (config5_value / config5_frac)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the H2D-inclusive / reference-mode legs (profiling runs)")
    ap.add_argument("--in-flight", type=int, default=4, help="batches in flight per GPU (1 = fully synchronous)")
    ap.add_argument("--no-alone-leg", action="store_true",
                    help="skip the in-flight-1 leg behind the timed region (profiling runs: a kernel trace then holds the pipelined "
                         "launches only); roofline.frac is then priced on the timed region's own event spans")
    ap.add_argument("--ref-n1", type=float, default=0.0,
                    help="frames/s of the N=1 run: rank 0 then prints weak_scaling_efficiency = value / (N x ref)")
    ap.add_argument("--no-noise-floor", action="store_true", help="skip the sensor-noise sweep (noise_floor_mm)")
    ap.add_argument("--solver", choices=("grid", "reference"), default="grid",
                    help="reference: the TIMED region runs ILCC_SOLVER_REFERENCE_LOCAL (K7a instead of K6 + K7r) -- profiling runs of "
                         "that mode (tools/gpu_profile_reference.sh); the headline and the driver's run are `grid`")
    args = ap.parse_args()
    ensure_world(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # batch size measured on one MI355X (1024 frames per step either way): 128: 249 k, 192: 266 k, 256: 275 k, 320: 273 k,
    # 384: 269 k, 512: 273 k, 1024: 269 k frames/s (config 5: 16: 2.87 k, 32: 3.30 k, 64: 3.54 k) -- 128 per-frame
    # workgroups fill only half of the 256 CUs.  Measured again once the grid search had its box pre-pass (round 3, the path no
    # longer VALU-saturated): 128 x 8: 405 k, 192 x 6: 487 k, 256 x 4: 542 k, 384 x 4: 559 k, 512 x 2: 571 k, 1024 x 1: 577 k
    # frames/s -- 512 x 2 (four batches = 2048 frames in flight) was the default until round 4's common pre-pass (k6_triple_prepass) halved
    # the grid search's share of the step; measured then: 256 x 4: 738 k, 384 x 2: 819 k, 512 x 2: 853 k, 768 x 2: 922 k, 1024 x 1: 923-928 k
    # frames/s resident (H2D-inclusive: 117.6-119.7 k at 512 x 2, 115.9 k at 1024 x 1: the link either way).  Config 5: 32 x 4: 46.4 k,
    # 64 x 2: 59.0 k (H2D-inclusive 23.8 k), 128 x 1: 61.9 k (22.6 k): 64 x 2 stays, its contract metric is the one still short of the link
    # Round 5 (k6_anchor's second round, K7r's silent points): config 5 64 x 2: 70.6 k resident / 24.4 k H2D-inclusive, 128 x 1: 78.6 k / 23.4 k
    # (link bound 26.9 k either way): 128 x 1 is the default now
    F = args.frames_per_batch or (1024 if args.config == 2 else 128)
    B = max(1, args.batches_per_step or 1)
    FS = F * B                                        # frames per step and GPU
    # distinct batches resident per GPU: step s submits the batches (s * B + b) mod NB -- round 5 walked the SAME 1024 frames every
    # step (the labelled points, walk layouts and bounds of step s were bit-identical to step s - 1's); four distinct batches = 1.9 GB
    # of config-2 input per GPU
    NB = max(B, args.distinct_batches or 4)
    NB = (NB + B - 1) // B * B                        # whole steps

    # synthetic inputs first (forked workers; nothing has touched the HIP runtime yet).  Weak scaling: every rank
    # owns its own FS frames (seeds disjoint per rank)
    from lidar_camera_calibration_amd import synth
    t_gen = time.perf_counter()
    cores = os.cpu_count() or 1
    # (ILCC_BENCH_GEN_WORKERS=1: no forked workers -- for runs under rocprofv3, whose tool library does not like forks)
    # the ranks of one node share the host's cores; LOCAL_WORLD_SIZE (torchrun) ranks generate at the same time.  Never fewer
    # than 2 workers per rank when the host has the cores for it (a 1-worker rank takes ~6 s for its 1024 frames)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    workers = int(os.environ.get("ILCC_BENCH_GEN_WORKERS", "0")) or max(2 if cores >= 2 * local_world else 1, min(16, cores // local_world))
    clouds, clicks, gts = generate(args.config, NB * F, 0xC0FFEE + rank * NB * F, workers)
    t_gen = time.perf_counter() - t_gen
    # the short config-5 leg of the default run (BASELINE's roofline run made driver-visible: roofline.config5_value / config5_frac)
    c5_inputs = None
    if world == 1 and args.config == 2 and not (args.no_extra_legs or args.no_config5_leg):
        c5_inputs = generate(5, 256, 0xC0FFEE, workers)
    # inputs of the sensor-noise sweep (noise_floor_mm): generated now, before anything touches the HIP runtime (forked workers)
    noise_inputs = None
    if world == 1 and args.config == 2 and not (args.no_extra_legs or args.no_noise_floor):
        noise_inputs = gen_noise_variants(min(256, FS), workers)

    import torch
    import torch.distributed as dist

    # test hooks for a 1-GPU box (never set by the driver): run the N > 1 code path with every rank on
    # device 0 and the records gathered over gloo instead of RCCL
    backend = os.environ.get("ILCC_BENCH_BACKEND", "nccl")
    if os.environ.get("ILCC_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    # ILCC_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, per-step gather, barriers) even with
    # one rank, so that the RCCL side can be exercised on a 1-GPU box
    dist_on = world > 1 or bool(os.environ.get("ILCC_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libilcc_hip has no CPU fallback")
    if local_world > torch.cuda.device_count() and not os.environ.get("ILCC_BENCH_SINGLE_DEVICE"):
        raise SystemExit("bench.py: %d ranks on this node but only %d HIP devices (one rank per device)"
                         % (local_world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # host-side placement: this rank's cores = the NUMA node of its GPU, BEFORE any pinned buffer is allocated
    # (first touch).  The ranks of one node share the host's memory bandwidth: 8(d)'s metric at N = 8 is that bandwidth's to lose
    from lidar_camera_calibration_amd.sharding import pin_rank_to_gpu_numa
    all_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pci = None
    placement = pin_rank_to_gpu_numa(pci, apply=not os.environ.get("ILCC_BENCH_NO_PIN"))
    print("bench placement: rank %d device %d pci %s numa_node %d cpus %d pinned %s"
          % (rank, local_rank, placement["pci"], placement["numa_node"], placement["cpus"], placement["pinned"]), file=sys.stderr, flush=True)
    rec_dev = dev if backend == "nccl" else torch.device("cpu")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                **({"device_id": dev} if backend == "nccl" else {}))

    from lidar_camera_calibration_amd import LidarCornersBatch
    from lidar_camera_calibration_amd import _native as N
    from lidar_camera_calibration_amd.sharding import gather_records, record_floats, verify_records

    if args.config == 5:
        board, n_points = synth.Board(9, 12, 0.10), synth.hdl64().n_points
        params = config5_params(N)
    else:
        board, n_points = synth.Board(), synth.vlp16().n_points
        params = N.default_params()            # ILCC_SOLVER_GRID: 61 x 40 x 40 candidates x 2 phases
    if args.solver == "reference":
        params.solver = N.SOLVER_REFERENCE_LOCAL
    bytes_per_frame = 16 * n_points + 12 * board.n_corners + 64          # 461 284 B for config 2
    clouds = clouds.reshape(NB, F, n_points, 4)
    clicks = clicks.reshape(NB, F, 3)
    gts = gts.reshape(NB, F, *gts.shape[1:])
    d_clouds = [torch.from_numpy(clouds[b]).to(dev) for b in range(NB)]
    d_clicks = [torch.from_numpy(clicks[b]).to(dev) for b in range(NB)]
    est = LidarCornersBatch(F, n_points, params, device=local_rank)
    # result traffic: the 500-byte gather records only (ILCC_RESULTS_COMPACT); the full 3.3 KB records stay in HBM
    est.set_result_mode(N.RESULTS_COMPACT)
    # ilcc_reserve: the sensor is known, so the handle's on-chip capacities are set up front and the warm-up batches take the
    # same kernels as the timed ones (a fresh handle grows them after its first batch: INTEGRATION.md, "Sizing")
    # (labelled points per frame: the closest boards of the sensor model hold 1 784 (config 2) and 6 170 (config 5): the next
    # multiple of 256 -- at 1 792 seven K6 workgroups fit a CU's LDS, at 2 048 six: 1 228 k vs 1 255 k frames/s)
    res_lab = int(os.environ.get("ILCC_BENCH_RESERVE_LABELLED", "0")) or (6400 if args.config == 5 else 1792)   # (env: LDS staging A/B)
    est.reserve(res_lab, 20000 if args.config == 5 else 2560)
    n_cand = params.n_th * params.n_ty * params.n_tz * 2
    depth = max(1, min(args.in_flight, int(os.environ.get("ILCC_BENCH_MAX_DEPTH", "4"))))

    # the path's only collective ships fixed-size corner records: one step's FS records per rank, packed ON the GPU
    # (ilcc_wait_records_device) into one of three rotating device buffers and handed to RCCL from there on a side
    # stream: no host round trip, nothing on the null stream
    rec_w = record_floats(board.n_corners)
    rec_bufs = [torch.zeros((FS, rec_w), dtype=torch.float32, device=dev) for _ in range(3)] if dist_on else []
    side = torch.cuda.Stream(device=dev) if dist_on else None
    pending = []          # the previous step's gather, still in flight (one collective per step)

    def issue_gather(step):
        buf = rec_bufs[step % 3]
        with torch.cuda.stream(side):           # never the null stream: it would serialise with the batches in flight
            rec = buf if rec_dev.type == "cuda" else buf.cpu()      # (gloo test hook: CPU tensors)
            while pending:                       # at most one collective outstanding
                w, _ = pending.pop(0)
                if w is not None:
                    w.wait()
            pending.append(gather_records(rec, world, rank, async_op=True, force_collective=dist_on))

    def drain_gather():
        gathered = None
        while pending:
            w, bufs = pending.pop(0)
            with torch.cuda.stream(side):
                if w is not None:
                    w.wait()
                gathered = torch.cat(bufs, 0) if bufs is not None else None
            side.synchronize()
        return gathered

    def run(n_steps, clouds_ptrs, step_times=None, keep=None, host_clicks=None, depth_override=None, last_on=None):
        """n_steps steps of B batches each, up to `depth` batches in flight (the library's submit/wait pipeline: the
        latency-bound stages of one batch overlap with the grid search of another).  Step s submits the distinct batches
        (rot + s * B + b) mod NB.  keep: list that receives the compact records of the LAST step, batch by batch.  last_on: the
        distinct batch the last step's first batch must be (the legs that compare per-frame results end on the same frames).
        host_clicks: the inputs are pinned HOST buffers and every batch's H2D copy is enqueued on the batch's own stream
        (ilcc_submit_batch)."""
        inflight = []
        nb = len(clouds_ptrs)
        rot = 0 if last_on is None else (last_on - (n_steps - 1) * B) % nb

        def finish():
            ticket, s, b = inflight.pop(0)
            if dist_on:
                buf = rec_bufs[s % 3]
                est.wait(ticket, buf.data_ptr() + 4 * rec_w * F * b, board.n_corners, tag_base=(rank * B + b) * F,
                         want_results=False)
                res = None
                if keep is not None and s == n_steps - 1:      # (the same records, read back from the step's device buffer)
                    res = buf[F * b:F * (b + 1)].cpu().numpy()
                if b == B - 1:
                    issue_gather(s)
            else:
                res = est.wait_compact(ticket)
            if keep is not None and s == n_steps - 1:
                keep.append(res)
            if step_times is not None and b == B - 1:
                step_times.append(time.perf_counter())

        for s in range(n_steps):
            for b in range(B):
                k = (rot + s * B + b) % nb
                if host_clicks is not None:
                    inflight.append((est.submit_host(clouds_ptrs[k], F, n_points, host_clicks[k]), s, b))
                else:
                    inflight.append((est.submit_device(clouds_ptrs[k], F, n_points, d_clicks[k].data_ptr()), s, b))
                if len(inflight) == (depth_override or depth):
                    finish()
        while inflight:
            finish()
        return drain_gather() if dist_on else None

    dptrs = [t.data_ptr() for t in d_clouds]

    def warm(ptrs, w_steps, host_clicks=None):
        """W steps, then blocks of 5 steps until the median step time of a block is within 2 % of the previous block's
        median and >= 0.3 s have passed (cap: 3 s).  Returns (extra steps, seconds)."""
        t0 = time.perf_counter()
        run(max(1, w_steps), ptrs, host_clicks=host_clicks)
        extra, prev = 0, None
        while True:
            ts = [time.perf_counter()]
            run(5, ptrs, step_times=ts, host_clicks=host_clicks)
            extra += 5
            med = float(np.median(np.diff(ts)))
            now = time.perf_counter() - t0
            done = (prev is not None and now >= 0.3 and abs(med - prev) <= 0.02 * max(med, prev)) or now >= 3.0
            prev = med
            if dist_on:   # every rank must issue the same number of steps (= gathers): stop only when all are steady
                flag = torch.tensor([0.0 if done else 1.0], dtype=torch.float32, device=rec_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                done = float(flag.item()) == 0.0
            if done:
                return extra, time.perf_counter() - t0

    if dist_on:
        # preflight: what the collective really spans (a SCALE log then proves N ranks on N devices)
        info = torch.tensor([float(rank), float(torch.cuda.current_device()), float(FS * rec_w * 4)], dtype=torch.float64, device=rec_dev)
        got = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(got, info)
        if rank == 0:
            print("bench preflight: backend %s, world %d; ranks %s on devices %s; gather of %d bytes per rank and step"
                  % (dist.get_backend(), dist.get_world_size(), [int(g[0]) for g in got], [int(g[1]) for g in got], int(got[0][2])),
                  file=sys.stderr, flush=True)
    extra_warm, warm_s = warm(dptrs, args.warmup)
    est.reset_timing()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    last = []
    t0 = time.perf_counter()
    gathered = run(args.steps, dptrs, keep=last)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    last0 = ((args.steps - 1) * B) % NB               # the distinct batches the last step walked: last0 ... last0 + B - 1
    gts = gts[last0:last0 + B].reshape(FS, *gts.shape[2:])
    clouds_last, clicks_last = clouds[last0:last0 + B], clicks[last0:last0 + B]   # (what the per-frame legs below re-run and compare)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rec_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            # the gather delivered EVERY rank's records of the last step, each rank's block in its place, intact:
            # tag = (rank * B + b) * F + f and a content check word per record (sharding.verify_records)
            assert gathered is not None and tuple(gathered.shape) == (world * FS, rec_w)
            verify_records(gathered.cpu().numpy(), np.arange(world * FS))
    tm = est.timing()

    # ONE batch at a time (in flight 1, nothing else on the chip): the only exclusive launch durations.  roofline.frac is priced
    # on these; the timed region's own event spans (batches overlapping) are reported beside them
    tm_alone = tm
    if not args.no_alone_leg and depth > 1:
        est.reset_timing()
        run(max(3, min(10, args.steps)), dptrs, depth_override=1)
        torch.cuda.synchronize()
        tm_alone = est.timing()

    # the same pipeline with every batch starting in pinned HOST memory (SURVEY.md 8d counts that copy): every rank, its own link
    h2d = None
    if not args.no_extra_legs:
        h2d = h2d_inclusive_leg(torch, dist if dist_on else None, rec_dev, est, clouds, d_clicks, F, NB, B, FS, n_points, world,
                                args.steps, run, warm)
    gen_per_rank = [round(t_gen, 2)]
    if dist_on:
        g = [torch.zeros(1, dtype=torch.float64, device=rec_dev) for _ in range(world)]
        dist.all_gather(g, torch.tensor([t_gen], dtype=torch.float64, device=rec_dev))
        gen_per_rank = [round(float(x.item()), 2) for x in g]

    # accuracy of the last step on this rank (FS frames)
    res = [Rec(r, board.n_corners) for batch in last for r in batch]
    ok = [f for f in range(FS) if res[f].status == N.OK]
    amb = [f for f in range(FS) if res[f].status == N.AMBIGUOUS]
    err_ok = np.array([synth.corner_error(res[f].corners_array(), gts[f], board) for f in ok])
    err_amb = np.array([synth.corner_error(res[f].corners_array(), gts[f], board) for f in amb])
    m_lab = float(np.mean([res[f].n_black + res[f].n_white for f in ok + amb])) if ok or amb else 0.0

    if rank == 0:
        total_frames = world * FS * args.steps
        fps = total_frames / elapsed
        launches = max(1, tm.grid_cost_launches)
        la = max(1, tm_alone.grid_cost_launches)
        # The K6 stage of one batch, launch by launch (HIP events on the batch's own stream: seed, refinement, anchor, common
        # pre-pass, full pass).  k6_ms_alone: one batch in flight, nothing else on the chip -- exclusive durations, what a
        # rocprofv3 kernel trace of `--in-flight 1` adds up to.  k6_ms_pipelined: the same event spans inside the timed region,
        # where up to four batches share the chip (a span then also holds what a launch waited for a CU).
        k6_ms_alone = tm_alone.grid_cost_kernel_ms_sum / la
        k6_ms = tm.grid_cost_kernel_ms_sum / launches
        k6_span_ms = tm.grid_cost_ms_sum / launches
        k6_bytes = bytes_per_frame * F
        evals_per_launch = tm.grid_cost_evals_sum / launches            # executed (after branch-and-bound cuts)
        evals_nominal = tm.grid_cost_evals_nominal_sum / launches       # what a cut-free exhaustive pass needs
        evals_interior = tm.grid_cost_evals_interior_sum / launches
        valu_ops_per_eval = (K6_VALU_OPS_INTERIOR * evals_interior + K6_VALU_OPS_BORDER * (evals_per_launch - evals_interior)) \
            / max(1.0, evals_per_launch)
        box_evals = tm.grid_cost_box_evals_sum / launches               # (point, tile) evaluations of the box pre-passes
        credited_lane_instr = evals_per_launch * valu_ops_per_eval + box_evals * K6_VALU_OPS_BOX
        valu_rate = credited_lane_instr / (max(k6_ms_alone, 1e-9) * 1e-3) / 1e12
        valu_rate_pipe = credited_lane_instr / (max(k6_ms, 1e-9) * 1e-3) / 1e12
        pmc = k6_pmc(args.config, F)
        credited_wave_instr = credited_lane_instr / 64.0
        rocprof = k6_rocprof(args.config, credited_lane_instr)
        mix = k6_mix_ceiling(evals_per_launch - evals_interior, evals_interior, box_evals)
        # K1's count pass: the path's HBM-bound kernel (reads every input point exactly once)
        k1_ms_alone = tm_alone.roi_count_ms_sum / max(1, tm_alone.batches)
        k1_GBps = 16.0 * n_points * F / (k1_ms_alone * 1e-3) / 1e9 if k1_ms_alone > 0 else None
        step_ms_alone = tm_alone.stage_ms_sum[6] / max(1, tm_alone.batches)
        low = [bool(res[f].flags & N.FLAG_LOW_COVERAGE) for f in range(FS)]
        acc = [f for f in ok if not low[f]]
        err_acc = np.array([synth.corner_error(res[f].corners_array(), gts[f], board) for f in acc])
        out = {
            "metric": "chessboard-corner frames/sec + max corner error (mm), VLP-16 cloud",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                # (the driver keeps the first 120 characters of a string: self-contained within that)
                "workload": ("configs[1] VLP-16 cloud 28800 pts, 7x5 board @0.15 m; step = %d frames/GPU (configs[3] shard) in %d batch(es) of %d"
                             % (FS, B, F))
                if args.config == 2 else
                ("configs[4] 64-ring cloud 131072 pts, 11x8 board @0.10 m, 129^3 x 2 grid; step = %d frames/GPU in %d batches of %d"
                 % (FS, B, F)),
                "workload_detail": "%d distinct batches resident per GPU (%.2f GB), step s walks batch s mod %d: no step repeats its predecessor's "
                                   "data; %.0f MB of input per GPU and step%s; synthetic frames, one random board pose each"
                                   % (NB, NB * F * n_points * 16 / 1e9, NB // B, FS * n_points * 16 / 1e6,
                                      " (> the 256 MB Infinity Cache)" if FS * n_points * 16 > 256e6 else ""),
                "distinct_batches": NB,
                "frames_per_step_per_gpu": FS,
                "frames_per_batch": F,
                "points_per_frame": n_points,
                "batches_in_flight": depth,
                "solver": "ILCC_SOLVER_GRID: exhaustive %dx%dx%d x 2 grid (%d candidates) + 27-point pattern search (step/%d) + basin check"
                          % (params.n_th, params.n_ty, params.n_tz, n_cand, params.refine_div),
                "parallelism": "frames sharded across %d GPU(s), one RCCL gather of corner records per step" % world
                               if world > 1 else "1 GPU",
            },
            "value_resident": fps,
            "value_h2d_inclusive": None,
            "link_GBps_achieved": None,
            "link_bound_frames_per_s": None,
            "link_frac": None,
            "value_note": "`value` = inputs resident in HBM when the timed region starts: the bench contract of this build says "
                          "so in as many words (a PCIe-inclusive rate `is never value`).  SURVEY.md 8(d) defines the metric with the "
                          "host->device copy of XYZI inside: that figure is `value_h2d_inclusive` (same pipeline, every batch starts "
                          "in pinned host memory), with the link's own rate beside it -- at one GPU config 2 is PCIe-bound "
                          "(`link_frac` of what the link alone delivers), config 5 is not",
            "warmup_extra_steps_until_steady": extra_warm,
            "warmup_s": round(warm_s, 3),
            "input_generation_s": round(t_gen, 2),
            "input_generation_s_per_rank": gen_per_rank,
            "input_generation_workers": workers,
            "host_placement": placement,
            "max_corner_error_mm_vs_ground_truth": 1e3 * float(err_ok.max()) if len(err_ok) else None,
            "median_corner_error_mm_vs_ground_truth": 1e3 * float(np.median(err_ok)) if len(err_ok) else None,
            "p99_corner_error_mm_vs_ground_truth": 1e3 * float(np.percentile(err_ok, 99)) if len(err_ok) else None,
            "frames_ok": "%d/%d" % (len(ok), FS),
            "frames_flagged_ambiguous": len(amb),
            "max_corner_error_mm_incl_ambiguous": 1e3 * float(max(err_ok.max(initial=0.0), err_amb.max(initial=0.0))),
            "accept_rule": {"rule": "status == ILCC_OK and not (flags & ILCC_FLAG_LOW_COVERAGE)  [include/ilcc_hip.h, 'accepting a frame']",
                            "frames_accepted": "%d/%d" % (len(acc), FS),
                            "max_corner_error_mm_accepted": 1e3 * float(err_acc.max()) if len(err_acc) else None,
                            "median_corner_error_mm_accepted": 1e3 * float(np.median(err_acc)) if len(err_acc) else None,
                            "frames_ok_but_low_coverage": int(sum(1 for f in ok if low[f]))},
            "labelled_points_per_frame": m_lab,
            "labelled_points_max": int(max([res[f].n_black + res[f].n_white for f in ok + amb] or [0])),
            "stage_ms_last_batch_overlapped": {k: round(getattr(tm, k), 4) for k in
                                               ("roi_crop", "cluster", "ransac_plane", "plane_frame_hist", "grid_cost",
                                                "refine_corners", "total")},
            "roofline": {
                # the first 24 scalars are what the driver's record keeps (BENCH_rNN.json parsed.roofline): the contract figures first
                "kernel": "k6_grid_cost (launches per batch: k6_locate [or seed, refinement, anchor], k6_group_prepass, full pass)",
                "bound": "valu",
                "achieved": valu_rate,
                "peak": VALU_ISSUE_PEAK_T,
                "unit": "T lane-instr/s",
                "frac": valu_rate / VALU_ISSUE_PEAK_T,
                "frac_what": "credited lane-instr of one batch's K6 launches / k6_ms_alone (HIP events, ONE batch in flight) / peak"
                             if tm_alone is not tm or depth == 1 else
                             "credited lane-instr of one batch's K6 launches / the timed region's event spans (--no-alone-leg) / peak",
                "k6_ms_alone": k6_ms_alone,
                "frac_rocprof_alone": rocprof["in_flight_1"]["frac_this_run"] if rocprof and "in_flight_1" in rocprof else None,
                "frac_whole_step": credited_lane_instr * B / (elapsed / args.steps) / 1e12 / VALU_ISSUE_PEAK_T,
                "traffic": pmc["traffic_bytes"] if pmc else None,
                # HBM (north_star's figure): the algorithmic bytes of SURVEY.md 8(d) over the whole step, and K1's count pass alone
                "hbm_frac": fps / max(1, world) * bytes_per_frame / 1e9 / HBM_PEAK_GBPS,
                "k1_hbm_frac": k1_GBps / HBM_PEAK_GBPS if k1_GBps else None,
                # SURVEY.md 8(d)'s metric as written (H2D copy inside), filled in below when that leg runs
                "h2d_inclusive_frames_per_s": None,
                "link_frac": None,
                "link_bound_frames_per_s": None,
                "uncredited_share": 1.0 - credited_wave_instr / pmc["valu_wave_instr"] if pmc else None,
                "wave_instr_per_frame": pmc["valu_wave_instr"] / F if pmc else None,
                # the issue ceiling of the credited instruction mix (half-rate floor / fract / min / max / cmp / cndmask priced at 2 units)
                "mix_ceiling": mix[0] if mix else None,
                "frac_of_mix_ceiling": valu_rate / mix[0] if mix else None,
                # the mode with reference-path parity and BASELINE's roofline run (config 5), filled in below when those legs run
                "reference_mode_frames_per_s": None,
                "reference_mode_link_frac": None,
                "config5_frames_per_s": None,
                "config5_frac": None,
                # ---- everything below is beyond the 24 scalars the driver keeps
                "alone_leg": tm_alone is not tm or depth == 1,
                "mix_ceiling_at_measured_rates": mix[1] if mix else None,
                "frac_of_mix_ceiling_at_measured_rates": valu_rate / mix[1] if mix else None,
                "mix_ceiling_file": mix[2] if mix else None,
                "k6_ms_pipelined": k6_ms,
                "k6_locate_ms_alone": tm_alone.grid_cost_locate_ms_sum / la,
                "k6_prepass_ms_alone": tm_alone.grid_cost_prepass_ms_sum / la,
                "k6_full_pass_ms_alone": tm_alone.grid_cost_full_ms_sum / la,
                "k6_locate_ms_pipelined": tm.grid_cost_locate_ms_sum / launches,
                "k6_prepass_ms_pipelined": tm.grid_cost_prepass_ms_sum / launches,
                "k6_full_pass_ms_pipelined": tm.grid_cost_full_ms_sum / launches,
                "frac_pipelined": valu_rate_pipe / VALU_ISSUE_PEAK_T,
                "frac_of_stage_span": credited_lane_instr / (k6_span_ms * 1e-3) / 1e12 / VALU_ISSUE_PEAK_T if k6_span_ms > 0 else None,
                "frac_rocprof_pipelined": rocprof["pipelined"]["frac_this_run"] if rocprof and "pipelined" in rocprof else None,
                "k6_ms_rocprof_alone": rocprof["in_flight_1"]["k6_ms_per_batch"] if rocprof and "in_flight_1" in rocprof else None,
                "k6_ms_rocprof_pipelined": rocprof["pipelined"]["k6_ms_per_batch"] if rocprof and "pipelined" in rocprof else None,
                "rocprof_files": ", ".join(v["file"] for v in rocprof.values()) if rocprof else None,
                "credited_lane_instr_per_batch": credited_lane_instr,
                "issued_wave_instr_per_batch": pmc["valu_wave_instr"] if pmc else None,
                # issued VALU wave-instructions of the full pass / what the chip issues in its duration at one wave64 instruction per
                # SIMD every TWO cycles (1024 SIMDs; MI355X_MICROARCH.md) -- round 5's DESIGN text priced this against a 4-cycle ceiling
                "issue_utilisation_full_pass_alone": (2.0 * pmc["full_pass"]["valu_wave_instr"] /
                                                      max(1.0, 1024.0 * pmc["full_pass"]["gui_active_cycles_per_xcd"])) if pmc else None,
                "pmc_file": pmc["file"] if pmc else None,
                "evals_executed_per_batch": evals_per_launch,
                "evals_nominal_per_batch": evals_nominal,
                "executed_fraction": evals_per_launch / evals_nominal if evals_nominal else None,
                "valu_instr_per_eval": valu_ops_per_eval,
                "box_evals_per_batch": box_evals,
                "valu_instr_per_box_eval": K6_VALU_OPS_BOX,
                "batches_timed": int(tm.grid_cost_launches),
                "batches_timed_alone": int(tm_alone.grid_cost_launches),
                "stage_span_ms": k6_span_ms,
                "walk_order_k5w_ms": tm.walk_order_ms_sum / launches,
                "step_ms_alone": step_ms_alone,
                "hbm_algorithmic_bytes_per_batch": k6_bytes,
                "hbm_GBps_whole_path": fps / max(1, world) * bytes_per_frame / 1e9,
                "k1_count_ms_alone": k1_ms_alone,
                "k1_hbm_GBps": k1_GBps,
                "hbm_peak_GBps": HBM_PEAK_GBPS,
                "link_GBps_achieved": None,
                "link_GBps_raw_hipMemcpy": None,
                "note": "VALU-bound by construction (points in LDS, no MFMA).  credited = executed evaluations x %g / %g VALU instr "
                        "(border / interior class) + box evaluations x %g (tools/k6_isa_count.sh: the terms only).  frac FALLS when "
                        "pruning removes credited work: track wave_instr_per_frame and k6_ms_alone instead.  mix_ceiling: the same credited "
                        "instructions priced per issue class (tools/k6_issue_ceiling.py)" % (
                            K6_VALU_OPS_BORDER, K6_VALU_OPS_INTERIOR, K6_VALU_OPS_BOX),
                "rocprof": rocprof,
                "issued_vs_credited": {
                    "issued_valu_wave_instr_per_launch": pmc["valu_wave_instr"],
                    "credited_valu_wave_instr_per_launch": credited_wave_instr,
                    "what": "issued = SQ_INSTS_VALU of the K6 launches of one batch (PMC file, batch alone on the chip); credited = this "
                            "run's executed evaluations x the term's ISA count / 64 lanes.  The difference is bound tests, tile "
                            "prologues, staging, address arithmetic, idle lanes of tail blocks",
                    "full_pass_alone": dict(pmc["full_pass"],
                                            ms_at_2p4GHz=pmc["full_pass"]["gui_active_cycles_per_xcd"] / 2.4e6,
                                            valu_busy_share=pmc["full_pass"]["valu_busy_quad_cycles"] / max(1.0, pmc["full_pass"]["gui_active_cycles_per_xcd"] * 256.0)),
                } if pmc else None,
            },
        }
        if h2d:
            out["value_h2d_inclusive"] = h2d["value"]
            out["link_GBps_achieved"] = h2d["link_GBps_achieved_per_gpu"]
            out["link_bound_frames_per_s"] = h2d["link_bound_frames_per_s"]
            out["link_frac"] = h2d["value"] / h2d["link_bound_frames_per_s"]
            out["pcie_inclusive"] = {k: v for k, v in h2d.items() if k not in ("hptrs", "hclicks", "keepalive")}
            out["roofline"].update({"h2d_inclusive_frames_per_s": h2d["value"], "link_GBps_achieved": h2d["link_GBps_achieved_per_gpu"],
                                    "link_GBps_raw_hipMemcpy": h2d["link_GBps_raw_hipMemcpy"],
                                    "link_bound_frames_per_s": h2d["link_bound_frames_per_s"],
                                    "link_frac": h2d["value"] / h2d["link_bound_frames_per_s"]})
        if args.ref_n1 > 0:
            out["weak_scaling_efficiency"] = fps / (world * args.ref_n1)
            out["weak_scaling_reference_n1"] = args.ref_n1
        if world == 1 and not args.no_extra_legs:
            out["grid_vs_reference_path_mm"], gpu_ref = reference_mode_leg(N, est, params, dptrs, d_clicks, res, F, B, FS, run,
                                                                            warm, args.steps, synth, gts, board,
                                                                            h2d["hptrs"] if h2d else None, h2d["hclicks"] if h2d else None,
                                                                            last_on=last0)
            out["reference_local_mode"] = out["grid_vs_reference_path_mm"].pop("reference_local_mode")
            rl = out["reference_local_mode"]
            out["roofline"]["reference_mode_frames_per_s"] = rl["value"]
            if h2d and rl.get("value_h2d_inclusive"):
                out["roofline"]["reference_mode_link_frac"] = rl["value_h2d_inclusive"] / h2d["link_bound_frames_per_s"]
                rl["link_frac"] = out["roofline"]["reference_mode_link_frac"]
            if c5_inputs is not None:
                out["config5"] = config5_leg(N, synth, c5_inputs, local_rank)
                out["roofline"]["config5_frames_per_s"] = out["config5"]["value"]
                out["roofline"]["config5_frac"] = out["config5"]["frac"]
            out["single_frame_latency_ms"] = single_frame_latency(N, params, clouds_last, clicks_last, n_points, local_rank)
            if args.config == 2:
                out["half_resolution_grid_variant"] = half_grid_leg(N, est, params, dptrs, FS, run, warm, args.steps, synth, gts, board,
                                                                    last_on=last0)
            if args.config == 2:
                out["online_caller"] = online_caller_leg(N, params, clouds_last.reshape(FS, n_points, 4), gts, n_points, local_rank)
            if noise_inputs is not None:
                out["noise_floor_mm"] = noise_floor_leg(N, params, synth, board, n_points, local_rank, noise_inputs)
            if not args.no_cpu_baseline and args.config == 2:
                if all_cpus and hasattr(os, "sched_setaffinity"):
                    os.sched_setaffinity(0, all_cpus)      # the CPU baseline's all-cores leg uses every core of the box again
                out["cpu_baseline"] = cpu_baseline(clouds_last.reshape(FS, n_points, 4), clicks_last.reshape(FS, 3), gts, board,
                                                   args.cpu_seconds, gpu_ref)
                cb = out["cpu_baseline"]
                dv = cb.get("gpu_vs_cpu_corner_deviation_mm") or {}
                ac = cb["all_cores"]
                cb.update({"all_cores_value": ac["value"], "all_cores": ac["cores"], "all_cores_sample": ac["sample"],
                           "gpu_vs_cpu_max_dev_mm": dv.get("max"), "frames_compared": dv.get("frames_compared"),
                           "status_agree": dv.get("status_agree"),
                           "gpu_vs_cpu_what": "GPU ILCC_SOLVER_REFERENCE_LOCAL vs this CPU path, same frames, max |dx| over corners",
                           "gpu_resident_over_cpu_1core": fps / cb["value"],
                           "gpu_h2d_inclusive_over_cpu_1core": (h2d["value"] / cb["value"]) if h2d else None,
                           "gpu_h2d_inclusive_over_cpu_all_cores": (h2d["value"] / ac["value"]) if h2d else None})
                # the mode that meets north_star's "within 1e-3 m of the reference CPU path": say so next to its rates
                out["reference_local_mode"]["gpu_vs_cpu_port_corner_deviation_mm"] = out["cpu_baseline"]["gpu_vs_cpu_corner_deviation_mm"]
        print(json.dumps(out), flush=True)
    est.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def reference_mode_leg(N, est, params, dptrs, d_clicks, res_grid, F, B, FS, run, warm, steps, synth, gts, board,
                       hptrs=None, hclicks=None, last_on=None):
    """The reference's own trajectory on the GPU (ILCC_SOLVER_REFERENCE_LOCAL: two Ceres-style solves from zero per
    colour phase, LidarCornersEst.cpp:398-409), through the SAME pipelined harness: its frames/s, and how far the
    headline GRID mode's corners are from it frame by frame (the reference's 50+50 iterations do not converge, so the
    two modes legitimately differ; BASELINE's 1e-3 m bar vs the CPU path is met by THIS mode, see cpu_baseline)."""
    import ctypes as C
    import torch
    p_ref = N.Params()
    C.memmove(C.byref(p_ref), C.byref(params), C.sizeof(N.Params))
    p_ref.solver = N.SOLVER_REFERENCE_LOCAL
    est.set_params(p_ref)
    warm(dptrs, 2)
    torch.cuda.synchronize()
    last = []
    t0 = time.perf_counter()
    n = max(3, steps // 2)
    run(n, dptrs, keep=last, last_on=last_on)      # (ends on the distinct batch the headline's last step walked: same frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res_ref = [Rec(r, board.n_corners) for batch in last for r in batch]
    dt_h2d = None
    n_h2d = None
    if hptrs is not None:            # the same mode with every batch crossing PCIe inside the timed region: the headline's H2D leg
        warm(hptrs, 2, host_clicks=hclicks)       # exactly (same steady-state rule, same number of steps: a step is 8 ms of link time, and
        n_h2d = max(64, steps)                     # round 5's 2 + 10 steps measured the pipeline's fill and drain into it: 0.79 of the link)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n_h2d, hptrs, host_clicks=hclicks)
        torch.cuda.synchronize()
        dt_h2d = time.perf_counter() - t0
    est.set_params(params)
    both = [f for f in range(FS) if res_ref[f].status == N.OK and res_grid[f].status in (N.OK, N.AMBIGUOUS)]
    dev = np.array([synth.corner_error(res_grid[f].corners_array(), res_ref[f].corners_array().astype(np.float64), board)
                    for f in both])
    both_ok = np.array([res_grid[f].status == N.OK for f in both])
    e_ref = np.array([synth.corner_error(res_ref[f].corners_array(), gts[f], board) for f in range(FS) if res_ref[f].status == N.OK])
    gpu_ref = [res_ref[f].corners_array() if res_ref[f].status == N.OK else None for f in range(FS)]
    blk = {
        "what": "max corner distance per frame, GPU ILCC_SOLVER_GRID vs GPU ILCC_SOLVER_REFERENCE_LOCAL, same frames "
                "(lattice symmetries folded)",
        "frames_compared": int(len(dev)),
        "median": 1e3 * float(np.median(dev)) if len(dev) else None,
        "p90": 1e3 * float(np.percentile(dev, 90)) if len(dev) else None,
        "max": 1e3 * float(dev.max()) if len(dev) else None,
        "n_above_1mm": int((dev > 1e-3).sum()),
        "max_excluding_frames_flagged_ambiguous": 1e3 * float(dev[both_ok].max()) if both_ok.any() else None,
        "reference_local_mode": {
            "value": FS * n / dt, "unit": "frames/s", "steps": n, "ms_per_step": 1e3 * dt / n,
            "value_h2d_inclusive": FS * n_h2d / dt_h2d if dt_h2d else None, "steps_h2d_inclusive": n_h2d,
            "what": "ILCC_SOLVER_REFERENCE_LOCAL: the reference's own trajectory (2 phases x (pass A + pass B) trust-region solves "
                    "from zero) on the GPU -- the mode whose corners match the reference CPU path (gpu_vs_cpu_port_corner_deviation_mm: "
                    "north_star's 1e-3 m clause); the headline ILCC_SOLVER_GRID mode finds a lower cost and lands elsewhere "
                    "(grid_vs_reference_path_mm)",
            "median_corner_error_mm_vs_ground_truth": 1e3 * float(np.median(e_ref)) if len(e_ref) else None,
            "max_corner_error_mm_vs_ground_truth": 1e3 * float(e_ref.max()) if len(e_ref) else None,
            "frames_ok": "%d/%d" % (len(e_ref), FS),
            "note": "same pipelined harness and inputs as `value`; this mode never runs K6, its K7a walks the "
                    "reference's trust-region iterations (2 phases x (50 + 50))",
        },
    }
    return blk, gpu_ref


def config5_params(N):
    """BASELINE configs[4]: 11 x 8 corners @0.10 m, ty, tz in [-g, g] step g/64 (129 x 129), theta in [-16, 16] deg step
    0.25 deg (129): 2 146 689 candidates x 2 phases."""
    p = N.default_params()
    p.board_w, p.board_h, p.grid_length = 9, 12, 0.10
    p.n_th = p.n_ty = p.n_tz = 129
    p.th_min, p.th_step = -16.0 * np.pi / 180.0, 0.25 * np.pi / 180.0
    p.ty_min = p.tz_min = -0.10
    p.ty_step = p.tz_step = 0.10 / 64
    return p


def k6_credit(tm):
    """Credited lane-instructions of ONE batch's K6 launches from the library's counters (executed evaluations by class, box evaluations)."""
    la = max(1, tm.grid_cost_launches)
    evals, interior, box = tm.grid_cost_evals_sum / la, tm.grid_cost_evals_interior_sum / la, tm.grid_cost_box_evals_sum / la
    return K6_VALU_OPS_INTERIOR * interior + K6_VALU_OPS_BORDER * (evals - interior) + K6_VALU_OPS_BOX * box


def config5_leg(N, synth, inputs, device, steps=10):
    """BASELINE's roofline run (configs[4]: 64-ring 131 072-point clouds, 11 x 8 board @0.10 m, 129^3 x 2 grid) as a SHORT leg of the
    default run, so that the driver's record carries it (roofline.config5_frames_per_s / config5_frac): two distinct 128-frame batches
    resident, four in flight, `steps` steps timed after a warm-up; then one batch at a time for the K6 launches' exclusive durations.
    `python bench.py --config 5` is the full version (H2D-inclusive leg, reference mode, rocprof cross-checks)."""
    import torch
    from lidar_camera_calibration_amd import LidarCornersBatch
    clouds, clicks, gts = inputs
    lidar, board = synth.hdl64(), synth.Board(9, 12, 0.10)
    F = 128
    nb = len(clouds) // F
    est = LidarCornersBatch(F, lidar.n_points, config5_params(N), device=device)
    est.set_result_mode(N.RESULTS_COMPACT)
    est.reserve(6400, 20000)
    d_c = [torch.from_numpy(clouds[b * F:(b + 1) * F]).cuda() for b in range(nb)]
    d_k = [torch.from_numpy(clicks[b * F:(b + 1) * F]).cuda() for b in range(nb)]

    def run(n, depth):
        inflight, last = [], None
        for s in range(n):
            inflight.append(est.submit_device(d_c[s % nb].data_ptr(), F, lidar.n_points, d_k[s % nb].data_ptr()))
            if len(inflight) == depth:
                last = est.wait_compact(inflight.pop(0))
        while inflight:
            last = est.wait_compact(inflight.pop(0))
        return last
    run(8, 4)
    torch.cuda.synchronize()
    est.reset_timing()
    t0 = time.perf_counter()
    rec = run(steps, 4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    est.reset_timing()
    run(4, 1)
    torch.cuda.synchronize()
    tm = est.timing()
    k6_ms = tm.grid_cost_kernel_ms_sum / max(1, tm.grid_cost_launches)
    credited = k6_credit(tm)
    lastb = (steps - 1) % nb
    res = [Rec(r, board.n_corners) for r in rec]
    ok = [f for f in range(F) if res[f].status == N.OK]
    err = np.array([synth.corner_error(res[f].corners_array(), gts[lastb * F + f], board) for f in ok])
    est.close()
    return {"value": steps * F / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "frames_per_batch": F,
            "distinct_batches": nb, "k6_ms_alone": k6_ms, "frac": credited / (max(k6_ms, 1e-9) * 1e-3) / 1e12 / VALU_ISSUE_PEAK_T,
            "credited_lane_instr_per_batch": credited, "frames_ok": "%d/%d" % (len(ok), F),
            "median_corner_error_mm_vs_ground_truth": 1e3 * float(np.median(err)) if len(err) else None,
            "bytes_per_frame": 16 * lidar.n_points + 12 * board.n_corners + 64,
            "hbm_frac": steps * F / dt * (16 * lidar.n_points + 12 * board.n_corners + 64) / 1e9 / HBM_PEAK_GBPS,
            "what": "configs[4], inputs resident, four batches in flight; frac = credited lane-instr of one batch's K6 launches / their "
                    "summed duration with ONE batch in flight / the 78.6 T issue peak"}


class Rec:
    """One compact result record (ILCC_RECORD_HEADER floats + 3 per corner; layout: sharding.pack_records) with the
    attribute names of ilcc_result that this file reads."""
    __slots__ = ("r", "nc")

    def __init__(self, row, n_corners):
        self.r, self.nc = row, n_corners

    status = property(lambda s: int(s.r[0]))
    n_corners = property(lambda s: int(s.r[1]))
    n_black = property(lambda s: int(s.r[13]))
    n_white = property(lambda s: int(s.r[14]))
    basin_margin = property(lambda s: float(s.r[15]))
    flags = property(lambda s: int(s.r[18]))
    n_roi = property(lambda s: int(s.r[19]))

    def corners_array(self):
        k = max(0, min(self.n_corners, self.nc))
        return np.array(self.r[20:20 + 3 * k], dtype=np.float32).reshape(k, 3)


def _gen_noise(args):
    from lidar_camera_calibration_amd import synth
    lo, n, sigma_r, footprint = args
    board, lidar = synth.Board(), synth.vlp16()
    clouds, clicks, gts = [], [], []
    for f in range(lo, lo + n):
        s = 0xC0FFEE + f
        prng = np.random.Generator(np.random.Philox(key=(s ^ 0x905E) & 0xFFFFFFFFFFFFFFFF))
        pose = synth.random_pose(prng)
        clouds.append(synth.make_frame(lidar, board, pose, s, sigma_r=sigma_r, footprint=footprint))
        clicks.append(synth.make_click(pose, s))
        gts.append(synth.true_corners(pose, board))
    return np.stack(clouds), np.stack(clicks), np.stack(gts)


def online_caller_leg(N, params, clouds, gts, n_points, device):
    """LidarCornersEst::get_chessboard_by_point as the online node calls it (lidar_chessboard_online.cpp:91-101)
    -- NO ROI crop, the whole 28 800-point cloud clustered at tolerance 0.10, the cluster around the predicted board centre,
    getPlane, gray zone -- through ilcc_chessboard_by_point_batch (host buffers in, records out, synchronous), 128 frames per
    call.  The whole-cloud clustering is K2's multi-workgroup path (28 800 points per frame > the LDS capacity)."""
    from lidar_camera_calibration_amd import LidarCornersBatch
    F = min(128, len(clouds))
    est = LidarCornersBatch(F, n_points, params, device=device)
    pts = np.ascontiguousarray(gts[:F].mean(axis=1), dtype=np.float32)   # the tracker's prediction: the board centre
    c = np.ascontiguousarray(clouds[:F])
    for _ in range(3):
        res = est.chessboard_by_point(c, pts)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        res = est.chessboard_by_point(c, pts)
        ts.append(time.perf_counter() - t0)
    tm = est.timing()
    found = sum(1 for r in res if r.status == N.OK)
    dt = float(np.median(ts))
    # the same calls in flight four at a time (ilcc_submit_chessboard_by_point / ilcc_wait_chessboard_by_point, pinned host
    # buffers): what a recorded sequence gets -- the copy of one call overlaps the kernels of the others
    import torch
    pc = [torch.from_numpy(c).pin_memory() for _ in range(4)]
    pp = [torch.from_numpy(pts).pin_memory() for _ in range(4)]
    def pipelined(n_calls):
        inflight, results = [], []
        t0 = time.perf_counter()
        for k in range(n_calls):
            if len(inflight) == 4:
                results.append(est.wait_chessboard_by_point(inflight.pop(0)))
            inflight.append(est.submit_chessboard_by_point(pc[k % 4].data_ptr(), F, n_points, pp[k % 4].data_ptr()))
        while inflight:
            results.append(est.wait_chessboard_by_point(inflight.pop(0)))
        dt_all = time.perf_counter() - t0
        return dt_all, sum(1 for res_k in results for r in res_k if r.status == N.OK)   # (counted off the clock: ctypes iteration is slow)
    pipelined(8)
    n_calls = 40
    runs = sorted(pipelined(n_calls) for _ in range(5))
    dtp, n_ok = runs[len(runs) // 2]   # median of 5 runs of 40 calls
    est.close()
    return {"value": F / dt, "unit": "frames/s", "frames_per_call": F, "ms_per_call": 1e3 * dt, "boards_found": "%d/%d" % (found, F),
            "cluster_ms_per_call": round(tm.cluster, 4),
            "value_four_calls_in_flight": n_calls * F / dtp, "ms_per_call_four_in_flight": 1e3 * dtp / n_calls,
            "boards_found_four_in_flight": "%d/%d" % (n_ok, n_calls * F),
            "runs_ms_per_call_four_in_flight": [round(1e3 * r[0] / n_calls, 3) for r in runs],
            "link_bound_frames_per_s_note": "a call moves 16 B x 28 800 points per frame over PCIe: the link figure of the main legs applies",
            "what": "ilcc_chessboard_by_point_batch, host in / host out (the 59 MB H2D copy of a call is inside), median of 10 calls; "
                    "the reference's online node runs this at the sensor's 10 Hz.  value_four_calls_in_flight: the asynchronous pair "
                    "(ilcc_submit_chessboard_by_point / ilcc_wait_chessboard_by_point), 40 calls, four in flight"}


NOISE_VARIANTS = ((0.0, 0.015), (0.001, 0.015), (0.003, 0.015), (0.010, 0.015), (0.0, 0.0))   # (sigma_r, beam footprint) in m


def gen_noise_variants(n_frames, workers):
    import multiprocessing as mp
    chunk = max(1, n_frames // max(1, workers))
    out = []
    for sigma_r, footprint in NOISE_VARIANTS:
        jobs = [(lo, min(chunk, n_frames - lo), sigma_r, footprint) for lo in range(0, n_frames, chunk)]
        if workers > 1 and len(jobs) > 1:
            with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
                parts = pool.map(_gen_noise, jobs)
        else:
            parts = [_gen_noise(j) for j in jobs]
        out.append(tuple(np.concatenate([p[k] for p in parts]) for k in range(3)))
    return out


def noise_floor_leg(N, params, synth, board, n_points, device, inputs):
    """Where the ~2.9 mm median comes from: the bench's first 256 board poses re-drawn with range noise
    sigma_r in {0, 1, 3, 10} mm (15 mm beam footprint, the bench's model) and once with an ideal point beam, through the
    SAME GPU path (ILCC_SOLVER_GRID; it equals the CPU oracle bit for bit, tests/).  The committed oracle study with more
    variants is profiles/r03_noise_floor_study.json (tools/noise_floor_study.py)."""
    from lidar_camera_calibration_amd import LidarCornersBatch
    n_frames = len(inputs[0][0])
    est = LidarCornersBatch(n_frames, n_points, params, device=device)
    out = {"what": "corner error vs ground truth (mm, frames with status OK) of the GPU ILCC_SOLVER_GRID path on the bench's first %d board "
                   "poses, sensor model varied: the median is set by the 15 mm beam footprint greying the square edges out and by "
                   "what 16 rings sample of the pattern, not by the range noise and not by the solver" % n_frames,
           "variants": []}
    for (sigma_r, footprint), (clouds, clicks, gts) in zip(NOISE_VARIANTS, inputs):
        res = est.extract(clouds, clicks)
        ok = [f for f in range(n_frames) if res[f].status == N.OK]
        e = np.array([1e3 * synth.corner_error(res[f].corners_array(), gts[f], board) for f in ok])
        out["variants"].append({"sigma_r_mm": 1e3 * sigma_r, "footprint_mm": 1e3 * footprint, "frames_ok": "%d/%d" % (len(ok), n_frames),
                                "median": float(np.median(e)) if len(e) else None, "p90": float(np.percentile(e, 90)) if len(e) else None,
                                "max": float(e.max()) if len(e) else None})
    est.close()
    return out


def single_frame_latency(N, params, clouds, clicks, n_points, device):
    """What the reference's node does per click: ONE frame, host buffers in, corners out (`ilcc_extract`, synchronous,
    H2D copy and result copy included).  Median of 30 calls after 5 warm-up calls, both solver modes."""
    import ctypes as C
    from lidar_camera_calibration_amd import LidarCornersBatch
    out = {}
    cloud = np.ascontiguousarray(clouds.reshape(-1, n_points, 4)[0])
    click = np.ascontiguousarray(clicks.reshape(-1, 3)[0])
    for name, solver in (("grid", N.SOLVER_GRID), ("reference_local", N.SOLVER_REFERENCE_LOCAL)):
        p = N.Params()
        C.memmove(C.byref(p), C.byref(params), C.sizeof(N.Params))
        p.solver = solver
        e = LidarCornersBatch(1, n_points, p, device=device)
        ts = []
        for k in range(35):
            t0 = time.perf_counter()
            e.extract(cloud[None], click[None])
            ts.append(time.perf_counter() - t0)
        e.close()
        out[name] = round(1e3 * float(np.median(ts[5:])), 4)
    out["what"] = "one frame through ilcc_extract (host in, host out), median of 30 calls; the reference runs this once per rviz click"
    return out


def half_grid_leg(N, est, params, dptrs, FS, run, warm, steps, synth, gts, board, last_on=None):
    """NOT the headline: the same pipeline with the exhaustive grid at HALF the resolution per axis (31 x 20 x 20 x 2 =
    24 800 candidates: theta step 1 deg, translation step g/10) and the refinement lattice kept at the same final
    resolution (refine_div 32).  The pattern search's capture range covers the coarser cells: accuracy statistics are
    the same (oracle study in DESIGN.md), the grid stage does an eighth of the nominal work."""
    import ctypes as C
    import torch
    p = N.Params()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(N.Params))
    p.n_th, p.th_step = 31, 2.0 * params.th_step
    p.n_ty = p.n_tz = 20
    p.ty_step, p.tz_step = 2.0 * params.ty_step, 2.0 * params.tz_step
    p.refine_div = 2 * params.refine_div
    est.set_params(p)
    warm(dptrs, 2)
    torch.cuda.synchronize()
    last = []
    n = max(3, steps // 2)
    t0 = time.perf_counter()
    run(n, dptrs, keep=last, last_on=last_on)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    est.set_params(params)
    res = [Rec(r, board.n_corners) for batch in last for r in batch]
    ok = [f for f in range(FS) if res[f].status == N.OK]
    amb = [f for f in range(FS) if res[f].status == N.AMBIGUOUS]
    e = np.array([synth.corner_error(res[f].corners_array(), gts[f], board) for f in ok])
    return {"value": FS * n / dt, "unit": "frames/s", "steps": n, "ms_per_step": 1e3 * dt / n,
            "grid": "31x20x20 x 2 phases, refine_div 32",
            "frames_ok": "%d/%d" % (len(ok), FS), "frames_flagged_ambiguous": len(amb),
            "median_corner_error_mm_vs_ground_truth": 1e3 * float(np.median(e)) if len(e) else None,
            "p99_corner_error_mm_vs_ground_truth": 1e3 * float(np.percentile(e, 99)) if len(e) else None,
            "max_corner_error_mm_vs_ground_truth": 1e3 * float(e.max()) if len(e) else None,
            "note": "reported beside the headline, never as `value`: `value` keeps SURVEY.md 8(d)'s suggested 61x40x40 grid"}


def h2d_inclusive_leg(torch, dist, rec_dev, est, clouds, d_clicks, F, NB, B, FS, n_points, world, steps, run, warm):
    """SURVEY.md 8(d)'s metric as written: the same pipeline, same steps, but every batch starts in pinned HOST memory and
    crosses PCIe inside the timed region.  Every rank runs it (each GPU has its own link); the timed region is bracketed
    by barriers and the MAX over ranks counts, like `value`.  Explicit copies (ilcc_submit_batch: hipMemcpyAsync on the
    batch's own stream, in front of its kernels) are the figure; at N = 1 the zero-copy variant (K1 reads the pinned
    buffer over PCIe) and what hipMemcpy alone delivers on the link are measured beside it."""
    import time
    pinned = [torch.from_numpy(clouds[b]).pin_memory() for b in range(NB)]      # the same NB distinct batches, rotated through
    pinned_clicks = [d_clicks[b].cpu().pin_memory() for b in range(NB)]
    nbytes = int(pinned[0].numel() * 4)
    hptrs = [t.data_ptr() for t in pinned]
    hclicks = [t.data_ptr() for t in pinned_clicks]
    # 8(d) defines the metric at steady state: a run of n steps also pays ONE pipeline fill and drain (a 472 MB copy before the first
    # kernel can start, the last batch's kernels behind the last copy: ~10 ms), which is 6 % of 20 steps and 1.8 % of 64
    n = max(64, steps)

    def timed(fn):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=rec_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    warm(hptrs, 2, host_clicks=hclicks)      # the same steady-state rule as the resident leg (pinned pages touched, link clocks up)
    dt_copy = timed(lambda: run(n, hptrs, host_clicks=hclicks))
    frames = world * FS * n
    copy = {"value": frames / dt_copy, "ms_per_step": 1e3 * dt_copy / n, "link_GBps_achieved_per_gpu": nbytes * B * n / dt_copy / 1e9}
    zero = None
    if dist is None:
        warm(hptrs, 2)
        dt_zero = timed(lambda: run(n, hptrs))
        zero = {"value": frames / dt_zero, "ms_per_step": 1e3 * dt_zero / n, "link_GBps_achieved_per_gpu": nbytes * B * n / dt_zero / 1e9}
    # what the link itself delivers for the same buffers with nothing else running on this GPU (all ranks at once)
    nbuf = 4
    bufs = [torch.empty(pinned[0].shape, dtype=pinned[0].dtype, device="cuda") for _ in range(nbuf)]
    cs = torch.cuda.Stream()

    def raw_copies(k):
        with torch.cuda.stream(cs):
            for b in range(k):
                bufs[b % nbuf].copy_(pinned[b % NB], non_blocking=True)
            cs.synchronize()
    raw_copies(2)
    raw1 = 16 * nbytes / timed(lambda: raw_copies(16)) / 1e9
    # ... and from two streams at once (the pipeline's copies ride on four: the runtime spreads them over its SDMA engines, and two
    # engines can move more than one -- the bound is the better of the two figures, so that link_frac cannot flatter the pipeline)
    cs2 = torch.cuda.Stream()

    def raw_copies2(k):
        for b in range(k):
            with torch.cuda.stream(cs if b % 2 == 0 else cs2):
                bufs[b % nbuf].copy_(pinned[b % NB], non_blocking=True)
        cs.synchronize()
        cs2.synchronize()
    raw_copies2(2)
    raw2 = 16 * nbytes / timed(lambda: raw_copies2(16)) / 1e9
    raw = max(raw1, raw2)
    del bufs
    best, how = (copy, "explicit copies") if zero is None or copy["value"] >= zero["value"] else (zero, "zero-copy")
    return {"value": best["value"], "unit": "frames/s", "steps": n, "ms_per_step": best["ms_per_step"], "how": how,
            "h2d_bytes_per_step_per_gpu": nbytes * B,
            "link_GBps_achieved_per_gpu": best["link_GBps_achieved_per_gpu"],
            "link_GBps_raw_hipMemcpy": raw, "link_GBps_raw_one_stream": raw1, "link_GBps_raw_two_streams": raw2,
            "link_bound_frames_per_s": world * raw * 1e9 / (nbytes / F),
            "explicit_copy": dict(copy, how="ilcc_submit_batch: hipMemcpyAsync of the batch on the batch's own stream, in front of "
                                            "its kernels (overlaps with the other batches in flight)"),
            "zero_copy": dict(zero, how="the batch stays in pinned host memory and K1 (which reads every input point exactly "
                                        "once) fetches it over PCIe while other batches compute") if zero else None,
            "hptrs": hptrs, "hclicks": hclicks, "keepalive": (pinned, pinned_clicks)}


def _cpu_all_cores(clouds, clicks, p, budget_s):
    """The same CPU path on every host core (frames in parallel; ctypes releases the GIL)."""
    import concurrent.futures as cf
    from oracle import binding as ob
    cores = os.cpu_count() or 1
    ob.extract(clouds[0], clicks[0], p)
    stop = time.perf_counter() + budget_s

    def worker(w):
        n, f = 0, w
        while time.perf_counter() < stop:
            ob.extract(clouds[f % len(clouds)], clicks[f % len(clouds)], p)
            n += 1
            f += cores
        return n

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "sample": "%d frames in %.1f s on %d threads" % (n, dt, cores)}


def cpu_baseline(clouds, clicks, gts, board, budget_s, gpu_ref=None):
    """Reference-faithful CPU path (oracle, ORC_SOLVER_REFERENCE_LOCAL, both phases), 1 thread: median of 5 runs
    over disjoint blocks of the same frames, after one warm-up frame."""
    from lidar_camera_calibration_amd import synth
    from oracle import binding as ob   # checker / baseline only; never on the product path
    p = ob.default_params()
    p.solver = ob.SOLVER_REFERENCE_LOCAL
    p.phase_mode = 2
    p.accum_float = 0   # same accumulation precision as the HIP path, so the corner comparison below is like for like
    ob.extract(clouds[0], clicks[0], p)
    t_probe = time.perf_counter()
    ob.extract(clouds[1 % len(clouds)], clicks[1 % len(clouds)], p)
    t_probe = time.perf_counter() - t_probe
    per_run = int(max(8, min(len(clouds) // 5, budget_s / 5.0 / max(t_probe, 1e-4))))
    rates, errs, dev, status_match, compared = [], [], [], 0, 0
    for run_i in range(5):
        lo = run_i * per_run
        t0 = time.perf_counter()
        rs = [ob.extract(clouds[f % len(clouds)], clicks[f % len(clouds)], p) for f in range(lo, lo + per_run)]
        rates.append(per_run / (time.perf_counter() - t0))
        for k, r in enumerate(rs):
            f = (lo + k) % len(clouds)
            if r.status == 0:
                errs.append(synth.corner_error(ob.result_corners(r), gts[f], board))
            if gpu_ref is not None:
                compared += 1
                status_match += int((r.status == 0) == (gpu_ref[f] is not None))
                if r.status == 0 and gpu_ref[f] is not None:
                    dev.append(float(np.abs(ob.result_corners(r) - gpu_ref[f]).max()))
    return {
        "value": float(np.median(rates)),
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "runs_frames_per_s": [round(r, 2) for r in rates],
        "all_cores": _cpu_all_cores(clouds, clicks, p, min(6.0, budget_s)),
        "sample": "median of 5 runs of %d frames each (disjoint blocks of the step's frames), single thread; restatement of "
                  "the reference path (crop, cluster, RANSAC, PCA, gray zone, 2 phases x Ceres-style pass A+B); omits "
                  "Ceres autodiff/heap and PCL kd-tree overheads, so it is faster than the real reference" % per_run,
        "max_corner_error_mm_vs_ground_truth": 1e3 * max(errs) if errs else None,
        "median_corner_error_mm_vs_ground_truth": 1e3 * float(np.median(errs)) if errs else None,
        "gpu_vs_cpu_corner_deviation_mm": {
            "what": "GPU ILCC_SOLVER_REFERENCE_LOCAL vs this CPU path, same frames, max |dx| over corners",
            "frames_compared": len(dev), "status_agree": "%d/%d" % (status_match, compared),
            "max": 1e3 * max(dev) if dev else None,
            "n_above_1mm": int(sum(d > 1e-3 for d in dev)),
            "note": "both sides accumulate centroid/covariance in double (PCL: float; switching the oracle to float "
                    "moves corners by < 0.1 mm)",
        } if gpu_ref is not None else None,
    }


if __name__ == "__main__":
    main()
