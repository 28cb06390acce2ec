"""CPU tests of the host side: C-ABI surface, file contracts, yaml reader, sharding (gloo, 2 ranks)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lidar_camera_calibration_amd import _native as N
from lidar_camera_calibration_amd import sharding, synth
from lidar_camera_calibration_amd.lidar_corners_est import (LidarCornersEst, read_lidar_corners,
                                                           save_corners2txt)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ilcc_hip.h")).read()
    declared = set(re.findall(r"\b(ilcc_[a-z0-9_]+)\s*\(", header))
    declared -= {"ilcc_handle", "ilcc_params", "ilcc_result", "ilcc_timing"}
    assert declared == set(N.EXPORTS)
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ilcc_abi_version() == N.ABI_VERSION == 5


def test_calib_library_exports_every_declared_symbol():
    from lidar_camera_calibration_amd import calib
    header = open(os.path.join(ROOT, "include", "ilcc_calib.h")).read()
    body = header[header.index('extern "C"'):]
    declared = set(re.findall(r"\b(ilcc_[a-z0-9_]+)\s*\(", body))
    assert len(declared) == 8
    lib = calib.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_library():
    """ctypes mirrors vs the compiled structs: default params round-trip through C."""
    p = N.default_params()
    assert tuple(p.roi_half) == (1.0, 1.5, 2.0)
    assert (p.cluster_tol, p.cluster_min, p.cluster_max) == (0.12, 100, 25000)
    assert p.ransac_thresh == 0.03 and p.hist_bins == 100 and p.gray_rate == 2.5 and p.huber_delta == 0.1
    assert (p.grid_length, p.board_w, p.board_h) == (0.15, 6, 8)
    assert p.solver == N.SOLVER_GRID and p.phase_mode == 2 and p.max_iterations == 50 and p.grid_prune == 1
    assert (p.n_th, p.n_ty, p.n_tz) == (61, 40, 40)
    assert p.tz_step == pytest.approx(0.0075) and p.tz_min == pytest.approx(-0.15)
    assert (p.refine_div, p.refine_max_rounds, p.refine_th_margin) == (16, 64, 32)
    assert p.ambiguity_eps == 1.0 and p.online_cluster_tol == 0.10     # LidarCornersEst.cpp:80
    assert p.min_cell_coverage == 0.9
    assert C.sizeof(N.Timing) == 7 * 4 + 4 + 8 + 4 * 8 + 5 * 8 + 8 + 8 * 8 + 8   # ABI 4: three doubles appended; ABI 5: two more, the batch count, the stage sums
    assert C.sizeof(N.Result) == 14 * 4 + 4 + 16 + 64 + 4 + 16 + 24 + 24 + 8 + 8 + 8 + 3 * 4 * 256  # no hidden padding surprises


def test_ros_node_source_type_checks(tmp_path):
    """host/get_lidar_corners_node.cpp carries the reference node's external surface (node name, private
    parameters, topics, bag reading, corner files) and needs ROS1 + PCL, which the image lacks.  It is type-checked
    against declaration-only headers (tests/ros_stub/, `g++ -fsyntax-only`): a typo in that file fails here.  Not a build
    of the reference, nothing links.  """
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "lidar_camera_calibration_amd", "host", "get_lidar_corners_node.cpp")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "tests", "ros_stub"),
           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "lidar_camera_calibration_amd", "host")]
    r = subprocess.run(cmd + [src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the surface the launch files and calib_lidar_cam.launch rely on is spelled in this file
    text = open(src).read()
    for token in ('"lidar_corners"', '"bag_path_prefix"', '"bag_num"', '"lidar_topic"', '"camera_name"', '"yaml_path"',
                  '"/clicked_point"', '"/velodyne_points"', '"/ChessBoard"', '"/pca_cloud"', '"/Optim_cloud"', '"/lidar_corners"',
                  '"/velodyne"', '"/process_data/"', '"_lidar_"'):
        assert token in text, token
    # and the check has teeth: a misspelt member of the estimator makes it fail
    bad = tmp_path / "node_typo.cpp"
    bad.write_text(text.replace("estimator_.EuclideanCluster()", "estimator_.EuclideanClusters()"))
    r = subprocess.run(cmd + [str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "EuclideanClusters" in r.stderr


def test_bench_reads_its_roofline_inputs_from_committed_profiles():
    """roofline.traffic / issued_vs_credited are derived at run time from the PMC summaries committed under profiles/
    (no scaled constants): the parser finds the K6 launches of the batch sizes the bench runs."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for (config, frames), paths in bench.PMC_FILES.items():
        have = [q for q in paths if os.path.exists(os.path.join(root, q))]
        if not have:                                                # no clean pass committed for this configuration:
            assert bench.k6_pmc(config, frames) is None            # traffic is null, never borrowed from another file
            continue
        pmc = bench.k6_pmc(config, frames)
        assert pmc is not None and pmc["file"] == have[0]          # the newest committed pass of that configuration
        assert pmc["traffic_bytes"] > 1e6 and pmc["valu_wave_instr"] > pmc["full_pass"]["valu_wave_instr"] > 1e7
        assert 0.15 < pmc["full_pass"]["valu_busy_quad_cycles"] / (pmc["full_pass"]["gui_active_cycles_per_xcd"] * 256.0) < 1.2
        # a batch alone on the chip cannot take longer than the same kernel does with three other batches beside it
        # (round 3's config-5 file failed this: a cold first dispatch was averaged in)
        alone_ms = pmc["full_pass"]["gui_active_cycles_per_xcd"] / 2.4e6
        rp = bench.k6_rocprof(config, 1.0)
        if rp and "pipelined" in rp:
            assert alone_ms <= 1.15 * rp["pipelined"]["k6_ms_per_batch"], (alone_ms, rp)   # (full pass alone <= the batch's K6 kernels pipelined)
    assert bench.k6_pmc(2, 100) is None          # no file for that batch size: traffic is null, never a scaled guess


def test_k6_credit_matches_the_isa():
    """bench.py credits each executed K6 evaluation with the VALU instruction count of the term: the constants there, the
    committed profiles/r06_k6_isa_count.json and a fresh run of tools/k6_isa_count.sh on the current source (hipcc -S,
    no GPU needed) must agree -- the credit is regenerated, not asserted."""
    import json
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = json.load(open(os.path.join(root, "profiles", "r06_k6_isa_count.json")))
    src = open(os.path.join(root, "bench.py")).read()
    border = float(re.search(r"^K6_VALU_OPS_BORDER = ([0-9.]+)", src, re.M).group(1))
    interior = float(re.search(r"^K6_VALU_OPS_INTERIOR = ([0-9.]+)", src, re.M).group(1))
    box = float(re.search(r"^K6_VALU_OPS_BOX = ([0-9.]+)", src, re.M).group(1))
    assert (border, interior, box) == (committed["border_valu_per_eval"], committed["interior_valu_per_eval"],
                                       committed["box_valu_per_tile_eval"])
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    fresh = json.loads(subprocess.check_output([os.path.join(root, "tools", "k6_isa_count.sh")], text=True))
    assert (fresh["border_valu_per_eval"], fresh["interior_valu_per_eval"], fresh["box_valu_per_tile_eval"]) == (border, interior, box), fresh


def test_defaults_agree_with_the_oracle(ob):
    p, o = N.default_params(), ob.default_params()
    for f in ("cluster_tol", "cluster_min", "cluster_max", "ransac_thresh", "ransac_hyp", "ransac_seed", "ransac_probability",
              "hist_bins", "gray_rate", "huber_delta", "grid_length", "board_w", "board_h", "phase_mode",
              "n_th", "n_ty", "n_tz", "th_min", "th_step", "ty_min", "ty_step", "tz_min", "tz_step",
              "refine_div", "refine_max_rounds", "refine_th_margin", "ambiguity_eps", "min_cell_coverage"):
        assert getattr(p, f) == getattr(o, f), f
    assert tuple(p.roi_half) == tuple(o.roi_half)


def test_strerror_and_no_device_failure():
    lib = N.lib()
    assert lib.ilcc_strerror(N.OK) == b"ok"
    assert b"cluster" in lib.ilcc_strerror(N.NO_CLUSTER)
    if _has_gpu():
        pytest.skip("device present")
    p = N.default_params()
    assert not lib.ilcc_create(-1, C.byref(p), 1, 1000)
    assert b"no CPU fallback" in lib.ilcc_last_error(None)
    with pytest.raises(Exception):
        est = LidarCornersEst()
        est.setROI(np.zeros((10, 4), np.float32), [0, 0, 0])
        est.EuclideanCluster()


def test_set_chessboard_param_reads_reference_yaml(golden_dir, tmp_path, capsys):
    p = N.default_params()
    assert N.lib().ilcc_set_chessboard_param(C.byref(p), os.path.join(golden_dir, "pointgrey.yaml").encode()) == 0
    assert (p.grid_length, p.board_w, p.board_h) == (0.15, 6, 8)      # 5+1, 7+1 sorted ascending
    y = tmp_path / "b.yaml"
    y.write_text("%YAML:1.0\ngrid_length: 0.10\ncorner_in_x: 11   # comment\ncorner_in_y: 8\n")
    assert N.lib().ilcc_set_chessboard_param(C.byref(p), str(y).encode()) == 0
    assert (p.grid_length, p.board_w, p.board_h) == (0.10, 9, 12)
    assert p.ty_step == pytest.approx(0.005) and p.ty_min == pytest.approx(-0.10)   # grid follows g
    assert N.lib().ilcc_set_chessboard_param(C.byref(p), b"/nonexistent.yaml") == N.IO_ERROR
    bad = tmp_path / "bad.yaml"
    bad.write_text("grid_length: 0.1\n")
    assert N.lib().ilcc_set_chessboard_param(C.byref(p), str(bad).encode()) == N.BAD_ARGUMENT
    est = LidarCornersEst.__new__(LidarCornersEst)
    est._lib, est.params, est._h = N.lib(), N.default_params(), None
    assert est.set_chessboard_param("/nonexistent.yaml") is False
    assert "can not open" in capsys.readouterr().out
    assert est.set_chessboard_param(os.path.join(golden_dir, "pointgrey.yaml")) is True


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_corner_file_writer_is_byte_identical_to_reference_files(golden_dir, tmp_path, n):
    """save_corners2txt through the C-ABI regenerates the reference's own files byte for byte, and
    read_lidar_corners (the consumer, ImageCornersEst.cpp:281-299) reads them back."""
    src = os.path.join(golden_dir, "pointgrey_lidar_%d.txt" % n)
    corners = np.loadtxt(src, dtype=np.float32)
    out = tmp_path / ("pointgrey_lidar_%d.txt" % n)
    save_corners2txt(corners, str(out))
    assert out.read_bytes() == open(src, "rb").read()
    back = read_lidar_corners(str(out), 35)
    assert back.shape == (35, 3) and np.array_equal(back.astype(np.float32), corners)
    assert len(read_lidar_corners(str(out), 10)) == 10
    assert len(read_lidar_corners(str(out), 100)) == 35


def test_synth_generator_is_deterministic_and_shaped():
    board = synth.Board()
    a, ka, ga, _ = synth.make_batch(2, seed=7)
    b, kb, gb, _ = synth.make_batch(2, seed=7)
    assert a.shape == (2, 28800, 4) and a.dtype == np.float32 and np.array_equal(a, b) and np.array_equal(ka, kb)
    assert ga.shape == (2, 35, 3)
    assert np.isfinite(a).all() and a[..., 3].min() >= 0 and a[..., 3].max() <= 255
    c, _, _, _ = synth.make_batch(1, synth.hdl64(), synth.Board(9, 12, 0.10), seed=1)
    assert c.shape == (1, 131072, 4)
    pose = synth.pose_from_fixture(0)
    assert synth.corner_error(synth.true_corners(pose, board)[::-1], synth.true_corners(pose, board), board) < 1e-12


def test_shard_ranges_and_record_packing():
    assert [sharding.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sharding.shard_range(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    res = (N.Result * 2)()
    res[0].status, res[0].n_corners, res[0].phase = 0, 35, 1
    for k in range(105):
        res[0].corners[k] = k * 0.5
    res[1].status = N.NO_CLUSTER
    rec = sharding.pack_records(res, 2, 35, tag_base=640)
    assert rec.shape == (2, N.RECORD_HEADER + 105) and rec[0, 1] == 35 and rec[1, 0] == N.NO_CLUSTER
    cs = sharding.unpack_corners(rec)
    assert cs[0].shape == (35, 3) and cs[0][34, 2] == 52.0 and cs[1].shape == (0, 3)
    # what rank 0 checks after the gather: whose record sits where, and that its contents are intact
    assert list(rec[:, 16]) == [640, 641]
    sharding.verify_records(rec, [640, 641])
    with pytest.raises(AssertionError, match="tag"):          # two ranks' blocks delivered in the wrong order
        sharding.verify_records(rec[::-1], [640, 641])
    bad = rec.copy()
    bad[0, N.RECORD_HEADER + 7] += 1e-3                        # one corner coordinate damaged in flight
    with pytest.raises(AssertionError, match="content"):
        sharding.verify_records(bad, [640, 641])
    swapped = rec.copy()
    swapped[0, N.RECORD_HEADER:N.RECORD_HEADER + 3], swapped[0, N.RECORD_HEADER + 3:N.RECORD_HEADER + 6] = \
        rec[0, N.RECORD_HEADER + 3:N.RECORD_HEADER + 6], rec[0, N.RECORD_HEADER:N.RECORD_HEADER + 3]
    with pytest.raises(AssertionError, match="content"):       # the fold is position dependent
        sharding.verify_records(swapped, [640, 641])


def _gloo_worker(rank, world, port, clouds, clicks, out_dir):
    import torch.distributed as dist
    from oracle import binding as ob_          # stand-in producer for the CPU test only
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = ob_.default_params()
    p.solver = ob_.SOLVER_REFERENCE_LOCAL

    def producer(cl, ck):
        return [ob_.extract(cl[i], ck[i], p) for i in range(len(ck))]
    g = sharding.run_sharded(producer, clouds, clicks, world, rank, 35)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), g)
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(ob, tmp_path):
    """world_size 2 over gloo: contiguous shards, one gather, records identical to a 1-rank run
    (3 frames -> ranks own 2 + 1, the short shard is padded and the padding dropped)."""
    import torch.multiprocessing as mp
    clouds, clicks, _, _ = synth.make_batch(3, fixture_poses=True)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, clouds, clicks, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    p = ob.default_params()
    p.solver = ob.SOLVER_REFERENCE_LOCAL
    want = sharding.pack_records([ob.extract(clouds[i], clicks[i], p) for i in range(3)], 3, 35)
    assert got.shape == want.shape == (3, N.RECORD_HEADER + 105)
    assert np.array_equal(got, want)


def _gloo_worker_synthetic(rank, world, port, n_frames, out_dir):
    """run_sharded with a producer that needs no solver: record f is a function of the GLOBAL frame index only."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clouds = np.arange(n_frames, dtype=np.float32).reshape(n_frames, 1, 1) * np.ones((1, 4, 4), np.float32)   # frame f holds the value f
    clicks = np.zeros((n_frames, 3), np.float32)

    def producer(cl, ck):
        out = []
        for i in range(len(ck)):
            r = N.Result()
            f = int(cl[i, 0, 0])
            r.status, r.n_corners, r.grid_index, r.n_roi = 0, 35, 1000 + f, 7 * f
            for k in range(105):
                r.corners[k] = f + k / 128.0
            out.append(r)
        return out
    g = sharding.run_sharded(producer, clouds, clicks, world, rank, 35)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered4.npy"), g)
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_gather_with_an_uneven_shard(tmp_path):
    """world_size 4 over gloo, F = 10 frames: ceil(10 / 4) = 3 per rank -> ranks own 3 + 3 + 3 + 1; every rank ships 3 records
    (the last one two padding records), rank 0 drops the padding, checks every tag against the global frame index and every
    content check word (sharding.verify_records inside run_sharded), and the block equals a 1-process packing."""
    import torch.multiprocessing as mp
    assert [sharding.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharding.shard_range(2, 4, 3) == (2, 2)                       # more ranks than frames: empty shards are legal
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker_synthetic, args=(4, port, 10, str(tmp_path)), nprocs=4, join=True)
    got = np.load(tmp_path / "gathered4.npy")
    assert got.shape == (10, N.RECORD_HEADER + 105)
    assert np.array_equal(got[:, 16], np.arange(10)) and np.array_equal(got[:, 3], 1000 + np.arange(10))
    assert np.array_equal(got[:, 19], 7 * np.arange(10))                 # header slot 19: n_roi
    assert np.array_equal(got[:, N.RECORD_HEADER], np.arange(10, dtype=np.float32))
    sharding.verify_records(got, np.arange(10))


def test_bench_gpus_flag_never_silently_runs_one_rank():
    """`python bench.py --gpus 8` on a box with fewer devices exits non-zero with a clear message and prints no JSON line; a
    launcher whose WORLD_SIZE disagrees with --gpus is refused too (the flag was parsed and ignored)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "ILCC_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode != 0 and "needs 64 HIP devices" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_rank_placement_reads_the_gpus_numa_node(tmp_path):
    """sharding.pin_rank_to_gpu_numa: PCI address -> numa_node -> that node's cpulist (a fake sysfs tree; nothing is pinned here)."""
    from lidar_camera_calibration_amd.sharding import gpu_numa_node, parse_cpulist, pin_rank_to_gpu_numa
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c5:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    allowed = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text("%d-%d\n" % (allowed[0], allowed[-1]))
    assert gpu_numa_node("0000:C5:00.0", str(tmp_path)) == 1 and gpu_numa_node("0000:00:00.0", str(tmp_path)) == -1
    info = pin_rank_to_gpu_numa("0000:c5:00.0", str(tmp_path), apply=False)
    assert info["numa_node"] == 1 and info["cpus"] == len(allowed) and info["pinned"] is False
    assert pin_rank_to_gpu_numa(None, str(tmp_path))["pinned"] is False
    assert pin_rank_to_gpu_numa("0000:00:00.0", str(tmp_path))["numa_node"] == -1
