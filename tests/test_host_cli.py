"""GPU tests of the C++ host side and of bench.py's multi-rank path (pytest -m gpu).

The reference's host code is C++ (one ROS node driving the class LidarCornersEst); north_star keeps it C++ over a
thin C-ABI.  These tests EXECUTE that side -- the ROS-free CLI `ilcc_corners` and the loop of
`ilcc2/test/get_lidar_corners.cpp:130-211` over `ilcc_host::LidarCornersEst` (`ilcc_host_selftest`) -- and hold its
output files against what the Python mirror writes for the same frames: byte for byte.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from lidar_camera_calibration_amd import LidarCornersEst, save_corners2txt, synth
from lidar_camera_calibration_amd import _native as N

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lidar_camera_calibration_amd")
CLI = os.path.join(PKG, "ilcc_corners")
SELFTEST = os.path.join(PKG, "ilcc_host_selftest")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _frames(n):
    board = synth.Board()
    out = []
    for k in range(n):
        pose = synth.pose_from_fixture(k)
        out.append((synth.make_frame(synth.vlp16(), board, pose, 0x51 + k), synth.make_click(pose, 0x51 + k)))
    return out


def _python_mirror_file(cloud, click, yaml, path, solver):
    m = LidarCornersEst(max_points_per_frame=len(cloud))
    m.params.solver = solver
    assert m.set_chessboard_param(yaml)
    m.setROI(cloud, click)
    assert m.EuclideanCluster()
    m.PCA()
    corners = []
    ok = m.get_corners(corners)
    if ok:
        save_corners2txt(m.m_cloud_corners[:, :3], path)
    sizes = (len(m.m_cloud_ROI), len(m.m_cloud_chessboard), len(m.m_cloud_PCA), len(m.m_cloud_optim) if ok else 0, len(corners))
    st = m.result.status
    m.close()
    return ok, st, sizes


@pytest.mark.parametrize("solver", ["grid", "reference"])
def test_cli_writes_the_file_the_python_mirror_writes(tmp_path, golden_dir, solver):
    """`ilcc_corners --cloud ... --click ... --yaml pointgrey.yaml --out ...` (C++: ilcc_host::LidarCornersEst over the
    C-ABI) vs the Python mirror on the same frame: identical bytes in the process_data-format file."""
    assert os.path.exists(CLI), "build() makes lidar_camera_calibration_amd/ilcc_corners"
    yaml = os.path.join(golden_dir, "pointgrey.yaml")
    cloud, click = _frames(1)[0]
    raw = tmp_path / "frame.bin"
    cloud.tofile(raw)
    out_cpp, out_py = tmp_path / "cpp_lidar_1.txt", tmp_path / "py_lidar_1.txt"
    r = subprocess.run([CLI, "--cloud", str(raw), "--click", *("%.9g" % v for v in click), "--yaml", yaml, "--out", str(out_cpp),
                        "--solver", solver], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "chessboard plane size:" in r.stdout and "add_corner" in r.stdout        # LidarCornersEst.cpp:184, get_lidar_corners.cpp:196
    ok, st, _ = _python_mirror_file(cloud, click, yaml, str(out_py), N.SOLVER_GRID if solver == "grid" else N.SOLVER_REFERENCE_LOCAL)
    assert ok and st == N.OK
    a, b = out_cpp.read_bytes(), out_py.read_bytes()
    assert a == b and a.count(b"\n") == 35
    # and the consumer's reader takes it back (ImageCornersEst::read_lidar_corners)
    from lidar_camera_calibration_amd import read_lidar_corners
    assert read_lidar_corners(str(out_cpp), 35).shape == (35, 3)


def test_cli_rejects_bad_input(tmp_path, golden_dir):
    raw = tmp_path / "empty.bin"
    np.zeros((100, 4), np.float32).tofile(raw)
    r = subprocess.run([CLI, "--cloud", str(raw), "--click", "50", "50", "50", "--out", str(tmp_path / "x.txt")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "change /click_point" in r.stdout              # EuclideanCluster() == false
    assert not (tmp_path / "x.txt").exists()
    r = subprocess.run([CLI, "--cloud", str(raw), "--click", "0", "0", "0", "--yaml", str(tmp_path / "missing.yaml"),
                        "--out", str(tmp_path / "x.txt")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "can not open" in r.stderr                     # LidarCornersEst.cpp:27


def test_cli_reads_the_first_cloud_of_a_bag(tmp_path, golden_dir):
    """--bag: rosbag V2.0 reader + PointCloud2 parser + K0, then the same path (get_lidar_corners.cpp:133-204)."""
    import rosbag_writer as W
    cloud, click = _frames(2)[1]
    a, fields, step = W.velodyne_points(cloud)
    msg = W.pointcloud2(a, fields, step)
    bag = W.BagWriter(str(tmp_path / "20181101_2.bag"), "lz4")
    bag.add_chunk([("/velodyne_points", "sensor_msgs/PointCloud2", W.POINTCLOUD2_MD5, (10, 0), msg)])
    bag.write()
    raw = tmp_path / "frame.bin"
    cloud.tofile(raw)
    yaml = os.path.join(golden_dir, "pointgrey.yaml")
    outs = []
    for src in (["--bag", str(tmp_path / "20181101_2.bag")], ["--cloud", str(raw)]):
        out = tmp_path / ("o%d.txt" % len(outs))
        r = subprocess.run([CLI, *src, "--click", *("%.9g" % v for v in click), "--yaml", yaml, "--out", str(out)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1]


@pytest.mark.parametrize("solver", ["grid", "reference"])
def test_cpp_mirror_runs_the_reference_node_loop(tmp_path, golden_dir, solver):
    """ilcc_host_selftest = the per-bag loop of get_lidar_corners.cpp:130-211 over ONE ilcc_host::LidarCornersEst
    object (set_chessboard_param once, then per frame setROI -> EuclideanCluster -> PCA -> get_corners ->
    save_corners2txt, members read in between), three frames + one frame whose click hits nothing."""
    assert os.path.exists(SELFTEST)
    yaml = os.path.join(golden_dir, "pointgrey.yaml")
    frames = _frames(3)
    frames.insert(2, (frames[0][0], np.array([40.0, 40.0, 40.0], np.float32)))     # "change /click_point ..."
    argv = [SELFTEST, yaml, str(tmp_path / "pointgrey"), solver]
    for k, (cloud, click) in enumerate(frames):
        raw = tmp_path / ("f%d.bin" % k)
        cloud.tofile(raw)
        argv += [str(raw), *("%.9g" % v for v in click)]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("frame ")]
    assert len(lines) == 4
    want_solver = N.SOLVER_GRID if solver == "grid" else N.SOLVER_REFERENCE_LOCAL
    for k, (cloud, click) in enumerate(frames):
        f = dict(zip(lines[k][2::2], lines[k][3::2]))
        py = tmp_path / ("py_%d.txt" % k)
        if k == 2:
            assert f["ok"] == "0" and int(f["status"]) == N.NO_ROI_POINTS and f["corners"] == "0"
            assert not os.path.exists(f["file"])
            continue
        ok, st, sizes = _python_mirror_file(cloud, click, yaml, str(py), want_solver)
        assert ok and f["ok"] == "1" and int(f["status"]) == st
        assert tuple(int(f[k2]) for k2 in ("roi", "chessboard", "pca", "optim", "corners")) == sizes
        assert open(f["file"], "rb").read() == py.read_bytes()
        assert f["file"].endswith("pointgrey_lidar_%d.txt" % (k + 1))                # get_lidar_corners.cpp:197


def test_cpp_multi_gpu_driver_gathers_with_rccl(tmp_path, golden_dir):
    """ilcc_corners_mgpu: one thread + one ilcc_handle per GPU, contiguous shards, H2D on the batch's stream, records
    packed on the GPU and ONE ncclGather (RCCL, C++) to rank 0, which checks tag + content word of every record and
    writes the files.  Runs on EVERY device of the box (n_gpus = 0): a 1-rank communicator on the 1-GPU box, N ranks on
    an N-GPU node, where each frame's record must come from rank f // ceil(F / N).  The files must be the Python
    mirror's byte for byte."""
    import torch
    mgpu = os.path.join(PKG, "ilcc_corners_mgpu")
    assert os.path.exists(mgpu)
    ndev = torch.cuda.device_count()
    yaml = os.path.join(golden_dir, "pointgrey.yaml")
    base = _frames(3)
    frames = [base[k % 3] for k in range(max(3, ndev + 1))]                         # more frames than GPUs: every rank works
    frames.insert(1, (base[0][0], np.array([40.0, 40.0, 40.0], np.float32)))        # a frame without a board
    F = len(frames)
    per = -(-F // ndev)
    argv = [mgpu, yaml, str(tmp_path / "pointgrey"), "0"]
    for k, (cloud, click) in enumerate(frames):
        raw = tmp_path / ("f%d.bin" % k)
        cloud.tofile(raw)
        argv += [str(raw), *("%.9g" % v for v in click)]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gathered %d records from %d GPU(s) with one ncclGather; %d files written" % (F, ndev, F - 1) in r.stdout
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("frame ")]
    assert len(lines) == F
    for k, (cloud, click) in enumerate(frames):
        f = dict(zip(lines[k][2::2], lines[k][3::2]))
        assert int(f["rank"]) == k // per                                           # whose record sits where
        if k == 1:
            assert int(f["status"]) == N.NO_ROI_POINTS and f["file"] == "-"
            continue
        py = tmp_path / ("py_%d.txt" % k)
        ok, st, _ = _python_mirror_file(cloud, click, yaml, str(py), N.SOLVER_GRID)
        assert ok and int(f["status"]) == st == N.OK
        assert open(f["file"], "rb").read() == py.read_bytes()
    # the accept rule (include/ilcc_hip.h): a frame whose record carries ILCC_FLAG_LOW_COVERAGE (the upper third of the board
    # occluded) gets no file unless --accept-low-coverage is given -- what the class mirror, the node and ilcc_corners do
    board = synth.Board()
    pose = synth.pose_from_fixture(0)
    cloud = synth.make_frame(synth.vlp16(), board, pose, 0xC0FFEE)
    click = synth.make_click(pose, 0xC0FFEE)
    up = np.array(pose.v if abs(pose.v[2]) > abs(pose.u[2]) else pose.u)
    up = up * np.sign(up[2])
    keep = ((cloud[:, :3] - pose.centre) @ up < 0.18) | (np.linalg.norm(cloud[:, :3] - pose.centre, axis=1) > 1.0)
    cut = np.ascontiguousarray(cloud[keep])
    raw = tmp_path / "cut.bin"
    cut.tofile(raw)
    tail = [str(raw), *("%.9g" % v for v in click), str(tmp_path / "f0.bin"), *("%.9g" % v for v in frames[0][1])]
    for switches, want_files in (([], None), (["--accept-ambiguous"], None), (["--accept-low-coverage", "--accept-ambiguous"], 2)):
        r = subprocess.run([mgpu, *switches, yaml, str(tmp_path / ("acc%d" % len(switches))), "1", *tail],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        ln = [x.split() for x in r.stdout.splitlines() if x.startswith("frame ")]
        f0 = dict(zip(ln[0][2::2], ln[0][3::2]))
        assert int(f0["status"]) in (N.OK, N.AMBIGUOUS) and int(f0["flags"]) & N.FLAG_LOW_COVERAGE and int(f0["corners"]) == 35
        if want_files is None:
            assert f0["file"] == "-" and "1 files written" in r.stdout and "rejected by the accept rule" in r.stdout
        else:
            assert os.path.exists(f0["file"]) and "%d files written" % want_files in r.stdout
    # a rank whose local work fails still joins the collective: the driver exits with an error, it does not hang
    # (round-2 advisor finding); the LAST rank fails, so that on an N-GPU node the other ranks really wait for it
    env = dict(os.environ, ILCC_MGPU_FAIL_RANK=str(ndev - 1))
    r = subprocess.run(argv, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and "rank %d (device %d) failed" % (ndev - 1, ndev - 1) in r.stderr, r.stdout + r.stderr


def _bench(env_extra, args, launcher=None, timeout=300, with_stderr=False):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):   # a plain invocation must look like one
        env.pop(k, None)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return (json.loads(lines[0]), r.stderr) if with_stderr else json.loads(lines[0])


def test_bench_multi_rank_path_with_one_rccl_rank():
    """bench.py's N > 1 code path -- process group over RCCL (backend "nccl"), records packed on the GPU, ONE gather per
    step on a side stream, rank 0 verifying tags + content checks of what arrived -- forced with a single rank, so that
    RCCL itself is exercised on the 1-GPU box."""
    out = _bench({"ILCC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29611"},
                 ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs", "--batches-per-step", "2", "--frames-per-batch", "256"])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 1000
    assert out["config"]["frames_per_step_per_gpu"] == 512


def test_bench_two_ranks_on_one_device_over_gloo():
    """Two ranks (both on device 0, records gathered over gloo): shard seeds, per-rank tags and the gathered block's
    verification on rank 0 with world_size 2."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29613"]
    out = _bench({"ILCC_BENCH_BACKEND": "gloo", "ILCC_BENCH_SINGLE_DEVICE": "1", "ILCC_BENCH_MAX_DEPTH": "2"},
                 ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batches-per-step", "2", "--frames-per-batch", "64"],
                 launcher=launcher)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 500


def test_bench_plain_invocation_with_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (no WORLD_SIZE): the flag itself must produce two ranks -- bench.py
    re-executes under torch.distributed.run -- and the line must say n_gpus 2 (round 4: the flag was parsed and ignored, a plain
    `--gpus 8` ran one rank and printed n_gpus 1).  Both ranks on device 0, records over gloo (the 1-GPU test hooks)."""
    out, err = _bench({"ILCC_BENCH_BACKEND": "gloo", "ILCC_BENCH_SINGLE_DEVICE": "1", "ILCC_BENCH_MAX_DEPTH": "2"},
                      ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batches-per-step", "2", "--frames-per-batch", "64"],
                      with_stderr=True)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 500
    assert "bench preflight: backend gloo, world 2; ranks [0, 1]" in err, err[-2000:]
    assert "re-executing as" in err
    # every rank reports where it runs: its GPU's PCI address, that GPU's NUMA node and the cores it pinned itself to
    import re
    place = re.findall(r"bench placement: rank (\d) device 0 pci ([0-9a-f:.]+) numa_node (-?\d+) cpus (\d+) pinned (True|False)", err)
    assert sorted(p[0] for p in place) == ["0", "1"], err[-2000:]
    assert all((p[4] == "True") == (int(p[2]) >= 0) and int(p[3]) >= 1 for p in place), place
    assert out["host_placement"]["pci"] == place[0][1]


def test_bench_refuses_more_ranks_than_devices():
    """--gpus N on a node with fewer than N devices exits non-zero with a clear message (no silent 1-rank run), and so does a
    launcher whose WORLD_SIZE disagrees with --gpus."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "ILCC_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(2, n))], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "HIP devices" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
