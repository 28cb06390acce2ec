"""Where does the N > 1 step path spend host time?  1 rank, RCCL, forced collective (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
from lidar_camera_calibration_amd.sharding import gather_records, pack_records

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
F = 128; board = synth.Board(); lidar = synth.vlp16()
clouds, clicks, gts, _ = synth.make_batch(F, lidar, board, seed=1)
d_clouds = torch.from_numpy(clouds).to(dev); d_clicks = torch.from_numpy(clicks).to(dev)
est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
T = {k: 0.0 for k in ("submit", "wait", "pack", "h2d", "pwait", "gather")}
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
pending = []
def finish(t):
    t0 = time.perf_counter(); res = est.wait(t); t1 = time.perf_counter(); T["wait"] += t1 - t0
    if mode == "none": return
    rec = pack_records(res, F, board.n_corners); t2 = time.perf_counter(); T["pack"] += t2 - t1
    if mode == "pack": return
    rec = torch.from_numpy(rec).to(dev); t3 = time.perf_counter(); T["h2d"] += t3 - t2
    if mode == "h2d": return
    while pending:
        w, _ = pending.pop(0); w.wait()
    t4 = time.perf_counter(); T["pwait"] += t4 - t3
    pending.append(gather_records(rec, 1, 0, async_op=True, force_collective=True)); T["gather"] += time.perf_counter() - t4
def run(n):
    tk = []
    for _ in range(n):
        t0 = time.perf_counter(); tk.append(est.submit_device(d_clouds.data_ptr(), F, lidar.n_points, d_clicks.data_ptr())); T["submit"] += time.perf_counter() - t0
        if len(tk) == 3: finish(tk.pop(0))
    while tk: finish(tk.pop(0))
run(20)
for k in T: T[k] = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(mode, "ms/step %.3f" % (10 * dt), {k: round(10 * v, 3) for k, v in T.items()})
dist.destroy_process_group()
