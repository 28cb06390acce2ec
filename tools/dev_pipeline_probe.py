"""Sensitivity probe: the bench's 4-deep pipeline (one 1024-frame batch per step) with parameter overrides, to see
what a stage is worth: e.g. `refine_div=0` (K7r keeps the grid argmin: no pattern search), `ransac_hyp=8`, `n_th=31` ...
usage: python tools/dev_pipeline_probe.py [steps=20] [name=value ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
over = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("depth=")]
DEPTH = int(([a.split("=")[1] for a in sys.argv[1:] if a.startswith("depth=")] or ["4"])[0])
F, B, n_points = 1024, 1, 28800
clouds, clicks, gts = bench.generate(2, F * B, 0xC0FFEE, 16)
import torch
from lidar_camera_calibration_amd import LidarCornersBatch
from lidar_camera_calibration_amd import _native as N
params = N.default_params()
for k, v in over:
    setattr(params, k, type(getattr(params, k))(float(v)) if not isinstance(getattr(params, k), int) else int(v))
dev = torch.device("cuda", 0)
d_clouds = [torch.from_numpy(clouds.reshape(B, F, n_points, 4)[b]).to(dev) for b in range(B)]
d_clicks = [torch.from_numpy(clicks.reshape(B, F, 3)[b]).to(dev) for b in range(B)]
est = LidarCornersBatch(F, n_points, params, device=0)
est.reserve(1792, 2560)


T = {"submit": 0.0, "wait": 0.0, "n": 0, "life": 0.0, "gpu": 0.0}


def run(n):
    inflight = []
    for s in range(n):
        for b in range(B):
            t0 = time.perf_counter()
            inflight.append((est.submit_device(d_clouds[b].data_ptr(), F, n_points, d_clicks[b].data_ptr()), t0))
            t1 = time.perf_counter()
            if len(inflight) == DEPTH:
                tk, ts = inflight.pop(0)
                est.wait(tk)
                T["life"] += time.perf_counter() - ts          # host: submit call -> wait returned
                T["gpu"] += 1e-3 * est.timing().total          # GPU: first kernel start -> last kernel end of that batch
            t2 = time.perf_counter()
            T["submit"] += t1 - t0
            T["wait"] += t2 - t1
            T["n"] += 1
    while inflight:
        est.wait(inflight.pop(0)[0])


run(40)
torch.cuda.synchronize()
T.update(submit=0.0, wait=0.0, n=0, life=0.0, gpu=0.0)
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("depth %d" % DEPTH, end=" ")
print("probe %s: %.0f frames/s, %.3f ms per step; host per batch: submit %.3f ms, wait (incl. blocking) %.3f ms"
      % (dict(over), F * B * steps / dt, 1e3 * dt / steps, 1e3 * T["submit"] / T["n"], 1e3 * T["wait"] / T["n"]))
print("   per batch: host submit->wait-returned %.3f ms, GPU first-kernel->last-kernel %.3f ms (difference = queueing before the first kernel + wake-up)"
      % (1e3 * T["life"] / T["n"], 1e3 * T["gpu"] / T["n"]))
