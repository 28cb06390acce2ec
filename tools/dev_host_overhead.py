import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_camera_calibration_amd import synth, LidarCornersBatch, _native as N
F = 128
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
dev = torch.device("cuda", 0)
dc = torch.from_numpy(clouds).to(dev); dk = torch.from_numpy(clicks).to(dev)
e = LidarCornersBatch(F, 28800, N.default_params())
for _ in range(4): e.extract_device(dc.data_ptr(), F, 28800, dk.data_ptr())
torch.cuda.synchronize()
for depth in (1, 2, 3):
    ts = tw = 0.0; K = 60; tickets = []
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter(); tickets.append(e.submit_device(dc.data_ptr(), F, 28800, dk.data_ptr())); ts += time.perf_counter() - a
        if len(tickets) == depth:
            a = time.perf_counter(); e.wait(tickets.pop(0)); tw += time.perf_counter() - a
    while tickets:
        a = time.perf_counter(); e.wait(tickets.pop(0)); tw += time.perf_counter() - a
    dt = time.perf_counter() - t0
    print("depth", depth, "ms/step %.3f" % (1e3 * dt / K), "submit %.3f ms" % (1e3 * ts / K), "wait %.3f ms" % (1e3 * tw / K))
