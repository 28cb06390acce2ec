"""Experiment: two handles driven by two host threads (ctypes releases the GIL) -- does overlapping
one batch's latency-bound stages with the other's K6 raise throughput?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_camera_calibration_amd import synth, LidarCornersBatch, _native as N
F = 128
clouds, clicks, gts, _ = synth.make_batch(F, seed=0xC0FFEE)
dev = torch.device("cuda", 0)
dc = torch.from_numpy(clouds).to(dev); dk = torch.from_numpy(clicks).to(dev)
for nthreads in (1, 2, 3):
    ests = [LidarCornersBatch(F, 28800, N.default_params()) for _ in range(nthreads)]
    for e in ests:
        for _ in range(3): e.extract_device(dc.data_ptr(), F, 28800, dk.data_ptr())
    K = 40
    torch.cuda.synchronize()
    def work(e, n):
        for _ in range(n): e.extract_device(dc.data_ptr(), F, 28800, dk.data_ptr())
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(e, K // nthreads)) for e in ests]
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    steps = (K // nthreads) * nthreads
    print("threads", nthreads, "ms/step %.3f" % (1e3 * dt / steps), "frames/s %.0f" % (steps * F / dt))
    for e in ests: e.close()
