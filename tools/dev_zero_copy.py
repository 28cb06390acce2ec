"""PCIe-inclusive variants (dev tool): explicit H2D copies vs K1 reading pinned host memory directly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_camera_calibration_amd import LidarCornersBatch, synth
from lidar_camera_calibration_amd import _native as N
board, lidar = synth.Board(), synth.vlp16()
F = 128
clouds, clicks, _, _ = synth.make_batch(F, lidar, board, seed=0xC0FFEE)
est = LidarCornersBatch(F, lidar.n_points, N.default_params(), device=0)
h = torch.from_numpy(clouds).pin_memory()
d_k = torch.from_numpy(clicks).cuda()
def run(ptr, n, depth=3):
    t = []
    for i in range(n):
        t.append(est.submit_device(ptr, F, lidar.n_points, d_k.data_ptr()))
        if len(t) == depth: est.wait(t.pop(0))
    while t: r = est.wait(t.pop(0))
    return r
d = h.cuda()
for name, ptr in (("device-resident", d.data_ptr()), ("zero-copy pinned host", h.data_ptr())):
    run(ptr, 10); torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(ptr, 60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.0f frames/s, %.3f ms/step, ok %d" % (F * 60 / dt, dt / 60 * 1e3, sum(1 for f in range(F) if r[f].status == 0)))

# copies and compute with NO dependency between them: does the link slow down under compute?
cs = torch.cuda.Stream()
dummy = torch.empty_like(d)
def copies(n):
    with torch.cuda.stream(cs):
        for _ in range(n):
            dummy.copy_(h, non_blocking=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
copies(3); cs.synchronize()
with torch.cuda.stream(cs): e0.record()
copies(20)
with torch.cuda.stream(cs): e1.record()
cs.synchronize(); print("copy alone: %.1f GB/s" % (20 * h.numel() * 4 / e0.elapsed_time(e1) / 1e6))
with torch.cuda.stream(cs): e0.record()
copies(40)
with torch.cuda.stream(cs): e1.record()
t0 = time.perf_counter(); run(d.data_ptr(), 60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
cs.synchronize()
print("copy under compute: %.1f GB/s; compute under copy: %.0f frames/s" % (40 * h.numel() * 4 / e0.elapsed_time(e1) / 1e6, F * 60 / dt))
