#!/usr/bin/env python3
"""Concurrent callers of one keyframe on C2: k scan contexts sharing the whole GPU against k contexts on k slices of
the compute units (bench.concurrent_contexts), a few rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import lidar_odometry_demo_amd as lom

work = bench.build_workload(1, 0, "C2")
dev = torch.device("cuda", 0)
grid = lom.VoxelGrid(0.5, 20)
grid.addCloud(work["map_xyz"], work["map_nrm"])
d_scan = torch.from_numpy(work["shard"]).to(dev)
torch.cuda.synchronize()
guess = lom.Pose3D()
lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, 600)
import time
t0 = time.perf_counter()
lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, 200)
one = 200 / (time.perf_counter() - t0)
print(f"one caller: {one:.0f} aligns/s", flush=True)
for r in range(int(os.environ.get("ROUNDS", "2"))):
    cc = bench.concurrent_contexts(lom, torch, grid, d_scan, guess, 200)
    print(" | ".join(f"{k}: {v['frames_per_s']:.0f}/s ({v['frames_per_s'] / one:.2f}x, fb {v['host_fallbacks']})" for k, v in cc.items()), flush=True)
