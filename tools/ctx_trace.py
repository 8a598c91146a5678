#!/usr/bin/env python3
"""K concurrent callers of one C2 keyframe, FORM = shared | partitioned, 150 aligns each: the workload whose kernel
trace (rocprofv3 --kernel-trace) shows what a solve's kernels cost when other callers' kernels are on the GPU.
    rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python3 tools/ctx_trace.py 3 shared
    python3 tools/ctx_trace_summary.py out"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import lidar_odometry_demo_amd as lom

k, form = int(sys.argv[1]), sys.argv[2]
work = bench.build_workload(1, 0, "C2")
grid = lom.VoxelGrid(0.5, 20)
grid.addCloud(work["map_xyz"], work["map_nrm"])
d_scan = torch.from_numpy(work["shard"]).to("cuda:0")
torch.cuda.synchronize()
guess = lom.Pose3D()
ctxs = [lom.ScanContext(grid, partition=((i, k) if form == "partitioned" else None)) for i in range(k)]
for c in ctxs:
    lom.align_repeat(c, d_scan.data_ptr(), d_scan.shape[0], guess, 50)
torch.cuda.synchronize()
start = threading.Barrier(k)


def work_fn(i):
    start.wait()
    lom.align_repeat(ctxs[i], d_scan.data_ptr(), d_scan.shape[0], guess, 150)


th = [threading.Thread(target=work_fn, args=(i,)) for i in range(k)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
for c in ctxs:   # (before the interpreter tears the HIP runtime down: a CU-masked stream destroyed from an exit handler
    c.close()    #  crashed under rocprofv3)
del grid
