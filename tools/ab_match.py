#!/usr/bin/env python3
"""Back-to-back launch trains of k_match on C2 and C3 (the numbers quoted in match.hip come from
rebuilding with other template arguments and running this)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lidar_odometry_demo_amd as lom
from lidar_odometry_demo_amd import synth

boxes = synth.make_boxes()
if os.environ.get("LOM_TABLE_MULT"):
    print("table mult", os.environ["LOM_TABLE_MULT"], flush=True)
cases = {}
for name, (nb, naz, nmap) in {"C2": (16, 1800, 500_000), "C3": (64, 2048, 2_000_000)}.items():
    scan, _, _, _ = synth.make_scan(nb, naz, boxes=boxes)
    mp, mn = synth.make_map_points(nmap, boxes=boxes)
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(mp, mn)
    d = torch.from_numpy(scan).to("cuda:0")
    torch.cuda.synchronize()
    cases[name] = (g, d)
REPS = int(os.environ.get("LOM_AB_REPS", "100"))
for rep in range(int(os.environ.get("LOM_AB_ROUNDS", "3"))):
    for v in ("built-in",):
        row = []
        for name, (g, d) in cases.items():
            us, by, rq = g.profileMatch(d.data_ptr(), d.shape[0], lom.Pose3D(), 0.3, reps=REPS)
            row.append(f"{name} {us:7.2f} us alg {by / us / 1e3:6.0f} GB/s req {rq / us / 1e3:6.0f} GB/s")
        print(f"variant {v}: " + " | ".join(row), flush=True)
