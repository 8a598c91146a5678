#!/usr/bin/env python3
"""Soak of the device-resident align on the C2 workload: the same align over and over, every pose compared bit for
bit with the first (the sums are added in a fixed order, so the result is a pure function of the inputs), fall-backs
and errors counted.  usage: soak_align.py [seconds] [C2|C5size]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lidar_odometry_demo_amd as lom  # noqa: E402
from lidar_odometry_demo_amd import synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
small = len(sys.argv) > 2 and sys.argv[2] != "C2"
boxes = synth.make_boxes()
scan, _, _, _ = synth.make_scan(16, 1800, boxes=boxes)
if small:
    scan = np.ascontiguousarray(scan[::3])      # ~8.9k points: the 256-thread k_lm, the four-loads k_match
mp, mn = synth.make_map_points(500_000, boxes=boxes)
g = lom.VoxelGrid(0.5, 20)
g.addCloud(mp, mn)
d_scan = torch.from_numpy(scan).cuda()
guess = lom.Pose3D((0.05, -0.04, 0.02), (0.99996, 0.0, 0.0017, 0.0087))
first, stats0 = lom.align_repeat(g, d_scan.data_ptr(), d_scan.shape[0], guess, 1)
ref = first.translation.tobytes() + first.rotation.tobytes()
aligns = mismatches = fallbacks = 0
t0 = time.time()
while time.time() - t0 < seconds:
    for _ in range(50):
        pose, st = lom.align_repeat(g, d_scan.data_ptr(), d_scan.shape[0], guess, 1)
        aligns += 1
        mismatches += (pose.translation.tobytes() + pose.rotation.tobytes()) != ref
        fallbacks += int(st.get("host_fallback", 0))
        assert st["outer_iterations"] == stats0["outer_iterations"] and st["evaluations"] == stats0["evaluations"]
print(f"{len(scan)} points: {aligns} aligns in {time.time() - t0:.1f} s, {mismatches} poses differ from the first, "
      f"{fallbacks} fall-backs; outer {stats0['outer_iterations']}, evaluations {stats0['evaluations']}")
sys.exit(1 if mismatches or fallbacks else 0)
