#!/usr/bin/env python3
"""Phase stamps of the device-resident solve (k_lm) on the C2 workload: LOM_OPT_DEBUG_LM_STAMPS makes the align
print, for workgroup 0's first lane, the shader-clock cycles of accumulate / reduce+exchange / policy
per evaluation and the split of one reduce+exchange (stderr).  Entries of the different k_lm launches
of one align overwrite each other: read them as samples, not as one timeline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (shares its HIP runtime with the library)
import lidar_odometry_demo_amd as lom  # noqa: E402
from lidar_odometry_demo_amd import synth  # noqa: E402

boxes = synth.make_boxes()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
nb, naz, nmap = {"C2": (16, 1800, 500_000), "C3": (64, 2048, 2_000_000), "C4": (128, 2048, 2_000_000)}[cfg]
scan, _, _, _ = synth.make_scan(nb, naz, boxes=boxes)
mp, mn = synth.make_map_points(nmap, boxes=boxes)
g = lom.VoxelGrid(0.5, 20)
g.addCloud(mp, mn)
m = lom.CloudMatcher()
guess = lom.Pose3D((0.05, -0.04, 0.02), (0.99996, 0.0, 0.0017, 0.0087))
for _ in range(3):
    m.align(g, scan, guess)
g.setOption(lom.capi.OPT_DEBUG_LM_STAMPS, 1)
m.align(g, scan, guess)
print(m.stats)
