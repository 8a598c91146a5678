import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lidar_odometry_demo_amd as lom
from lidar_odometry_demo_amd import synth
boxes = synth.make_boxes()
scan, _, _, _ = synth.make_scan(16, 1800, boxes=boxes)
mp, mn = synth.make_map_points(500_000, boxes=boxes)
g = lom.VoxelGrid(0.5, 20); g.addCloud(mp, mn)
m = lom.CloudMatcher()
guess = lom.Pose3D((0.05, -0.04, 0.02), (0.99996, 0.0, 0.0017, 0.0087))
for i in range(3):
    p = m.align(g, scan, guess)
os.environ["LOM_DEBUG_LM"] = "1"
if os.environ.get("LOM_EXP_DUMMY"): os.environ["LOM_TEST_SERVER_TIMEOUT_TICKS"] = "12345"
p = m.align(g, scan, guess)
print(m.stats)
