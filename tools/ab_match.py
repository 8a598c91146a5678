#!/usr/bin/env python3
"""k_match on C2, C3, C4 and the streaming path's matching cloud: back-to-back launch trains at the identity pose
(lom_profile_match) and whole aligns.  (The numbers quoted in match.hip / DESIGN.md for other template arguments
come from rebuilding and running this.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lidar_odometry_demo_amd as lom
from lidar_odometry_demo_amd import synth

boxes = synth.make_boxes()
cases = {}
maps = {}
for name, (nb, naz, nmap) in {"C2": (16, 1800, 500_000), "C5m": (16, 1800, 500_000), "C3": (64, 2048, 2_000_000),
                              "C4": (128, 2048, 2_000_000)}.items():
    if os.environ.get("LOM_AB_CASES") and name not in os.environ["LOM_AB_CASES"].split(","):
        continue
    scan, _, _, _ = synth.make_scan(nb, naz, boxes=boxes)
    if name == "C5m":
        scan = np.ascontiguousarray(scan[::3])      # ~8.9k points, like the streaming path's matching cloud
    if nmap not in maps:
        mp, mn = synth.make_map_points(nmap, boxes=boxes)
        g = lom.VoxelGrid(0.5, 20)
        g.addCloud(mp, mn)
        maps[nmap] = g
    d = torch.from_numpy(scan).to("cuda:0")
    torch.cuda.synchronize()
    cases[name] = (maps[nmap], d)
REPS = int(os.environ.get("LOM_AB_REPS", "100"))
# the forms of the searches of outer iterations >= 2 in one process: with / without the temporal pruning bound
# (LOM_OPT_NO_TEMPORAL_BOUND), without / with the reference-algorithm counts (LOM_OPT_COUNT_CANDIDATES: all 27 slots);
# the train after a warm-up launch is what those iterations run
MODES = [("temporal       ", 0, 0), ("plain          ", 1, 0), ("temporal+counts", 0, 1), ("plain+counts   ", 1, 1)]
for rep in range(int(os.environ.get("LOM_AB_ROUNDS", "3"))):
    for label, off, cnt in MODES:
        row = []
        for name, (g, d) in cases.items():
            g.setOption(lom.capi.OPT_NO_TEMPORAL_BOUND, off)
            g.setOption(lom.capi.OPT_COUNT_CANDIDATES, cnt)
            us, by, rq, _ = g.profileMatch(d.data_ptr(), d.shape[0], lom.Pose3D(), 0.3, reps=REPS)
            lom.align_repeat(g, d.data_ptr(), d.shape[0], lom.Pose3D(), 20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, tot = lom.align_repeat(g, d.data_ptr(), d.shape[0], lom.Pose3D(), 100)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 10.0
            row.append(f"{name} {us:6.2f} us ({rq / 1e6:6.1f} MB req, {rq / us / 1e3:5.0f} GB/s) align {ms:.4f} ms")
        print(label + ": " + " | ".join(row), flush=True)
