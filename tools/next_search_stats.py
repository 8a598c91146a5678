#!/usr/bin/env python3
"""The one-lane-per-query search of outer iterations >= 2 (k_match_next) on C2 / C3 / C4: how many queries of the last
search had no usable bound from their previous winner (searched by their wave together), and whole aligns with and
without the kernel (LOM_OPT_NO_NEXT_SEARCH), same box, back to back."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lidar_odometry_demo_amd as lom
from lidar_odometry_demo_amd import synth

boxes = synth.make_boxes()
maps = {}
for name, (nb, naz, nmap) in {"C2": (16, 1800, 500_000), "C3": (64, 2048, 2_000_000), "C4": (128, 2048, 2_000_000)}.items():
    scan, _, _, _ = synth.make_scan(nb, naz, boxes=boxes)
    if nmap not in maps:
        mp, mn = synth.make_map_points(nmap, boxes=boxes)
        g = lom.VoxelGrid(0.5, 20)
        g.addCloud(mp, mn)
        maps[nmap] = g
    g = maps[nmap]
    d = torch.from_numpy(scan).to("cuda:0")
    torch.cuda.synchronize()
    guess = lom.Pose3D()
    out = []
    for off in (0, 1, 0, 1):
        g.setOption(lom.capi.OPT_NO_NEXT_SEARCH, off)
        lom.align_repeat(g, d.data_ptr(), d.shape[0], guess, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, tot = lom.align_repeat(g, d.data_ptr(), d.shape[0], guess, 200)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 200 * 1e3)
    g.setOption(lom.capi.OPT_NO_NEXT_SEARCH, 0)
    lom.align_repeat(g, d.data_ptr(), d.shape[0], guess, 1)
    slow = lom.capi.lib().lom_debug_next_search_slow(g.handle)
    print(f"{name}: {len(scan)} queries, last search: {slow} without a usable bound ({100.0 * slow / len(scan):.2f} %); "
          f"ms per align one-lane / rows / one-lane / rows: " + " / ".join(f"{v:.4f}" for v in out), flush=True)
