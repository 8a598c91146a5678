#!/usr/bin/env python3
"""The map build of C2 (500k points) and C3 / C4 (2M points) through the partitioned bulk insert and through the
four-kernel path (LOM_OPT_NO_BULK_INSERT): all kernels of one insert into an empty map under one HIP event pair
(lom_profile_insert, scratch and slabs sized before the bracket), a few repetitions each on fresh maps, then the two
maps compared bytewise.  A third line per size: the same points in scan order (sorted by azimuth cell -- what a
keyframe built from registered scans looks like) instead of the synthetic map's random order."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lidar_odometry_demo_amd as lom
from lidar_odometry_demo_amd import synth

REPS = int(os.environ.get("LOM_AB_REPS", "5"))
boxes = synth.make_boxes()


def digest(g):
    xyz, nrm = g.getCloud()
    return hashlib.sha256(xyz.tobytes() + nrm.tobytes()).hexdigest()[:16], g.size(), g.pointCount()


def build(d_xyz, d_nrm, no_bulk):
    times, last = [], None
    for _ in range(REPS):
        g = lom.VoxelGrid(0.5, 20)
        g.setOption(lom.capi.OPT_NO_BULK_INSERT, 1 if no_bulk else 0)
        times.append(g.profileInsert(d_xyz.data_ptr(), d_nrm.data_ptr(), d_xyz.shape[0]))
        last = g
    return times, last


for n in (500_000, 2_000_000):
    mp, mn = synth.make_map_points(n, boxes=boxes)
    order = np.lexsort((np.hypot(mp[:, 0], mp[:, 1]), np.floor(np.degrees(np.arctan2(mp[:, 1], mp[:, 0])) * 5)))
    for label, xyz, nrm in (("random order", mp, mn), ("azimuth order", np.ascontiguousarray(mp[order]),
                                                       np.ascontiguousarray(mn[order]))):
        d_xyz, d_nrm = torch.from_numpy(xyz).to("cuda:0"), torch.from_numpy(nrm).to("cuda:0")
        torch.cuda.synchronize()
        tb, gb = build(d_xyz, d_nrm, False)
        tf, gf = build(d_xyz, d_nrm, True)
        db, df = digest(gb), digest(gf)
        print(f"{n:>9} points, {label}: bulk {min(tb):8.1f} us (median {np.median(tb):8.1f})   four-kernel {min(tf):8.1f} us "
              f"(median {np.median(tf):8.1f})   voxels {db[1]}, stored {db[2]}, maps {'EQUAL' if db == df else 'DIFFER'} "
              f"{db[0]}, redone {gb.debugCounter()}", flush=True)
