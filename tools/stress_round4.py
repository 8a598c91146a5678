#!/usr/bin/env python3
"""One-off stress of what round 4 added, against the CPU oracle (tests/test_gpu_parity.py holds the committed subset):
  * bulk inserts (above 65,536 points): random sizes, voxel sizes, caps, cluster shapes from wide to crowded (the crowded
    ones are sent back to the four-kernel path), second batches over the first, batches without normals -- the map
    bytewise;
  * the second of two searches of one scan (previous winner's bound, records left alone where the winner stays):
    identical poses, millimetre moves, large moves, searches after a map change -- every entry;
  * radius cleanups (erase in place, holes closed at a quarter; half of them with the scan enqueued behind an align),
    points coming back into erased voxels: the map bytewise, searches with their creation indices;
  * aligns: pose within 1e-4 m / 1e-4 rad, iteration and evaluation counts equal.
usage: python tools/stress_round4.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import lidar_odometry_demo_amd as lom  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import scenes  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(20261005)
bad = 0
redone = 0
cleanups = 0
taken_behind = 0


def same_map(g, og):
    if g.size() != og.size() or g.pointCount() != og.pointCount():
        return False
    a, an = g.getCloud()
    b, bn = og.getCloud()
    return a.tobytes() == b.tobytes() and an.tobytes() == bn.tobytes()


def same_pairs(a, b):
    return (np.array_equal(a["index"], b["index"]) and a["sq_dist"].tobytes() == b["sq_dist"].tobytes()
            and a["origin"].tobytes() == b["origin"].tobytes() and a["normal"].tobytes() == b["normal"].tobytes())


for case in range(n_cases):
    K = int(rng.choice([1, 3, 7, 20, 20, 33]))
    voxel = float(rng.choice([0.2, 0.25, 0.4, 0.5, 1.0]))
    g, og = lom.VoxelGrid(voxel, K), oracle.VoxelGrid(voxel, K)
    spread = float(rng.choice([0.05, 0.3, 1.0, 4.0]))
    n_centers = int(rng.choice([3, 40, 500, 5000]))
    centers = rng.uniform(-25, 25, (n_centers, 3))
    for rnd in range(3):
        n = int(rng.integers(66_000, 400_000))
        pts = (centers[rng.integers(0, n_centers, n)] + rng.normal(0, spread, (n, 3))).astype(np.float32)
        if rng.random() < 0.3:
            pts[rng.random(n) < 0.2] = pts[0]                      # heavy duplicates
        nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
        if rnd == 2 and rng.random() < 0.5:
            g.addCloudWithoutNormals(pts)
            og.addCloudWithoutNormals(pts)
        else:
            g.addCloud(pts, nrm)
            og.addCloud(pts, nrm)
        if not same_map(g, og):
            bad += 1
            print("MAP MISMATCH", case, rnd, voxel, K, n, n_centers, spread, flush=True)
    redone += g.debugCounter()
    # searches in pairs
    q = (centers[rng.integers(0, n_centers, 6000)] + rng.normal(0, spread + 0.1, (6000, 3))).astype(np.float32)
    base = (rng.uniform(-0.3, 0.3, 3), scenes.angle_axis_q(rng.uniform(-0.05, 0.05), scenes._unit(rng.standard_normal(3))))
    moves = [((0, 0, 0), 0.0), ((1e-3, -1e-3, 5e-4), 1e-4), ((0.02, 0.01, -0.01), 2e-3), ((0.6, -0.4, 0.2), 0.05)]
    for dt_, da in moves:
        second = (np.asarray(base[0]) + np.asarray(dt_), scenes.angle_axis_q(da, (0, 0, 1)))
        for d in (0.3, 0.9):
            a = g.findMatchingPairsAfter(q, lom.Pose3D(*base), lom.Pose3D(*second), d)
            b = og.findMatchingPairs(q, oracle.Pose3D(*second), d)
            if not same_pairs(a, b):
                bad += 1
                print("SEARCH MISMATCH", case, voxel, K, dt_, da, d, flush=True)
    # radius cleanups that erase in place (holes until a quarter of the slabs are empty, then closed), points coming back
    # into erased voxels, cleanups whose scan ran behind an align: the map bytewise, searches entry by entry
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    for rnd in range(int(rng.integers(3, 9))):
        c = centers[rng.integers(0, n_centers)] + rng.normal(0, 3.0, 3)
        r = float(rng.choice([8.0, 15.0, 25.0, 40.0, 70.0]))
        behind = rng.random() < 0.5
        if behind:
            g.radiusCleanupAfterAlign(r)
            scan_c = np.ascontiguousarray(q[:1500])
            gp = (rng.uniform(-0.03, 0.03, 3), scenes.angle_axis_q(rng.uniform(-0.01, 0.01), (0, 0, 1)))
            p, op = m.align(g, scan_c, lom.Pose3D(*gp)), om.align(og, scan_c, oracle.Pose3D(*gp))
            c = np.asarray(p.translation, np.float32)  # (the oracle's cleanup takes the product's centre: the map is what is compared)
        cleanups_behind_before = g.debugCounter(lom.capi.COUNTER_CLEANUPS_BEHIND_ALIGN)
        g.radiusCleanup(c, r)
        og.radiusCleanup(c, r)
        taken_behind += g.debugCounter(lom.capi.COUNTER_CLEANUPS_BEHIND_ALIGN) - cleanups_behind_before
        cleanups += 1
        if g.size() != og.size():
            bad += 1
            print("SIZE MISMATCH after cleanup", case, rnd, g.size(), og.size(), flush=True)
        back = rng.integers(0, n_centers, 20_000)
        pts = (centers[back] + rng.normal(0, spread, (len(back), 3))).astype(np.float32)
        nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
        g.addCloud(pts, nrm)
        og.addCloud(pts, nrm)
        if not same_map(g, og):
            bad += 1
            print("MAP MISMATCH after cleanup + insert", case, rnd, voxel, K, r, flush=True)
        a = g.findMatchingPairs(q[:3000], lom.Pose3D(*base), 0.3)
        b = og.findMatchingPairs(q[:3000], oracle.Pose3D(*base), 0.3)
        if not (same_pairs(a, b) and np.array_equal(a["index"], b["index"])):
            bad += 1
            print("SEARCH MISMATCH after cleanup", case, rnd, voxel, K, r, flush=True)
    # an align on this map
    scan = np.ascontiguousarray(q[: int(rng.choice([300, 2000, 6000]))])
    guess = (rng.uniform(-0.05, 0.05, 3), scenes.angle_axis_q(rng.uniform(-0.01, 0.01), (0, 0, 1)))
    p, op = m.align(g, scan, lom.Pose3D(*guess)), om.align(og, scan, oracle.Pose3D(*guess))
    dt, dr = scenes.pose_delta(p.translation, p.rotation, op.translation, op.rotation)
    if (dt > 1e-4 or dr > 1e-4 or m.stats["outer_iterations"] != om.stats["outer_iterations"]
            or m.stats["lm_iterations"] != om.stats["lm_iterations"] or m.stats["evaluations"] != om.stats["points_evaluated"]):
        bad += 1
        print("ALIGN MISMATCH", case, voxel, K, dt, dr, m.stats["outer_iterations"], om.stats["outer_iterations"], flush=True)
    print(f"case {case}: voxel {voxel} cap {K} centers {n_centers} spread {spread}: voxels {g.size()}, bulk inserts sent back so far {redone}, "
          f"mismatches so far {bad}", flush=True)
print(f"{n_cases} cases: mismatches {bad}; bulk inserts sent back to the four-kernel path {redone}; radius cleanups {cleanups}, "
      f"{taken_behind} of them with the scan that ran behind an align")
sys.exit(1 if bad else 0)
