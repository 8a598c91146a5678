#!/usr/bin/env python3
"""One-off stress: random sub-scans, guesses and map parameters; the device-resident loop against
the host-driven loop of the same library (LOM_HOST_LM=1) -- iteration counts equal, poses < 1e-6."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import lidar_odometry_demo_amd as lom  # noqa: E402
from tests import scenes  # noqa: E402

rng = np.random.default_rng(20261004)
sm = scenes.small_synth_case()
worst = (0.0, 0.0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mismatch = 0
for case in range(n_cases):
    voxel = float(rng.choice([0.5, 0.37, 1.0, 0.25]))
    K = int(rng.choice([20, 7, 33, 1]))
    g = lom.VoxelGrid(voxel, K)
    keep = rng.random(len(sm["map_xyz"])) < rng.uniform(0.2, 1.0)
    g.addCloud(sm["map_xyz"][keep], sm["map_nrm"][keep])
    m = lom.CloudMatcher()
    for trial in range(3):
        n = int(rng.choice([0, 1, 5, 64, 513, 1500, len(sm["scan"])]))
        sel = np.sort(rng.choice(len(sm["scan"]), n, replace=False))
        scan = np.ascontiguousarray(sm["scan"][sel]).reshape(-1, 3)
        scale = float(rng.choice([0.0, 0.02, 0.15, 0.3]))
        t = rng.uniform(-1, 1, 3) * scale
        q = scenes.angle_axis_q(rng.uniform(-0.3, 0.3) * scale, scenes._unit(rng.standard_normal(3)))
        os.environ.pop("LOM_HOST_LM", None)
        dev = m.align(g, scan, lom.Pose3D(t, q))
        ds = dict(m.stats)
        os.environ["LOM_HOST_LM"] = "1"
        host = m.align(g, scan, lom.Pose3D(t, q))
        hs = dict(m.stats)
        os.environ.pop("LOM_HOST_LM", None)
        dt, dr = scenes.pose_delta(dev.translation, dev.rotation, host.translation, host.rotation)
        worst = (max(worst[0], dt), max(worst[1], dr))
        same = all(ds[k] == hs[k] for k in ("outer_iterations", "lm_iterations", "evaluations", "queries", "cand_total", "valid_last"))
        if not same or dt > 1e-6 or dr > 1e-6:
            mismatch += 1
            print("MISMATCH", case, trial, n, voxel, K, dt, dr, {k: (ds[k], hs[k]) for k in ("outer_iterations", "lm_iterations", "evaluations")}, flush=True)
print(f"{n_cases * 3} aligns, mismatches {mismatch}, worst pose delta {worst[0]:.3e} m {worst[1]:.3e} rad")
sys.exit(1 if mismatch else 0)
