#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build
container (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Outputs (data only -- inputs and expected outputs, no reference source):
  intersection00056_xyzn.npy   (n,6) f32: the one data file the reference ships
        (test/test_data/intersection00056.pcd, x/y/z columns), with normals
        from a radius-0.25 PCA restating what test/test.cpp:196-224 does with
        pcl::NormalEstimation; rows whose normal is NaN are dropped as in
        test.cpp:219-221.  A second array holds the full xyz cloud.
  c1_matching_test.json        MatchingTest protocol (test/test.cpp:226-262) run
        by the CPU oracle on that cloud: 7 guess poses -> final poses + stats.
  synth_small.json             seeded synthetic scene (8 beams x 256 az vs 40k-pt
        map): per-query winner indices (hash + first 64) and final pose from the
        CPU oracle.
The oracle-generated numbers pin GPU-vs-oracle parity and guard regressions;
they are NOT outputs of the reference binary (which cannot be built here).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from lidar_odometry_demo_amd import synth  # noqa: E402
from tests import scenes  # noqa: E402

PCD = "/root/reference/test/test_data/intersection00056.pcd"


def read_pcd_xyz(path):
    """Binary PCD v0.7 reader honouring FIELDS/SIZE/COUNT (fixture: 32 B/point)."""
    with open(path, "rb") as f:
        raw = f.read()
    head_end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    hdr = {}
    for line in raw[:head_end].decode("ascii").splitlines():
        if line and not line.startswith("#"):
            k, *v = line.split()
            hdr[k] = v
    sizes = [int(s) for s in hdr["SIZE"]]
    counts = [int(c) for c in hdr["COUNT"]]
    offs, o = {}, 0
    for name, s, c in zip(hdr["FIELDS"], sizes, counts):
        offs.setdefault(name, o)
        o += s * c
    rec = o
    n = int(hdr["POINTS"][0])
    body = np.frombuffer(raw, dtype=np.uint8, count=n * rec, offset=head_end).reshape(n, rec)
    cols = [body[:, offs[a]:offs[a] + 4].copy().view("<f4")[:, 0] for a in ("x", "y", "z")]
    return np.stack(cols, axis=1).astype(np.float32)


def main():
    xyz = read_pcd_xyz(PCD)
    assert xyz.shape == (59691, 3), xyz.shape
    nrm = scenes.estimate_normals_radius(xyz, 0.25)
    ok = ~np.isnan(nrm).any(axis=1)
    xyzn = np.concatenate([xyz[ok], nrm[ok]], axis=1).astype(np.float32)
    np.save(os.path.join(HERE, "intersection00056_xyz.npy"), xyz)
    np.save(os.path.join(HERE, "intersection00056_xyzn.npy"), xyzn)
    print("fixture:", xyz.shape, "with normals:", xyzn.shape)

    res = scenes.run_matching_test(O, xyz, xyzn)
    with open(os.path.join(HERE, "c1_matching_test.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("c1:", res["keyframe_voxels"], res["keyframe_points"], res["source_points"])

    sm = scenes.small_synth_case()
    g = O.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    corr = g.findMatchingPairs(sm["scan"], O.Pose3D(), 0.3)
    m = O.CloudMatcher()
    pose = m.align(g, sm["scan"], O.Pose3D())
    idx = corr["index"].astype("<i8")
    out = {
        "scan_points": int(len(sm["scan"])),
        "map_voxels": g.size(),
        "map_points": g.pointCount(),
        "winner_sha256": hashlib.sha256(idx.tobytes()).hexdigest(),
        "winner_first64": idx[:64].tolist(),
        "n_valid": int((idx >= 0).sum()),
        "cand_total": int(corr["n_cand"].sum()),
        "occ_total": int(corr["n_occ"].sum()),
        "final_t": [float(v) for v in pose.translation],
        "final_q_wxyz": [float(v) for v in pose.rotation],
        "stats": {k: v for k, v in m.stats.items() if not k.endswith("seconds")},
    }
    with open(os.path.join(HERE, "synth_small.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("synth_small:", out["n_valid"], out["final_t"])


if __name__ == "__main__":
    main()
