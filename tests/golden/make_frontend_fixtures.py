#!/usr/bin/env python3
"""Writes tests/golden/frontend_restated.npz: what the INDEPENDENT numpy restatement of the reference's per-frame
stages (tests/restate_frontend.py, written from src/utils/*.h) gives for three seeded frames of the synthetic
sequence (lidar_odometry_demo_amd/synth.py, seeds fixed there) under three start poses.  Data only: per frame the
organised cloud's shape, the counts, SHA-256 digests of the deskewed coordinates / planar cell indices / normals /
range-filter mask, and every 16th planar point (cell, normal) in clear for diagnostics.  (The restatement is plain
numpy + libm and also runs on the GPU box: the tests compare full arrays against a live run of it, and a CPU test
checks that the live run still reproduces these digests.)  tests/test_frontend_restatement.py compares the oracle, the product's host stages and its
device front end with these.  Also recorded: how many planar points each alternative READING of the C++ text moves
(restate_frontend.py's docstring).  Run in the build container:  python tests/golden/make_frontend_fixtures.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from lidar_odometry_demo_amd import synth  # noqa: E402
from tests import restate_frontend as R  # noqa: E402
from tests import scenes  # noqa: E402

# (frame of the sequence, start pose = relative.inverse() of lidar_odometry.cpp:30, end pose = identity)
CASES = [
    (3, ((0.45, -0.02, 0.01), scenes.angle_axis_q(0.0087, (0, 0, 1)))),
    (11, ((0.3, -0.1, 0.02), scenes.angle_axis_q(0.3, scenes._unit((0.1, 0.2, 1.0))))),
    (17, ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))),
]


DIGESTED = ("deskewed_xyz", "planar_cell", "planar_nrm", "kept")


def digest(a):
    a = np.ascontiguousarray(a.astype(np.int32) if a.dtype == np.int64 else a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    out = {}
    for i, (k, (st, sq)) in enumerate(CASES):
        frame = synth.make_sequence_frame(k)
        base = R.front_end(frame, st, sq, (0, 0, 0), (1, 0, 0, 0))
        out[f"f{i}_frame"] = np.int64(k)
        out[f"f{i}_start"] = np.array(list(st) + list(sq), np.float32)
        out[f"f{i}_shape"] = base["shape"]
        out[f"f{i}_counts"] = np.array([len(frame), len(base["planar_cell"]), int(base["kept"].sum())], np.int64)
        out[f"f{i}_digests"] = np.array([digest(base[key]) for key in DIGESTED])
        out[f"f{i}_sample_cell"] = base["planar_cell"][::16].astype(np.int32)
        out[f"f{i}_sample_nrm"] = base["planar_nrm"][::16]
        moved = {}
        for name, kw in (("atan2=float", dict(atan2="float")), ("sqrt=float", dict(sqrt="float")), ("dot=sse2", dict(dot="sse2"))):
            alt = R.front_end(frame, st, sq, (0, 0, 0), (1, 0, 0, 0), **kw)
            a, b = set(base["planar_cell"].tolist()), set(alt["planar_cell"].tolist())
            common = np.intersect1d(base["planar_cell"], alt["planar_cell"])
            ia = np.searchsorted(base["planar_cell"], common)
            ib = np.searchsorted(alt["planar_cell"], common)
            dn = np.abs(base["planar_nrm"][ia] - alt["planar_nrm"][ib]).max(axis=1) if len(common) else np.zeros(0)
            moved[name] = (len(a ^ b), int((dn > 1e-6).sum()), len(a))
            print(f"frame {k}: reading {name}: {len(a ^ b)} of {len(a)} planar cells differ, {int((dn > 1e-6).sum())} shared "
                  f"cells with a normal off by > 1e-6")
        out[f"f{i}_readings"] = np.array([moved[n] for n in ("atan2=float", "sqrt=float", "dot=sse2")], np.int64)
        print(f"frame {k}: {len(base['planar_cell'])} planar points, {int(base['kept'].sum())} after the range filter, "
              f"organised cloud {tuple(base['shape'])}")
    np.savez_compressed(os.path.join(HERE, "frontend_restated.npz"), **out)


if __name__ == "__main__":
    main()
