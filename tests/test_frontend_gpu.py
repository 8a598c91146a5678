"""The per-frame front end on the device (csrc/frontend.hip: time normalisation, deskew, classifier, range
filter -- SURVEY.md 8f rows f2 / f3) against the oracle's restatement (oracle/pipeline.c), bit for bit.
The reference holds no vector for these stages (parity unpinned); the oracle follows
src/utils/{point_time_normalize,cloud_transform,cloud_classifier,range_filter}.h line by line."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

from lidar_odometry_demo_amd import synth
from tests import scenes

pytestmark = pytest.mark.gpu


def _oracle_chain(oracle, frame, start, end, lo, hi):
    norm = oracle.pointTimeNormalize(frame)
    desk = oracle.transformNonRigid(norm, oracle.Pose3D(*start), oracle.Pose3D(*end))
    xyz, nrm, nu, grid = oracle.classify(desk)
    fx, fn = oracle.rangeFilter(xyz, nrm, lo, hi)
    return desk, len(xyz), fx, fn, grid


POSES = [
    (((0, 0, 0), (1, 0, 0, 0)), ((0, 0, 0), (1, 0, 0, 0))),                                    # identity: linear branch of slerp
    (((0.45, -0.02, 0.01), scenes.angle_axis_q(0.0087, (0, 0, 1))), ((0, 0, 0), (1, 0, 0, 0))),   # a frame of the sequence
    (((0.3, -0.1, 0.02), scenes.angle_axis_q(0.3, scenes._unit((0.1, 0.2, 1.0)))), ((0, 0, 0), (1, 0, 0, 0))),
    (((0.3, -0.1, 0.02), -scenes.angle_axis_q(0.05, scenes._unit((0.1, 0.2, 1.0)))), ((0.01, 0, 0), (1, 0, 0, 0))),  # dot < 0
    (((0, 0, 0), scenes.angle_axis_q(1.9, (0, 0, 1))), ((0, 0, 0), (1, 0, 0, 0))),               # theta > pi/4: reduced-argument branch of sinf
]


@pytest.mark.parametrize("pi", range(len(POSES)))
def test_frontend_bit_exact_vs_oracle(lom, oracle, pi):
    fe = lom.FrontEnd()
    start, end = POSES[pi]
    for k in (0, 3, 17):
        frame = synth.make_sequence_frame(k)
        desk, n_planar, fx, fn, grid = _oracle_chain(oracle, frame, start, end, 4.0, 80.0)
        got = fe.process(frame, lom.Pose3D(*start), lom.Pose3D(*end), 4.0, 80.0)
        assert not got["redo_on_host"]
        assert got["deskewed"].tobytes() == desk.tobytes()
        assert got["grid"] == grid
        assert got["planar_points"] == n_planar
        assert got["xyz"].tobytes() == fx.tobytes() and got["normals"].tobytes() == fn.tobytes()
        assert len(fx) > 1000


def test_frontend_no_filter_equals_classifier(lom, oracle):
    """min 0 / max inf: the filtered cloud IS the classifier's planar cloud."""
    fe = lom.FrontEnd()
    frame = synth.make_sequence_frame(5)
    ident = ((0, 0, 0), (1, 0, 0, 0))
    desk = oracle.transformNonRigid(oracle.pointTimeNormalize(frame), oracle.Pose3D(), oracle.Pose3D())
    xyz, nrm, _, grid = oracle.classify(desk)
    got = fe.process(frame, lom.Pose3D(*ident), lom.Pose3D(*ident), 0.0, 3.0e18)
    assert got["planar_points"] == len(xyz) == len(got["xyz"])
    assert got["xyz"].tobytes() == xyz.tobytes() and got["normals"].tobytes() == nrm.tobytes()


def test_frontend_ragged_rings_and_small_frames(lom, oracle):
    """Rings of unequal size (W = the largest), ring ids keyed by uint8, missing rings, frames of a few
    points, consecutive frames of different shape on one front end (the cell table must be back at rest)."""
    fe = lom.FrontEnd()
    rng = np.random.default_rng(3)
    ident = ((0, 0, 0), (1, 0, 0, 0))
    base = synth.make_sequence_frame(2)
    cases = []
    keep = rng.random(len(base)) < np.where(base["ring"] % 3 == 0, 0.35, 0.9)       # unequal rings
    cases.append(base[keep])
    sub = base[(base["ring"] != 4) & (base["ring"] != 9)].copy()                      # missing rings
    sub["ring"][sub["ring"] == 2] = 258                                               # 258 & 0xFF == 2
    cases.append(sub)
    cases.append(base[:300].copy())
    cases.append(base[:9].copy())
    cases.append(base)
    for i, f in enumerate(cases):
        f = np.ascontiguousarray(f)
        desk, n_planar, fx, fn, grid = _oracle_chain(oracle, f, *POSES[1], 4.0, 80.0)
        got = fe.process(f, lom.Pose3D(*POSES[1][0]), lom.Pose3D(*POSES[1][1]), 4.0, 80.0)
        assert not got["redo_on_host"], i
        assert got["deskewed"].tobytes() == desk.tobytes(), i
        assert got["grid"] == grid and got["planar_points"] == n_planar, i
        assert got["xyz"].tobytes() == fx.tobytes() and got["normals"].tobytes() == fn.tobytes(), i


def test_device_sinf_equals_libm(lom):
    """The per-point slerp coefficients are sin((1 - t) theta) / sin(theta): the device evaluates glibc's
    own sinf algorithm; here against this box's libm over [0, pi/2] (every float below 2^-9, a million
    above)."""
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.sinf.restype = C.c_float
    libm.sinf.argtypes = [C.c_float]
    fe = lom.FrontEnd()
    rng = np.random.default_rng(11)
    x = np.concatenate([
        np.float32(rng.uniform(0, np.pi / 2, 200_000)),
        np.float32(np.exp(rng.uniform(np.log(1e-6), np.log(1.5), 200_000))),
        np.float32([0.0, 0.75, 0.7853982, 1.5707964, 2.0 ** -12, 2.0 ** -13, 0.74999994]),
    ]).astype(np.float32)
    got = fe.sinf(x)
    want = np.array([libm.sinf(float(v)) for v in x], np.float32)
    assert got.tobytes() == want.tobytes()


def test_frontend_64_beam_frame(lom, oracle):
    """A 64-beam x 2048 frame (~130k points): the organised cloud needs several cells per thread of the in-kernel
    scan (k_fe_planar<4>); still bit-equal to the oracle's chain."""
    fe = lom.FrontEnd()
    frame = synth.make_sequence_frame(4, n_beams=64, n_az=2048)
    assert len(frame) > 100_000
    start, end = POSES[1]
    desk, n_planar, fx, fn, grid = _oracle_chain(oracle, frame, start, end, 4.0, 80.0)
    got = fe.process(frame, lom.Pose3D(*start), lom.Pose3D(*end), 4.0, 80.0)
    assert not got["redo_on_host"]
    assert got["deskewed"].tobytes() == desk.tobytes()
    assert got["grid"] == grid and got["planar_points"] == n_planar
    assert got["xyz"].tobytes() == fx.tobytes() and got["normals"].tobytes() == fn.tobytes()
