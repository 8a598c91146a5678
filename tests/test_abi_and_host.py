"""CPU-only tests of the product: the C-ABI library loads and exports every
symbol include/lidar_odometry_amd.h declares, fails loudly without a GPU, and
its host logic (Pose3D algebra, align driver) matches the oracle.  No compute
calls into HIP kernels here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import scenes
from tests.conftest import ROOT


def test_library_exports_every_declared_symbol(lom):
    hdr = open(os.path.join(ROOT, "include", "lidar_odometry_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lom_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert sorted(lom.capi.EXPORTED) == declared
    L = lom.capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.lom_abi_version() == 2


def test_no_gpu_fails_loudly(lom):
    L = lom.capi.lib()
    if L.lom_device_count() > 0:
        pytest.skip("a GPU is visible; the loud-failure path is for CPU-only hosts")
    with pytest.raises(lom.LomError) as e:
        lom.VoxelGrid(0.5, 20)
    assert e.value.code == lom.capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_create_argument_errors(lom):
    L = lom.capi.lib()
    h = C.c_void_p()
    assert L.lom_map_create(0.0, 20, 0, 0, C.byref(h)) == lom.capi.ERR_ARG
    assert L.lom_map_create(0.5, 0, 0, 0, C.byref(h)) == lom.capi.ERR_ARG
    assert L.lom_map_create(0.5, 20, 0, 0, None) == lom.capi.ERR_ARG
    assert L.lom_map_size(None) == lom.capi.ERR_ARG


def test_pose_algebra_bit_exact_vs_oracle_and_reference_vectors(lom, oracle):
    for (t1, q1), (t2, q2) in scenes.pose_pairs():
        a, b = lom.Pose3D(t1, q1), lom.Pose3D(t2, q2)
        oa, ob = oracle.Pose3D(t1, q1), oracle.Pose3D(t2, q2)
        for got, want in ((a.compose(b), oa.compose(ob)), (a.relativeTo(b), oa.relativeTo(ob)),
                          (a.inverse(), oa.inverse()), (b.inverse(), ob.inverse())):
            assert got.translation.tobytes() == want.translation.tobytes()
            assert got.rotation.tobytes() == want.rotation.tobytes()
        assert a.rotationMatrix().tobytes() == oa.rotationMatrix().tobytes()
        # test.cpp:137: compose translation against Isometry3f algebra, 1e-6 (see test_oracle.py)
        M = scenes.se3_matrix(t1, q1) @ scenes.se3_matrix(t2, q2)
        ulp = float(np.spacing(np.float32(np.abs(M[:3, 3]).max())))
        assert np.linalg.norm(M[:3, 3].astype(np.float64) - a.compose(b).translation) < max(1e-6, 2 * ulp)


def test_rigid_transform_vectors(lom, oracle):
    nrm = scenes.RIGID_POINTS[::-1].copy()
    for t, q in scenes.rigid_poses():
        got = lom.transform_points(lom.Pose3D(t, q), scenes.RIGID_POINTS)
        want = oracle.transform_points(oracle.Pose3D(t, q), scenes.RIGID_POINTS)
        assert got.tobytes() == want.tobytes()
        M = scenes.se3_matrix(t, q).astype(np.float64)
        ref = scenes.RIGID_POINTS.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
        assert np.abs(got - ref).max() < 5e-7      # test.cpp:186 (1e-7 between two f32 paths)
        g2, n2 = lom.transform_points(lom.Pose3D(t, q), scenes.RIGID_POINTS, nrm)
        w2, wn2 = oracle.transform_points(oracle.Pose3D(t, q), scenes.RIGID_POINTS, nrm)
        assert g2.tobytes() == w2.tobytes() and n2.tobytes() == wn2.tobytes()


def _oracle_hooks(lom, oracle, shard, allreduce=None):
    """lom_align_hooks whose evaluators are the oracle's C functions (test stand-in
    for the HIP kernels, so the product's host driver can run without a GPU)."""
    OL = oracle.lib()
    me = lom.capi.MATCH_EVAL_FN(C.cast(OL.orc_shard_match_eval, C.c_void_p).value)
    ef = lom.capi.EVAL_FIXED_FN(C.cast(OL.orc_shard_eval_fixed, C.c_void_p).value)
    ar = lom.capi.ALLREDUCE_FN(allreduce) if allreduce else lom.capi.ALLREDUCE_FN()
    return lom.capi.AlignHooks(shard.handle, me, ef, ar), (me, ef, ar)


def _driver_align(lom, hooks, guess_t=(0, 0, 0), guess_q=(1, 0, 0, 0)):
    ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
    st = lom.capi.AlignStats()
    rc = lom.capi.lib().lom_align_with_hooks(C.byref(hooks), lom.capi.f3(guess_t), lom.capi.f4(guess_q), ot, oq,
                                             C.byref(st))
    assert rc == 0, rc
    return np.array(ot[:], np.float32), np.array(oq[:], np.float32), st.asdict()


def test_host_align_driver_matches_oracle_align(lom, oracle):
    """Product LM/outer loop (reduced 6x6 Cholesky) vs oracle (row-wise Householder QR):
    poses within 1e-4 m / 1e-4 rad, same outer-iteration count."""
    sm = scenes.small_synth_case()
    g = oracle.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    shard = oracle.Shard(g, sm["scan"])
    hooks, keep = _oracle_hooks(lom, oracle, shard)
    for gt, gq in (((0, 0, 0), (1, 0, 0, 0)), ((0.05, -0.1, 0.02), scenes.angle_axis_q(0.01, (0, 0, 1)))):
        t, q, st = _driver_align(lom, hooks, gt, gq)
        m = oracle.CloudMatcher()
        ref = m.align(g, sm["scan"], oracle.Pose3D(gt, gq))
        dt, dr = scenes.pose_delta(t, q, ref.translation, ref.rotation)
        assert dt < 1e-4 and dr < 1e-4, (dt, dr)
        assert st["outer_iterations"] == m.stats["outer_iterations"]
        assert st["queries"] == m.stats["queries"]
        assert st["cand_total"] == m.stats["cand_total"]
        assert abs(st["final_cost"] - m.stats["final_cost"]) < 1e-6 * max(1.0, m.stats["final_cost"])


def test_host_align_driver_on_fixture(lom, oracle, fixture_cloud):
    xyz, xyzn = fixture_cloud
    keyframe = oracle.VoxelGrid(0.25, 20)
    keyframe.addCloud(xyzn[:, :3], xyzn[:, 3:])
    vf = oracle.VoxelGrid(0.5, 1)
    vf.addCloudWithoutNormals(xyz)
    sub = vf.getCloudWithoutNormals()
    t, q = scenes.matching_guess_poses()[6]
    guess = oracle.Pose3D(t, q)
    cloud = oracle.transform_points(guess.inverse(), sub)
    shard = oracle.Shard(keyframe, cloud)
    hooks, keep = _oracle_hooks(lom, oracle, shard)
    gt, gq, st = _driver_align(lom, hooks)
    ref = oracle.CloudMatcher().align(keyframe, cloud, oracle.Pose3D())
    dt, dr = scenes.pose_delta(gt, gq, ref.translation, ref.rotation)
    assert dt < 1e-4 and dr < 1e-4, (dt, dr)


def test_zero_points_prior_only(lom, oracle):
    g = oracle.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(np.array([[50, 50, 50]], np.float32))
    shard = oracle.Shard(g, scenes.UNIQUE_POINTS)
    hooks, keep = _oracle_hooks(lom, oracle, shard)
    gq = scenes.angle_axis_q(0.1, (0, 0, 1))
    t, q, st = _driver_align(lom, hooks, (1, 2, 3), gq)
    assert np.allclose(t, (1, 2, 3), atol=1e-6)
    assert abs(abs(float(np.dot(q, gq))) - 1) < 1e-6
    assert st["outer_iterations"] == 5 and st["valid_last"] == 0


def test_hook_error_propagates(lom):
    def bad(user, pt, pq, q, t, out):
        return 7

    me = lom.capi.MATCH_EVAL_FN(bad)
    ef = lom.capi.EVAL_FIXED_FN(lambda user, q, t, out: 0)
    hooks = lom.capi.AlignHooks(None, me, ef, lom.capi.ALLREDUCE_FN())
    ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
    rc = lom.capi.lib().lom_align_with_hooks(C.byref(hooks), lom.capi.f3((0, 0, 0)), lom.capi.f4((1, 0, 0, 0)),
                                             ot, oq, None)
    assert rc == lom.capi.ERR_HOOK
