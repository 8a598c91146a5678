"""Direct parity of the residual / Jacobian / robust-loss evaluation and of the LM trace.

reference: PointToPlaneErrorAnalytic::Evaluate (src/cloud_matcher.cpp:38-103), ceres::HuberLoss(0.15)
(:134), Ceres' QuaternionManifold plus-Jacobian (:121) and the sums ceres::Solve forms inside DENSE_QR.

A Jacobian column with a wrong scale still converges to the same fixed point, so converged poses alone
do not pin this code.  Here the GPU's 28 reduced sums and 4 counters are compared with the oracle's
(`orc_shard_match_eval` / `orc_shard_eval_fixed`, oracle/oracle.c) sum by sum:

* `lom_debug_eval_sums`  -- k_match + the host-driven path's evaluation kernel at a chosen f64 point;
* `lom_debug_lm_trace`   -- every evaluation of one solve inside k_lm (device-resident path): the point
  the policy proposed and the totals it received after the in-kernel reduction and exchange.

Bars: counters exact; sums within 1e-12 of the oracle's relative to the scale of their block (the two
sides add the same f64 terms in different orders).  The align tests below also hold the LM trace
numbers (recorded iterations, evaluated points, last step norm, final cost) against the oracle's.
"""
import numpy as np
import pytest

from tests import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("counted_search")]

REL = 1e-12


def assert_sums_close(got, ref, what=""):
    """Entry by entry within 1e-12 of the entry's own scale, plus the rounding floor of the block: the device contracts
    a * b + c to one FMA in its f64 residual / Jacobian arithmetic, the oracle (built like the reference, without FMA)
    does not, so an entry the oracle cancels to an exact zero -- one point on an axis-aligned plane -- comes out as a few
    1e-16 of the largest Jacobian terms involved."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got[28:32].tolist() == ref[28:32].tolist(), (what, got[28:32], ref[28:32])   # valid, cand, occ, queries
    # A = sum w J J^T: every entry is bounded by the geometric mean of its diagonal entries
    diag = {a: ref[a * 6 - (a * (a - 1)) // 2] for a in range(6)}
    floor = 64 * np.finfo(np.float64).eps * max(abs(v) for v in diag.values())   # ~1.4e-14 of the largest diagonal entry
    k = 0
    for a in range(6):
        for b in range(a, 6):
            scale = np.sqrt(abs(diag[a] * diag[b]))
            assert abs(got[k] - ref[k]) <= REL * scale + floor, (what, "A", a, b, got[k], ref[k])
            k += 1
    # g = sum w J r  <=  sqrt(A_aa * 2 cost) (Cauchy-Schwarz; rho <= r^2)
    gmax = max(np.sqrt(abs(diag[a]) * 2.0 * max(ref[27], 0.0)) + abs(ref[21 + a]) for a in range(6))
    for a in range(6):
        scale = np.sqrt(abs(diag[a]) * 2.0 * max(ref[27], 0.0)) + abs(ref[21 + a])
        assert abs(got[21 + a] - ref[21 + a]) <= REL * scale + 64 * np.finfo(np.float64).eps * gmax, (what, "g", a, got[21 + a], ref[21 + a])
    assert abs(got[27] - ref[27]) <= REL * max(abs(ref[27]), 1e-300), (what, "cost", got[27], ref[27])


def eval_poses():
    """(f32 pose of the search, f64 point of the evaluation or None = the widened pose)."""
    z = scenes.angle_axis_q(0.004, (0, 0, 1))
    # a pose as it stands between two outer iterations: f32, NOT renormalised (cloud_matcher.cpp:161-167)
    raw = np.array([0.99993, 0.0031, -0.0042, 0.0105], np.float32)
    return [
        (((0, 0, 0), (1, 0, 0, 0)), None),
        (((0.02, -0.01, 0.0), z), None),
        (((0.05, 0.03, -0.02), raw), None),
        # |r| straddling the Huber knee 0.15 (:134): a 0.14 m offset along every axis
        (((0.14, -0.14, 0.14), scenes.angle_axis_q(0.01, scenes._unit((0.3, -0.2, 1.0)))), None),
        # evaluation away from the search pose, non-unit f64 quaternion (an LM candidate)
        (((0.02, -0.01, 0.0), z), ((1.00004, 0.0021, -0.0013, 0.0047), (0.051, -0.032, 0.017))),
    ]


def _check_case(lom, oracle, g, og, scan, tag):
    m = lom.CloudMatcher()
    sh = oracle.Shard(og, scan)
    huber_both = False
    for i, ((pt, pq), point) in enumerate(eval_poses()):
        pose = lom.Pose3D(pt, pq)
        q = np.asarray(pose.rotation, np.float64) if point is None else np.asarray(point[0], np.float64)
        t = np.asarray(pose.translation, np.float64) if point is None else np.asarray(point[1], np.float64)
        ref = sh.match_eval(pose.translation, pose.rotation, q, t)
        got = m.debugEvalSums(g, scan, pose, q, t)
        assert_sums_close(got, ref, (tag, i))
        if ref[28] > 0:
            huber_both = huber_both or ref[27] > 0
    assert huber_both


def test_eval_sums_fixture_c1(lom, oracle, fixture_cloud):
    """C1: the reference's data file, MatchingTest's grids (test.cpp:226-231)."""
    xyz, xyzn = fixture_cloud
    g, og = lom.VoxelGrid(0.25, 20), oracle.VoxelGrid(0.25, 20)
    g.addCloud(xyzn[:, :3], xyzn[:, 3:])
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    vf = oracle.VoxelGrid(0.5, 1)
    vf.addCloudWithoutNormals(xyz)
    _check_case(lom, oracle, g, og, vf.getCloudWithoutNormals(), "C1")


def test_eval_sums_synth_and_ragged_sizes(lom, oracle):
    sm = scenes.small_synth_case()
    g, og = lom.VoxelGrid(0.5, 20), oracle.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    _check_case(lom, oracle, g, og, sm["scan"], "synth")
    m = lom.CloudMatcher()
    for n in (1, 2, 63, 64, 65, 511, 513, 1025):      # fewer points than lanes, ragged last workgroup
        scan = np.ascontiguousarray(sm["scan"][:n])
        sh = oracle.Shard(og, scan)
        pose = lom.Pose3D((0.03, 0.01, -0.01), scenes.angle_axis_q(0.002, (0, 0, 1)))
        ref = sh.match_eval(pose.translation, pose.rotation, pose.rotation.astype(np.float64),
                            pose.translation.astype(np.float64))
        assert_sums_close(m.debugEvalSums(g, scan, pose), ref, ("ragged", n))


def test_eval_sums_zero_normals_and_huber_split(lom, oracle):
    """addCloudWithoutNormals stores (0,0,0) normals (voxel_grid.h:103,107): such matches are valid
    correspondences with zero residual and zero Jacobian.  Then a map of two parallel planes whose
    residuals sit on either side of the Huber knee: the weights must be 1 and 0.15/|r|."""
    sm = scenes.small_synth_case()
    g, og = lom.VoxelGrid(0.5, 20), oracle.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(sm["map_xyz"])
    og.addCloudWithoutNormals(sm["map_xyz"])
    m = lom.CloudMatcher()
    sh = oracle.Shard(og, sm["scan"])
    pose = lom.Pose3D()
    ref = sh.match_eval(pose.translation, pose.rotation, (1, 0, 0, 0), (0, 0, 0))
    got = m.debugEvalSums(g, sm["scan"], pose)
    assert ref[28] > 1000 and not ref[:28].any()
    assert got.tolist() == ref.tolist()
    # two planes z = 0 (normal +z): scan points 0.10 and 0.20 above it -> |r| = 0.10 (inlier), 0.20 (outlier)
    rng = np.random.default_rng(5)
    xy = rng.uniform(-3, 3, (4000, 2)).astype(np.float32)
    plane = np.c_[xy, np.zeros(len(xy), np.float32)].astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(xy), 1))
    g2, og2 = lom.VoxelGrid(0.5, 20), oracle.VoxelGrid(0.5, 20)
    g2.addCloud(plane, nrm)
    og2.addCloud(plane, nrm)
    q = rng.uniform(-2.5, 2.5, (600, 2)).astype(np.float32)
    scan = np.c_[q, np.where(np.arange(len(q)) % 2 == 0, 0.10, 0.20)].astype(np.float32)
    sh2 = oracle.Shard(og2, scan)
    ref = sh2.match_eval((0, 0, 0), (1, 0, 0, 0), (1, 0, 0, 0), (0, 0, 0))
    got = m.debugEvalSums(g2, scan, lom.Pose3D())
    assert_sums_close(got, ref, "huber")
    n_in, n_out = np.sum(scan[:, 2] < 0.15), np.sum(scan[:, 2] > 0.15)
    assert ref[28] == len(scan)
    # sum 0.5 rho: 0.5 r^2 inside the knee, 0.5 (2 a |r| - a^2) outside (ceres::HuberLoss)
    want = n_in * 0.5 * np.float64(np.float32(0.10)) ** 2 + n_out * 0.5 * (2 * 0.15 * np.float64(np.float32(0.20)) - 0.15 ** 2)
    assert abs(got[27] - want) < 1e-9 * want
    # A[5][5] = sum w n_z^2 = n_in + n_out * 0.15 / 0.20
    want_tt = n_in + n_out * 0.15 / np.float64(np.float32(0.20))
    assert abs(got[20] - want_tt) < 1e-9 * want_tt


def _trace_vs_oracle(lom, oracle, g, og, scan, guess, outer_index, tag):
    m = lom.CloudMatcher()
    pose, trace = m.debugLmTrace(g, scan, guess, outer_index)
    assert 1 <= len(trace) <= 5
    sh = oracle.Shard(og, scan)
    x0, s0 = trace[0]
    # evaluation 0 is made at the widened f32 pose of the search (cloud_matcher.cpp:122-131)
    pq, pt = x0[:4].astype(np.float32), x0[4:].astype(np.float32)
    assert pq.astype(np.float64).tolist() == x0[:4].tolist() and pt.astype(np.float64).tolist() == x0[4:].tolist()
    assert_sums_close(s0, sh.match_eval(pt, pq, x0[:4], x0[4:]), (tag, outer_index, 0))
    for e, (x, s) in enumerate(trace[1:], 1):
        ref = sh.eval_fixed(x[:4], x[4:])
        ref[28:31] = s0[28:31]                 # counters travel with evaluation 0 only
        got = s.copy()
        got[28:31] = s0[28:31]
        assert_sums_close(got, ref, (tag, outer_index, e))
        assert not np.array_equal(x, trace[e - 1][0])
    return pose, m.stats, len(trace)


def test_lm_trace_every_evaluation_fixture_c1(lom, oracle, fixture_cloud):
    xyz, xyzn = fixture_cloud
    g, og = lom.VoxelGrid(0.25, 20), oracle.VoxelGrid(0.25, 20)
    g.addCloud(xyzn[:, :3], xyzn[:, 3:])
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    vf = oracle.VoxelGrid(0.5, 1)
    vf.addCloudWithoutNormals(xyz)
    sub = vf.getCloudWithoutNormals()
    om = oracle.CloudMatcher()
    for gi, (t, q) in enumerate(scenes.matching_guess_poses()[:4]):
        guess_cloud = oracle.transform_points(oracle.Pose3D(t, q).inverse(), sub)
        evals = 0
        for outer in (0, 1, 4):
            pose, st, ne = _trace_vs_oracle(lom, oracle, g, og, guess_cloud, lom.Pose3D(), outer, ("C1", gi))
            evals += ne
        ref = om.align(og, guess_cloud, oracle.Pose3D())
        _assert_lm_numbers(st, om.stats)
        dt, dr = scenes.pose_delta(pose.translation, pose.rotation, ref.translation, ref.rotation)
        assert dt < 1e-4 and dr < 1e-4
        assert evals >= 3


def test_lm_trace_synth(lom, oracle):
    sm = scenes.small_synth_case()
    g, og = lom.VoxelGrid(0.5, 20), oracle.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    guess = lom.Pose3D((0.2, -0.2, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))     # needs a sixth outer iteration
    om = oracle.CloudMatcher()
    ref = om.align(og, sm["scan"], oracle.Pose3D(guess.translation, guess.rotation))
    for outer in range(om.stats["outer_iterations"]):
        pose, st, ne = _trace_vs_oracle(lom, oracle, g, og, sm["scan"], guess, outer, "synth")
    _assert_lm_numbers(st, om.stats)
    # a trace request beyond the last outer iteration records nothing
    m = lom.CloudMatcher()
    _, trace = m.debugLmTrace(g, sm["scan"], guess, 30)
    assert trace == []


def test_lm_trace_c2_size(lom, oracle):
    """The same, evaluation by evaluation, on a C2-sized cloud (26.6k points: 512-thread workgroups of the
    device-resident solve and the two-loads-in-flight search; the cases above run the 256-thread / four-loads
    variants that clouds of up to 16,384 points take)."""
    c = scenes.synth_case(16, 1800, 200_000)
    assert len(c["scan"]) > 16384
    g, og = lom.VoxelGrid(0.5, 20), oracle.VoxelGrid(0.5, 20)
    g.addCloud(c["map_xyz"], c["map_nrm"])
    og.addCloud(c["map_xyz"], c["map_nrm"])
    guess = lom.Pose3D((0.05, -0.04, 0.02), scenes.angle_axis_q(0.0175, (0, 0, 1)))
    om = oracle.CloudMatcher(nthreads=8)
    ref = om.align(og, c["scan"], oracle.Pose3D(guess.translation, guess.rotation))
    for outer in (0, 2, om.stats["outer_iterations"] - 1):
        pose, st, ne = _trace_vs_oracle(lom, oracle, g, og, c["scan"], guess, outer, "C2-size")
    _assert_lm_numbers(st, om.stats)
    dt, dr = scenes.pose_delta(pose.translation, pose.rotation, ref.translation, ref.rotation)
    assert dt < 1e-4 and dr < 1e-4


def _assert_lm_numbers(st, ost):
    """Recorded iterations, evaluated points, last step norm and final cost of the whole align against
    the oracle's Ceres restatement."""
    assert st["outer_iterations"] == ost["outer_iterations"]
    assert st["lm_iterations"] == ost["lm_iterations"], (st, ost)
    assert st["evaluations"] == ost["points_evaluated"], (st, ost)
    assert abs(st["last_step_norm"] - ost["last_step_norm"]) < 1e-9, (st["last_step_norm"], ost["last_step_norm"])
    assert abs(st["final_cost"] - ost["final_cost"]) <= 1e-9 * max(1.0, abs(ost["final_cost"]))
    for k in ("queries", "cand_total", "occ_total", "valid_last"):
        assert st[k] == ost[k], k


@pytest.mark.parametrize("seed", range(4))
def test_align_lm_numbers_randomized(lom, oracle, seed):
    rng = np.random.default_rng(900 + seed)
    sm = scenes.small_synth_case()
    voxel = float(rng.choice([0.5, 0.37, 1.0, 0.25]))
    g, og = lom.VoxelGrid(voxel, 20), oracle.VoxelGrid(voxel, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    for trial in range(5):
        n = int(rng.choice([7, 63, 500, len(sm["scan"])]))
        sel = np.sort(rng.choice(len(sm["scan"]), n, replace=False))
        scan = np.ascontiguousarray(sm["scan"][sel])
        scale = float(rng.choice([0.02, 0.15, 0.25]))
        t = rng.uniform(-1, 1, 3) * scale
        q = scenes.angle_axis_q(rng.uniform(-0.2, 0.2) * scale, scenes._unit(rng.standard_normal(3)))
        m.align(g, scan, lom.Pose3D(t, q))
        om.align(og, scan, oracle.Pose3D(t, q))
        _assert_lm_numbers(m.stats, om.stats)


def test_sharding_linearity_of_the_sums(lom, oracle):
    """The path shards by source range: sums over contiguous ranges add up to the sums
    over the whole scan; counters exactly, f64 sums to rounding.  C2-sized scan and map."""
    c = scenes.synth_case(16, 1800, 500_000)
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(c["map_xyz"], c["map_nrm"])
    m = lom.CloudMatcher()
    pose = lom.Pose3D((0.03, -0.02, 0.01), scenes.angle_axis_q(0.004, (0, 0, 1)))
    whole = m.debugEvalSums(g, c["scan"], pose)
    n = len(c["scan"])
    for world in (2, 3, 8):
        parts = [m.debugEvalSums(g, np.ascontiguousarray(c["scan"][n * r // world: n * (r + 1) // world]), pose)
                 for r in range(world)]
        total = np.sum(parts, axis=0)
        assert_sums_close(total, whole, ("linearity", world))
    og = oracle.VoxelGrid(0.5, 20)
    og.addCloud(c["map_xyz"], c["map_nrm"])
    sh = oracle.Shard(og, c["scan"])
    assert_sums_close(whole, sh.match_eval(pose.translation, pose.rotation, pose.rotation.astype(np.float64),
                                           pose.translation.astype(np.float64)), "C2 vs oracle")
