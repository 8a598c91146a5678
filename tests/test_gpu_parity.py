"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs and against the committed golden fixtures.

Bars: bit-exact for the integer/index work (voxel membership, order, winner
index, counters) and for copied f32 payloads; 1e-4 m / 1e-4 rad for poses
(BASELINE.json north_star)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from tests import scenes
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


@pytest.fixture(autouse=True, params=["product", "counted"])
def search_mode(request, monkeypatch):
    """Every test of this module runs twice.  "product": the handles as a caller gets them -- a neighbour voxel that the
    distance bound prunes is not even looked up, and the reference-ALGORITHM counts (n_cand / n_occ, cand_total /
    occ_total: SURVEY.md 8d's cand(q), measurement only -- the reference returns no such thing) read 0.  "counted":
    LOM_COUNT_CANDIDATES=1 at create (= LOM_OPT_COUNT_CANDIDATES): all 27 slots per query, counts equal to the oracle's.
    Winners, distances, poses and iteration counts must equal the oracle's either way."""
    monkeypatch.setenv("LOM_COUNT_CANDIDATES", "1" if request.param == "counted" else "0")
    return request.param


def _counted():
    return os.environ.get("LOM_COUNT_CANDIDATES") == "1"


def _keys(*keys):
    """stats keys to compare with the oracle: the reference-algorithm counts only where the product produced them"""
    return tuple(k for k in keys if _counted() or k not in ("cand_total", "occ_total"))


def _both(lom, oracle, voxel, K):
    return lom.VoxelGrid(voxel, K), oracle.VoxelGrid(voxel, K)


def _assert_same_map(g, og):
    assert g.size() == og.size()
    assert g.pointCount() == og.pointCount()
    xyz, nrm = g.getCloud()
    oxyz, onrm = og.getCloud()
    assert xyz.tobytes() == oxyz.tobytes()
    assert nrm.tobytes() == onrm.tobytes()
    assert g.getSparseCloudWithoutNormals().tobytes() == og.getSparseCloudWithoutNormals().tobytes()
    assert g.getCloudWithoutNormals().tobytes() == oxyz.tobytes()


def _assert_same_pairs(c, oc):
    assert np.array_equal(c["index"], oc["index"])
    assert c["origin"].tobytes() == oc["origin"].tobytes()
    assert c["normal"].tobytes() == oc["normal"].tobytes()
    assert c["sq_dist"].tobytes() == oc["sq_dist"].tobytes()
    if _counted():
        assert np.array_equal(c["n_cand"], oc["n_cand"])
        assert np.array_equal(c["n_occ"], oc["n_occ"])
    else:
        assert not c["n_cand"].any() and not c["n_occ"].any()


# ---- reference unit tests through the product (test.cpp:26-75) -------------

def test_unique_points(lom):
    g = lom.VoxelGrid(0.5, 1)
    g.addCloud(scenes.UNIQUE_POINTS, np.zeros_like(scenes.UNIQUE_POINTS))
    assert g.size() == 7
    xyz, _ = g.getCloud()
    left = [tuple(p) for p in scenes.UNIQUE_POINTS]
    for p in xyz:
        left.remove(tuple(p))
    assert not left


def test_duplicate_points(lom):
    g = lom.VoxelGrid(0.5, 1)
    g.addCloud(scenes.DUPLICATE_POINTS, np.zeros_like(scenes.DUPLICATE_POINTS))
    assert g.size() == 2
    xyz, _ = g.getCloud()
    assert len(xyz) == 2 and tuple(xyz[0]) != tuple(xyz[1])


# ---- insert / export / cleanup parity ---------------------------------------

def test_insert_semantics_small(lom, oracle):
    pts = np.array([[-0.4, 0, 0], [0.4, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [-0.6, 0, 0]], np.float32)
    g, og = _both(lom, oracle, 0.5, 3)
    g.addCloudWithoutNormals(pts)
    og.addCloudWithoutNormals(pts)
    _assert_same_map(g, og)
    assert g.size() == 2 and g.pointCount() == 4


def test_empty_inputs(lom):
    g = lom.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(np.zeros((0, 3), np.float32))
    assert g.size() == 0 and g.pointCount() == 0
    assert len(g.getCloudWithoutNormals()) == 0
    c = g.findMatchingPairs(scenes.UNIQUE_POINTS, lom.Pose3D(), 0.3)
    assert (c["index"] == -1).all()
    g.radiusCleanup((0, 0, 0), 1.0)
    pose = lom.CloudMatcher().align(g, np.zeros((0, 3), np.float32), lom.Pose3D((1, 2, 3), (1, 0, 0, 0)))
    assert np.allclose(pose.translation, (1, 2, 3))


@pytest.mark.parametrize("voxel,K", [(0.25, 20), (0.5, 1), (0.1, 1), (0.2, 20), (0.5, 3)])
def test_insert_parity_fixture(lom, oracle, fixture_cloud, voxel, K):
    xyz, xyzn = fixture_cloud
    g, og = _both(lom, oracle, voxel, K)
    # several calls: later calls must append to voxels created by earlier ones, up to the cap
    n = len(xyzn)
    cuts = [0, n // 3, n // 3 + 1, 2 * n // 3, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g.addCloud(xyzn[a:b, :3], xyzn[a:b, 3:])
        og.addCloud(xyzn[a:b, :3], xyzn[a:b, 3:])
    _assert_same_map(g, og)


def test_insert_interleaved_records(lom, oracle, fixture_cloud):
    """pcl::PointNormal-shaped 48-byte records passed without repacking."""
    _, xyzn = fixture_cloud
    n = 5000
    rec = np.zeros((n, 12), np.float32)
    rec[:, 0:3] = xyzn[:n, :3]
    rec[:, 4:7] = xyzn[:n, 3:]
    g, og = _both(lom, oracle, 0.25, 20)
    g.addCloudInterleaved(rec, 48, 16)
    og.addCloud(xyzn[:n, :3], xyzn[:n, 3:])
    _assert_same_map(g, og)


def test_insert_heavy_duplicates_and_order(lom, oracle):
    """Many points per voxel, shuffled: the kept set is the first K in input order."""
    rng = np.random.default_rng(7)
    pts = (rng.random((20000, 3)) * 2.0 - 1.0).astype(np.float32)   # 4^3..5^3 voxels of 0.5 m
    nrm = rng.standard_normal((20000, 3)).astype(np.float32)
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    _assert_same_map(g, og)
    g.addCloud(pts[::-1].copy(), nrm)          # all voxels already full: no change
    og.addCloud(pts[::-1].copy(), nrm)
    _assert_same_map(g, og)


@pytest.mark.parametrize("voxel", [0.1, 0.3, 0.5])
def test_fused_downsample_equals_grid_of_cap_one(lom, oracle, fixture_cloud, voxel):
    """lom_voxel_downsample == VoxelGrid(voxel, 1).addCloud(cloud); getCloud() (lidar_odometry.cpp:37-47)."""
    _, xyzn = fixture_cloud
    og = oracle.VoxelGrid(voxel, 1)
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    oxyz, onrm = og.getCloud()
    ws = lom.VoxelGrid(1.0, 1)
    xyz, nrm = ws.downsample(xyzn[:, :3], xyzn[:, 3:], voxel)
    assert xyz.tobytes() == oxyz.tobytes() and nrm.tobytes() == onrm.tobytes()
    assert ws.size() == 0                                       # the workspace is left empty
    xyz0, nrm0 = ws.downsample(xyzn[:, :3], None, voxel)        # addCloudWithoutNormals flavour
    assert xyz0.tobytes() == oxyz.tobytes() and not nrm0.any()
    ws.addCloud(xyzn[:100, :3], xyzn[:100, 3:])                 # and is still a working grid
    assert ws.size() > 0
    with pytest.raises(lom.LomError):
        ws.downsample(np.array([[1e9, 0, 0]], np.float32), None, voxel)


@pytest.mark.parametrize("seed", range(6))
def test_randomized_map_and_search_parity(lom, oracle, seed):
    """Ragged random inputs: clustered points (many collisions per voxel), negative coordinates and the
    double-width voxel 0 of the truncating index, random cap / voxel size, several inserts with
    cleanups in between, queries at random poses -- everything compared exactly with the oracle."""
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([1, 2, 3, 7, 20, 33]))
    voxel = float(rng.choice([0.1, 0.25, 0.37, 0.5, 1.0]))
    g, og = _both(lom, oracle, voxel, K)
    for rnd in range(4):
        n = int(rng.integers(1, 6000))
        centers = rng.uniform(-6, 6, size=(int(rng.integers(1, 40)), 3))
        pts = (centers[rng.integers(0, len(centers), n)] + rng.normal(0, rng.uniform(0.01, 0.8), (n, 3))).astype(np.float32)
        pts[rng.random(n) < 0.05] *= np.float32(0.01)          # crowd the double-width voxel around 0
        nrm = rng.standard_normal((n, 3)).astype(np.float32)
        if rng.random() < 0.3:
            g.addCloudWithoutNormals(pts)
            og.addCloudWithoutNormals(pts)
        else:
            g.addCloud(pts, nrm)
            og.addCloud(pts, nrm)
        _assert_same_map(g, og)
        if rnd in (1, 2):
            c = rng.uniform(-3, 3, 3).astype(np.float32)
            r = float(rng.uniform(2, 9))
            g.radiusCleanup(c, r)
            og.radiusCleanup(c, r)
            _assert_same_map(g, og)
        q = (rng.uniform(-7, 7, (int(rng.integers(1, 3000)), 3))).astype(np.float32)
        pose = (rng.uniform(-0.5, 0.5, 3), scenes.angle_axis_q(rng.uniform(-0.3, 0.3), scenes._unit(rng.standard_normal(3))))
        d = float(rng.choice([0.05, 0.3, 0.3, 1.0]))
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(*pose), d), og.findMatchingPairs(q, oracle.Pose3D(*pose), d))


def test_large_cap_voxels(lom, oracle):
    """max_points beyond 2048: the search's other prefix path (27 x K no longer fits 16 bits per set), voxels that
    hold thousands of points, caps that cut in the middle of a batch; map, pairs and align against the oracle."""
    rng = np.random.default_rng(77)
    for K in (3000, 2049, 40000):
        g, og = _both(lom, oracle, 2.0, K)
        centers = rng.uniform(-5, 5, (6, 3))
        for rnd in range(2):
            n = 9000
            pts = (centers[rng.integers(0, len(centers), n)] + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
            nrm = scenes._unit(rng.standard_normal((n, 3))).astype(np.float32)
            g.addCloud(pts, nrm)
            og.addCloud(pts, nrm)
            _assert_same_map(g, og)
        assert g.pointCount() // max(g.size(), 1) > 100            # crowded voxels
        q = (centers[rng.integers(0, len(centers), 800)] + rng.normal(0, 0.6, (800, 3))).astype(np.float32)
        pose = ((0.02, -0.03, 0.01), scenes.angle_axis_q(0.01, (0, 0, 1)))
        for d in (0.05, 0.3):
            _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(*pose), d), og.findMatchingPairs(q, oracle.Pose3D(*pose), d))
        m, om = lom.CloudMatcher(), oracle.CloudMatcher()
        p = m.align(g, q, lom.Pose3D(*pose))
        r = om.align(og, q, oracle.Pose3D(*pose))
        dt, dr = scenes.pose_delta(p.translation, p.rotation, r.translation, r.rotation)
        assert dt < 1e-4 and dr < 1e-4 and m.stats["outer_iterations"] == om.stats["outer_iterations"]
        assert m.stats["cand_total"] == (om.stats["cand_total"] if _counted() else 0)


def test_lattice_of_equal_parities(lom, oracle):
    """A regular lattice: every voxel index even (then a second batch at odd indices) -- keys that differ in few bits
    are what a multiplicative slot hash has to spread over the table.  Map and pairs against the oracle."""
    g, og = _both(lom, oracle, 0.5, 4)
    ax = np.arange(-24, 24, dtype=np.float32)                        # 48^3 = 110,592 voxels, index = 2 * k
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3) + np.float32(0.1)
    rng = np.random.default_rng(5)
    pts = np.ascontiguousarray(pts[rng.permutation(len(pts))])
    nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    assert g.size() == og.size() == 48 ** 3
    _assert_same_map(g, og)
    odd = np.ascontiguousarray(pts[::5] + np.float32(0.5))
    g.addCloudWithoutNormals(odd)
    og.addCloudWithoutNormals(odd)
    _assert_same_map(g, og)
    q = rng.uniform(-12, 12, (4000, 3)).astype(np.float32)
    for d in (0.3, 1.0):
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(), d), og.findMatchingPairs(q, oracle.Pose3D(), d))


@pytest.mark.parametrize("slots", [2, 64])
def test_table_size_never_changes_results(lom, oracle, monkeypatch, slots):
    """LOM_TABLE_SLOTS_PER_VOXEL (read at create) sizes the slot table after a bulk insert: 2 slots per voxel means long
    probe chains in the search and in later inserts, 64 a table mostly empty.  Map, pairs and an align against the
    oracle either way."""
    monkeypatch.setenv("LOM_TABLE_SLOTS_PER_VOXEL", str(slots))
    rng = np.random.default_rng(91)
    g, og = _both(lom, oracle, 0.25, 6)
    centers = rng.uniform(-8, 8, (400, 3))
    pts = (centers[rng.integers(0, len(centers), 90000)] + rng.normal(0, 0.6, (90000, 3))).astype(np.float32)   # bulk: > 65536
    nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    _assert_same_map(g, og)
    more = np.ascontiguousarray(pts[::9] + np.float32(0.3))          # a small batch into the (possibly dense) table
    g.addCloudWithoutNormals(more)
    og.addCloudWithoutNormals(more)
    _assert_same_map(g, og)
    q = (centers[rng.integers(0, len(centers), 6000)] + rng.normal(0, 0.7, (6000, 3))).astype(np.float32)
    pose = ((0.03, -0.02, 0.01), scenes.angle_axis_q(0.015, (0, 0, 1)))
    for d in (0.3, 1.0):
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(*pose), d), og.findMatchingPairs(q, oracle.Pose3D(*pose), d))
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    p = m.align(g, q, lom.Pose3D(*pose))
    r = om.align(og, q, oracle.Pose3D(*pose))
    dt, dr = scenes.pose_delta(p.translation, p.rotation, r.translation, r.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and m.stats["outer_iterations"] == om.stats["outer_iterations"]


def test_more_than_a_million_voxels(lom, oracle):
    """A map beyond 16 x 65536 voxels: the cleanup's multi-launch scan (its in-kernel scan covers 1,048,576), table
    growth while inserting, bulk inserts of more than 65536 points, a second batch over the first; bytewise against
    the oracle before and after two cleanups."""
    rng = np.random.default_rng(2026)
    g, og = _both(lom, oracle, 0.1, 3)
    pts = rng.uniform(-60, 60, (1_250_000, 3)).astype(np.float32)
    nrm = rng.standard_normal((len(pts), 3)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    assert g.size() == og.size() > 16 * 65536
    more = np.ascontiguousarray(pts[::7] + np.float32(0.01))
    g.addCloudWithoutNormals(more)
    og.addCloudWithoutNormals(more)
    _assert_same_map(g, og)
    for center, r in (((5, -3, 2), 75.0), ((0, 0, 0), 30.0)):
        c = np.array(center, np.float32)
        g.radiusCleanup(c, r)
        og.radiusCleanup(c, r)
        _assert_same_map(g, og)
    q = rng.uniform(-25, 25, (5000, 3)).astype(np.float32)
    _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(), 0.3), og.findMatchingPairs(q, oracle.Pose3D(), 0.3))


def test_out_of_range_rejected_and_nothing_inserted(lom):
    g = lom.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(scenes.UNIQUE_POINTS)
    bad = np.array([[0.3, 0.3, 0.3], [1e9, 0, 0]], np.float32)
    with pytest.raises(lom.LomError) as e:
        g.addCloudWithoutNormals(bad)
    assert e.value.code == lom.capi.ERR_RANGE
    nanp = np.array([[np.nan, 0, 0]], np.float32)
    with pytest.raises(lom.LomError):
        g.addCloudWithoutNormals(nanp)
    assert g.size() == 7 and g.pointCount() == 7


def test_set_voxel_size_clears_and_set_max_points(lom):
    g = lom.VoxelGrid(0.5, 1)
    g.addCloudWithoutNormals(scenes.UNIQUE_POINTS)
    g.setVoxelSize(0.25)                       # voxel_grid.h:61-66
    assert g.size() == 0 and g.pointCount() == 0
    g.setMaxPoints(5)
    g.addCloudWithoutNormals(np.repeat(scenes.UNIQUE_POINTS, 7, axis=0))
    assert g.size() == 7 and g.pointCount() == 35


def test_set_max_points_on_a_map_that_holds_voxels(lom, oracle):
    """voxel_grid.h:56-59 sets max_points_ and nothing else: stored voxels keep what they hold, and :86-90 appends
    to a voxel only while size() < max_points_.  Raise (the slabs are re-strided), lower below what voxels already
    hold, raise again, with inserts, a cleanup and searches in between -- map, pairs and an align against the oracle."""
    rng = np.random.default_rng(31)
    g, og = _both(lom, oracle, 0.5, 3)
    centers = rng.uniform(-4, 4, (30, 3))

    def batch(n):
        pts = (centers[rng.integers(0, len(centers), n)] + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
        nrm = scenes._unit(rng.standard_normal((n, 3))).astype(np.float32)
        g.addCloud(pts, nrm)
        og.addCloud(pts, nrm)
        _assert_same_map(g, og)

    batch(4000)
    full3 = g.pointCount()
    for k in (9, 2, 2, 6, 40, 1):
        g.setMaxPoints(k)
        og.setMaxPoints(k)
        _assert_same_map(g, og)                      # the call itself changes nothing
        before = g.pointCount()
        batch(3000)
        if k <= 3:
            assert g.pointCount() - before <= 3000   # only new voxels (and those below k) take points
        if k in (6, 40):                             # (40: right after the slabs were re-strided)
            c = np.array((0.5, -0.5, 0.2), np.float32)
            g.radiusCleanup(c, 4.0 if k == 6 else 5.0)
            og.radiusCleanup(c, 4.0 if k == 6 else 5.0)
            _assert_same_map(g, og)
            batch(1500)
    assert g.pointCount() > full3
    q = (centers[rng.integers(0, len(centers), 1500)] + rng.normal(0, 0.5, (1500, 3))).astype(np.float32)
    pose = ((0.02, -0.03, 0.01), scenes.angle_axis_q(0.01, (0, 0, 1)))
    for d in (0.3, 1.0):
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(*pose), d), og.findMatchingPairs(q, oracle.Pose3D(*pose), d))
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    p = m.align(g, q, lom.Pose3D(*pose))
    r = om.align(og, q, oracle.Pose3D(*pose))
    dt, dr = scenes.pose_delta(p.translation, p.rotation, r.translation, r.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and m.stats["outer_iterations"] == om.stats["outer_iterations"]


def test_radius_cleanup_parity(lom, oracle, fixture_cloud):
    _, xyzn = fixture_cloud
    g, og = _both(lom, oracle, 0.25, 20)
    g.addCloud(xyzn[:, :3], xyzn[:, 3:])
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    for center, r in (((0, 5, 0), 80.0), ((10, 5, -20), 40.0), ((10, 5, -20), 40.0), ((0, 0, 0), 5.0)):
        g.radiusCleanup(center, r)
        og.radiusCleanup(center, r)
        _assert_same_map(g, og)
    # insert after cleanup: slabs were compacted, creation order continues
    g.addCloud(xyzn[:3000, :3], xyzn[:3000, 3:])
    og.addCloud(xyzn[:3000, :3], xyzn[:3000, 3:])
    _assert_same_map(g, og)
    c = g.findMatchingPairs(xyzn[:2000, :3], lom.Pose3D(), 0.3)
    oc = og.findMatchingPairs(xyzn[:2000, :3], oracle.Pose3D(), 0.3)
    _assert_same_pairs(c, oc)


@pytest.mark.parametrize("dense", [False, True])
def test_radius_cleanup_leaves_holes_until_they_are_a_quarter(lom, oracle, monkeypatch, dense):
    """voxel_grid.h:236-246 erases voxels; the slabs of the others do not move for that: an erased voxel's slab stays in the
    creation order, empty, its key a claimed slot without a voxel (k_cleanup_mark) -- until a quarter of the slabs are
    holes, then they are closed (k_compact).  A walk of a small radius across the map: every step erases a few voxels and
    re-creates some erased earlier (at the END of the creation order, as after an erase in the reference); size, exports,
    searches (with the creation index counting held voxels only) and aligns equal the oracle's at every step -- with the
    holes (default) and with LOM_DENSE_CLEANUP=1 (every cleanup closes its holes at once, as until round 4)."""
    if dense:
        monkeypatch.setenv("LOM_DENSE_CLEANUP", "1")
    sm = scenes.small_synth_case()
    g, og = _both(lom, oracle, 0.5, 20)
    xyz, nrm = sm["map_xyz"], sm["map_nrm"]
    g.addCloud(xyz, nrm)
    og.addCloud(xyz, nrm)
    rng = np.random.default_rng(77)
    q = np.ascontiguousarray(xyz[rng.choice(len(xyz), 3000, replace=False)] + np.float32(0.02))
    sizes, holes = [], []
    for step in range(14):
        centre = np.array([-12.0 + 2.0 * step, 3.0 * np.sin(step), 0.0], np.float32)
        g.radiusCleanup(centre, 38.0)
        og.radiusCleanup(centre, 38.0)
        assert g.size() == og.size(), step
        sizes.append(g.size())
        holes.append(g.debugCounter(lom.capi.COUNTER_EMPTY_SLABS))
        # part of the original cloud comes back: erased voxels are created anew, kept ones take what they have room for
        sel = rng.choice(len(xyz), 6000, replace=False)
        sel.sort()
        g.addCloud(xyz[sel], nrm[sel])
        og.addCloud(xyz[sel], nrm[sel])
        assert g.size() == og.size() and g.pointCount() == og.pointCount(), step
        if step % 3 == 0 or step == 13:
            _assert_same_map(g, og)
            _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(), 0.3), og.findMatchingPairs(q, oracle.Pose3D(), 0.3))
    assert len(set(sizes)) > 5, "the walk was meant to erase voxels at every step"
    if dense:
        assert max(holes) == 0
    else:  # holes pile up, are closed once they are a quarter of the slabs, pile up again
        assert max(holes) > 0 and any(b < a for a, b in zip(holes, holes[1:])), holes
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    guess = ((0.05, -0.02, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))
    p, op = m.align(g, sm["scan"], lom.Pose3D(*guess)), om.align(og, sm["scan"], oracle.Pose3D(*guess))
    dt, dr = scenes.pose_delta(p.translation, p.rotation, op.translation, op.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD and m.stats["outer_iterations"] == om.stats["outer_iterations"]
    # the cap changed with holes in place, a cleanup that erases everything, and the map goes on
    g.setMaxPoints(7)
    og.setMaxPoints(7)
    g.addCloud(xyz[:5000], nrm[:5000])
    og.addCloud(xyz[:5000], nrm[:5000])
    _assert_same_map(g, og)
    g.radiusCleanup((500.0, 0.0, 0.0), 1.0)
    og.radiusCleanup((500.0, 0.0, 0.0), 1.0)
    assert g.size() == og.size() == 0
    g.addCloud(xyz[:5000], nrm[:5000])
    og.addCloud(xyz[:5000], nrm[:5000])
    _assert_same_map(g, og)


def test_radius_cleanup_scan_behind_align(lom, oracle):
    """lidar_odometry.cpp:65-67: radiusCleanup(translation the align has just returned).  Armed before the align, the
    cleanup's scan runs behind the align's last solve with the centre taken from the align's result in HBM; the
    radiusCleanup that follows takes it only for exactly that centre and radius on an untouched map -- and the map is
    the oracle's either way."""
    sm = scenes.small_synth_case()
    taken = lom.capi.COUNTER_CLEANUPS_BEHIND_ALIGN
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    guess = ((0.05, -0.02, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))
    far = ((0.2, -0.2, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))  # needs a sixth outer iteration: the scan finds the align unfinished
    # (radius, what happens between arming and the cleanup, expected to be taken)
    cases = [(30.0, "plain", True), (6.0, "plain", True), (6.0, "other_centre", False), (6.0, "other_radius", False),
             (6.0, "map_touched", False), (6.0, "six_iterations", False), (6.0, "scan_gives_up", True), (6.0, "host_lm", False),
             (6.0, "not_armed", False)]
    for radius, what, expect in cases:
        g, og = _both(lom, oracle, 0.5, 20)
        g.addCloud(sm["map_xyz"], sm["map_nrm"])
        og.addCloud(sm["map_xyz"], sm["map_nrm"])
        g.radiusCleanup((0, 0, 0), 1e6)  # (sizes the cleanup's scratch: a scan behind an align does not allocate)
        if what == "host_lm":
            g.setOption(lom.capi.OPT_HOST_LM, 1)
        if what != "not_armed":
            g.radiusCleanupAfterAlign(5.0 if what == "other_radius" else radius)
        if what == "scan_gives_up":
            g.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, 1)
        use = far if what == "six_iterations" else guess
        p = m.align(g, sm["scan"], lom.Pose3D(*use))
        op = om.align(og, sm["scan"], oracle.Pose3D(*use))
        assert m.stats["outer_iterations"] == om.stats["outer_iterations"], what
        if what == "map_touched":
            g.addCloud(sm["map_xyz"][:50] + np.float32(0.01), sm["map_nrm"][:50])
            og.addCloud(sm["map_xyz"][:50] + np.float32(0.01), sm["map_nrm"][:50])
        centre = np.asarray(p.translation, np.float32)
        if what == "other_centre":
            centre = centre + np.float32(1e-3)
        before, redos = g.debugCounter(taken), g.debugCounter()
        g.radiusCleanup(centre, radius)
        og.radiusCleanup(centre, radius)
        assert g.debugCounter(taken) - before == (1 if expect else 0), what
        assert g.debugCounter() - redos == (1 if what == "scan_gives_up" else 0), what
        if radius < 10.0:
            assert g.size() < len(np.unique(np.floor(sm["map_xyz"] / 0.5), axis=0)), "the cleanup was meant to remove voxels"
        _assert_same_map(g, og)
        # the map goes on as the oracle's: an insert and a search after the compaction
        g.addCloud(sm["map_xyz"][:3000], sm["map_nrm"][:3000])
        og.addCloud(sm["map_xyz"][:3000], sm["map_nrm"][:3000])
        _assert_same_map(g, og)
        # an arming nobody used does not reach the align after the next one
        p2 = m.align(g, sm["scan"], lom.Pose3D(*guess))
        before = g.debugCounter(taken)
        g.radiusCleanup(np.asarray(p2.translation, np.float32), radius)
        assert g.debugCounter(taken) == before, what


# ---- correspondence parity ----------------------------------------------------

def test_strict_min_first_wins(lom, oracle):
    pts = np.array([[0.6, 0.1, 0.1], [-0.1, 0.1, 0.1]], np.float32)
    nrm = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    q = np.array([[0.25, 0.1, 0.1]], np.float32)
    for d in (0.5, 0.35, 0.3):
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(), d), og.findMatchingPairs(q, oracle.Pose3D(), d))
    assert g.findMatchingPairs(q, lom.Pose3D(), 0.5)["index"][0] == 20


def test_find_pairs_parity_fixture(lom, oracle, fixture_cloud):
    xyz, xyzn = fixture_cloud
    g, og = _both(lom, oracle, 0.25, 20)
    g.addCloud(xyzn[:, :3], xyzn[:, 3:])
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    vf = oracle.VoxelGrid(0.5, 1)
    vf.addCloudWithoutNormals(xyz)
    sub = vf.getCloudWithoutNormals()
    valid = []
    for t, q in scenes.matching_guess_poses():
        c = g.findMatchingPairs(sub, lom.Pose3D(t, q), 0.3)
        oc = og.findMatchingPairs(sub, oracle.Pose3D(t, q), 0.3)
        _assert_same_pairs(c, oc)
        valid.append(int((c["index"] >= 0).sum()))
    assert valid[0] > 5000 and min(valid) > 1000


def test_find_pairs_parity_synth_and_golden(lom, oracle):
    with open(os.path.join(GOLDEN, "synth_small.json")) as f:
        gold = json.load(f)
    sm = scenes.small_synth_case()
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    _assert_same_map(g, og)
    c = g.findMatchingPairs(sm["scan"], lom.Pose3D(), 0.3)
    _assert_same_pairs(c, og.findMatchingPairs(sm["scan"], oracle.Pose3D(), 0.3))
    idx = c["index"].astype("<i8")
    assert hashlib.sha256(idx.tobytes()).hexdigest() == gold["winner_sha256"]
    if _counted():
        assert int(c["n_cand"].sum()) == gold["cand_total"] and int(c["n_occ"].sum()) == gold["occ_total"]
    # a rotated / translated pose exercises the f64 transform + f32 cast path
    pose = (0.31, -0.2, 0.05), scenes.angle_axis_q(0.03, scenes._unit((0.1, 0.2, 1.0)))
    _assert_same_pairs(g.findMatchingPairs(sm["scan"], lom.Pose3D(*pose), 0.3),
                       og.findMatchingPairs(sm["scan"], oracle.Pose3D(*pose), 0.3))


def test_find_pairs_at_the_index_range_limit(lom, oracle):
    """Voxels in the outermost index layer (|i| = 2^20 - 1): their outward neighbours cannot exist;
    the search must neither wrap nor read them.  Same pairs as the oracle."""
    voxel = 0.25
    edge = (2 ** 20 - 1) * voxel                      # first coordinate of the last voxel layer
    rng = np.random.default_rng(77)
    base = np.array([edge + 0.1, -edge - 0.1, 0.0])
    pts = (base + rng.uniform(-0.6, 0.1, (4000, 3)) * np.array([1, -1, 1])).astype(np.float32)
    pts = pts[(np.abs(pts / np.float32(voxel)) < 2 ** 20).all(axis=1)]
    nrm = rng.standard_normal((len(pts), 3)).astype(np.float32)
    g, og = _both(lom, oracle, voxel, 20)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    _assert_same_map(g, og)
    q = (base + rng.uniform(-0.7, 0.12, (3000, 3)) * np.array([1, -1, 1])).astype(np.float32)
    q = q[(np.abs(q / np.float32(voxel)) < 2 ** 20).all(axis=1)]
    c, oc = g.findMatchingPairs(q, lom.Pose3D(), 0.3), og.findMatchingPairs(q, oracle.Pose3D(), 0.3)
    _assert_same_pairs(c, oc)
    assert (c["index"] >= 0).sum() > 100


def test_zero_normals_are_valid_matches(lom):
    g = lom.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(scenes.UNIQUE_POINTS)
    c = g.findMatchingPairs(scenes.UNIQUE_POINTS + np.float32(0.01), lom.Pose3D(), 0.3)
    assert (c["index"] >= 0).all() and not c["normal"].any()


# ---- align parity ---------------------------------------------------------------

def test_matching_test_protocol_and_golden(lom, oracle, fixture_cloud):
    """test/test.cpp:191-264 through the product; tolerances of the reference test, then the
    1e-4 bar against the oracle's golden poses."""
    xyz, xyzn = fixture_cloud
    res = scenes.run_matching_test(lom, xyz, xyzn)
    with open(os.path.join(GOLDEN, "c1_matching_test.json")) as f:
        gold = json.load(f)
    assert res["source_points"] == gold["source_points"] == 9043
    assert res["keyframe_voxels"] == gold["keyframe_voxels"]
    assert res["keyframe_points"] == gold["keyframe_points"]
    for c, gcase in zip(res["cases"], gold["cases"]):
        assert c["err_t_norm"] < 0.05 and c["rot_err"] < 0.01          # test.cpp:261-262
        dt, dr = scenes.pose_delta(c["final_t"], c["final_q_wxyz"], gcase["final_t"], gcase["final_q_wxyz"])
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
        assert c["stats"]["outer_iterations"] == gcase["stats"]["outer_iterations"]
        assert c["stats"]["cand_total"] == (gcase["stats"]["cand_total"] if _counted() else 0)
        assert c["stats"]["valid_last"] == gcase["stats"]["valid_last"]


def test_align_parity_synth(lom, oracle):
    with open(os.path.join(GOLDEN, "synth_small.json")) as f:
        gold = json.load(f)
    sm = scenes.small_synth_case()
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    for guess in (((0, 0, 0), (1, 0, 0, 0)), ((0.08, -0.03, 0.0), scenes.angle_axis_q(0.012, (0, 0, 1)))):
        p = m.align(g, sm["scan"], lom.Pose3D(*guess))
        o = om.align(og, sm["scan"], oracle.Pose3D(*guess))
        dt, dr = scenes.pose_delta(p.translation, p.rotation, o.translation, o.rotation)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
        assert m.stats["outer_iterations"] == om.stats["outer_iterations"]
        assert m.stats["queries"] == om.stats["queries"]
        assert m.stats["cand_total"] == (om.stats["cand_total"] if _counted() else 0)
    p = m.align(g, sm["scan"], lom.Pose3D())
    dt, dr = scenes.pose_delta(p.translation, p.rotation, gold["final_t"], gold["final_q_wxyz"])
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD


@pytest.mark.parametrize("seed", range(5))
def test_align_parity_randomized(lom, oracle, seed):
    """Random sub-scans, voxel sizes (power of two: exact-reciprocal index; others: IEEE division),
    caps and guess poses -- small guesses converge in the minimum five outer iterations, larger ones
    need more (the device loop then enqueues pair by pair): same iteration counts, poses within the bar."""
    rng = np.random.default_rng(500 + seed)
    sm = scenes.small_synth_case()
    voxel = float(rng.choice([0.5, 0.37, 1.0, 0.25]))
    K = int(rng.choice([20, 7, 33]))
    g, og = _both(lom, oracle, voxel, K)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    outers = []
    for trial in range(4):
        n = int(rng.choice([1, 7, 63, 500, len(sm["scan"])]))
        sel = np.sort(rng.choice(len(sm["scan"]), n, replace=False))
        scan = np.ascontiguousarray(sm["scan"][sel])
        scale = float(rng.choice([0.02, 0.15, 0.25]))
        t = rng.uniform(-1, 1, 3) * scale
        q = scenes.angle_axis_q(rng.uniform(-0.2, 0.2) * scale, scenes._unit(rng.standard_normal(3)))
        got = m.align(g, scan, lom.Pose3D(t, q))
        ref = om.align(og, scan, oracle.Pose3D(t, q))
        dt, dr = scenes.pose_delta(got.translation, got.rotation, ref.translation, ref.rotation)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (seed, trial, n, voxel, dt, dr)
        for k in _keys("outer_iterations", "queries", "cand_total", "occ_total", "valid_last"):
            assert m.stats[k] == om.stats[k], (seed, trial, k)
        outers.append(m.stats["outer_iterations"])
    assert min(outers) >= 5


def test_align_beyond_the_five_enqueued_iterations(lom, oracle):
    """A guess 0.28 m off needs a sixth outer iteration: the device loop enqueues five pairs ahead and
    then one pair per report until the stop rule fires."""
    sm = scenes.small_synth_case()
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    og.addCloud(sm["map_xyz"], sm["map_nrm"])
    m, om = lom.CloudMatcher(), oracle.CloudMatcher()
    guess_q = scenes.angle_axis_q(0.01, (0, 0, 1))
    got = m.align(g, sm["scan"], lom.Pose3D((0.2, -0.2, 0.0), guess_q))
    ref = om.align(og, sm["scan"], oracle.Pose3D((0.2, -0.2, 0.0), guess_q))
    assert om.stats["outer_iterations"] > 5
    assert m.stats["outer_iterations"] == om.stats["outer_iterations"]
    dt, dr = scenes.pose_delta(got.translation, got.rotation, ref.translation, ref.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)


def test_host_driven_path_matches_device_loop(lom, monkeypatch):
    """LOM_HOST_LM=1 keeps the outer loop and the LM policy on the host (resident evaluation server,
    the path the multi-GPU exchange uses); the default single-GPU path runs both on the device.
    Same policy source (lm_core.hpp): same iteration counts, poses within the bar."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    dev = m.align(g, sm["scan"], lom.Pose3D())
    dev_stats = dict(m.stats)
    g.setOption(lom.capi.OPT_HOST_LM, 1)
    host = m.align(g, sm["scan"], lom.Pose3D())
    dt, dr = scenes.pose_delta(dev.translation, dev.rotation, host.translation, host.rotation)
    assert dt < 1e-6 and dr < 1e-6, (dt, dr)
    for k in _keys("outer_iterations", "lm_iterations", "evaluations", "queries", "cand_total", "occ_total", "valid_last"):
        assert m.stats[k] == dev_stats[k], k


def test_evaluation_server_timeout_recovery(lom, monkeypatch):
    """Host-driven path: the resident evaluation server leaves when the host stays away (bounded
    spin); the host then relaunches it.  With a 1-tick timeout every LM iteration takes that path:
    same bits."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.setOption(lom.capi.OPT_HOST_LM, 1)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    ref = m.align(g, sm["scan"], lom.Pose3D())
    ref_stats = dict(m.stats)
    g.setOption(lom.capi.OPT_DEVICE_PATIENCE_TICKS, 1)
    got = m.align(g, sm["scan"], lom.Pose3D())
    assert got.translation.tobytes() == ref.translation.tobytes()
    assert got.rotation.tobytes() == ref.rotation.tobytes()
    for k in _keys("outer_iterations", "evaluations", "queries", "cand_total", "valid_last"):
        assert m.stats[k] == ref_stats[k]
    g.setOption(lom.capi.OPT_DEVICE_PATIENCE_TICKS, 5_000_000)
    again = m.align(g, sm["scan"], lom.Pose3D())
    assert again.translation.tobytes() == ref.translation.tobytes()


def test_device_loop_gives_up_cleanly_and_falls_back(lom, monkeypatch):
    """Every wait inside the device-resident solve is bounded: with a 1-tick patience the workgroups
    give up waiting for each other and the grid drains (what happens when they are not all resident:
    a caller sharing the GPU, a CU mask).  The align is then redone by the host-driven loop, whose
    workgroups never wait for each other: the call still returns the pose -- the bits of the
    host-driven path -- and says so in the stats; the handle is on the device loop again afterwards."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    ref = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["host_fallback"] == 0
    g.setOption(lom.capi.OPT_HOST_LM, 1)
    host = m.align(g, sm["scan"], lom.Pose3D())
    host_stats = dict(m.stats)
    g.setOption(lom.capi.OPT_HOST_LM, 0)
    g.setOption(lom.capi.OPT_DEVICE_PATIENCE_TICKS, 1)
    got = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["host_fallback"] == 1
    assert got.translation.tobytes() == host.translation.tobytes()
    assert got.rotation.tobytes() == host.rotation.tobytes()
    for k in _keys("outer_iterations", "lm_iterations", "evaluations", "queries", "cand_total", "valid_last"):
        assert m.stats[k] == host_stats[k], k
    g.setOption(lom.capi.OPT_DEVICE_PATIENCE_TICKS, 5_000_000)
    again = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["host_fallback"] == 0
    assert again.translation.tobytes() == ref.translation.tobytes()
    assert again.rotation.tobytes() == ref.rotation.tobytes()
    # a give-up late in the chain (the k_lm of the fourth outer iteration, pairs still enqueued behind it): the
    # kernels behind it return at once, the align is redone on the host-driven path, the next one is back on the device
    g.setOption(lom.capi.OPT_TEST_GIVE_UP_AT_OUTER, 3)
    late = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["host_fallback"] == 1
    assert late.translation.tobytes() == host.translation.tobytes() and late.rotation.tobytes() == host.rotation.tobytes()
    again = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["host_fallback"] == 0
    assert again.translation.tobytes() == ref.translation.tobytes()


def test_comm_path_single_rank(lom):
    """The multi-GPU exchange path (k_eval -> k_sum_records -> RCCL all-gather -> rank-ordered host
    sum) with a one-rank communicator: same pose and stats as the single-GPU server path.  More
    ranks need more GPUs than a gpurun box has; the N>1 logic itself is covered on CPU by
    tests/test_dist_gloo.py."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    ref = m.align(g, sm["scan"], lom.Pose3D())
    ref_stats = dict(m.stats)
    L = lom.capi.lib()
    ident = C.create_string_buffer(lom.capi.COMM_ID_BYTES)
    lom.capi.check(L.lom_comm_unique_id(ident))
    lom.capi.check(L.lom_comm_init(g.handle, 0, 1, ident.raw), g.handle)
    assert L.lom_comm_init(g.handle, 0, 1, ident.raw) == lom.capi.ERR_STATE      # already initialised
    try:
        got = m.align(g, sm["scan"], lom.Pose3D())
        assert got.translation.tobytes() == ref.translation.tobytes()
        assert got.rotation.tobytes() == ref.rotation.tobytes()
        for k in _keys("outer_iterations", "evaluations", "queries", "cand_total", "valid_last"):
            assert m.stats[k] == ref_stats[k]
    finally:
        lom.capi.check(L.lom_comm_finalize(g.handle), g.handle)
    again = m.align(g, sm["scan"], lom.Pose3D())                                 # back on the server path
    assert again.translation.tobytes() == ref.translation.tobytes()


def test_sampled_profiling_events(lom):
    """lom_map_set_profiling(period): HIP event pairs around the correspondence launches of every
    period-th align only; the other aligns run uninstrumented; results do not change."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    ref = m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["profiled_launches"] == 0 and m.stats["match_kernel_ms"] == 0.0
    g.setProfiling(2)
    seen = []
    for _ in range(4):
        got = m.align(g, sm["scan"], lom.Pose3D())
        assert got.translation.tobytes() == ref.translation.tobytes()
        seen.append((m.stats["profiled_launches"], m.stats["match_launches"], m.stats["match_kernel_ms"]))
    assert [p for p, _, _ in seen] == [seen[0][1], 0, seen[2][1], 0]
    assert seen[0][2] > 0.0 and seen[1][2] == 0.0
    g.setProfiling(0)
    m.align(g, sm["scan"], lom.Pose3D())
    assert m.stats["profiled_launches"] == 0


def test_align_is_a_pure_function_of_its_inputs(lom):
    """Sums are added in a fixed order at every level (lane, wave, workgroup, grid): the same align 300 times gives
    the same pose bit for bit, on a cloud of each kernel shape (256- / 512-thread workgroups of the solve).
    tools/soak_align.py is the long version (275k aligns)."""
    sm = scenes.small_synth_case()
    c = scenes.synth_case(16, 1800, 120_000)
    for scan, mx, mn in ((sm["scan"], sm["map_xyz"], sm["map_nrm"]), (c["scan"], c["map_xyz"], c["map_nrm"])):
        g = lom.VoxelGrid(0.5, 20)
        g.addCloud(mx, mn)
        m = lom.CloudMatcher()
        guess = lom.Pose3D((0.05, -0.04, 0.02), scenes.angle_axis_q(0.0175, (0, 0, 1)))
        first = m.align(g, scan, guess)
        ref = first.translation.tobytes() + first.rotation.tobytes()
        ev = m.stats["evaluations"]
        for _ in range(300):
            p = m.align(g, scan, guess)
            assert p.translation.tobytes() + p.rotation.tobytes() == ref
            assert m.stats["evaluations"] == ev and not m.stats["host_fallback"]


def test_zero_matches_returns_guess(lom):
    g = lom.VoxelGrid(0.5, 20)
    g.addCloudWithoutNormals(np.array([[50, 50, 50]], np.float32))
    m = lom.CloudMatcher()
    gq = scenes.angle_axis_q(0.1, (0, 0, 1))
    out = m.align(g, scenes.UNIQUE_POINTS, lom.Pose3D((1, 2, 3), gq))
    assert np.allclose(out.translation, (1, 2, 3), atol=1e-6)
    assert abs(abs(float(np.dot(out.rotation, gq))) - 1) < 1e-6
    assert m.stats["outer_iterations"] == 5 and m.stats["valid_last"] == 0


def test_align_device_resident_source(lom):
    """Source cloud handed over as a device pointer (torch is only the allocator here)."""
    import torch

    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    host = m.align(g, sm["scan"], lom.Pose3D())
    d = torch.from_numpy(sm["scan"]).to("cuda:0")
    torch.cuda.synchronize()
    dev = m.alignDevice(g, d.data_ptr(), d.shape[0], lom.Pose3D())
    assert host.translation.tobytes() == dev.translation.tobytes()
    assert host.rotation.tobytes() == dev.rotation.tobytes()
    # device-resident insert
    mp = torch.from_numpy(sm["map_xyz"]).to("cuda:0")
    mn = torch.from_numpy(sm["map_nrm"]).to("cuda:0")
    torch.cuda.synchronize()
    g2 = lom.VoxelGrid(0.5, 20)
    g2.addCloudDevice(mp.data_ptr(), mn.data_ptr(), mp.shape[0])
    assert g2.getCloud()[0].tobytes() == g.getCloud()[0].tobytes()


def test_caller_owned_stream(lom):
    """lom_map_set_stream: the handle's work runs on a caller-owned hipStream_t (here one that torch
    created); results are unchanged and the handle can go back to its own stream."""
    import torch

    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = lom.CloudMatcher()
    ref = m.align(g, sm["scan"], lom.Pose3D())
    stream = torch.cuda.Stream()
    lom.capi.check(lom.capi.lib().lom_map_set_stream(g.handle, C.c_void_p(stream.cuda_stream)), g.handle)
    got = m.align(g, sm["scan"], lom.Pose3D())
    g.addCloud(sm["map_xyz"][:1000], sm["map_nrm"][:1000])
    n_after = g.pointCount()
    lom.capi.check(lom.capi.lib().lom_map_set_stream(g.handle, None), g.handle)
    assert got.translation.tobytes() == ref.translation.tobytes()
    assert got.rotation.tobytes() == ref.rotation.tobytes()
    assert g.pointCount() == n_after


# ---- full BASELINE.json sizes: size-independent properties ----------------------

@pytest.fixture(scope="module")
def c2_case():
    return scenes.synth_case(16, 1800, 500_000)


def test_c2_full_size_properties(lom, oracle, c2_case):
    """C2 (VLP16 ~30k-pt scan vs 500k-pt map, 0.5 m voxels).  The oracle still finishes
    in seconds at this size, so: exact map parity, exact winners, pose bar; plus
    properties: determinism (bitwise repeatable), idempotence of insert, sharding
    linearity of the reduced sums."""
    c = c2_case
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(c["map_xyz"], c["map_nrm"])
    og.addCloud(c["map_xyz"], c["map_nrm"])
    assert g.size() == og.size() and g.pointCount() == og.pointCount()
    assert g.getCloud()[0].tobytes() == og.getCloud()[0].tobytes()
    pairs = g.findMatchingPairs(c["scan"], lom.Pose3D(), 0.3)
    _assert_same_pairs(pairs, og.findMatchingPairs(c["scan"], oracle.Pose3D(), 0.3, nthreads=4))
    m, om = lom.CloudMatcher(), oracle.CloudMatcher(nthreads=4)
    p = m.align(g, c["scan"], lom.Pose3D())
    o = om.align(og, c["scan"], oracle.Pose3D())
    dt, dr = scenes.pose_delta(p.translation, p.rotation, o.translation, o.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
    assert m.stats["outer_iterations"] == om.stats["outer_iterations"]
    # recovers the simulated motion where the scene constrains it (y: walls, z: ground, yaw)
    assert abs(p.translation[1] - c["true_t"][1]) < 0.01 and abs(p.translation[2] - c["true_t"][2]) < 0.01
    assert scenes.pose_delta((0, 0, 0), p.rotation, (0, 0, 0), c["true_q"])[1] < 2e-3
    # determinism: same call, same bits
    p2 = m.align(g, c["scan"], lom.Pose3D())
    assert p.translation.tobytes() == p2.translation.tobytes() and p.rotation.tobytes() == p2.rotation.tobytes()
    # idempotence: re-inserting the same cloud into full/partial voxels changes nothing for
    # voxels already at the cap and never reorders existing points
    before = g.getCloud()[0]
    g.addCloud(c["map_xyz"], c["map_nrm"])
    og.addCloud(c["map_xyz"], c["map_nrm"])
    after = g.getCloud()[0]
    assert after.tobytes() == og.getCloud()[0].tobytes()
    assert g.size() == og.size()
    assert len(after) >= len(before)


def test_c3_full_size_properties(lom, oracle):
    """C3 (64-beam ~130k-pt scan vs 2M-pt map): counts and checksums of winners against the
    oracle (search only: ~1 s of oracle time), plus align determinism and pose bar."""
    c = scenes.synth_case(64, 2048, 2_000_000)
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(c["map_xyz"], c["map_nrm"])
    og.addCloud(c["map_xyz"], c["map_nrm"])
    assert g.size() == og.size() and g.pointCount() == og.pointCount()
    pairs = g.findMatchingPairs(c["scan"], lom.Pose3D(), 0.3)
    opairs = og.findMatchingPairs(c["scan"], oracle.Pose3D(), 0.3, nthreads=4)
    assert np.array_equal(pairs["index"], opairs["index"])
    assert int(pairs["n_cand"].sum()) == (int(opairs["n_cand"].sum()) if _counted() else 0)
    m, om = lom.CloudMatcher(), oracle.CloudMatcher(nthreads=4)
    p = m.align(g, c["scan"], lom.Pose3D())
    o = om.align(og, c["scan"], oracle.Pose3D())
    dt, dr = scenes.pose_delta(p.translation, p.rotation, o.translation, o.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
    p2 = m.align(g, c["scan"], lom.Pose3D())
    assert p.translation.tobytes() == p2.translation.tobytes()


@pytest.fixture(scope="module")
def c4_case():
    return scenes.synth_case(128, 2048, 2_000_000)


def test_c4_full_size_single_rank(lom, oracle, c4_case):
    """C4 (BASELINE.json configs[3]: 128-beam x 2048 scan, <= 262,144 returns, vs the 2M-point map) with all
    index ranges on one rank: winners and counters exact, pose within the bar, LM numbers equal the
    oracle's.  The sharded runs (2 and 3 ranks on this GPU) are in tests/test_p2p_gpu.py."""
    c = c4_case
    assert 200_000 < len(c["scan"]) <= 262_144
    g, og = _both(lom, oracle, 0.5, 20)
    g.addCloud(c["map_xyz"], c["map_nrm"])
    og.addCloud(c["map_xyz"], c["map_nrm"])
    assert g.size() == og.size() and g.pointCount() == og.pointCount()
    pairs = g.findMatchingPairs(c["scan"], lom.Pose3D(), 0.3)
    opairs = og.findMatchingPairs(c["scan"], oracle.Pose3D(), 0.3, nthreads=8)
    _assert_same_pairs(pairs, opairs)
    m, om = lom.CloudMatcher(), oracle.CloudMatcher(nthreads=8)
    p = m.align(g, c["scan"], lom.Pose3D())
    o = om.align(og, c["scan"], oracle.Pose3D())
    dt, dr = scenes.pose_delta(p.translation, p.rotation, o.translation, o.rotation)
    assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (dt, dr)
    for k in _keys("outer_iterations", "lm_iterations", "queries", "cand_total", "occ_total", "valid_last"):
        assert m.stats[k] == om.stats[k], k
    assert m.stats["evaluations"] == om.stats["points_evaluated"]
    assert abs(m.stats["last_step_norm"] - om.stats["last_step_norm"]) < 1e-9
    p2 = m.align(g, c["scan"], lom.Pose3D())
    assert p.translation.tobytes() == p2.translation.tobytes() and p.rotation.tobytes() == p2.rotation.tobytes()
    # the 8 contiguous ranges of the 8-GPU configuration, one after the other on this GPU: the ranges'
    # reduced sums add up to the whole scan's (what the exchange step computes)
    from tests.test_eval_parity import assert_sums_close

    pose = lom.Pose3D((0.03, -0.02, 0.01), scenes.angle_axis_q(0.004, (0, 0, 1)))
    n = len(c["scan"])
    parts = [m.debugEvalSums(g, np.ascontiguousarray(c["scan"][n * r // 8: n * (r + 1) // 8]), pose) for r in range(8)]
    assert_sums_close(np.sum(parts, axis=0), m.debugEvalSums(g, c["scan"], pose), "C4 ranges")


def test_deferred_insert_verdict(lom):
    """lom_map_add_points_device_nowait only enqueues; lom_map_status reports a point out of range afterwards, and
    such a call has inserted nothing (its voxels' keys may stay claimed without payload: invisible to every query)."""
    import ctypes as C

    import torch

    L = lom.capi.lib()
    g = lom.VoxelGrid(0.5, 20)
    good = torch.from_numpy(np.array([[0.1, 0.1, 0.1], [1.1, 0.1, 0.1], [0.2, 0.1, 0.1]], np.float32)).to("cuda:0")
    bad = torch.from_numpy(np.array([[5.1, 0.1, 0.1], [1e9, 0.0, 0.0], [7.2, 0.1, 0.1]], np.float32)).to("cuda:0")
    torch.cuda.synchronize()
    lom.capi.check(L.lom_map_add_points_device_nowait(g.handle, good.data_ptr(), None, 3, 12), g.handle)
    assert L.lom_map_status(g.handle) == 0
    assert g.size() == 2 and g.pointCount() == 3
    assert L.lom_map_add_points_device_nowait(g.handle, bad.data_ptr(), None, 3, 12) == 0        # enqueued
    assert L.lom_map_status(g.handle) == lom.capi.ERR_RANGE                                     # the verdict, later
    assert g.size() == 2 and g.pointCount() == 3                                                 # nothing inserted
    assert L.lom_map_status(g.handle) == 0                                                       # reported once
    c = g.findMatchingPairs(np.array([[5.1, 0.1, 0.1], [0.1, 0.1, 0.1]], np.float32), lom.Pose3D(), 0.3)
    assert c["index"][0] == -1 and c["index"][1] >= 0
    lom.capi.check(L.lom_map_add_points_device_nowait(g.handle, good.data_ptr(), None, 3, 12), g.handle)
    assert L.lom_map_status(g.handle) == 0 and g.pointCount() == 6
    # a later insert INTO a voxel whose key an aborted call left behind works like into a new voxel
    ok2 = torch.from_numpy(np.array([[5.1, 0.1, 0.1]], np.float32)).to("cuda:0")
    torch.cuda.synchronize()
    lom.capi.check(L.lom_map_add_points_device(g.handle, ok2.data_ptr(), None, 1, 12), g.handle)
    assert g.size() == 3 and g.pointCount() == 7
    xyz, _ = g.getCloud()
    assert tuple(xyz[-1]) == (np.float32(5.1), np.float32(0.1), np.float32(0.1))


# ---- the partitioned bulk insert (batches above 65,536 points) ------------------------------------------------------

def _bulk_cloud(rng, n, n_centers, spread, lo=-20, hi=20):
    centers = rng.uniform(lo, hi, (n_centers, 3))
    pts = (centers[rng.integers(0, n_centers, n)] + rng.normal(0, spread, (n, 3))).astype(np.float32)
    return pts, scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)


@pytest.mark.parametrize("mode", ["bulk", "four_kernel", "sent_back"])
def test_bulk_insert_paths_give_the_same_map(lom, oracle, mode):
    """Batches above 65,536 points go through the partition pass (k_bi_*), the four-kernel path (LOM_OPT_NO_BULK_INSERT),
    or -- a partition beyond the workgroup's LDS, forced here by LOM_OPT_TEST_BULK_PARTITION_MAX -- start on the first and
    are redone by the second.  Bytewise the oracle's map every way: a first batch (every voxel new, caps cutting buckets),
    a second one over it (voxels that exist, with and without room), one without normals, interleaved 48-byte records."""
    rng = np.random.default_rng(31)
    for voxel, K, n, nc, spread in ((0.5, 20, 150_000, 3000, 0.8), (0.25, 3, 70_000, 500, 0.5), (1.0, 7, 300_001, 40, 3.0)):
        g, og = _both(lom, oracle, voxel, K)
        if mode == "four_kernel":
            g.setOption(lom.capi.OPT_NO_BULK_INSERT, 1)
        if mode == "sent_back":
            g.setOption(lom.capi.OPT_TEST_BULK_PARTITION_MAX, 16)
        redos = 0
        pts, nrm = _bulk_cloud(rng, n, nc, spread)
        g.addCloud(pts, nrm)
        og.addCloud(pts, nrm)
        redos += mode == "sent_back"
        assert g.size() == og.size() and g.debugCounter() == redos
        _assert_same_map(g, og)
        more, mnrm = _bulk_cloud(rng, 80_000, nc, spread * 1.5)
        g.addCloud(more, mnrm)
        og.addCloud(more, mnrm)
        redos += mode == "sent_back"
        _assert_same_map(g, og)
        assert g.debugCounter() == redos
        bare = np.ascontiguousarray(pts[::2] + np.float32(0.05))
        g.addCloudWithoutNormals(bare)
        og.addCloudWithoutNormals(bare)
        _assert_same_map(g, og)
        rec = np.zeros((70_001, 12), np.float32)                       # x y z pad nx ny nz pad ... as one 48-byte record
        rec[:, 0:3], rec[:, 4:7] = _bulk_cloud(rng, len(rec), nc, spread)
        g.addCloudInterleaved(rec, 48, 16)
        og.addCloud(np.ascontiguousarray(rec[:, 0:3]), np.ascontiguousarray(rec[:, 4:7]))
        _assert_same_map(g, og)
        q = pts[::50]
        _assert_same_pairs(g.findMatchingPairs(q, lom.Pose3D(), 0.3), og.findMatchingPairs(q, oracle.Pose3D(), 0.3))


def test_bulk_insert_crowded_voxels_are_sent_back(lom, oracle):
    """100,000 points in a handful of voxels: every partition that holds one is far beyond the 1,024 points a workgroup
    groups in LDS, the bulk insert writes nothing and the four-kernel path redoes it (first K in input order)."""
    rng = np.random.default_rng(32)
    g, og = _both(lom, oracle, 0.5, 20)
    pts = (rng.integers(0, 2, (100_000, 3)) * 0.5 + rng.random((100_000, 3)) * 0.49).astype(np.float32)
    nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    assert g.size() == og.size() == 8 and g.debugCounter() == 1
    _assert_same_map(g, og)
    wide, wnrm = _bulk_cloud(rng, 90_000, 2000, 1.0)                   # and an ordinary bulk batch over it
    g.addCloud(wide, wnrm)
    og.addCloud(wide, wnrm)
    assert g.debugCounter() == 1
    _assert_same_map(g, og)


def test_bulk_insert_beyond_two_million_points(lom, oracle):
    """Above 2,097,152 points the partitions number 16,384 and grow past 128 points on average: the group kernel's
    1,024-point shape takes over (k_bi_group<1024>), the claim / scatter passes run eight points per thread.  One batch of
    2.3 M points (every voxel new, caps cutting buckets), a second one over it; bytewise against the oracle."""
    rng = np.random.default_rng(34)
    g, og = _both(lom, oracle, 0.4, 12)
    pts, nrm = _bulk_cloud(rng, 2_300_000, 20_000, 1.2, lo=-60, hi=60)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    assert g.size() == og.size() and g.debugCounter() == 0
    _assert_same_map(g, og)
    more, mnrm = _bulk_cloud(rng, 2_150_000, 20_000, 1.6, lo=-60, hi=60)
    g.addCloud(more, mnrm)
    og.addCloud(more, mnrm)
    assert g.debugCounter() == 0
    _assert_same_map(g, og)


def test_bulk_insert_range_error_inserts_nothing(lom, oracle):
    """voxel_grid.h casts unchecked; here a coordinate beyond the index range rejects the whole call (LOM_ERR_RANGE) --
    for a bulk batch too: the map is what it was, and the same batch without the bad point goes in afterwards."""
    import torch

    L = lom.capi.lib()
    rng = np.random.default_rng(33)
    g, og = _both(lom, oracle, 0.5, 20)
    first, fn = _bulk_cloud(rng, 30_000, 300, 1.0)
    g.addCloud(first, fn)
    og.addCloud(first, fn)
    pts, nrm = _bulk_cloud(rng, 100_000, 1000, 1.0)
    bad = pts.copy()
    bad[77_777, 1] = np.float32(3e9)
    d_bad = torch.from_numpy(bad).to("cuda:0")
    d_nrm = torch.from_numpy(nrm).to("cuda:0")
    torch.cuda.synchronize()
    assert L.lom_map_add_points_device(g.handle, d_bad.data_ptr(), d_nrm.data_ptr(), len(bad), 12) == lom.capi.ERR_RANGE
    _assert_same_map(g, og)
    d_ok = torch.from_numpy(pts).to("cuda:0")
    torch.cuda.synchronize()
    lom.capi.check(L.lom_map_add_points_device(g.handle, d_ok.data_ptr(), d_nrm.data_ptr(), len(pts), 12), g.handle)
    og.addCloud(pts, nrm)
    _assert_same_map(g, og)


# ---- an in-kernel scan that gives up: the call changes nothing and is redone with the multi-launch scan -------

@pytest.mark.parametrize("fail_from", [1, 3])
def test_map_maintenance_survives_a_grid_give_up(lom, oracle, fixture_cloud, fail_from):
    """The single-pass kernels of insert / down-sampling / cleanup wait for their predecessor workgroups with a
    bounded patience.  LOM_OPT_TEST_GRID_GIVE_UP makes the workgroups from `fail_from` on give up in the next
    such call -- those before it have already written (slab ids, outputs), those from it on have no prefix.
    The call must change nothing the redo could trip over: results stay bytewise the oracle's, the handle
    reports the redo, and the scratch is at rest (the following plain calls are exact as well)."""
    _, xyzn = fixture_cloud
    xyz, nrm = xyzn[:, :3], xyzn[:, 3:]
    g, og = _both(lom, oracle, 0.25, 20)
    cuts = [0, 20000, 40000, len(xyz)]                     # three single-pass batches (<= 65536 points)
    redo = 0
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if i != 1:                                         # first (all voxels new) and third (mixed) batch give up
            g.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, fail_from)
            redo += 1
        g.addCloud(xyz[a:b], nrm[a:b])
        og.addCloud(xyz[a:b], nrm[a:b])
        assert g.size() == og.size()                       # resolves the pending insert
        assert g.debugCounter() == redo
    _assert_same_map(g, og)
    # cleanup: the scan that flags and numbers the voxels gives up -> flags + multi-launch scan
    g.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, fail_from)
    g.radiusCleanup((5.0, 3.0, 10.0), 40.0)
    og.radiusCleanup((5.0, 3.0, 10.0), 40.0)
    assert g.debugCounter() == redo + 1
    _assert_same_map(g, og)
    g.addCloud(xyz[:30000], nrm[:30000])                   # and the map is a working map afterwards
    og.addCloud(xyz[:30000], nrm[:30000])
    _assert_same_map(g, og)
    # the search sees the redone insert without anybody asking for the status in between
    g2, og2 = _both(lom, oracle, 0.25, 20)
    g2.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, fail_from)
    g2.addCloud(xyz[:50000], nrm[:50000])
    og2.addCloud(xyz[:50000], nrm[:50000])
    src = np.ascontiguousarray(xyz[::7])
    _assert_same_pairs(g2.findMatchingPairs(src, lom.Pose3D(), 0.3), og2.findMatchingPairs(src, oracle.Pose3D(), 0.3))
    assert g2.debugCounter() == 1
    # down-sampler workspace: outputs equal, workspace left empty and reusable
    ods = oracle.VoxelGrid(0.3, 1)
    ods.addCloud(xyz, nrm)
    oxyz, onrm = ods.getCloud()
    ws = lom.VoxelGrid(1.0, 1)
    ws.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, fail_from)
    dx, dn = ws.downsample(xyz, nrm, 0.3)
    assert dx.tobytes() == oxyz.tobytes() and dn.tobytes() == onrm.tobytes()
    assert ws.debugCounter() == 1 and ws.size() == 0
    dx, dn = ws.downsample(xyz, nrm, 0.3)
    assert dx.tobytes() == oxyz.tobytes() and dn.tobytes() == onrm.tobytes()


# ---- several callers, one keyframe (const VoxelGrid&, voxel_grid.h:206 / cloud_matcher.h:15) -----------------------

@pytest.mark.parametrize("again", range(3))
def test_scan_contexts_align_concurrently_against_one_keyframe(lom, again):
    """Scan contexts (lom_scan_create) own stream, per-scan buffers and solve state; their kernels read the one
    keyframe.  Three threads, one context each, different scans and guesses, many aligns each, all at the same time
    (the C calls release the GIL): every result carries the bits of the same align issued alone through the map
    handle, and so do the map handle's own aligns afterwards.  No solve loses the device loop: a solve's workgroups
    wait up to 50 ms for each other, the kernels of the other callers it may have to wait behind last microseconds
    (measured: 0 fall-backs in 3 x 2 runs of 120 aligns here, 0 of 4,400 in bench.py's concurrent_contexts)."""
    import threading

    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    scan = sm["scan"]
    jobs = [(np.ascontiguousarray(scan[0::2]), lom.Pose3D((0.0, 0.0, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))),
            (np.ascontiguousarray(scan[1::2]), lom.Pose3D((0.2, -0.2, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))),
            (scan, lom.Pose3D((0.05, 0.02, 0.0), (1, 0, 0, 0)))]
    m = lom.CloudMatcher()
    want = []
    for cloud, guess in jobs:                                # serial, through the map handle
        p = m.align(g, cloud, guess)
        want.append((p.translation.tobytes(), p.rotation.tobytes(), dict(m.stats)))
    ctxs = [lom.ScanContext(g) for _ in jobs]
    got = [[] for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(i):
        try:
            mm = lom.CloudMatcher()
            cloud, guess = jobs[i]
            start.wait()
            for _ in range(40):
                p = mm.align(ctxs[i], cloud, guess)
                got[i].append((p.translation.tobytes(), p.rotation.tobytes(), dict(mm.stats)))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    fell_back = 0
    for i, w in enumerate(want):
        assert len(got[i]) == 40
        for t_, q_, st in got[i]:
            fell_back += st["host_fallback"]
            assert t_ == w[0] and q_ == w[1], i
            for k in _keys("outer_iterations", "lm_iterations", "evaluations", "queries", "cand_total", "valid_last"):
                assert st[k] == w[2][k], (i, k)
    assert fell_back == 0, fell_back
    c = ctxs[0].handle and lom.capi.lib().lom_scan_find_pairs                     # the search entry on a context
    pairs = (lom.capi.Correspondence * len(scan))()
    n_valid = c(ctxs[0].handle, scan.ctypes.data, len(scan), 12, lom.capi.f3((0, 0, 0)), lom.capi.f4((1, 0, 0, 0)), 0.3, pairs)
    ref = g.findMatchingPairs(scan, lom.Pose3D(), 0.3)
    assert n_valid == int((ref["index"] >= 0).sum())
    p = m.align(g, jobs[0][0], jobs[0][1])                   # and the map handle still gives its own bits
    assert p.translation.tobytes() == want[0][0] and p.rotation.tobytes() == want[0][1]
    # the map changes (its kernels run on the map's stream): the contexts' next aligns see the new points, like the
    # reference's synchronous calls
    rng = np.random.default_rng(3)
    extra = (sm["map_xyz"][::5] + rng.normal(0, 0.01, sm["map_xyz"][::5].shape)).astype(np.float32)
    g.addCloud(extra, sm["map_nrm"][::5])
    for i, (cloud, guess) in enumerate(jobs):
        a = m.align(ctxs[i], cloud, guess)
        b = m.align(g, cloud, guess)
        assert a.translation.tobytes() == b.translation.tobytes() and a.rotation.tobytes() == b.rotation.tobytes(), i
    for cx in ctxs:
        cx.close()


# ---- the temporal pruning bound of the align's searches (csrc/match.hip) ---------------------------------------------

@pytest.mark.parametrize("voxel,K", [(0.5, 20), (0.2, 20), (0.25, 3)])
def test_temporal_bound_never_changes_a_search(lom, oracle, fixture_cloud, voxel, K):
    """Outer iterations >= 2 of an align prune with the previous iteration's winner (its distance at the new pose
    bounds the minimum from above).  lom_debug_find_pairs_after runs exactly that pair of searches for ANY two poses;
    the second search must equal the oracle's search at the second pose entry for entry -- winner, f32 distance,
    counts -- whether the pose moved by a millimetre (the bound is tight), not at all (ties with the old winner), by
    metres (the old winner has left the 27 voxels: found out, searched again at the plain bound) or back again, with
    voxels smaller than the search radius (0.2 m: the old winner may lie two voxels away) and with short voxels."""
    _, xyzn = fixture_cloud
    g, og = _both(lom, oracle, voxel, K)
    g.addCloud(xyzn[:, :3], xyzn[:, 3:])
    og.addCloud(xyzn[:, :3], xyzn[:, 3:])
    src = np.ascontiguousarray(xyzn[::6, :3])
    z = (0, 0, 1)
    poses = [((0, 0, 0), (1, 0, 0, 0)),
             ((0.001, -0.001, 0.0005), scenes.angle_axis_q(0.0002, z)),
             ((0.02, -0.03, 0.01), scenes.angle_axis_q(0.004, z)),
             ((0.15, 0.1, -0.05), scenes.angle_axis_q(0.02, scenes._unit((0.2, 1, 0.1)))),
             ((1.5, -2.0, 0.3), scenes.angle_axis_q(0.3, z)),
             ((40.0, 0.0, 0.0), (1, 0, 0, 0))]                 # nothing in reach at all: no winner to take a bound from
    want = {i: og.findMatchingPairs(src, oracle.Pose3D(*p), 0.3) for i, p in enumerate(poses)}
    for a, b in [(0, 0), (0, 1), (1, 0), (0, 2), (2, 3), (3, 2), (0, 4), (4, 0), (5, 0), (0, 5), (4, 4), (3, 1)]:
        got = g.findMatchingPairsAfter(src, lom.Pose3D(*poses[a]), lom.Pose3D(*poses[b]), 0.3)
        _assert_same_pairs(got, want[b])
    # other gates: 5 cm (most old winners are beyond it at the new pose), 1 m (27 voxels do not cover it)
    for d in (0.05, 1.0):
        w = og.findMatchingPairs(src, oracle.Pose3D(*poses[2]), d)
        _assert_same_pairs(g.findMatchingPairsAfter(src, lom.Pose3D(*poses[0]), lom.Pose3D(*poses[2]), d), w)
        _assert_same_pairs(g.findMatchingPairsAfter(src, lom.Pose3D(*poses[4]), lom.Pose3D(*poses[2]), d), w)


@pytest.mark.parametrize("seed", range(4))
def test_temporal_bound_randomized(lom, oracle, seed):
    """Random maps (crowded double-width voxel around 0, duplicates: ties), random pose pairs."""
    rng = np.random.default_rng(4100 + seed)
    voxel, K = float(rng.choice([0.2, 0.4, 0.5, 1.0])), int(rng.choice([1, 2, 7, 20]))
    g, og = _both(lom, oracle, voxel, K)
    n = int(rng.integers(2000, 30000))
    centers = rng.uniform(-6, 6, size=(int(rng.integers(3, 60)), 3))
    pts = (centers[rng.integers(0, len(centers), n)] + rng.normal(0, rng.uniform(0.05, 0.8), (n, 3))).astype(np.float32)
    pts[rng.random(n) < 0.05] *= np.float32(0.01)
    pts[n // 2:n // 2 + n // 10] = pts[:n // 10]               # exact duplicates: equal distances
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    q = (pts[rng.integers(0, n, 4000)] + rng.normal(0, 0.1, (4000, 3))).astype(np.float32)
    for _ in range(4):
        p0 = (rng.uniform(-0.3, 0.3, 3), scenes.angle_axis_q(rng.uniform(-0.05, 0.05), scenes._unit(rng.standard_normal(3))))
        step = float(rng.choice([0.0, 1e-3, 0.05, 0.6]))
        p1 = (p0[0] + rng.uniform(-step, step, 3), scenes.angle_axis_q(rng.uniform(-0.05, 0.05), scenes._unit(rng.standard_normal(3))))
        d = float(rng.choice([0.1, 0.3, 0.3, 0.7]))
        _assert_same_pairs(g.findMatchingPairsAfter(q, lom.Pose3D(*p0), lom.Pose3D(*p1), d),
                           og.findMatchingPairs(q, oracle.Pose3D(*p1), d))


def test_no_temporal_switch_gives_the_same_align(lom, oracle):
    """LOM_OPT_NO_TEMPORAL_BOUND only changes what is read: pose bits, iteration counts and candidate counts of an
    align are the same with and without it (device-resident loop and host-driven loop)."""
    sm = scenes.small_synth_case()
    res = {}
    for host in (0, 1):
        for off in (0, 1):
            g = lom.VoxelGrid(0.5, 20)
            g.setOption(lom.capi.OPT_HOST_LM, host)
            g.setOption(lom.capi.OPT_COUNT_CANDIDATES, 1)
            g.setOption(lom.capi.OPT_NO_TEMPORAL_BOUND, off)
            g.addCloud(sm["map_xyz"], sm["map_nrm"])
            m = lom.CloudMatcher()
            p = m.align(g, sm["scan"], lom.Pose3D((0.2, -0.2, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1))))
            res[(host, off)] = (p.translation.tobytes(), p.rotation.tobytes(),
                                tuple(m.stats[k] for k in ("outer_iterations", "lm_iterations", "evaluations", "queries",
                                                           "cand_total", "occ_total", "valid_last")))
    assert res[(0, 0)] == res[(0, 1)]
    assert res[(1, 0)] == res[(1, 1)]
    assert res[(0, 0)][2] == res[(1, 0)][2]


def test_second_search_far_moves_rim_and_empty_space(lom, oracle):
    """The second of two searches of one scan (what outer iterations >= 2 of an align run, with the previous winner's
    bound where there is one), entry for entry against the oracle: millimetre and far moves between the two, queries at
    the rim of the index range (neighbours beyond it cannot exist), queries in empty space, voxels smaller than the
    search radius."""
    rng = np.random.default_rng(17)
    g, og = _both(lom, oracle, 0.5, 20)
    far = np.float32(0.5 * ((1 << 20) - 4))
    pts = np.concatenate([rng.uniform(-3, 3, (4000, 3)), rng.uniform(-0.9, 0.9, (500, 3)) + [far, 0, 0],
                          rng.uniform(-0.9, 0.9, (500, 3)) - [0, far, 0]]).astype(np.float32)
    nrm = scenes._unit(rng.standard_normal(pts.shape)).astype(np.float32)
    g.addCloud(pts, nrm)
    og.addCloud(pts, nrm)
    src = np.concatenate([pts[::3] + rng.normal(0, 0.05, pts[::3].shape).astype(np.float32),
                          rng.uniform(-30, 30, (300, 3)).astype(np.float32)]).astype(np.float32)
    poses = [((0, 0, 0), (1, 0, 0, 0)), ((0.01, 0.0, -0.01), scenes.angle_axis_q(0.001, (0, 0, 1))),
             ((0.4, 0.3, 0.0), (1, 0, 0, 0)), ((0, 0, 0), (1, 0, 0, 0))]
    for a, b in zip(poses[:-1], poses[1:]):
        for d in (0.3, 1.2):
            got = g.findMatchingPairsAfter(src, lom.Pose3D(*a), lom.Pose3D(*b), d)
            _assert_same_pairs(got, og.findMatchingPairs(src, oracle.Pose3D(*b), d))


# ---- getCorrespondence's squared threshold (voxel_grid.h:164: a double) ------------------------------------------------

def test_squared_threshold_is_taken_as_it_is(lom, oracle):
    """lom_match_find_pairs_sq hands the reference's `double max_correspondence_distance_sq` over without a root: a
    stored point passes iff (double)(f32 squared distance) < max_sq (voxel_grid.h:184-186).  Thresholds that are no
    square of an f32, exactly a candidate's squared distance (strict: rejected), the next double above it
    (accepted), and the square root's round trip that the mirror used before (differs by an ulp for some)."""
    rng = np.random.default_rng(12)
    g, og = _both(lom, oracle, 0.5, 5)
    pts = rng.uniform(-2, 2, (3000, 3)).astype(np.float32)
    g.addCloudWithoutNormals(pts)
    og.addCloudWithoutNormals(pts)
    qs = rng.uniform(-2, 2, (60, 3)).astype(np.float32)
    loose = og.findMatchingPairs(qs, oracle.Pose3D(), 2.0)
    checked = 0
    for q, c in zip(qs, loose):
        if c["index"] < 0:
            continue
        d2 = float(c["sq_dist"])                               # the winner's f32 squared distance, exactly
        for thr in (d2, np.nextafter(d2, np.inf), np.nextafter(d2, 0.0), d2 * (1 + 2e-8), d2 * (1 - 2e-8), 0.0899999, 0.09,
                    float(np.float32(0.3) * np.float32(0.3)), 1e-30, 0.0, -1.0, 1e300):
            want = og.getCorrespondence(q, thr)
            got = g.getCorrespondence(q, thr)
            assert got["index"] == want["index"], (q, thr)
            assert got["sq_dist"].tobytes() == want["sq_dist"].tobytes()
            if _counted():
                assert got["n_cand"] == want["n_cand"] and got["n_occ"] == want["n_occ"]
            checked += 1
        assert g.getCorrespondence(q, d2)["index"] != c["index"]            # strict: the winner itself is rejected ...
        assert g.getCorrespondence(q, np.nextafter(d2, np.inf))["index"] == c["index"]   # ... and passes one ulp above
    assert checked > 300


# ---- contexts created and first used by several threads at once, behind an insert nobody has looked at -------------------

def test_contexts_created_concurrently_settle_a_pending_insert_once(lom, oracle, fixture_cloud):
    """The C++ mirror's worker threads create their scan contexts -- and make their first call -- together.  If the
    keyframe's last insert was only enqueued (lom_map_add_points_device_nowait) and its in-kernel scan gave up
    (LOM_OPT_TEST_GRID_GIVE_UP), exactly ONE of them redoes it, under the map's settle lock: the map is bytewise the
    oracle's afterwards (a second redo would fill voxels with duplicates), the redo counter says 1, and every thread's
    search equals the oracle's."""
    import threading
    import torch

    _, xyzn = fixture_cloud
    xyz, nrm = np.ascontiguousarray(xyzn[:50000, :3]), np.ascontiguousarray(xyzn[:50000, 3:])
    L = lom.capi.lib()
    src = np.ascontiguousarray(xyz[::9])
    for trial in range(3):
        g, og = _both(lom, oracle, 0.25, 20)
        og.addCloud(xyz, nrm)
        want = og.findMatchingPairs(src, oracle.Pose3D(), 0.3)
        d_xyz, d_nrm = torch.from_numpy(xyz).cuda(), torch.from_numpy(nrm).cuda()
        torch.cuda.synchronize()
        g.setOption(lom.capi.OPT_TEST_GRID_GIVE_UP, 1 + trial)
        lom.capi.check(L.lom_map_add_points_device_nowait(g.handle, d_xyz.data_ptr(), d_nrm.data_ptr(), len(xyz), 12), g.handle)
        n_threads = 6
        start = threading.Barrier(n_threads)
        errors, got = [], [None] * n_threads

        def worker(i):
            try:
                start.wait()
                cx = lom.ScanContext(g)                        # lom_scan_create: settles the map
                pairs = (lom.capi.Correspondence * len(src))()
                rc = L.lom_scan_find_pairs(cx.handle, src.ctypes.data, len(src), 12, lom.capi.f3((0, 0, 0)),
                                           lom.capi.f4((1, 0, 0, 0)), 0.3, pairs)
                got[i] = (rc, np.frombuffer(pairs, dtype=lom.capi.CORR_DTYPE).copy())
                cx.close()
            except Exception as e:  # noqa: BLE001
                errors.append((i, repr(e)))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        assert g.debugCounter() == 1, g.debugCounter()
        _assert_same_map(g, og)
        for rc, pairs in got:
            assert rc == int((want["index"] >= 0).sum())
            _assert_same_pairs(pairs, want)
        del d_xyz, d_nrm


def test_uneven_slices_hold_their_solve(lom):
    """Slices of the compute units that are no multiple of the XCD count (256 / 6 = 42 or 43, 256 / 5, 256 / 7): the
    dispatcher deals a grid's workgroups round-robin over the XCDs whatever the CU mask says, so the solve kernel -- whose
    workgroups wait for each other -- fits as often as the XCD with the fewest enabled CUs allows.  A VLP16-sized scan
    (more workgroups than any such slice holds): no align waits out its patience (host_fallback == 0), same bits as on the
    whole GPU."""
    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    rng = np.random.default_rng(5)
    scan = np.ascontiguousarray(np.tile(sm["scan"], (14, 1)) + rng.normal(0, 0.01, (14 * len(sm["scan"]), 3)).astype(np.float32))
    guess = lom.Pose3D((0.05, -0.02, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))
    m = lom.CloudMatcher()
    want = m.align(g, scan, guess)
    want_stats = dict(m.stats)
    assert want_stats["host_fallback"] == 0
    for part in ((0, 6), (3, 6), (5, 6), (1, 5), (4, 7), (2, 3)):
        cx = lom.ScanContext(g, partition=part)
        for _ in range(3):
            p = m.align(cx, scan, guess)
            assert m.stats["host_fallback"] == 0, part
            assert p.translation.tobytes() == want.translation.tobytes() and p.rotation.tobytes() == want.rotation.tobytes(), part
            for k in _keys("outer_iterations", "lm_iterations", "evaluations", "queries", "valid_last"):
                assert m.stats[k] == want_stats[k], (part, k)
        cx.close()


def test_partitioned_contexts_run_side_by_side_with_the_same_bits(lom):
    """lom_scan_create_on_partition: k contexts on k disjoint slices of the compute units (a CU mask on each context's
    stream).  Four threads, many aligns each, all at once: every result carries the bits of the same align issued alone
    through the map handle -- the slice changes where the kernels run, not what they compute -- and no solve ever finds
    its workgroups short of room (host_fallback == 0: each context's kernels have their slice to themselves)."""
    import threading

    sm = scenes.small_synth_case()
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    scan = sm["scan"]
    jobs = [(np.ascontiguousarray(scan[0::2]), lom.Pose3D((0.0, 0.0, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))),
            (np.ascontiguousarray(scan[1::2]), lom.Pose3D((0.2, -0.2, 0.0), scenes.angle_axis_q(0.01, (0, 0, 1)))),
            (scan, lom.Pose3D((0.05, 0.02, 0.0), (1, 0, 0, 0))),
            (np.ascontiguousarray(scan[::3]), lom.Pose3D((-0.1, 0.05, 0.02), scenes.angle_axis_q(-0.008, (0, 0, 1))))]
    m = lom.CloudMatcher()
    want = []
    for cloud, guess in jobs:
        p = m.align(g, cloud, guess)
        want.append((p.translation.tobytes(), p.rotation.tobytes(), dict(m.stats)))
    ctxs = [lom.ScanContext(g, partition=(i, len(jobs))) for i in range(len(jobs))]
    got = [[] for _ in jobs]
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(i):
        try:
            mm = lom.CloudMatcher()
            cloud, guess = jobs[i]
            start.wait()
            for _ in range(60):
                p = mm.align(ctxs[i], cloud, guess)
                got[i].append((p.translation.tobytes(), p.rotation.tobytes(), dict(mm.stats)))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i, w in enumerate(want):
        assert len(got[i]) == 60
        for t_, q_, st in got[i]:
            assert st["host_fallback"] == 0
            assert t_ == w[0] and q_ == w[1], i
            for k in _keys("outer_iterations", "lm_iterations", "evaluations", "queries", "cand_total", "valid_last"):
                assert st[k] == w[2][k], (i, k)
    # bad arguments are refused
    h = C.c_void_p()
    assert lom.capi.lib().lom_scan_create_on_partition(g.handle, 3, 3, C.byref(h)) == -1
    assert lom.capi.lib().lom_scan_create_on_partition(g.handle, 0, 9, C.byref(h)) == -1
    for cx in ctxs:
        cx.close()
