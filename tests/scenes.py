"""Shared test inputs and protocols (test infrastructure).

* the reference's own unit-test vectors (test/test.cpp), restated as data;
* the MatchingTest protocol (test/test.cpp:191-264) over any implementation
  exposing VoxelGrid / CloudMatcher with the reference's method names;
* seeded synthetic cases sized for CPU-seconds.
"""
import numpy as np

from lidar_odometry_demo_amd import synth


# ---- reference test vectors (test/test.cpp) --------------------------------

# VoxelGrid.UniquePoints, test.cpp:28-35 (voxel 0.5, cap 1)
UNIQUE_POINTS = np.array(
    [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32)
# VoxelGrid.DuplicatePoints, test.cpp:59-63
DUPLICATE_POINTS = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0], [1, 0, 0]], np.float32)
# CloudTransformer.RigidTransform sample cloud, test.cpp:154-162
RIGID_POINTS = np.array(
    [[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)


def angle_axis_q(angle, axis):
    """Eigen::Quaternionf(AngleAxisf(angle, axis)) as wxyz f32 (axis used as given)."""
    a = np.float32(angle)
    ax = np.asarray(axis, np.float32)
    ha = np.float32(0.5) * a
    s = np.float32(np.sin(ha))
    return np.array([np.float32(np.cos(ha)), s * ax[0], s * ax[1], s * ax[2]], np.float32)


def _unit(v):
    v = np.asarray(v, np.float32)
    return v / np.float32(np.sqrt(np.float32((v * v).sum())))


def pose_pairs():
    """Pose3D.ComposeRelativeInverse, test.cpp:79-108: list of ((t,q),(t,q))."""
    z = (0, 0, 1)
    return [
        (((0, 0, 0), (1, 0, 0, 0)), ((0, 0, 0), (1, 0, 0, 0))),
        (((0, 0, 0), angle_axis_q(0.2, z)), ((0, 0, 0), angle_axis_q(0.2, z))),
        (((0, 0, 0), angle_axis_q(0, z)), ((1, 0, 0), angle_axis_q(np.pi * 0.5, z))),
        (((1, 0, 0), angle_axis_q(0, z)), ((1, 1, 1), angle_axis_q(-np.pi, z))),
        (((100, 100, 100), angle_axis_q(0, z)), ((150, 150, 150), angle_axis_q(0, z))),
        (((100, 100, 100), angle_axis_q(0.1, z)), ((150, 150, 150), angle_axis_q(-0.2, z))),
        (((1, 0.5, -0.5), angle_axis_q(0.456, _unit((0.1, 0.2, 1)))),
         ((-1, -0.6, 0), angle_axis_q(-0.245, _unit((-0.2, 0, 0))))),
    ]


def rigid_poses():
    """CloudTransformer.RigidTransform, test.cpp:165-174."""
    d = np.pi / 180.0
    return [
        ((0, 0, 0), angle_axis_q(0.0, (0, 0, 1))),
        ((0, 0, 0), angle_axis_q(80.0 * d, (0, 0, 1))),
        ((0, 0, 0), angle_axis_q(80.0 * d, (0, 1, 0))),
        ((0, 0, 0), angle_axis_q(90.0 * d, _unit((1, 0, 1)))),
        ((1, 1, 1), angle_axis_q(45.0 * d, _unit((0, 0.5, 0.5)))),
        ((-2, 2, 0), angle_axis_q(45.0 * d, _unit((0.5, 0.5, 0)))),
    ]


def matching_guess_poses():
    """CloudMatcher.MatchingTest, test.cpp:235-243."""
    d = np.pi / 180.0
    ident = (1, 0, 0, 0)
    return [
        ((0.0, 0.0, 0.0), ident),
        ((0.0, 0.0, 0.1), ident),
        ((0.1, 0.1, 0.1), ident),
        ((-0.1, -0.1, -0.1), ident),
        ((0.1, -0.1, 0), ident),
        ((0.0, 0.0, 0.0), angle_axis_q(-1.0 * d, (0, 0, 1))),
        ((-0.2, 0.0, 0.0), angle_axis_q(2.0 * d, (0, 0, 1))),
    ]


def se3_matrix(t, q):
    """f32 4x4 of a (t, q wxyz) pose -- the Eigen::Isometry3f side of test.cpp:118-132
    (translate then rotate; rotation matrix from the f32 quaternion as given)."""
    w, x, y, z = [np.float32(v) for v in q]
    two = np.float32(2)
    one = np.float32(1)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = np.array(
        [[one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
         [two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x)],
         [two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]], np.float32)
    M[:3, 3] = np.asarray(t, np.float32)
    return M


def se3_inverse(M):
    """Isometry inverse in f32: [R^T | -R^T t]."""
    out = np.eye(4, dtype=np.float32)
    Rt = M[:3, :3].T
    out[:3, :3] = Rt
    out[:3, 3] = -(Rt @ M[:3, 3])
    return out


def matrix_to_quat(R):
    """wxyz from a rotation matrix (Shepperd), f64."""
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


# ---- normals for the shipped data file -----------------------------------

def estimate_normals_radius(xyz, radius, details=False):
    """Restates pcl::NormalEstimation with setRadiusSearch(radius) as used by
    test/test.cpp:196-205: covariance of all points within `radius` (the point
    itself included), eigenvector of the smallest eigenvalue, flipped toward the
    viewpoint (0,0,0); NaN when fewer than 3 neighbours."""
    from scipy.spatial import cKDTree

    x = np.asarray(xyz, np.float64)
    n = len(x)
    pairs = cKDTree(x).query_pairs(radius, output_type="ndarray")
    i = np.concatenate([pairs[:, 0], pairs[:, 1]])
    j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    d = x[j] - x[i]
    cnt = np.bincount(i, minlength=n) + 1  # + self
    s1 = np.stack([np.bincount(i, d[:, a], minlength=n) for a in range(3)], 1)
    s2 = np.zeros((n, 3, 3))
    for a in range(3):
        for b in range(a, 3):
            s2[:, a, b] = s2[:, b, a] = np.bincount(i, d[:, a] * d[:, b], minlength=n)
    mu = s1 / cnt[:, None]
    cov = s2 / cnt[:, None, None] - mu[:, :, None] * mu[:, None, :]
    w, v = np.linalg.eigh(cov)
    nrm = v[:, :, 0]
    flip = (-(x) * nrm).sum(1) < 0  # (vp - p).n < 0, vp = 0
    nrm[flip] *= -1
    nrm[cnt < 3] = np.nan
    if details:   # + neighbour counts and the covariance's eigenvalues (ascending): how well defined a normal is
        return nrm.astype(np.float32), cnt, w
    return nrm.astype(np.float32)


# ---- MatchingTest protocol -------------------------------------------------

def run_matching_test(impl, xyz_full, xyzn, helper=None, n_poses=7):
    """test/test.cpp:226-262 with `impl` providing VoxelGrid/CloudMatcher/Pose3D.
    `helper` provides Pose3D algebra + transform_points for input preparation
    (defaults to impl)."""
    H = helper or impl
    keyframe = impl.VoxelGrid(0.25, 20)                       # :226
    keyframe.addCloud(xyzn[:, :3], xyzn[:, 3:])               # :227
    vf = impl.VoxelGrid(0.5, 1)                               # :229
    vf.addCloudWithoutNormals(xyz_full)                       # :230
    sub = vf.getCloudWithoutNormals()                         # :231
    matcher = impl.CloudMatcher()
    out = {"keyframe_voxels": keyframe.size(), "keyframe_points": keyframe.pointCount(),
           "source_points": int(len(sub)), "cases": []}
    for t, q in matching_guess_poses()[:n_poses]:
        guess = H.Pose3D(t, q)
        guess_cloud = H.transform_points(guess.inverse(), sub)   # :248
        final = matcher.align(keyframe, guess_cloud, impl.Pose3D())  # :250
        fh = H.Pose3D(final.translation, final.rotation)
        err = fh.relativeTo(guess)                                # :254
        rot_err = 1.0 - abs(float(np.dot(final.rotation.astype(np.float64),
                                         np.asarray(guess.rotation, np.float64))))  # :259
        st = {k: v for k, v in (matcher.stats or {}).items() if not k.endswith("seconds")}
        out["cases"].append({
            "guess_t": [float(v) for v in guess.translation],
            "guess_q_wxyz": [float(v) for v in guess.rotation],
            "final_t": [float(v) for v in final.translation],
            "final_q_wxyz": [float(v) for v in final.rotation],
            "err_t_norm": float(np.linalg.norm(err.translation.astype(np.float64))),
            "rot_err": rot_err,
            "stats": st,
        })
    return out


# ---- seeded synthetic cases ------------------------------------------------

def small_synth_case():
    """8 beams x 256 azimuth steps vs a 40k-point map (radius 40 m)."""
    boxes = synth.make_boxes()
    scan, ring, az, q = synth.make_scan(8, 256, boxes=boxes)
    mp, mn = synth.make_map_points(40_000, radius=40.0, boxes=boxes)
    return {"scan": scan, "map_xyz": mp, "map_nrm": mn, "true_q": q,
            "true_t": np.array([0.10, -0.05, 0.02])}


def synth_case(n_beams, n_az, map_points, radius=80.0):
    boxes = synth.make_boxes()
    scan, ring, az, q = synth.make_scan(n_beams, n_az, boxes=boxes)
    mp, mn = synth.make_map_points(map_points, radius=radius, boxes=boxes)
    return {"scan": scan, "ring": ring, "map_xyz": mp, "map_nrm": mn, "true_q": q,
            "true_t": np.array([0.10, -0.05, 0.02])}


def pose_delta(t_a, q_a, t_b, q_b):
    """(translation distance [m], rotation angle [rad]) between two poses."""
    qa = np.asarray(q_a, np.float64)
    qb = np.asarray(q_b, np.float64)
    qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
    d = min(1.0, abs(float(np.dot(qa, qb))))
    return (float(np.linalg.norm(np.asarray(t_a, np.float64) - np.asarray(t_b, np.float64))),
            2.0 * float(np.arccos(d)))
