"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
NSUMS = 32

EXPORT_FULL, EXPORT_FULL_NO_NORMALS, EXPORT_FIRST_PER_VOXEL = 0, 1, 2


def build(force=False):
    """Compile oracle.c with the committed Makefile (gcc, no third-party code)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("oracle.c", "pipeline.c", "oracle.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class Pose(C.Structure):
    _fields_ = [("t", C.c_float * 3), ("q", C.c_float * 4)]


class Corr(C.Structure):
    _fields_ = [
        ("index", C.c_int64),
        ("origin", C.c_float * 3),
        ("normal", C.c_float * 3),
        ("sq_dist", C.c_float),
        ("n_cand", C.c_uint32),
        ("n_occ", C.c_uint32),
    ]


CORR_DTYPE = np.dtype(
    [("index", "<i8"), ("origin", "<f4", 3), ("normal", "<f4", 3), ("sq_dist", "<f4"),
     ("n_cand", "<u4"), ("n_occ", "<u4")],
    align=True,
)


class AlignStats(C.Structure):
    _fields_ = [
        ("outer_iterations", C.c_int),
        ("lm_iterations", C.c_int),
        ("evaluations", C.c_int),
        ("queries", C.c_int64),
        ("valid_last", C.c_int64),
        ("cand_total", C.c_int64),
        ("occ_total", C.c_int64),
        ("final_cost", C.c_double),
        ("last_step_norm", C.c_double),
        ("search_seconds", C.c_double),
        ("solve_seconds", C.c_double),
        ("points_evaluated", C.c_int),
        ("pad", C.c_int),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    pp = C.POINTER(Pose)
    L.orc_pose_compose.argtypes = [pp, pp, pp]
    L.orc_pose_inverse.argtypes = [pp, pp]
    L.orc_pose_relative_to.argtypes = [pp, pp, pp]
    L.orc_pose_rotation_matrix.argtypes = [pp, fp]
    L.orc_transform_points.argtypes = [pp, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_map_create.restype = C.c_void_p
    L.orc_map_create.argtypes = [C.c_float, C.c_size_t]
    L.orc_map_destroy.argtypes = [C.c_void_p]
    L.orc_map_clear.argtypes = [C.c_void_p, C.c_float]
    L.orc_map_set_max_points.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_map_add_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.orc_map_radius_cleanup.argtypes = [C.c_void_p, fp, C.c_float]
    L.orc_map_size.restype = C.c_size_t
    L.orc_map_size.argtypes = [C.c_void_p]
    L.orc_map_point_count.restype = C.c_size_t
    L.orc_map_point_count.argtypes = [C.c_void_p]
    L.orc_map_export.restype = C.c_size_t
    L.orc_map_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_find_pairs.restype = C.c_int64
    L.orc_find_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, fp, fp,
                                 C.c_float, C.c_void_p, C.c_int]
    L.orc_align.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, fp, fp, fp, fp,
                            C.POINTER(AlignStats), C.c_int]
    L.orc_shard_create.restype = C.c_void_p
    L.orc_shard_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.orc_shard_destroy.argtypes = [C.c_void_p]
    L.orc_shard_match_eval.argtypes = [C.c_void_p, fp, fp, dp, dp, dp]
    L.orc_shard_eval_fixed.argtypes = [C.c_void_p, dp, dp, dp]
    _lib = L
    return L


def _f3(a):
    return (C.c_float * 3)(*[float(v) for v in a])


def _f4(a):
    return (C.c_float * 4)(*[float(v) for v in a])


def _xyz(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected (n, 3) float32")
    return a


class Pose3D:
    """f32 SE(3) value type, mirrors reference src/pose_3d.h."""

    def __init__(self, translation=(0, 0, 0), rotation_wxyz=(1, 0, 0, 0)):
        self.translation = np.asarray(translation, dtype=np.float32).copy()
        self.rotation = np.asarray(rotation_wxyz, dtype=np.float32).copy()

    def _c(self):
        return Pose(_f3(self.translation), _f4(self.rotation))

    @staticmethod
    def _from(c):
        return Pose3D(np.array(c.t[:], np.float32), np.array(c.q[:], np.float32))

    def compose(self, other):
        o = Pose()
        lib().orc_pose_compose(C.byref(self._c()), C.byref(other._c()), C.byref(o))
        return Pose3D._from(o)

    def inverse(self):
        o = Pose()
        lib().orc_pose_inverse(C.byref(self._c()), C.byref(o))
        return Pose3D._from(o)

    def relativeTo(self, target):
        o = Pose()
        lib().orc_pose_relative_to(C.byref(self._c()), C.byref(target._c()), C.byref(o))
        return Pose3D._from(o)

    def rotationMatrix(self):
        R = (C.c_float * 9)()
        lib().orc_pose_rotation_matrix(C.byref(self._c()), R)
        return np.array(R[:], np.float32).reshape(3, 3)


def transform_points(pose, xyz, nrm=None):
    """CloudTransformer::transform / transformWithNormals (utils/cloud_transform.h:43-97)."""
    xyz = _xyz(xyz)
    out = np.empty_like(xyz)
    nout = None
    if nrm is not None:
        nrm = _xyz(nrm)
        nout = np.empty_like(nrm)
    lib().orc_transform_points(C.byref(pose._c()), xyz.ctypes.data,
                               nrm.ctypes.data if nrm is not None else None, len(xyz), 12,
                               out.ctypes.data, nout.ctypes.data if nrm is not None else None, 12)
    return (out, nout) if nrm is not None else out


class VoxelGrid:
    """CPU restatement of reference src/voxel_grid.h (method names kept)."""

    def __init__(self, voxel_size=0.5, max_points=10):
        self._h = lib().orc_map_create(float(voxel_size), int(max_points))
        if not self._h:
            raise ValueError("orc_map_create failed")
        self.max_points = int(max_points)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # module globals may be gone at interpreter exit
            lib().orc_map_destroy(self._h)
            self._h = None

    @staticmethod
    def _chk(rc):
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")

    def setVoxelSize(self, v):
        self._chk(lib().orc_map_clear(self._h, float(v)))

    def setMaxPoints(self, k):
        self._chk(lib().orc_map_set_max_points(self._h, int(k)))
        self.max_points = int(k)

    def addCloud(self, xyz, normals):
        xyz, normals = _xyz(xyz), _xyz(normals)
        assert len(xyz) == len(normals)
        self._chk(lib().orc_map_add_points(self._h, xyz.ctypes.data, normals.ctypes.data, len(xyz), 12))

    def addCloudWithoutNormals(self, xyz):
        xyz = _xyz(xyz)
        self._chk(lib().orc_map_add_points(self._h, xyz.ctypes.data, None, len(xyz), 12))

    def size(self):
        return int(lib().orc_map_size(self._h))

    def pointCount(self):
        return int(lib().orc_map_point_count(self._h))

    def _export(self, mode, want_normals):
        n = lib().orc_map_export(self._h, mode, None, None, 0)
        xyz = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normals else None
        lib().orc_map_export(self._h, mode, xyz.ctypes.data,
                             nrm.ctypes.data if want_normals else None, n)
        return xyz, nrm

    def getCloud(self):
        return self._export(EXPORT_FULL, True)

    def getCloudWithoutNormals(self):
        return self._export(EXPORT_FULL_NO_NORMALS, False)[0]

    def getSparseCloudWithoutNormals(self):
        return self._export(EXPORT_FIRST_PER_VOXEL, False)[0]

    def radiusCleanup(self, point, radius):
        self._chk(lib().orc_map_radius_cleanup(self._h, _f3(point), float(radius)))

    def findMatchingPairs(self, xyz, pose, max_dist=0.3, nthreads=1):
        """Returns a structured array (CORR_DTYPE) in query order; index<0 = no match."""
        xyz = _xyz(xyz)
        out = np.zeros(len(xyz), CORR_DTYPE)
        assert out.itemsize == C.sizeof(Corr)
        rc = lib().orc_find_pairs(self._h, xyz.ctypes.data, len(xyz), 12, _f3(pose.translation),
                                  _f4(pose.rotation), float(max_dist), out.ctypes.data, nthreads)
        if rc < 0:
            raise RuntimeError(f"oracle error {rc}")
        return out


    def getCorrespondence(self, query, max_correspondence_distance_sq):
        """voxel_grid.h:164-204: one f32 query in the map frame, the squared threshold a double."""
        out = np.zeros(1, CORR_DTYPE)
        fn = lib().orc_get_correspondence
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        fn.restype = C.c_int
        rc = fn(self._h, _f3(query), float(max_correspondence_distance_sq), out.ctypes.data)
        if rc < 0:
            raise RuntimeError(f"oracle error {rc}")
        return out[0]


class CloudMatcher:
    """CPU restatement of reference src/cloud_matcher.cpp:105-178."""

    def __init__(self, nthreads=1):
        self.nthreads = nthreads
        self.stats = None

    def align(self, keyframe, xyz, guess):
        xyz = _xyz(xyz)
        ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
        st = AlignStats()
        rc = lib().orc_align(keyframe._h, xyz.ctypes.data, len(xyz), 12, _f3(guess.translation),
                             _f4(guess.rotation), ot, oq, C.byref(st), self.nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")
        self.stats = st.asdict()
        return Pose3D(np.array(ot[:], np.float32), np.array(oq[:], np.float32))


class Shard:
    """Per-rank evaluator over a contiguous source range (tests only)."""

    def __init__(self, keyframe, xyz):
        self._xyz = _xyz(xyz)  # keep alive
        self._kf = keyframe
        self._h = lib().orc_shard_create(keyframe._h, self._xyz.ctypes.data, len(self._xyz), 12)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().orc_shard_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def match_eval(self, pose_t, pose_q, q, t):
        """New correspondences at the f32 pose, then the NSUMS reduced sums at the f64 point (q, t)."""
        out = (C.c_double * NSUMS)()
        rc = lib().orc_shard_match_eval(self._h, _f3(pose_t), _f4(pose_q), (C.c_double * 4)(*[float(v) for v in q]),
                                        (C.c_double * 3)(*[float(v) for v in t]), out)
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")
        return np.array(out[:], np.float64)

    def eval_fixed(self, q, t):
        """The sums at (q, t) for the correspondences of the last match_eval."""
        out = (C.c_double * NSUMS)()
        lib().orc_shard_eval_fixed(self._h, (C.c_double * 4)(*[float(v) for v in q]),
                                   (C.c_double * 3)(*[float(v) for v in t]), out)
        return np.array(out[:], np.float64)


# ---- callers of the hot path (oracle/pipeline.c), tests only ---------------------------------

POINT_XYZIRT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"),
                         ("ring", "<u2"), ("pad1", "<u2"), ("time", "<f4"), ("pad2", "<f4")])


class OdomParams(C.Structure):
    _fields_ = [("lidar_min_range", C.c_float), ("lidar_max_range", C.c_float), ("keyframe_voxel_size", C.c_float),
                ("keyframe_max_points_cnt", C.c_uint32), ("keyframe_matching_voxel_size", C.c_float),
                ("keyframe_update_voxel_size", C.c_float), ("keyframe_cleanup_range", C.c_float),
                ("angular_divergence_threshold", C.c_float)]


class OdomFrameStats(C.Structure):
    _fields_ = [("planar_points", C.c_int64), ("filtered_points", C.c_int64), ("update_points", C.c_int64),
                ("matching_points", C.c_int64), ("keyframe_voxels", C.c_int64), ("queries", C.c_int64),
                ("outer_iterations", C.c_int32), ("initialised_keyframe", C.c_int32),
                ("unstable_rotation", C.c_int32), ("pad", C.c_int32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


def _pipeline_lib():
    L = lib()
    if getattr(L, "_pipeline_ready", False):
        return L
    pp = C.POINTER(Pose)
    L.orc_time_normalize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_time_normalize.restype = None
    L.orc_transform_non_rigid.argtypes = [C.c_void_p, C.c_size_t, pp, pp, C.c_void_p]
    L.orc_transform_non_rigid.restype = None
    L.orc_range_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_range_filter.restype = C.c_size_t
    L.orc_classify.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t),
                               C.POINTER(C.c_size_t)]
    L.orc_classify.restype = C.c_size_t
    L.orc_odom_default_params.argtypes = [C.POINTER(OdomParams)]
    L.orc_odom_default_params.restype = None
    L.orc_odom_create.argtypes = [C.POINTER(OdomParams)]
    L.orc_odom_create.restype = C.c_void_p
    L.orc_odom_destroy.argtypes = [C.c_void_p]
    L.orc_odom_destroy.restype = None
    L.orc_odom_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.orc_odom_get_pose.argtypes = [C.c_void_p, pp]
    L.orc_odom_get_pose.restype = None
    L.orc_odom_get_stats.argtypes = [C.c_void_p, C.POINTER(OdomFrameStats)]
    L.orc_odom_get_stats.restype = None
    L.orc_odom_keyframe.argtypes = [C.c_void_p]
    L.orc_odom_keyframe.restype = C.c_void_p
    L._pipeline_ready = True
    return L


def _cloud(points):
    a = np.ascontiguousarray(points, dtype=POINT_XYZIRT)
    assert a.ndim == 1
    return a


def pointTimeNormalize(points):
    a = _cloud(points)
    out = np.empty_like(a)
    _pipeline_lib().orc_time_normalize(a.ctypes.data, len(a), out.ctypes.data)
    return out


def transformNonRigid(points, start_pose, end_pose):
    a = _cloud(points)
    out = np.empty_like(a)
    _pipeline_lib().orc_transform_non_rigid(a.ctypes.data, len(a), C.byref(start_pose._c()), C.byref(end_pose._c()),
                                            out.ctypes.data)
    return out


def rangeFilter(xyz, normals, min_range, max_range):
    xyz = _xyz(xyz)
    normals = _xyz(normals) if normals is not None else None
    oxyz = np.empty_like(xyz)
    onrm = np.empty_like(xyz) if normals is not None else None
    n = _pipeline_lib().orc_range_filter(xyz.ctypes.data, normals.ctypes.data if normals is not None else None,
                                         len(xyz), float(min_range), float(max_range), oxyz.ctypes.data,
                                         onrm.ctypes.data if normals is not None else None)
    return (oxyz[:n], onrm[:n]) if normals is not None else oxyz[:n]


def classify(points):
    a = _cloud(points)
    xyz = np.empty((max(len(a), 1), 3), np.float32)
    nrm = np.empty((max(len(a), 1), 3), np.float32)
    nu = C.c_size_t()
    grid = (C.c_size_t * 2)()
    n = _pipeline_lib().orc_classify(a.ctypes.data, len(a), xyz.ctypes.data, nrm.ctypes.data, C.byref(nu), grid)
    return xyz[:n].copy(), nrm[:n].copy(), int(nu.value), (int(grid[0]), int(grid[1]))


class LidarOdometry:
    """CPU restatement of reference src/lidar_odometry.{h,cpp}."""

    def __init__(self, nthreads=1, **params):
        L = _pipeline_lib()
        p = OdomParams()
        L.orc_odom_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        self._h = L.orc_odom_create(C.byref(p))
        self.nthreads = nthreads

    def __del__(self):
        if getattr(self, "_h", None) and _pipeline_lib is not None:
            _pipeline_lib().orc_odom_destroy(self._h)
            self._h = None

    def processCloud(self, input_cloud):
        a = _cloud(input_cloud)
        rc = _pipeline_lib().orc_odom_process(self._h, a.ctypes.data, len(a), self.nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")

    def getCurrentPose(self):
        p = Pose()
        _pipeline_lib().orc_odom_get_pose(self._h, C.byref(p))
        return Pose3D._from(p)

    def _export(self, mode):
        kf = _pipeline_lib().orc_odom_keyframe(self._h)
        n = lib().orc_map_export(kf, mode, None, None, 0)
        xyz = np.empty((n, 3), np.float32)
        lib().orc_map_export(kf, mode, xyz.ctypes.data, None, n)
        return xyz

    def getFullKeyFrameCloudWithNormals(self):
        kf = _pipeline_lib().orc_odom_keyframe(self._h)
        n = lib().orc_map_export(kf, EXPORT_FULL, None, None, 0)
        xyz, nrm = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        lib().orc_map_export(kf, EXPORT_FULL, xyz.ctypes.data, nrm.ctypes.data, n)
        return xyz, nrm

    def getKeyFrameCloud(self):
        return self._export(EXPORT_FIRST_PER_VOXEL)

    def getFullKeyFrameCloud(self):
        return self._export(EXPORT_FULL_NO_NORMALS)

    @property
    def stats(self):
        s = OdomFrameStats()
        _pipeline_lib().orc_odom_get_stats(self._h, C.byref(s))
        return s.asdict()
