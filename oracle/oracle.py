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
        for f in ("oracle.c", "oracle.h", "Makefile")
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
        if getattr(self, "_h", None):
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
        if getattr(self, "_h", None):
            lib().orc_shard_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h
