"""MI355X-native scan-matching core behind the reference's VoxelGrid /
CloudMatcher / Pose3D interface (vovo-4K/lidar_odometry_demo:
src/voxel_grid.h, src/cloud_matcher.h, src/pose_3d.h).

Python mirror of the reference classes over the C ABI of
include/lidar_odometry_amd.h -- same names, argument meaning and error
behaviour, so parity tests read like the reference's own tests.  All compute
runs in the HIP library; there is no CPU path in this package.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import LomError  # noqa: F401

__all__ = ["Pose3D", "VoxelGrid", "CloudMatcher", "ScanContext", "LidarOdometry", "transform_points", "pointTimeNormalize",
           "transformNonRigid", "rangeFilter", "classify", "loadPCDFile", "fromROSMsg", "toROSMsg", "estimateNormals", "FrontEnd", "LomError", "capi"]


class Pose3D:
    """reference src/pose_3d.h:10-59 (f32 translation + wxyz quaternion)."""

    def __init__(self, translation=(0, 0, 0), rotation_wxyz=(1, 0, 0, 0)):
        self.translation = np.asarray(translation, dtype=np.float32).copy()
        self.rotation = np.asarray(rotation_wxyz, dtype=np.float32).copy()

    def _c(self):
        return capi.Pose(capi.f3(self.translation), capi.f4(self.rotation))

    @staticmethod
    def _from(c):
        return Pose3D(np.array(c.t[:], np.float32), np.array(c.q[:], np.float32))

    def compose(self, another):          # pose_3d.h:29-32
        o = capi.Pose()
        capi.lib().lom_pose_compose(C.byref(self._c()), C.byref(another._c()), C.byref(o))
        return Pose3D._from(o)

    def inverse(self):                   # pose_3d.h:34-39
        o = capi.Pose()
        capi.lib().lom_pose_inverse(C.byref(self._c()), C.byref(o))
        return Pose3D._from(o)

    def relativeTo(self, target):        # pose_3d.h:23-27
        o = capi.Pose()
        capi.lib().lom_pose_relative_to(C.byref(self._c()), C.byref(target._c()), C.byref(o))
        return Pose3D._from(o)

    def rotationMatrix(self):            # pose_3d.h:41-43
        R = (C.c_float * 9)()
        capi.lib().lom_pose_rotation_matrix(C.byref(self._c()), R)
        return np.array(R[:], np.float32).reshape(3, 3)

    def __repr__(self):
        return f"Pose3D(t={self.translation.tolist()}, q_wxyz={self.rotation.tolist()})"


def transform_points(pose, xyz, normals=None):
    """CloudTransformer::transform / transformWithNormals (src/utils/cloud_transform.h:43-97)."""
    xyz = capi.xyz_array(xyz)
    out = np.empty_like(xyz)
    nout = None
    if normals is not None:
        normals = capi.xyz_array(normals)
        nout = np.empty_like(normals)
    capi.check(capi.lib().lom_transform_points(
        C.byref(pose._c()), xyz.ctypes.data, normals.ctypes.data if normals is not None else None,
        len(xyz), 12, out.ctypes.data, nout.ctypes.data if normals is not None else None, 12))
    return (out, nout) if normals is not None else out


class VoxelGrid:
    """reference src/voxel_grid.h:17-257, device-resident."""

    def __init__(self, voxel_size=0.5, max_points=10, capacity_hint=0, device=0):
        h = C.c_void_p()
        rc = capi.lib().lom_map_create(float(voxel_size), int(max_points), int(capacity_hint), int(device),
                                       C.byref(h))
        if rc != 0:
            capi.check(rc, None)
        self._h = h
        self.max_points = int(max_points)
        self.device = int(device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and capi is not None:          # at interpreter shutdown the module may already be gone
            capi.lib().lom_map_destroy(h)
            self._h = None

    close = __del__

    @property
    def handle(self):
        return self._h

    def setOption(self, option, value):
        """lom_map_set_option: run-time switches of the handle (capi.OPT_*)."""
        capi.check(capi.lib().lom_map_set_option(self._h, int(option), int(value)), self._h)

    def debugCounter(self, which=capi.COUNTER_GRID_REDOS):
        return capi.check(capi.lib().lom_map_debug_counter(self._h, int(which)), self._h)

    def setMaxPoints(self, max_points):                    # voxel_grid.h:56-59
        capi.check(capi.lib().lom_map_set_max_points(self._h, int(max_points)), self._h)
        self.max_points = int(max_points)

    def setVoxelSize(self, voxel_size):                    # voxel_grid.h:61-66 (clears)
        capi.check(capi.lib().lom_map_clear(self._h, float(voxel_size)), self._h)

    def addCloud(self, xyz, normals):                      # voxel_grid.h:77-93
        xyz, normals = capi.xyz_array(xyz), capi.xyz_array(normals)
        if len(xyz) != len(normals):
            raise ValueError("xyz and normals differ in length")
        capi.check(capi.lib().lom_map_add_points(self._h, xyz.ctypes.data, normals.ctypes.data, len(xyz), 12),
                   self._h)

    def addCloudInterleaved(self, records, stride_bytes, normal_offset_bytes=None):
        """pcl-style records (e.g. 48-byte PointNormal: xyz at 0, normal at 16) without repacking."""
        buf = np.ascontiguousarray(records)
        n = buf.nbytes // stride_bytes
        base = buf.ctypes.data
        nrm = base + normal_offset_bytes if normal_offset_bytes is not None else None
        capi.check(capi.lib().lom_map_add_points(self._h, base, nrm, n, stride_bytes), self._h)

    def addCloudWithoutNormals(self, xyz):                 # voxel_grid.h:95-110
        xyz = capi.xyz_array(xyz)
        capi.check(capi.lib().lom_map_add_points(self._h, xyz.ctypes.data, None, len(xyz), 12), self._h)

    def addCloudDevice(self, d_xyz_ptr, d_nrm_ptr, n, stride_bytes=12):
        capi.check(capi.lib().lom_map_add_points_device(self._h, d_xyz_ptr, d_nrm_ptr, int(n), int(stride_bytes)),
                   self._h)

    def size(self):                                        # voxel_grid.h:248-251
        return int(capi.check(capi.lib().lom_map_size(self._h), self._h))

    def pointCount(self):
        return int(capi.check(capi.lib().lom_map_point_count(self._h), self._h))

    def _export(self, mode, want_normals):
        n = capi.check(capi.lib().lom_map_export(self._h, mode, None, None, 0), self._h)
        xyz = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normals else None
        if n:
            capi.check(capi.lib().lom_map_export(self._h, mode, xyz.ctypes.data,
                                                 nrm.ctypes.data if want_normals else None, n), self._h)
        return xyz, nrm

    def getCloud(self):                                    # voxel_grid.h:112-130
        return self._export(capi.EXPORT_FULL, True)

    def getCloudWithoutNormals(self):                      # voxel_grid.h:133-147
        return self._export(capi.EXPORT_FULL_NO_NORMALS, False)[0]

    def getSparseCloudWithoutNormals(self):                # voxel_grid.h:150-162
        return self._export(capi.EXPORT_FIRST_PER_VOXEL, False)[0]

    def downsample(self, xyz, normals, voxel_size):
        """VoxelGrid(voxel_size, 1).addCloud(...).getCloud() in one fused pass (lidar_odometry.cpp:37-47);
        this grid is only the workspace and is left empty."""
        xyz = capi.xyz_array(xyz)
        normals = capi.xyz_array(normals) if normals is not None else None
        oxyz = np.empty_like(xyz)
        onrm = np.empty_like(xyz)
        n = capi.check(capi.lib().lom_voxel_downsample(
            self._h, float(voxel_size), xyz.ctypes.data, normals.ctypes.data if normals is not None else None,
            len(xyz), 12, oxyz.ctypes.data, onrm.ctypes.data, len(xyz)), self._h)
        return oxyz[:n].copy(), onrm[:n].copy()

    def radiusCleanup(self, point, radius):                # voxel_grid.h:236-246
        capi.check(capi.lib().lom_map_radius_cleanup(self._h, capi.f3(point), float(radius)), self._h)

    def radiusCleanupAfterAlign(self, radius):
        """Arm the next align on this grid to enqueue the scan of radiusCleanup(<its result translation>, radius) behind
        itself (lidar_odometry.cpp:65-67's pattern); the radiusCleanup that follows takes it if its arguments match."""
        capi.check(capi.lib().lom_map_radius_cleanup_after_align(self._h, float(radius)), self._h)

    def findMatchingPairs(self, xyz, transform, max_correspondence_distance=0.3):
        """voxel_grid.h:206-234; one entry per source point in source order (index < 0: no match)."""
        xyz = capi.xyz_array(xyz)
        out = np.zeros(len(xyz), capi.CORR_DTYPE)
        capi.check(capi.lib().lom_match_find_pairs(
            self._h, xyz.ctypes.data, len(xyz), 12, capi.f3(transform.translation), capi.f4(transform.rotation),
            float(max_correspondence_distance), out.ctypes.data), self._h)
        return out

    def findMatchingPairsSq(self, xyz, transform, max_correspondence_distance_sq):
        """The search with the squared threshold as getCorrespondence takes it (voxel_grid.h:164: a double)."""
        xyz = capi.xyz_array(xyz)
        out = np.zeros(len(xyz), capi.CORR_DTYPE)
        capi.check(capi.lib().lom_match_find_pairs_sq(
            self._h, xyz.ctypes.data, len(xyz), 12, capi.f3(transform.translation), capi.f4(transform.rotation),
            float(max_correspondence_distance_sq), out.ctypes.data), self._h)
        return out

    def findMatchingPairsAfter(self, xyz, previous_transform, transform, max_correspondence_distance=0.3):
        """Parity entry: the search at `transform` with the temporal pruning bound taken from a search at
        `previous_transform` (what outer iterations >= 2 of an align run); equals findMatchingPairs(xyz, transform)."""
        xyz = capi.xyz_array(xyz)
        out = np.zeros(len(xyz), capi.CORR_DTYPE)
        capi.check(capi.lib().lom_debug_find_pairs_after(
            self._h, xyz.ctypes.data, len(xyz), 12, capi.f3(previous_transform.translation),
            capi.f4(previous_transform.rotation), capi.f3(transform.translation), capi.f4(transform.rotation),
            float(max_correspondence_distance), out.ctypes.data), self._h)
        return out

    def getCorrespondence(self, query, max_correspondence_distance_sq):
        """voxel_grid.h:164-204 for a single already-transformed f32 query point; the threshold is the reference's
        `double max_correspondence_distance_sq`, handed over as it is."""
        return self.findMatchingPairsSq(np.asarray(query, np.float32).reshape(1, 3), Pose3D(), max_correspondence_distance_sq)[0]

    def profileMatch(self, d_src_ptr, n, transform, max_correspondence_distance=0.3, reps=20, stride_bytes=12):
        """(average k_match launch duration [us] over a back-to-back train under one event pair, algorithmic
        bytes per launch, bytes the kernel requests per launch, average of the same launches with one event
        pair each [us])."""
        us, by, rq, pr = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        capi.check(capi.lib().lom_profile_match(
            self._h, d_src_ptr, int(n), int(stride_bytes), capi.f3(transform.translation),
            capi.f4(transform.rotation), float(max_correspondence_distance), int(reps), C.byref(us), C.byref(by),
            C.byref(rq), C.byref(pr)), self._h)
        return us.value, by.value, rq.value, pr.value

    def profileInsert(self, d_xyz_ptr, d_nrm_ptr, n, stride_bytes=12):
        """addCloud of device-resident points, all kernels of the insert under one HIP event pair: microseconds."""
        us = C.c_double()
        capi.check(capi.lib().lom_profile_insert(self._h, d_xyz_ptr, d_nrm_ptr, int(n), int(stride_bytes), C.byref(us)),
                   self._h)
        return us.value

    def setProfiling(self, period):
        """HIP event pairs around the correspondence launches of every `period`-th align (True = 1, False = 0)."""
        capi.check(capi.lib().lom_map_set_profiling(self._h, int(period)), self._h)


class ScanContext:
    """lom_scan: what a further caller of ONE keyframe owns (stream, per-scan buffers, solve state) -- the reference's
    search and align take the grid by const reference (voxel_grid.h:206, cloud_matcher.h:15), so several threads may
    align against one keyframe at a time.  Pass it to CloudMatcher.align / alignDevice / align_repeat in place of the
    grid.  One caller per context; nobody changes the grid while contexts are in use."""

    def __init__(self, keyframe, partition=None):
        """partition = (index, count): the context's stream runs on that slice of the GPU's compute units
        (lom_scan_create_on_partition) -- for `count` callers side by side."""
        h = C.c_void_p()
        if partition is None:
            capi.check(capi.lib().lom_scan_create(keyframe.handle, C.byref(h)), keyframe.handle)
        else:
            capi.check(capi.lib().lom_scan_create_on_partition(keyframe.handle, int(partition[0]), int(partition[1]),
                                                               C.byref(h)), keyframe.handle)
        self._h = h
        self.keyframe = keyframe          # keeps the grid alive

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and capi is not None:
            capi.lib().lom_scan_destroy(h)
            self._h = None

    close = __del__

    @property
    def handle(self):
        return self._h

    def setOption(self, option, value):
        if capi.lib().lom_scan_set_option(self._h, int(option), int(value)) != 0:
            raise LomError(-1, "lom_scan_set_option")

    def _check(self, rc):
        if rc < 0:
            text = capi.lib().lom_scan_last_error(self._h)
            raise LomError(int(rc), text.decode() if text else "")
        return rc


def _align_entry(keyframe, name):
    """(function, checker) for a grid or a scan context"""
    L = capi.lib()
    if isinstance(keyframe, ScanContext):
        return getattr(L, "lom_scan_" + name), keyframe._check
    return getattr(L, "lom_match_" + name), (lambda rc: capi.check(rc, keyframe.handle))


def align_repeat(keyframe, d_src_ptr, n, position_guess, reps, stride_bytes=12):
    """`reps` back-to-back aligns of a device-resident scan issued from compiled code
    (lom_match_align_repeat): (pose of the last one, accumulated stats)."""
    ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
    st = capi.AlignStats()
    fn, chk = _align_entry(keyframe, "align_repeat")
    chk(fn(keyframe.handle, d_src_ptr, int(n), int(stride_bytes), capi.f3(position_guess.translation),
           capi.f4(position_guess.rotation), int(reps), ot, oq, C.byref(st)))
    return Pose3D(np.array(ot[:], np.float32), np.array(oq[:], np.float32)), st.asdict()


class CloudMatcher:
    """reference src/cloud_matcher.h:13-17 / src/cloud_matcher.cpp:105-178."""

    def __init__(self):
        self.stats = None

    def align(self, keyframe, planar_cloud, position_guess):
        xyz = capi.xyz_array(planar_cloud)
        ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
        st = capi.AlignStats()
        fn, chk = _align_entry(keyframe, "align")
        chk(fn(keyframe.handle, xyz.ctypes.data, len(xyz), 12, capi.f3(position_guess.translation),
               capi.f4(position_guess.rotation), ot, oq, C.byref(st)))
        self.stats = st.asdict()
        return Pose3D(np.array(ot[:], np.float32), np.array(oq[:], np.float32))

    def debugEvalSums(self, keyframe, planar_cloud, transform, q=None, t=None):
        """One search at the f32 pose `transform`, then the LOM_NSUMS reduced sums of
        PointToPlaneErrorAnalytic::Evaluate (cloud_matcher.cpp:38-103) at the f64 point (q, t)
        (default: the widened pose) -- parity entry, host-driven path's kernels."""
        xyz = capi.xyz_array(planar_cloud)
        q = np.asarray(transform.rotation if q is None else q, np.float64)
        t = np.asarray(transform.translation if t is None else t, np.float64)
        out = (C.c_double * 32)()
        capi.check(capi.lib().lom_debug_eval_sums(
            keyframe.handle, xyz.ctypes.data, len(xyz), 12, capi.f3(transform.translation), capi.f4(transform.rotation),
            (C.c_double * 4)(*q), (C.c_double * 3)(*t), out), keyframe.handle)
        return np.array(out[:], np.float64)

    def debugLmTrace(self, keyframe, planar_cloud, position_guess, outer_index=0):
        """align() on the device-resident path plus what k_lm's policy saw in outer iteration
        `outer_index`: [(x[7], sums[32]), ...] per evaluation.  Returns (pose, trace)."""
        xyz = capi.xyz_array(planar_cloud)
        ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
        st = capi.AlignStats()
        raw = (C.c_double * 200)()
        ne = C.c_int()
        capi.check(capi.lib().lom_debug_lm_trace(
            keyframe.handle, xyz.ctypes.data, len(xyz), 12, capi.f3(position_guess.translation),
            capi.f4(position_guess.rotation), int(outer_index), raw, C.byref(ne), ot, oq, C.byref(st)), keyframe.handle)
        self.stats = st.asdict()
        a = np.array(raw[:], np.float64).reshape(5, 40)
        trace = [(a[e, :7].copy(), a[e, 8:40].copy()) for e in range(ne.value)]
        return Pose3D(np.array(ot[:], np.float32), np.array(oq[:], np.float32)), trace

    def alignDevice(self, keyframe, d_src_ptr, n, position_guess, stride_bytes=12):
        """Source cloud already resident in HBM (device pointer, e.g. torch tensor.data_ptr())."""
        ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
        st = capi.AlignStats()
        fn, chk = _align_entry(keyframe, "align_device")
        chk(fn(keyframe.handle, d_src_ptr, int(n), int(stride_bytes), capi.f3(position_guess.translation),
               capi.f4(position_guess.rotation), ot, oq, C.byref(st)))
        self.stats = st.asdict()
        return Pose3D(np.array(ot[:], np.float32), np.array(oq[:], np.float32))


# ---- callers of the path: host code in the library over the C ABI --------

def _cloud(points):
    a = np.ascontiguousarray(points, dtype=capi.POINT_XYZIRT)
    if a.ndim != 1:
        raise ValueError("expected a 1-d array of POINT_XYZIRT records")
    return a


def pointTimeNormalize(points):
    """utils::pointTimeNormalize (src/utils/point_time_normalize.h:15-39)."""
    a = _cloud(points)
    out = np.empty_like(a)
    capi.lib().lom_point_time_normalize(a.ctypes.data, len(a), out.ctypes.data)
    return out


def transformNonRigid(points, start_pose, end_pose):
    """CloudTransformer::transformNonRigid (src/utils/cloud_transform.h:15-40)."""
    a = _cloud(points)
    out = np.empty_like(a)
    capi.lib().lom_transform_non_rigid(a.ctypes.data, len(a), C.byref(start_pose._c()), C.byref(end_pose._c()),
                                       out.ctypes.data)
    return out


def rangeFilter(xyz, normals, min_range, max_range):
    """utils::rangeFilter (src/utils/range_filter.h:13-28) on packed xyz (+ normals)."""
    xyz = capi.xyz_array(xyz)
    normals = capi.xyz_array(normals) if normals is not None else None
    oxyz = np.empty_like(xyz)
    onrm = np.empty_like(xyz) if normals is not None else None
    n = capi.lib().lom_range_filter(xyz.ctypes.data, normals.ctypes.data if normals is not None else None, len(xyz),
                                    float(min_range), float(max_range), oxyz.ctypes.data,
                                    onrm.ctypes.data if normals is not None else None)
    return (oxyz[:n], onrm[:n]) if normals is not None else oxyz[:n]


def classify(points):
    """CloudClassifier::classify (src/utils/cloud_classifier.h:19-168):
    (planar xyz, planar normals, number of unclassified points, (height, width) of the organised cloud)."""
    a = _cloud(points)
    xyz = np.empty((max(len(a), 1), 3), np.float32)
    nrm = np.empty((max(len(a), 1), 3), np.float32)
    nu = C.c_size_t()
    grid = (C.c_size_t * 2)()
    n = capi.lib().lom_cloud_classify(a.ctypes.data, len(a), xyz.ctypes.data, nrm.ctypes.data, C.byref(nu), grid)
    return xyz[:n].copy(), nrm[:n].copy(), int(nu.value), (int(grid[0]), int(grid[1]))


def estimateNormals(xyz, radius, device=0, with_counts=False):
    """pcl::NormalEstimation with setRadiusSearch(radius), viewpoint (0, 0, 0) (test/test.cpp:196-205) on the
    device: (n, 3) float32 normals, NaN where fewer than 3 neighbours exist."""
    xyz = capi.xyz_array(xyz)
    nrm = np.empty_like(xyz)
    cnt = np.empty(len(xyz), np.uint32) if with_counts else None
    capi.check(capi.lib().lom_estimate_normals(xyz.ctypes.data, len(xyz), 12, float(radius), int(device), nrm.ctypes.data,
                                               cnt.ctypes.data if with_counts else None))
    return (nrm, cnt) if with_counts else nrm


class FrontEnd:
    """The per-frame front end on the device (csrc/frontend.hip): pointTimeNormalize + transformNonRigid +
    CloudClassifier::classify + rangeFilter, results left in HBM."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = capi.lib().lom_frontend_create(int(device), None, C.byref(h))
        if rc != 0:
            raise LomError(int(rc), "lom_frontend_create failed")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and capi is not None:
            capi.lib().lom_frontend_destroy(h)
            self._h = None

    def _check(self, rc):
        if rc < 0:
            raise LomError(int(rc), capi.lib().lom_frontend_last_error(self._h).decode())
        return rc

    def process(self, points, start_pose, end_pose, min_range, max_range):
        """Returns dict(deskewed, planar_points, xyz, normals, grid, redo_on_host)."""
        a = _cloud(points)
        L = capi.lib()
        self._check(L.lom_frontend_process(self._h, a.ctypes.data, len(a), C.byref(start_pose._c()),
                                           C.byref(end_pose._c()), float(min_range), float(max_range)))
        counts = (C.c_uint32 * 4)()
        redo = self._check(L.lom_frontend_wait(self._h, counts))
        desk = np.empty(len(a), capi.POINT_XYZIRT)
        self._check(L.lom_frontend_fetch(self._h, 0, desk.ctypes.data, None, len(a)))
        nf = int(counts[1])
        xyz, nrm = np.empty((nf, 3), np.float32), np.empty((nf, 3), np.float32)
        self._check(L.lom_frontend_fetch(self._h, 1, xyz.ctypes.data, nrm.ctypes.data, nf))
        return dict(deskewed=desk, planar_points=int(counts[0]), xyz=xyz, normals=nrm,
                    grid=(int(counts[2]), int(counts[3])), redo_on_host=bool(redo))

    def sinf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self._check(capi.lib().lom_debug_sinf(self._h, x.ctypes.data, x.size, out.ctypes.data))
        return out


def loadPCDFile(path, with_normals=False):
    """pcl::io::loadPCDFile<pcl::PointXYZ> (test/test.cpp:194) without PCL: (n, 3) float32 xyz
    (and normals when asked; zeros if the file has none)."""
    info = capi.PcdInfo()
    n = capi.lib().lom_pcd_read(str(path).encode(), None, None, 0, C.byref(info))
    if n < 0:
        raise LomError(int(n), capi.lib().lom_pcd_last_error().decode())
    xyz = np.empty((n, 3), np.float32)
    nrm = np.empty((n, 3), np.float32) if with_normals else None
    m = capi.lib().lom_pcd_read(str(path).encode(), xyz.ctypes.data, nrm.ctypes.data if with_normals else None, n, None)
    if m < 0:
        raise LomError(int(m), capi.lib().lom_pcd_last_error().decode())
    return (xyz, nrm) if with_normals else xyz


# sensor_msgs/msg/PointField datatypes
PF_INT8, PF_UINT8, PF_INT16, PF_UINT16, PF_INT32, PF_UINT32, PF_FLOAT32, PF_FLOAT64 = range(1, 9)


def fromROSMsg(data, fields, width, height=1, point_step=None, row_step=None, is_bigendian=False):
    """pcl::fromROSMsg into PointCloud<PointXYZIRT> (src/lidar_odometry_node.cpp:47-48) on the members of a
    sensor_msgs/PointCloud2: `data` the payload bytes, `fields` a list of (name, offset, datatype, count).
    Returns (POINT_XYZIRT records, names of the point type's fields the message did not carry)."""
    buf = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).ravel()
    arr = (capi.Pc2Field * max(len(fields), 1))()
    for i, (name, offset, datatype, count) in enumerate(fields):
        arr[i] = capi.Pc2Field(name.encode(), offset, datatype, count)
    if point_step is None:
        raise ValueError("point_step is required")
    view = capi.Pc2View(height, width, arr, len(fields), 1 if is_bigendian else 0, point_step,
                        row_step if row_step is not None else width * point_step, buf.ctypes.data, buf.size)
    missing = C.c_uint32(0)
    n = capi.lib().lom_pointcloud2_unpack(C.byref(view), None, 0, C.byref(missing))
    if n < 0:
        raise LomError(int(n), capi.lib().lom_pointcloud2_last_error().decode())
    out = np.empty(n, capi.POINT_XYZIRT)
    m = capi.lib().lom_pointcloud2_unpack(C.byref(view), out.ctypes.data, n, None)
    if m < 0:
        raise LomError(int(m), capi.lib().lom_pointcloud2_last_error().decode())
    names = ("x", "y", "z", "intensity", "ring", "time")
    return out, [names[k] for k in range(6) if missing.value >> k & 1]


def toROSMsg(cloud):
    """pcl::toROSMsg (src/lidar_odometry_node.cpp:61,71): an (n, 3) float32 array becomes a PointCloud<PointXYZ>
    message, POINT_XYZIRT records a PointCloud<PointXYZIRT> one.  Returns a dict of the PointCloud2 members."""
    arr = (capi.Pc2Field * 6)()
    step = C.c_uint32(0)
    cloud = np.asarray(cloud)
    if cloud.dtype == capi.POINT_XYZIRT:
        nf = capi.lib().lom_pointcloud2_layout(1, arr, C.byref(step))
        data = np.ascontiguousarray(cloud).tobytes()
        n = len(cloud)
    else:
        xyz = np.ascontiguousarray(cloud, np.float32).reshape(-1, 3)
        nf = capi.lib().lom_pointcloud2_layout(0, arr, C.byref(step))
        n = len(xyz)
        out = np.empty(16 * n, np.uint8)
        w = capi.lib().lom_pointcloud2_pack_xyz(xyz.ctypes.data, n, 12, out.ctypes.data, out.size)
        if w < 0:
            raise LomError(int(w), capi.lib().lom_pointcloud2_last_error().decode())
        data = out.tobytes()
    return {"height": 1, "width": n, "fields": [(arr[i].name.decode(), arr[i].offset, arr[i].datatype, arr[i].count) for i in range(nf)],
            "is_bigendian": False, "point_step": step.value, "row_step": step.value * n, "data": data, "is_dense": True}


class LidarOdometry:
    """reference src/lidar_odometry.{h,cpp}; `params` overrides LidarOdometry::Params defaults by name."""

    def __init__(self, device=0, **params):
        p = capi.OdometryParams()
        capi.lib().lom_odometry_default_params(C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown parameter {k}")
            setattr(p, k, v)
        h = C.c_void_p()
        rc = capi.lib().lom_odometry_create(C.byref(p), int(device), C.byref(h))
        if rc != 0:
            capi.check(rc, None)
        self._h = h
        self.params = p

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and capi is not None:
            capi.lib().lom_odometry_destroy(h)
            self._h = None

    def setOption(self, option, value):
        """lom_odometry_set_option: switches of the pipeline and of its keyframe handle (capi.OPT_*)."""
        rc = capi.lib().lom_odometry_set_option(self._h, int(option), int(value))
        if rc != 0:
            raise LomError(int(rc), "lom_odometry_set_option")

    def debugCounter(self, which=capi.COUNTER_GRID_REDOS):
        return int(capi.lib().lom_odometry_debug_counter(self._h, int(which)))

    def processCloud(self, input_cloud):                   # lidar_odometry.cpp:22-77
        a = _cloud(input_cloud)
        rc = capi.lib().lom_odometry_process_cloud(self._h, a.ctypes.data, len(a))
        if rc != 0:
            text = capi.lib().lom_odometry_last_error(self._h)
            raise LomError(int(rc), text.decode() if text else "")

    def hintNext(self, cloud):
        """The frame that will come after the next processCloud (lom_odometry_hint_next): its upload is sent ahead while
        that call's align runs.  The array is kept alive here until the hint is replaced."""
        self._hinted = _cloud(cloud)
        rc = capi.lib().lom_odometry_hint_next(self._h, self._hinted.ctypes.data, len(self._hinted))
        if rc != 0:
            raise LomError(int(rc), "lom_odometry_hint_next")
        return self._hinted

    def processSequence(self, clouds):
        """processCloud of every frame of `clouds`, in order, issued from compiled code (lom_odometry_process_sequence): the
        reference's caller is the C++ node -- no interpreter between two frames."""
        arrs = [_cloud(c) for c in clouds]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ns = (C.c_size_t * len(arrs))(*[len(a) for a in arrs])
        done = C.c_size_t(0)
        rc = capi.lib().lom_odometry_process_sequence(self._h, ptrs, ns, len(arrs), C.byref(done))
        if rc != 0:
            text = capi.lib().lom_odometry_last_error(self._h)
            raise LomError(int(rc), (text.decode() if text else "") + f" (frame {done.value} of the sequence)")

    def getCurrentPose(self):                              # lidar_odometry.cpp:87-89
        p = capi.Pose()
        capi.check(capi.lib().lom_odometry_get_pose(self._h, C.byref(p)))
        return Pose3D._from(p)

    def getTempCloud(self):                                # lidar_odometry.h:73-75
        """The deskewed input cloud of the last processCloud (None before the first frame)."""
        n = capi.check(capi.lib().lom_odometry_get_temp_cloud(self._h, None, 0))
        if n == 0:
            return None
        out = np.empty(n, capi.POINT_XYZIRT)
        capi.check(capi.lib().lom_odometry_get_temp_cloud(self._h, out.ctypes.data, n))
        return out

    def debugSetState(self, previous, current, keyframe_xyz=None, keyframe_normals=None):
        """Test hook: overwrite the two poses and, if given, rebuild the keyframe from a full export
        (creation order, insertion order inside a voxel: re-inserting it reproduces the map)."""
        if keyframe_xyz is not None:
            kf = capi.lib().lom_odometry_keyframe(self._h)
            xyz, nrm = capi.xyz_array(keyframe_xyz), capi.xyz_array(keyframe_normals)
            capi.check(capi.lib().lom_map_clear(kf, float(self.params.keyframe_voxel_size)), kf)
            capi.check(capi.lib().lom_map_add_points(kf, xyz.ctypes.data, nrm.ctypes.data, len(xyz), 12), kf)
        capi.check(capi.lib().lom_odometry_debug_set_state(self._h, C.byref(previous._c()), C.byref(current._c())))

    def getFullKeyFrameCloudWithNormals(self):
        kf = capi.lib().lom_odometry_keyframe(self._h)
        n = capi.check(capi.lib().lom_map_export(kf, capi.EXPORT_FULL, None, None, 0), kf)
        xyz, nrm = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        if n:
            capi.check(capi.lib().lom_map_export(kf, capi.EXPORT_FULL, xyz.ctypes.data, nrm.ctypes.data, n), kf)
        return xyz, nrm

    def _keyframe_export(self, mode):
        kf = capi.lib().lom_odometry_keyframe(self._h)
        n = capi.check(capi.lib().lom_map_export(kf, mode, None, None, 0), kf)
        xyz = np.empty((n, 3), np.float32)
        if n:
            capi.check(capi.lib().lom_map_export(kf, mode, xyz.ctypes.data, None, n), kf)
        return xyz

    def getKeyFrameCloud(self):                            # lidar_odometry.cpp:79-81
        return self._keyframe_export(capi.EXPORT_FIRST_PER_VOXEL)

    def getFullKeyFrameCloud(self):                        # lidar_odometry.cpp:83-85
        return self._keyframe_export(capi.EXPORT_FULL_NO_NORMALS)

    @property
    def stats(self):
        s = capi.OdometryFrameStats()
        capi.check(capi.lib().lom_odometry_get_stats(self._h, C.byref(s)))
        return s.asdict()
