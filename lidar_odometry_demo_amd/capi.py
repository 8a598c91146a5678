"""ctypes binding of liblidar_odometry_amd.so (include/lidar_odometry_amd.h).

There is no fallback: if the HIP library is missing this module raises, and
without a gfx950 device every map constructor fails with LOM_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOM_LIB_PATH") or os.path.join(_HERE, "liblidar_odometry_amd.so")  # LOM_LIB_PATH: A/B builds (tools/)
NSUMS = 32
COMM_ID_BYTES = 128

OK, ERR_ARG, ERR_OOM, ERR_RANGE, ERR_HIP, ERR_NO_DEVICE, ERR_COMM, ERR_STATE, ERR_HOOK = (
    0, -1, -2, -3, -4, -5, -6, -7, -8)
_ERR_NAMES = {ERR_ARG: "LOM_ERR_ARG", ERR_OOM: "LOM_ERR_OOM", ERR_RANGE: "LOM_ERR_RANGE",
              ERR_HIP: "LOM_ERR_HIP", ERR_NO_DEVICE: "LOM_ERR_NO_DEVICE", ERR_COMM: "LOM_ERR_COMM",
              ERR_STATE: "LOM_ERR_STATE", ERR_HOOK: "LOM_ERR_HOOK"}

EXPORT_FULL, EXPORT_FULL_NO_NORMALS, EXPORT_FIRST_PER_VOXEL = 0, 1, 2


class LomError(RuntimeError):
    def __init__(self, code, text=""):
        self.code = code
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {text}")


class Pose(C.Structure):
    _fields_ = [("t", C.c_float * 3), ("q", C.c_float * 4)]


class Correspondence(C.Structure):
    _fields_ = [("index", C.c_int64), ("origin", C.c_float * 3), ("normal", C.c_float * 3),
                ("sq_dist", C.c_float), ("n_cand", C.c_uint32), ("n_occ", C.c_uint32)]


CORR_DTYPE = np.dtype(
    [("index", "<i8"), ("origin", "<f4", 3), ("normal", "<f4", 3), ("sq_dist", "<f4"),
     ("n_cand", "<u4"), ("n_occ", "<u4")], align=True)


class AlignStats(C.Structure):
    _fields_ = [
        ("outer_iterations", C.c_int32), ("lm_iterations", C.c_int32), ("evaluations", C.c_int32),
        ("match_launches", C.c_int32), ("queries", C.c_int64), ("valid_last", C.c_int64),
        ("cand_total", C.c_int64), ("occ_total", C.c_int64), ("final_cost", C.c_double),
        ("last_step_norm", C.c_double), ("match_kernel_ms", C.c_double),
        ("algorithmic_bytes", C.c_double), ("host_launch_ms", C.c_double), ("host_wait_ms", C.c_double),
        ("profiled_launches", C.c_int64), ("host_fallback", C.c_int32), ("lm_workgroups", C.c_int32),
        ("lm_kernel_ms", C.c_double), ("lm_profiled_launches", C.c_int64),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# lidar_point::PointXYZIRT (src/lidar_point_type.h:13-21), 32 bytes
POINT_XYZIRT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"),
                         ("ring", "<u2"), ("pad1", "<u2"), ("time", "<f4"), ("pad2", "<f4")])
assert POINT_XYZIRT.itemsize == 32


class OdometryParams(C.Structure):
    _fields_ = [("lidar_min_range", C.c_float), ("lidar_max_range", C.c_float), ("keyframe_voxel_size", C.c_float),
                ("keyframe_max_points_cnt", C.c_uint32), ("keyframe_matching_voxel_size", C.c_float),
                ("keyframe_update_voxel_size", C.c_float), ("keyframe_cleanup_range", C.c_float),
                ("angular_divergence_threshold", C.c_float)]


class OdometryFrameStats(C.Structure):
    _fields_ = [("planar_points", C.c_int64), ("filtered_points", C.c_int64), ("update_points", C.c_int64),
                ("matching_points", C.c_int64), ("keyframe_voxels", C.c_int64), ("queries", C.c_int64),
                ("outer_iterations", C.c_int32), ("initialised_keyframe", C.c_int32),
                ("unstable_rotation", C.c_int32), ("host_stages", C.c_int32), ("queries_total", C.c_int64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class Pc2Field(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("datatype", C.c_uint8), ("count", C.c_uint32)]


class Pc2View(C.Structure):
    _fields_ = [("height", C.c_uint32), ("width", C.c_uint32), ("fields", C.POINTER(Pc2Field)), ("n_fields", C.c_uint32),
                ("is_bigendian", C.c_uint8), ("point_step", C.c_uint32), ("row_step", C.c_uint32),
                ("data", C.c_void_p), ("data_bytes", C.c_size_t)]


class PcdInfo(C.Structure):
    _fields_ = [("points", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("point_step", C.c_uint32),
                ("has_normals", C.c_int32), ("data_kind", C.c_int32)]


MATCH_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
EVAL_FIXED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.POINTER(C.c_double))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class AlignHooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("match_eval", MATCH_EVAL_FN), ("eval_fixed", EVAL_FIXED_FN),
                ("allreduce", ALLREDUCE_FN)]


# every symbol include/lidar_odometry_amd.h declares
EXPORTED = [
    "lom_abi_version", "lom_device_count", "lom_device_local_cpus", "lom_pose_identity", "lom_pose_compose", "lom_pose_inverse",
    "lom_pose_relative_to", "lom_pose_rotation_matrix", "lom_transform_points", "lom_map_create",
    "lom_map_destroy", "lom_last_error", "lom_map_clear", "lom_map_set_max_points", "lom_map_add_points",
    "lom_map_add_points_device", "lom_map_add_points_device_nowait", "lom_map_status", "lom_map_radius_cleanup", "lom_map_radius_cleanup_after_align", "lom_map_size", "lom_map_point_count",
    "lom_map_export", "lom_voxel_downsample", "lom_voxel_downsample_device", "lom_upload_points",
    "lom_transform_points_device", "lom_map_get_stream", "lom_match_find_pairs", "lom_match_find_pairs_sq", "lom_debug_find_pairs_after", "lom_match_align", "lom_match_align_device", "lom_match_align_repeat", "lom_debug_match_stamps", "lom_debug_eval_sums", "lom_debug_lm_trace",
    "lom_map_set_profiling", "lom_profile_match", "lom_profile_insert", "lom_map_set_stream", "lom_comm_unique_id", "lom_comm_init",
    "lom_comm_finalize", "lom_comm_host_id", "lom_host_comm_create", "lom_host_comm_allreduce",
    "lom_host_comm_destroy", "lom_host_comm_allgather", "lom_comm_attach_host", "lom_comm_attach_p2p", "lom_align_with_hooks", "lom_point_time_normalize", "lom_transform_non_rigid",
    "lom_range_filter", "lom_cloud_classify", "lom_odometry_default_params", "lom_odometry_create",
    "lom_odometry_destroy", "lom_odometry_process_cloud", "lom_odometry_process_sequence", "lom_odometry_hint_next", "lom_odometry_get_pose", "lom_odometry_get_stats", "lom_odometry_get_temp_cloud", "lom_odometry_debug_set_state",
    "lom_odometry_keyframe", "lom_odometry_last_error", "lom_pcd_read", "lom_pcd_last_error", "lom_estimate_normals", "lom_frontend_create", "lom_frontend_destroy", "lom_frontend_last_error",
    "lom_frontend_process", "lom_frontend_results", "lom_frontend_wait", "lom_frontend_fetch", "lom_frontend_stream", "lom_frontend_stage", "lom_map_set_align_idle_hook", "lom_frontend_done_event", "lom_map_wait_event", "lom_frontend_sequence", "lom_map_status_words", "lom_debug_sinf",
    "lom_voxel_downsample_device_nowait", "lom_map_read_device_words", "lom_map_read_device_words_begin", "lom_map_read_device_words_end",
    "lom_pointcloud2_unpack", "lom_pointcloud2_layout", "lom_pointcloud2_pack_xyz", "lom_pointcloud2_last_error",
    "lom_map_set_option", "lom_map_debug_counter", "lom_odometry_set_option", "lom_odometry_debug_counter",
    "lom_frontend_set_option", "lom_host_comm_set_timeout", "lom_host_comm_abort", "lom_host_comm_last_error",
    "lom_scan_create", "lom_scan_destroy", "lom_scan_last_error", "lom_scan_set_option", "lom_scan_set_stream",
    "lom_scan_get_stream", "lom_scan_create_on_partition", "lom_scan_align", "lom_scan_align_device", "lom_scan_align_repeat", "lom_scan_find_pairs", "lom_scan_find_pairs_sq",
]

# lom_option / counters of include/lidar_odometry_amd.h
OPT_HOST_LM, OPT_DEVICE_PATIENCE_TICKS, OPT_DEBUG_LM_STAMPS, OPT_DEBUG_TIMING, OPT_NO_TEMPORAL_BOUND, OPT_COUNT_CANDIDATES = 1, 2, 3, 4, 5, 6
OPT_TEST_GIVE_UP_AT_OUTER, OPT_TEST_GRID_GIVE_UP, OPT_TEST_FORCE_HOST_REDO = 100, 101, 102
OPT_NO_BULK_INSERT, OPT_TEST_BULK_PARTITION_MAX = 7, 106
OPT_TEST_GRID_GIVE_UP_MATCHING_DS, OPT_TEST_GRID_GIVE_UP_UPDATE_DS, OPT_TEST_GRID_GIVE_UP_KEYFRAME = 103, 104, 105
COUNTER_GRID_REDOS = 0
COUNTER_CLEANUPS_BEHIND_ALIGN = 1
COUNTER_FRAMES_SENT_AHEAD = 2
COUNTER_EMPTY_SLABS = 3

_lib = None


def _share_hip_runtime_with_torch():
    """A PyTorch-ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so (same
    SONAME as /opt/rocm's).  Two HSA runtimes in one process cannot both own the GPU
    ("No HIP GPUs are available" in whichever comes second), so when such a wheel is
    installed its copy is loaded first and this library binds to it by SONAME.  torch
    itself is not imported."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch's runtime is already resident; ours will resolve to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Loads the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lidar_odometry_demo_amd/csrc`.  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    fp, dp, pp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(Pose), C.c_void_p
    L.lom_abi_version.restype = C.c_int
    L.lom_device_count.restype = C.c_int
    L.lom_device_local_cpus.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.lom_pose_identity.argtypes = [pp]
    L.lom_pose_compose.argtypes = [pp, pp, pp]
    L.lom_pose_inverse.argtypes = [pp, pp]
    L.lom_pose_relative_to.argtypes = [pp, pp, pp]
    L.lom_pose_rotation_matrix.argtypes = [pp, fp]
    L.lom_transform_points.argtypes = [pp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t]
    L.lom_map_create.argtypes = [C.c_float, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.lom_map_destroy.argtypes = [vp]
    L.lom_map_destroy.restype = None
    L.lom_last_error.argtypes = [vp]
    L.lom_last_error.restype = C.c_char_p
    L.lom_map_clear.argtypes = [vp, C.c_float]
    L.lom_map_set_max_points.argtypes = [vp, C.c_size_t]
    L.lom_map_add_points.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t]
    L.lom_map_add_points_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t]
    L.lom_map_add_points_device_nowait.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t]
    L.lom_map_status.argtypes = [vp]
    L.lom_map_radius_cleanup.argtypes = [vp, fp, C.c_float]
    L.lom_map_radius_cleanup_after_align.argtypes = [vp, C.c_float]
    L.lom_map_size.argtypes = [vp]
    L.lom_map_size.restype = C.c_int64
    L.lom_map_point_count.argtypes = [vp]
    L.lom_map_point_count.restype = C.c_int64
    L.lom_map_export.argtypes = [vp, C.c_int, vp, vp, C.c_size_t]
    L.lom_map_export.restype = C.c_int64
    L.lom_voxel_downsample.argtypes = [vp, C.c_float, vp, vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t]
    L.lom_voxel_downsample.restype = C.c_int64
    L.lom_voxel_downsample_device.argtypes = [vp, C.c_float, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.lom_voxel_downsample_device.restype = C.c_int64
    L.lom_upload_points.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.lom_upload_points.restype = C.c_int
    L.lom_transform_points_device.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.lom_transform_points_device.restype = C.c_int
    L.lom_map_get_stream.argtypes = [vp]
    L.lom_map_get_stream.restype = vp
    L.lom_match_find_pairs.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_float, vp]
    L.lom_match_find_pairs.restype = C.c_int64
    L.lom_match_find_pairs_sq.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_double, vp]
    L.lom_match_find_pairs_sq.restype = C.c_int64
    L.lom_debug_find_pairs_after.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, fp, fp, C.c_float, vp]
    L.lom_debug_find_pairs_after.restype = C.c_int64
    L.lom_match_align.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, fp, fp, C.POINTER(AlignStats)]
    L.lom_match_align_device.argtypes = L.lom_match_align.argtypes
    L.lom_debug_match_stamps.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_float, vp, C.c_size_t,
                                         C.POINTER(C.c_uint32)]
    L.lom_match_align_repeat.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_int, fp, fp, C.POINTER(AlignStats)]
    L.lom_debug_eval_sums.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, dp, dp, dp]
    L.lom_debug_lm_trace.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_int, dp, C.POINTER(C.c_int), fp, fp,
                                     C.POINTER(AlignStats)]
    L.lom_map_set_profiling.argtypes = [vp, C.c_int]
    L.lom_map_set_stream.argtypes = [vp, vp]
    L.lom_profile_match.argtypes = [vp, vp, C.c_size_t, C.c_size_t, fp, fp, C.c_float, C.c_int, dp, dp, dp, dp]
    L.lom_profile_insert.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, dp]
    L.lom_comm_unique_id.argtypes = [C.c_char_p]
    L.lom_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.lom_comm_finalize.argtypes = [vp]
    L.lom_comm_host_id.argtypes = [C.c_char_p]
    L.lom_host_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
    L.lom_host_comm_allreduce.argtypes = [vp, dp, C.c_int]
    L.lom_host_comm_destroy.argtypes = [vp]
    L.lom_host_comm_destroy.restype = None
    L.lom_comm_attach_host.argtypes = [vp, vp]
    L.lom_comm_attach_p2p.argtypes = [vp, vp]
    L.lom_comm_attach_p2p.restype = C.c_int
    L.lom_host_comm_allgather.argtypes = [vp, vp, C.c_size_t, vp]
    L.lom_host_comm_allgather.restype = C.c_int
    L.lom_align_with_hooks.argtypes = [C.POINTER(AlignHooks), fp, fp, fp, fp, C.POINTER(AlignStats)]
    L.lom_point_time_normalize.argtypes = [vp, C.c_size_t, vp]
    L.lom_point_time_normalize.restype = None
    L.lom_transform_non_rigid.argtypes = [vp, C.c_size_t, pp, pp, vp]
    L.lom_transform_non_rigid.restype = None
    L.lom_range_filter.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp]
    L.lom_range_filter.restype = C.c_size_t
    L.lom_cloud_classify.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.lom_cloud_classify.restype = C.c_size_t
    L.lom_odometry_default_params.argtypes = [C.POINTER(OdometryParams)]
    L.lom_odometry_default_params.restype = None
    L.lom_odometry_create.argtypes = [C.POINTER(OdometryParams), C.c_int, C.POINTER(vp)]
    L.lom_odometry_destroy.argtypes = [vp]
    L.lom_odometry_destroy.restype = None
    L.lom_odometry_process_cloud.argtypes = [vp, vp, C.c_size_t]
    L.lom_odometry_hint_next.argtypes = [vp, vp, C.c_size_t]
    L.lom_odometry_process_sequence.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_size_t)]
    L.lom_odometry_get_pose.argtypes = [vp, pp]
    L.lom_odometry_get_stats.argtypes = [vp, C.POINTER(OdometryFrameStats)]
    L.lom_odometry_get_temp_cloud.argtypes = [vp, vp, C.c_size_t]
    L.lom_odometry_get_temp_cloud.restype = C.c_int64
    L.lom_odometry_debug_set_state.argtypes = [vp, pp, pp]
    L.lom_odometry_keyframe.argtypes = [vp]
    L.lom_odometry_keyframe.restype = vp
    L.lom_odometry_last_error.argtypes = [vp]
    L.lom_odometry_last_error.restype = C.c_char_p
    L.lom_frontend_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.lom_frontend_destroy.argtypes = [vp]
    L.lom_frontend_destroy.restype = None
    L.lom_frontend_last_error.argtypes = [vp]
    L.lom_frontend_last_error.restype = C.c_char_p
    L.lom_frontend_process.argtypes = [vp, vp, C.c_size_t, pp, pp, C.c_float, C.c_float]
    L.lom_frontend_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.lom_frontend_wait.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.lom_frontend_fetch.argtypes = [vp, C.c_int, vp, vp, C.c_size_t]
    L.lom_frontend_fetch.restype = C.c_int64
    L.lom_frontend_stream.argtypes = [vp]
    L.lom_frontend_stream.restype = vp
    L.lom_debug_sinf.argtypes = [vp, vp, C.c_size_t, vp]
    L.lom_voxel_downsample_device_nowait.argtypes = [vp, C.c_float, vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp),
                                                     C.POINTER(vp), C.POINTER(vp)]
    L.lom_map_read_device_words.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(C.c_uint32)]
    L.lom_map_read_device_words_begin.argtypes = [vp, C.POINTER(vp), C.c_int]
    L.lom_map_read_device_words_end.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.lom_estimate_normals.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, vp, vp]
    L.lom_estimate_normals.restype = C.c_int64
    L.lom_pcd_read.argtypes = [C.c_char_p, vp, vp, C.c_size_t, C.POINTER(PcdInfo)]
    L.lom_pcd_read.restype = C.c_int64
    L.lom_pcd_last_error.restype = C.c_char_p
    L.lom_pointcloud2_unpack.argtypes = [C.POINTER(Pc2View), vp, C.c_size_t, C.POINTER(C.c_uint32)]
    L.lom_pointcloud2_unpack.restype = C.c_int64
    L.lom_pointcloud2_layout.argtypes = [C.c_int, C.POINTER(Pc2Field), C.POINTER(C.c_uint32)]
    L.lom_pointcloud2_pack_xyz.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    L.lom_pointcloud2_pack_xyz.restype = C.c_int64
    L.lom_pointcloud2_last_error.restype = C.c_char_p
    L.lom_map_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.lom_map_debug_counter.argtypes = [vp, C.c_int]
    L.lom_map_debug_counter.restype = C.c_int64
    L.lom_odometry_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.lom_odometry_debug_counter.argtypes = [vp, C.c_int]
    L.lom_odometry_debug_counter.restype = C.c_int64
    L.lom_frontend_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.lom_host_comm_set_timeout.argtypes = [vp, C.c_double]
    L.lom_host_comm_abort.argtypes = [vp]
    L.lom_host_comm_last_error.argtypes = [vp]
    L.lom_host_comm_last_error.restype = C.c_char_p
    L.lom_scan_create.argtypes = [vp, C.POINTER(vp)]
    L.lom_scan_create_on_partition.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.lom_scan_create_on_partition.restype = C.c_int
    L.lom_scan_destroy.argtypes = [vp]
    L.lom_scan_destroy.restype = None
    L.lom_scan_last_error.argtypes = [vp]
    L.lom_scan_last_error.restype = C.c_char_p
    L.lom_scan_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.lom_scan_set_stream.argtypes = [vp, vp]
    L.lom_scan_get_stream.argtypes = [vp]
    L.lom_scan_get_stream.restype = vp
    L.lom_scan_align.argtypes = L.lom_match_align.argtypes
    L.lom_scan_align_device.argtypes = L.lom_match_align.argtypes
    L.lom_scan_align_repeat.argtypes = L.lom_match_align_repeat.argtypes
    L.lom_scan_find_pairs.argtypes = L.lom_match_find_pairs.argtypes
    L.lom_scan_find_pairs.restype = C.c_int64
    L.lom_scan_find_pairs_sq.argtypes = L.lom_match_find_pairs_sq.argtypes
    L.lom_scan_find_pairs_sq.restype = C.c_int64
    _lib = L
    return L


def pin_to_device_numa_node(device=0):
    """Restrict this process to the CPUs of the NUMA node the GPU is attached to (no-op when sysfs
    does not tell).  Returns the cpu set used, or None."""
    buf = C.create_string_buffer(512)
    if lib().lom_device_local_cpus(int(device), buf, 512) != 0 or not buf.value:
        return None
    cpus = set()
    for part in buf.value.decode().split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return None
    os.sched_setaffinity(0, cpus)
    return cpus


def f3(a):
    return (C.c_float * 3)(*[float(v) for v in a])


def f4(a):
    return (C.c_float * 4)(*[float(v) for v in a])


def xyz_array(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected an (n, 3) float32 array")
    return a


def check(rc, handle=None):
    if rc < 0:
        text = lib().lom_last_error(handle)
        raise LomError(int(rc), text.decode() if text else "")
    return rc
