/*
 * lidar_odometry_amd.h -- C ABI of the MI355X-native scan-matching core.
 *
 * Drop-in boundary for the hot path of vovo-4K/lidar_odometry_demo: the
 * reference's VoxelGrid (src/voxel_grid.h:17-257), VoxelWithPlanes
 * (src/voxel_with_planes.h:10-36), Pose3D (src/pose_3d.h:10-59) and
 * CloudMatcher::align (src/cloud_matcher.h:15-16, src/cloud_matcher.cpp:105-178)
 * are replaced by these entry points; the ROS2 node and LidarOdometry keep
 * their C++ shape and call through include/lidar_odometry_amd.hpp (a header-only
 * mirror of the reference classes over this ABI).  See INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; never throws.
 *   - every function returns 0 (LOM_OK) or a negative lom_status; functions
 *     that return a count return int64 (negative = lom_status).
 *   - points are 3 consecutive f32 (x,y,z) every `stride_bytes` bytes
 *     (12 = packed, 16 = pcl::PointXYZ, 48 = pcl::PointNormal with the normal
 *     at byte offset 16).  Normals use the same stride.
 *   - poses are f32 {t[3], q[4] = w,x,y,z} like the reference's Pose3D.
 *   - a handle owns all device memory, one HIP stream and one pinned result
 *     buffer; it is single-caller (not internally locked).  Independent
 *     handles may be used from different threads; further callers of ONE map
 *     take a scan context each (lom_scan_create).
 *   - there is NO CPU fallback: without a gfx950 device lom_map_create fails
 *     with LOM_ERR_NO_DEVICE.
 */
#ifndef LIDAR_ODOMETRY_AMD_H
#define LIDAR_ODOMETRY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOM_ABI_VERSION 2

typedef enum {
    LOM_OK = 0,
    LOM_ERR_ARG = -1,       /* null/invalid argument                               */
    LOM_ERR_OOM = -2,       /* host or device allocation failed                    */
    LOM_ERR_RANGE = -3,     /* |coordinate / voxel_size| >= 2^20 or non-finite      */
    LOM_ERR_HIP = -4,       /* HIP runtime error, text via lom_last_error()        */
    LOM_ERR_NO_DEVICE = -5, /* no usable gfx950 device                             */
    LOM_ERR_COMM = -6,      /* RCCL error                                          */
    LOM_ERR_STATE = -7,     /* call not valid in this state                        */
    LOM_ERR_HOOK = -8       /* a user hook returned non-zero                       */
} lom_status;

int lom_abi_version(void);
/* number of visible HIP devices (0 when none; never initialises a context) */
int lom_device_count(void);
/* CPU list (sysfs syntax, e.g. "64-127,192-255") of the NUMA node a device is attached to; callers
 * that care about the latency of the host<->device round trips run on those CPUs */
int lom_device_local_cpus(int device, char *out, size_t cap);

/* ---- Pose3D (src/pose_3d.h:10-59), f32 ---------------------------------- */
typedef struct {
    float t[3];
    float q[4]; /* w, x, y, z */
} lom_pose;

void lom_pose_identity(lom_pose *out);                                         /* pose_3d.h:15-18 */
void lom_pose_compose(const lom_pose *a, const lom_pose *b, lom_pose *out);    /* :29-32 */
void lom_pose_inverse(const lom_pose *a, lom_pose *out);                       /* :34-39 */
void lom_pose_relative_to(const lom_pose *a, const lom_pose *target, lom_pose *out); /* :23-27 */
void lom_pose_rotation_matrix(const lom_pose *a, float R_rowmajor[9]);         /* :41-43 */
/* CloudTransformer::transform / transformWithNormals (src/utils/cloud_transform.h:43-97),
 * host-side f32; nrm_in/nrm_out may be NULL. */
int lom_transform_points(const lom_pose *pose, const float *xyz_in, const float *nrm_in, size_t n,
                         size_t stride_bytes_in, float *xyz_out, float *nrm_out,
                         size_t stride_bytes_out);

/* ---- VoxelGrid (src/voxel_grid.h:17-257) -------------------------------- */
typedef struct lom_map lom_map;

/* VoxelGrid(float voxel_size, size_t max_points), voxel_grid.h:48-52.
 * capacity_hint = expected number of voxels (0 = default); device = HIP device index. */
int lom_map_create(float voxel_size, size_t max_points, size_t capacity_hint, int device,
                   lom_map **out);
void lom_map_destroy(lom_map *m);
const char *lom_last_error(const lom_map *m); /* NULL handle: error of the last failed create */

int lom_map_clear(lom_map *m, float voxel_size);            /* setVoxelSize, :61-66 (clears) */
int lom_map_set_max_points(lom_map *m, size_t max_points);  /* setMaxPoints, :56-59; at any time: stored voxels keep what they hold (:86-90) */
/* addCloud (:77-93) when nrm != NULL, addCloudWithoutNormals (:95-110) when nrm == NULL.
 * Deterministic: a voxel keeps the first max_points points in call/input order. */
int lom_map_add_points(lom_map *m, const float *xyz, const float *nrm, size_t n, size_t stride_bytes);
/* same with device-resident input (pointers valid on the handle's device) */
int lom_map_add_points_device(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n,
                              size_t stride_bytes);
/* the same, enqueued only: the range verdict (a call with a point out of range inserts nothing) is not
 * waited for; lom_map_status() waits for everything enqueued on the handle and returns the first deferred
 * error since the last check (LOM_ERR_RANGE, LOM_ERR_HIP) -- for callers that keep a frame in HBM and look at
 * the host only once per frame */
int lom_map_add_points_device_nowait(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n,
                                     size_t stride_bytes);
int lom_map_status(lom_map *m);
int lom_map_radius_cleanup(lom_map *m, const float center[3], float radius); /* :236-246 */
/* src/lidar_odometry.cpp:65-67 calls radiusCleanup with the translation the align has just returned.  A caller that will
 * do the same arms this before lom_match_align_device: the device-resident align then enqueues the cleanup's scan (it
 * only reads the map and writes scratch) behind its last solve, with the centre taken from its own result in HBM, and
 * the lom_map_radius_cleanup that follows takes that scan's result if -- and only if -- its centre and radius are
 * bit for bit what the scan used and the map has not been touched in between; otherwise it runs as if this had never
 * been called.  One shot (the next align only), map handles only (not scan contexts); results never differ. */
int lom_map_radius_cleanup_after_align(lom_map *m, float radius);
int64_t lom_map_size(const lom_map *m);        /* number of voxels, :248-251 */
int64_t lom_map_point_count(const lom_map *m); /* number of stored points */

typedef enum {
    LOM_EXPORT_FULL = 0,            /* getCloud, :112-130                       */
    LOM_EXPORT_FULL_NO_NORMALS = 1, /* getCloudWithoutNormals, :133-147         */
    LOM_EXPORT_FIRST_PER_VOXEL = 2  /* getSparseCloudWithoutNormals, :150-162   */
} lom_export_mode;
/* Packed (12-byte) host output in voxel creation order, insertion order inside a
 * voxel.  Returns the number of points the export holds; writes at most `cap`.
 * xyz_out / nrm_out may be NULL (count only). */
int64_t lom_map_export(lom_map *m, int mode, float *xyz_out, float *nrm_out, size_t cap);

/* The reference's down-sampling idiom in one pass: VoxelGrid(voxel_size, 1).addCloud(cloud)
 * followed by getCloud() / getCloudWithoutNormals() (src/lidar_odometry.cpp:37-38,42,46-47,50):
 * the first point of every voxel in input order, returned in order of first appearance -- exactly
 * what lom_map_create(voxel,1) + lom_map_add_points + lom_map_export(LOM_EXPORT_FULL) return.
 * `workspace` is any map handle; it is cleared and left empty.  nrm / nrm_out may be NULL
 * (nrm == NULL with nrm_out set yields zero normals, like addCloudWithoutNormals).  Returns the
 * number of voxels; writes at most `cap` points. */
int64_t lom_voxel_downsample(lom_map *workspace, float voxel_size, const float *xyz, const float *nrm, size_t n,
                             size_t stride_bytes, float *xyz_out, float *nrm_out, size_t cap);

/* ---- device-resident variants: a caller that keeps a frame in HBM (processCloud does) ---- */
/* Copy host points (and normals) into the handle's device staging buffers; the device pointers
 * stay valid until the next lom_upload_points / host-input call on this handle.  Records keep the
 * caller's stride. */
int lom_upload_points(lom_map *m, const float *xyz, const float *nrm, size_t n, size_t stride_bytes,
                      const float **d_xyz_out, const float **d_nrm_out);
/* lom_voxel_downsample with input and output in device memory: returns the number of voxels and
 * device pointers to packed 12-byte points (and normals, if d_nrm_out != NULL; zero normals when
 * d_nrm == NULL) inside the workspace handle, valid until the next call on that workspace. */
int64_t lom_voxel_downsample_device(lom_map *workspace, float voxel_size, const float *d_xyz, const float *d_nrm,
                                    size_t n, size_t stride_bytes, const float **d_xyz_out,
                                    const float **d_nrm_out);
/* the same, enqueued only: the input size may live on the device (*d_n, with n_bound its upper bound known
 * to the host, <= 262144 then; d_n == NULL: n_bound points), and the number of voxels is left in a device word
 * (*d_count_out) for the kernels that consume the result; lom_map_status() reports a point out of range. */
int lom_voxel_downsample_device_nowait(lom_map *workspace, float voxel_size, const float *d_xyz, const float *d_nrm,
                                       size_t n_bound, const uint32_t *d_n, size_t stride_bytes,
                                       const float **d_xyz_out, const float **d_nrm_out, const uint32_t **d_count_out);
/* for callers that fold a handle's deferred verdict into a read-back of their own: device words that hold the
 * sequence number of the last call that failed (range / a grid time-out) and the sequence number of the
 * handle's last call -- a word equal to *seq means that call failed */
int lom_map_status_words(lom_map *m, const uint32_t **d_range, const uint32_t **d_grid, uint32_t *seq);
/* up to 32 device words of any handle, read behind everything enqueued so far on this handle's stream: one
 * single-wave kernel that stores them into the handle's pinned block, no copy engine.  _begin enqueues it,
 * _end waits for its words (one read in flight per handle; work enqueued between the two does not delay it);
 * lom_map_read_device_words is both. */
int lom_map_read_device_words(lom_map *m, const uint32_t *const *d_ptrs, int n, uint32_t *out);
int lom_map_read_device_words_begin(lom_map *m, const uint32_t *const *d_ptrs, int n);
int lom_map_read_device_words_end(lom_map *m, uint32_t *out);
/* lom_transform_points on the device (same f32 arithmetic): packed 12-byte output in the handle's
 * staging buffers, valid until the next upload / host-input call on this handle. */
int lom_transform_points_device(lom_map *m, const lom_pose *pose, const float *d_xyz, const float *d_nrm, size_t n,
                                size_t stride_bytes, const float **d_xyz_out, const float **d_nrm_out);

/* ---- getCorrespondence / findMatchingPairs (voxel_grid.h:164-234) -------- */
typedef struct {
    int64_t index;   /* voxel_creation_index * max_points + in_voxel_index, or -1 */
    float origin[3]; /* Correspondence::plane_origin  */
    float normal[3]; /* Correspondence::plane_normal  */
    float sq_dist;   /* f32 squared distance to the winner (0 if none) */
    uint32_t n_cand; /* stored points scanned for this query */
    uint32_t n_occ;  /* occupied voxels among the 27 neighbours */
} lom_correspondence;

/* Deterministic findMatchingPairs: one entry per source point, in source order
 * (the reference's push order under its mutex is nondeterministic).  Returns the
 * number of valid correspondences. */
int64_t lom_match_find_pairs(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
                             const float t[3], const float q_wxyz[4], float max_dist,
                             lom_correspondence *out);
/* The same search with the SQUARED threshold handed over as the reference's getCorrespondence takes it
 * (`double max_correspondence_distance_sq`, src/voxel_grid.h:164): a stored point is accepted iff
 * (double)(f32 squared distance) < max_dist_sq (:184-186).  lom_match_find_pairs squares its f32 argument in f32, as
 * findMatchingPairs does (:215); callers that hold a squared threshold use this entry instead of taking a root. */
int64_t lom_match_find_pairs_sq(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
                                const float t[3], const float q_wxyz[4], double max_dist_sq,
                                lom_correspondence *out);
/* Parity entry for the temporal pruning bound of the align's searches (csrc/match.hip): a search at the pose
 * (t_prev, q_prev), then the search at (t, q) with the first one's winners as upper bounds -- what outer iterations
 * >= 2 of an align run.  The result must equal lom_match_find_pairs at (t, q) entry for entry, whatever the two poses
 * are (a winner that has left a query's 27 voxels is found out and that query searched again at the plain bound). */
int64_t lom_debug_find_pairs_after(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
                                   const float t_prev[3], const float q_prev_wxyz[4], const float t[3],
                                   const float q_wxyz[4], float max_dist, lom_correspondence *out);

/* ---- CloudMatcher::align (src/cloud_matcher.cpp:105-178) ---------------- */
/* Reduced normal equations produced per evaluation (f64):
 * [0..20] upper triangle (row-major, a<=b) of sum w J J^T, tangent order
 * rotation(3), translation(3); [21..26] sum w J r; [27] sum 0.5*rho(r^2);
 * [28] valid correspondences; [29] stored points scanned; [30] occupied
 * neighbour voxels; [31] source points searched.  The translation prior is NOT
 * included. */
#define LOM_NSUMS 32

typedef struct {
    int32_t outer_iterations;  /* executed outer iterations (<=35)                  */
    int32_t lm_iterations;     /* recorded LM iterations incl. iteration 0, total   */
    int32_t evaluations;       /* residual evaluations, total                       */
    int32_t match_launches;    /* correspondence-kernel launches (= outer its)      */
    int64_t queries;           /* source points x outer iterations (this rank)      */
    int64_t valid_last;        /* valid correspondences of the last outer iteration */
    int64_t cand_total;        /* stored points scanned, all outer iterations       */
    int64_t occ_total;         /* occupied neighbour voxels, all outer iterations   */
    double final_cost;
    double last_step_norm;
    double match_kernel_ms;    /* HIP-event time of the correspondence launches (profiling on) */
    double algorithmic_bytes;  /* sum over queries of 444 + 12*cand + 12*valid     */
    double host_launch_ms;     /* host time spent inside kernel-launch calls                   */
    double host_wait_ms;       /* host time spent waiting for evaluation results               */
    int64_t profiled_launches; /* correspondence launches that carried the HIP event pair      */
    int32_t host_fallback;     /* 1: the device-resident loop gave up (its workgroups were not all
                                  resident in time) and the align was redone by the host-driven loop */
    int32_t lm_workgroups;     /* workgroups of the device-resident solve kernel (k_lm), one per CU: the CUs a solve keeps busy */
    double lm_kernel_ms;          /* HIP-event time of the k_lm launches of the device-resident loop (profiling on) */
    int64_t lm_profiled_launches; /* k_lm launches that carried the events                                        */
} lom_align_stats;

/* Replaces CloudMatcher::align (cloud_matcher.h:15-16): up to 35 outer iterations of
 * {correspondence search, Levenberg-Marquardt solve with max 4 iterations}, f32 pose write-back and
 * the reference's stop rule.  On one GPU the whole loop runs on the device (a chain of kernels
 * enqueued ahead of time, pose handed from kernel to kernel in HBM); with an attached exchange
 * (below) or LOM_HOST_LM=1 in the environment the loop is driven from the host, one round trip per
 * LM iteration.  Same policy source either way (csrc/lm_core.hpp).  Zero correspondences is not an
 * error: the prior-only problem is solved and the pose stays at the guess. */
int lom_match_align(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
                    const float guess_t[3], const float guess_q_wxyz[4], float out_t[3],
                    float out_q_wxyz[4], lom_align_stats *stats_or_null);
/* same, source cloud already resident on the handle's device */
int lom_match_align_device(lom_map *m, const float *d_src_xyz, size_t n, size_t stride_bytes,
                           const float guess_t[3], const float guess_q_wxyz[4], float out_t[3],
                           float out_q_wxyz[4], lom_align_stats *stats_or_null);

/* `reps` independent aligns of the same device-resident scan from the same guess, back to back, as
 * a compiled caller would issue them; `total` accumulates the counters and times of all of them
 * (valid_last / final_cost / last_step_norm are the last align's).  Used by bench.py so that the
 * timed region holds the hot path and not an interpreter's per-call overhead. */
int lom_match_align_repeat(lom_map *m, const float *d_src_xyz, size_t n, size_t stride_bytes,
                           const float guess_t[3], const float guess_q_wxyz[4], int reps, float out_t[3],
                           float out_q_wxyz[4], lom_align_stats *total_or_null);

/* Diagnostic build of the correspondence kernel with shader-clock stamps after each phase of every
 * workgroup's first query (8 u64 per workgroup: entry, point transformed, slots probed, prefix in
 * LDS, candidates scanned, minimum known, record stored, exit).  Not a timing of the product kernel. */
int lom_debug_match_stamps(lom_map *m, const float *d_src_xyz, size_t n, size_t stride_bytes, const float t[3],
                           const float q_wxyz[4], float max_dist, unsigned long long *stamps_out,
                           size_t cap_blocks, uint32_t *n_blocks_out);

/* Parity entries for PointToPlaneErrorAnalytic::Evaluate (src/cloud_matcher.cpp:38-103) and the reduction
 * ceres::Solve performs inside the reference (the tests compare them with the oracle sum by sum).
 * lom_debug_eval_sums: one correspondence search at the f32 pose (pose_t, pose_q) with the align's 0.3 m
 * gate, then ONE evaluation of the LOM_NSUMS reduced sums at the f64 point (q, t) -- q need not be a unit
 * quaternion, exactly as inside an align -- through the kernels of the host-driven path. */
int lom_debug_eval_sums(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes, const float pose_t[3],
                        const float pose_q_wxyz[4], const double q[4], const double t[3], double out[LOM_NSUMS]);
/* lom_debug_lm_trace: lom_match_align on the device-resident path, also returning what the LM policy
 * inside k_lm saw during outer iteration `outer_index`: trace_out[e * 40 + 0..6] = the point [q, t] of
 * evaluation e, trace_out[e * 40 + 8 .. + 39] = its LOM_NSUMS totals (after the in-kernel reduction and
 * exchange); *n_evals_out = evaluations of that solve (<= 5; 0 when the align ended earlier).
 * trace_out holds 200 doubles. */
int lom_debug_lm_trace(lom_map *m, const float *src_xyz, size_t n, size_t stride_bytes, const float guess_t[3],
                       const float guess_q_wxyz[4], int outer_index, double *trace_out, int *n_evals_out,
                       float out_t[3], float out_q_wxyz[4], lom_align_stats *stats_or_null);

/* Record a HIP event pair around the correspondence launches of lom_match_align*
 * (stats->match_kernel_ms over stats->profiled_launches launches).  `period` = 0: off; 1: every
 * align; N: every N-th align of this handle (an event pair costs the stream ~5 us per launch --
 * sampling keeps a live in-loop measurement from distorting the loop it measures). */
int lom_map_set_profiling(lom_map *m, int period);
/* Roofline probe: `reps` back-to-back launches of the correspondence kernel on a device-resident
 * scan at pose (t,q), bracketed by ONE HIP event pair on the handle's stream, so the per-event
 * packet overhead is amortised.  Returns the average launch duration in microseconds and the
 * algorithmic bytes of one launch (SURVEY.md 8d formula, counted by the kernel), and optionally the
 * bytes the kernel itself requests (candidates of pruned voxels are not read; + 52 B of output per
 * query).  pair_avg_us_out (optional): the average of the same launches bracketed by one event pair EACH --
 * minus avg_us_out that is what an event pair adds to one short kernel, the correction for the sampled
 * in-loop measurement of lom_match_align*. */
int lom_profile_match(lom_map *m, const float *d_src_xyz, size_t n, size_t stride_bytes, const float t[3],
                      const float q_wxyz[4], float max_dist, int reps, double *avg_us_out,
                      double *algorithmic_bytes_out, double *requested_bytes_out, double *pair_avg_us_out);
/* Roofline probe for the insert chain: lom_map_add_points_device bracketed by one HIP event pair on the handle's
 * stream (all kernels of the insert, no host wait in between); microseconds from the first kernel to the last. */
int lom_profile_insert(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n, size_t stride_bytes,
                       double *total_us_out);
/* Run-time switches of a handle.  The environment is read ONCE, by lom_map_create (LOM_HOST_LM, LOM_DEBUG_LM,
 * LOM_DEBUG_TIMING: the production-side switches); afterwards only this call changes them -- nothing on the
 * align path looks at the environment.  Options of the 100 range exist for the tests. */
typedef enum {
    LOM_OPT_HOST_LM = 1,               /* 1: outer loop and LM policy on the host (one round trip per LM iteration) */
    LOM_OPT_DEVICE_PATIENCE_TICKS = 2, /* bound of every in-kernel wait, ticks of 10 ns; default 5,000,000 = 50 ms.
                                          Waits for a peer RANK take ten times that; the host-side agreement after
                                          a device-to-device align outlasts both (see lom_comm_attach_p2p).  A patience
                                          below 40 ticks -- shorter than the head start k_lm's gather sleeps before its
                                          first poll -- counts as timed out whatever the poll would find: that is what
                                          makes "1 tick" a deterministic way for the tests to force the give-up path, and
                                          it means values below 0.4 us are not a usable patience for anything else. */
    LOM_OPT_DEBUG_LM_STAMPS = 3,       /* 1: print k_lm's phase stamps after every align (stderr) */
    LOM_OPT_DEBUG_TIMING = 4,          /* 1: print host launch / wait times per evaluation (stderr) */
    LOM_OPT_NO_TEMPORAL_BOUND = 5,     /* 1: every correspondence search prunes at max_dist only.  Default 0: the searches
                                          of outer iterations >= 2 of an align also prune with the previous iteration's
                                          winner (exact, verified per query: csrc/match.hip "temporal bound"); results
                                          are the same either way, this switch exists for A/B timing (LOM_NO_TEMPORAL=1
                                          in the environment at create) */
    LOM_OPT_COUNT_CANDIDATES = 6,      /* 1: every search also produces the counts of the reference ALGORITHM -- occupied voxels
                                          among the 27 neighbours and their stored points, per query (lom_correspondence.n_cand
                                          / n_occ) and summed (lom_align_stats.cand_total / occ_total / algorithmic_bytes: SURVEY
                                          8d's cand(q)) -- which takes all 27 slot loads per query.  Default 0: a neighbour voxel
                                          that the distance bound prunes is not looked up at all (same winners, same poses: the
                                          result cannot depend on a voxel none of whose points could win) and those fields read
                                          0.  LOM_COUNT_CANDIDATES=1 in the environment at create.  bench.py times the default
                                          and takes the algorithmic bytes from a counted replay of the same align. */
    LOM_OPT_NO_BULK_INSERT = 7,        /* 1: inserts of more than 65,536 points take the four-kernel path of the smaller ones
                                          (three device-scope atomics per point) instead of the partitioned bulk insert
                                          (csrc/voxel_map.hip "bulk insert").  Same map either way, bytewise; the switch exists
                                          for A/B timing (LOM_NO_BULK_INSERT=1 in the environment at create) */
    LOM_OPT_TEST_GIVE_UP_AT_OUTER = 100, /* k: the k_lm of outer iteration k of the NEXT align behaves as if its
                                          workgroups had timed out waiting (one shot; -1 = off) */
    LOM_OPT_TEST_GRID_GIVE_UP = 101,   /* b >= 0: in the NEXT map-maintenance call with an in-kernel scan, workgroups
                                          b, b+1, ... give up waiting for their predecessors (one shot; -1 = off;
                                          + 65536 per such call to let pass first).  lom_frontend_set_option /
                                          lom_odometry_set_option also take b + 0x40000000: workgroup b ALONE gives up, the
                                          ones behind it get their prefix (a hole in the middle of the front end's output) */
    /* lom_odometry_set_option only: */
    LOM_OPT_TEST_FORCE_HOST_REDO = 102,        /* 1: every frame is handed back to the host stages */
    LOM_OPT_TEST_GRID_GIVE_UP_MATCHING_DS = 103, /* LOM_OPT_TEST_GRID_GIVE_UP on the matching down-sampler ... */
    LOM_OPT_TEST_GRID_GIVE_UP_UPDATE_DS = 104,   /* ... the next frame's update down-sampler ... */
    LOM_OPT_TEST_GRID_GIVE_UP_KEYFRAME = 105,    /* ... the keyframe (its next insert or cleanup) */
    /* lom_map_set_option again: */
    LOM_OPT_TEST_BULK_PARTITION_MAX = 106        /* p > 0: a partition of the bulk insert may hold p points (at most the
                                                    1,024 its workgroup has LDS for; 0 = that limit): a bulk insert with a
                                                    larger partition writes nothing and is redone by the four-kernel path
                                                    (counted by LOM_COUNTER_GRID_REDOS) */
} lom_option;
int lom_map_set_option(lom_map *m, int option, int64_t value);
/* diagnostics: LOM_COUNTER_GRID_REDOS = calls of this handle redone with the multi-launch scan after an
 * in-kernel scan gave up (such a call changes nothing; see csrc/grid_scan.hpp) */
enum { LOM_COUNTER_GRID_REDOS = 0,
       LOM_COUNTER_CLEANUPS_BEHIND_ALIGN = 1, /* radius cleanups that took the scan enqueued behind an align */
       LOM_COUNTER_FRAMES_SENT_AHEAD = 2,     /* lom_odometry only: frames found in pinned memory already (staged during the previous frame's align) */
       LOM_COUNTER_EMPTY_SLABS = 3            /* slabs whose voxel a radius cleanup erased in place and that are not closed yet */ };
int64_t lom_map_debug_counter(const lom_map *m, int which);

/* make the handle's stream wait for a hipEvent_t recorded elsewhere */
int lom_map_wait_event(lom_map *m, void *hip_event);
/* The device-resident align keeps the calling thread waiting for ~0.1 ms with nothing to do.  A caller with host work
 * that does not depend on the align's result (staging the next frame) hands it over here: fn(user) is called ONCE, by the
 * next lom_match_align_device on this handle, on the calling thread, after the align's kernels are enqueued and before it
 * waits for their report.  One shot; fn must not use this handle.  (An align that takes the host-driven path does not
 * call it: the caller checks whether its work was done.) */
int lom_map_set_align_idle_hook(lom_map *m, void (*fn)(void *user), void *user);
/* run the handle's work on a caller-owned hipStream_t (NULL = handle's own stream) */
int lom_map_set_stream(lom_map *m, void *hip_stream);
/* the hipStream_t the handle currently works on (to put several handles on one stream) */
void *lom_map_get_stream(lom_map *m);

/* ---- scan contexts: several callers aligning against ONE keyframe ------------------------------------- */
/* VoxelGrid::getCorrespondence / findMatchingPairs are const (src/voxel_grid.h:164,206) and CloudMatcher::align takes
 * `const VoxelGrid&` (src/cloud_matcher.h:15-16): any number of threads may align against one keyframe at a time.
 * A map handle is single-caller; a lom_scan is what each further caller owns -- stream, per-scan buffers, solve state,
 * report block -- while its kernels read the map's table and slabs.  Contexts of one map may be used concurrently
 * from different threads (one caller per context), as long as nobody changes the map meanwhile (lom_map_add_points*,
 * lom_map_radius_cleanup, lom_map_clear: the reference's non-const members); destroy them before the map.
 * One solve keeps 53 of the 256 CUs busy on a VLP16-sized scan: concurrent contexts are how one GPU is filled. */
typedef struct lom_scan lom_scan;
int lom_scan_create(lom_map *map, lom_scan **out);
/* The same, on a slice of the GPU: the context's stream runs on partition `part` of `nparts` equal, disjoint slices of
 * the device's compute units (a CU mask on the stream: the same number of CUs on every XCD; 1 <= nparts <= 8), and its
 * grids are sized for that slice.  One align alone leaves most of the GPU idle between the launches of its dependent
 * chain, and unpartitioned contexts queue behind each other's full-GPU search grids; k callers on k slices run side by
 * side (bench.py concurrent_contexts: 4 callers against 1).  Results do not depend on the slice: bit for bit those of
 * lom_scan_create / of the map handle.  A stream handed over with lom_scan_set_stream overrides the partition. */
int lom_scan_create_on_partition(lom_map *map, int part, int nparts, lom_scan **out);
void lom_scan_destroy(lom_scan *s);
const char *lom_scan_last_error(const lom_scan *s);
int lom_scan_set_option(lom_scan *s, int option, int64_t value);   /* lom_map_set_option's switches, per context */
int lom_scan_set_stream(lom_scan *s, void *hip_stream_or_null);
void *lom_scan_get_stream(lom_scan *s);
/* lom_match_align / _align_device / _align_repeat / lom_match_find_pairs on the context */
int lom_scan_align(lom_scan *s, const float *src_xyz, size_t n, size_t stride_bytes, const float guess_t[3],
                   const float guess_q_wxyz[4], float out_t[3], float out_q_wxyz[4], lom_align_stats *stats_or_null);
int lom_scan_align_device(lom_scan *s, const float *d_src_xyz, size_t n, size_t stride_bytes, const float guess_t[3],
                          const float guess_q_wxyz[4], float out_t[3], float out_q_wxyz[4], lom_align_stats *stats_or_null);
int lom_scan_align_repeat(lom_scan *s, const float *d_src_xyz, size_t n, size_t stride_bytes, const float guess_t[3],
                          const float guess_q_wxyz[4], int reps, float out_t[3], float out_q_wxyz[4],
                          lom_align_stats *total_or_null);
int64_t lom_scan_find_pairs(lom_scan *s, const float *src_xyz, size_t n, size_t stride_bytes, const float t[3],
                            const float q_wxyz[4], float max_dist, lom_correspondence *out);
int64_t lom_scan_find_pairs_sq(lom_scan *s, const float *src_xyz, size_t n, size_t stride_bytes, const float t[3],
                               const float q_wxyz[4], double max_dist_sq, lom_correspondence *out);

/* ---- multi-GPU: source points range-sharded, map replicated ------------- */
/* One all-gather of LOM_NSUMS f64 per residual evaluation over RCCL, summed in
 * rank order on every rank (results independent of the collective's tree). */
#define LOM_COMM_ID_BYTES 128
int lom_comm_unique_id(char id_out[LOM_COMM_ID_BYTES]);   /* rank 0: ncclGetUniqueId */
int lom_comm_init(lom_map *m, int rank, int nranks, const char id[LOM_COMM_ID_BYTES]);
int lom_comm_finalize(lom_map *m);
/* Same contract between the ranks of ONE node without a device-side collective: each rank's host
 * receives its own sums from the resident evaluation server, the hosts exchange the 256 bytes
 * through POSIX shared memory and add them in rank order.  lom_comm_host_id() on rank 0, broadcast
 * the id, lom_host_comm_create() on every rank, lom_comm_attach_host() on the map.  The exchange
 * object is plain host code (usable without a GPU, e.g. as the allreduce hook of
 * lom_align_with_hooks); the caller owns it. */
typedef struct lom_host_comm lom_host_comm;
int lom_comm_host_id(char id_out[LOM_COMM_ID_BYTES]);
int lom_host_comm_create(int rank, int nranks, const char id[LOM_COMM_ID_BYTES], lom_host_comm **out);
/* in place, count <= LOM_NSUMS.  LOM_ERR_COMM: a rank did not arrive within the deadline (lom_host_comm_set_timeout) or had
 * abandoned an exchange; the object stays broken on every rank from then on.  On the exchange a rank gives up on, a peer
 * that had just completed it may still return LOM_OK -- it fails at its next exchange, at once (csrc/comm.cpp exchange()). */
int lom_host_comm_allreduce(lom_host_comm *c, double *buf, int count);
/* Deadline of one exchange (default 60 s).  A rank that reaches it ABANDONS the exchange object: it marks its
 * slots, so that every rank still waiting for it -- or arriving later -- fails with LOM_ERR_COMM as well instead
 * of pairing with slots their owner has walked away from; all later calls on the object fail at once.
 * lom_host_comm_abort does the same on purpose (a rank that cannot continue tells its peers).
 * lom_host_comm_last_error says which exchange was abandoned, and by whom. */
int lom_host_comm_set_timeout(lom_host_comm *c, double seconds);
int lom_host_comm_abort(lom_host_comm *c);
const char *lom_host_comm_last_error(const lom_host_comm *c);
/* every rank contributes `bytes` (<= 256) raw bytes; all_out receives nranks * bytes in rank order */
int lom_host_comm_allgather(lom_host_comm *c, const void *mine, size_t bytes, void *all_out);
void lom_host_comm_destroy(lom_host_comm *c);
int lom_comm_attach_host(lom_map *m, lom_host_comm *c_or_null);
/* Third transport (ranks of one node, <= 8): the device-resident solve of the single-GPU path on
 * every rank, with the ranks' reduced sums exchanged by the GPUs themselves -- every rank's k_lm
 * stores its 32 words into a small buffer in each peer's HBM (IPC-mapped, over xGMI) and adds the
 * ranks' words in rank order: no host round trip per evaluation.  `c` carries rank / nranks and the
 * exchange of the IPC handles.  The call runs a self-test of the device-to-device exchange on all
 * ranks and returns LOM_ERR_COMM on EVERY rank if it fails on any (fall back to
 * lom_comm_attach_host).  Ranks must issue the same sequence of aligns.
 * Failure handling: a rank whose kernel gives up waiting (LOM_OPT_DEVICE_PATIENCE_TICKS) tells its peers through
 * an abort word in their exchange buffers, so they leave their waits at once instead of after their own patience;
 * after EVERY align the ranks agree on its outcome through `c` (deadline: 30 s + 12 x the patience for a peer
 * rank): all ranks keep the device result, or all redo the align through the host exchange (equal poses), or --
 * when a rank failed for good, or the agreement itself timed out -- all return an error. */
int lom_comm_attach_p2p(lom_map *m, lom_host_comm *c);

/* ---- host-side align driver over user evaluators ------------------------ */
/* lom_match_align* = this driver over the HIP kernels.  Exposed so that the
 * driver (outer loop, LM policy, reduction order) can be exercised with other
 * evaluators and collectives, e.g. world_size-2 gloo tests on CPU. */
typedef struct {
    void *user;
    /* new correspondences at the f32 pose, then sums at (q,t) (f64, = widened pose) */
    int (*match_eval)(void *user, const float pose_t[3], const float pose_q[4], const double q[4],
                      const double t[3], double out[LOM_NSUMS]);
    /* sums at (q,t) for the correspondences of the last match_eval */
    int (*eval_fixed)(void *user, const double q[4], const double t[3], double out[LOM_NSUMS]);
    /* optional: in-place sum over ranks of buf[count]; NULL = single rank */
    int (*allreduce)(void *user, double *buf, int count);
} lom_align_hooks;

int lom_align_with_hooks(const lom_align_hooks *hooks, const float guess_t[3],
                         const float guess_q_wxyz[4], float out_t[3], float out_q_wxyz[4],
                         lom_align_stats *stats_or_null);

/* ---- callers of the path, ROS-free; host code over the ABI above -- */
/* lidar_point::PointXYZIRT (src/lidar_point_type.h:13-21): 32 bytes, same field offsets */
typedef struct {
    float x, y, z, pad0;
    float intensity;
    uint16_t ring;
    uint16_t pad1;
    float time;
    float pad2;
} lom_point_xyzirt;

/* utils::pointTimeNormalize, src/utils/point_time_normalize.h:15-39 */
void lom_point_time_normalize(const lom_point_xyzirt *in, size_t n, lom_point_xyzirt *out);
/* CloudTransformer::transformNonRigid (deskew), src/utils/cloud_transform.h:15-40 */
void lom_transform_non_rigid(const lom_point_xyzirt *in, size_t n, const lom_pose *start_pose,
                             const lom_pose *end_pose, lom_point_xyzirt *out);
/* utils::rangeFilter, src/utils/range_filter.h:13-28; packed xyz (+ optional normals); returns kept count */
size_t lom_range_filter(const float *xyz, const float *nrm, size_t n, float min_range, float max_range,
                        float *xyz_out, float *nrm_out);
/* CloudClassifier::classify, src/utils/cloud_classifier.h:19-168: planar points with normals
 * (outputs sized for n points); the unclassified cloud's size and the organised grid {height,
 * width} are optional outputs.  Returns the number of planar points. */
size_t lom_cloud_classify(const lom_point_xyzirt *in, size_t n, float *xyz_out, float *nrm_out,
                          size_t *unclassified_out, size_t grid_out[2]);

/* ---- the same four callers on the device: a frame stays in HBM from its upload to its pose ------------- */
/* Per-frame front end = pointTimeNormalize + transformNonRigid + CloudClassifier::classify + rangeFilter
 * (the references above) as four HIP kernels; results bit-equal to the host functions above.  Sizes the
 * host does not know (planar / filtered point counts, organised-cloud shape) stay on the device as words
 * the consumers read (lom_voxel_downsample_device_nowait takes such a count).  A frame the device cannot
 * decide bit-exactly (an azimuth within 2e-14 rad of a bin boundary, where the device's atan2 could pick
 * another cell than the host's; an organised cloud beyond the device buffers) is reported by
 * lom_frontend_wait() == 1 and is redone by the caller with the host functions. */
typedef struct lom_frontend lom_frontend;
int lom_frontend_create(int device, void *hip_stream_or_null, lom_frontend **out);
void lom_frontend_destroy(lom_frontend *f);
const char *lom_frontend_last_error(const lom_frontend *f);
int lom_frontend_set_option(lom_frontend *f, int option, int64_t value); /* LOM_OPT_TEST_GRID_GIVE_UP: the next frame's scan */
/* enqueue one frame: upload, time normalisation, deskew from start_pose to end_pose, classification, range filter */
int lom_frontend_process(lom_frontend *f, const lom_point_xyzirt *pts, size_t n, const lom_pose *start_pose,
                         const lom_pose *end_pose, float min_range, float max_range);
/* device pointers of the last frame's filtered planar cloud (packed xyz, normals), the host's upper bound of
 * its size, and the device words {planar points, filtered points, grid height, grid width, ...} */
int lom_frontend_results(lom_frontend *f, const float **d_xyz, const float **d_nrm, const uint32_t **d_counts,
                         uint32_t *bound);
/* wait for the frame: 0 = done on the device, 1 = redo this frame on the host, < 0 = lom_status;
 * counts_out = {planar, filtered, height, width} */
int lom_frontend_wait(lom_frontend *f, uint32_t counts_out[4]);
/* host copies: what = 0 the deskewed cloud (lom_point_xyzirt records into out_a), what = 1 the filtered planar
 * cloud (packed xyz into out_a, normals into out_b); returns the number of points available */
int64_t lom_frontend_fetch(lom_frontend *f, int what, void *out_a, void *out_b, size_t cap);
/* pinned staging buffer for a frame of n points: a caller that writes the frame there itself (e.g. straight from
 * its message) passes the same pointer to lom_frontend_process and saves the copy */
int lom_frontend_stage(lom_frontend *f, size_t n, lom_point_xyzirt **out);
/* hipEvent_t recorded behind the last frame's kernels; lom_map_wait_event makes a handle's stream wait for it
 * (the front end runs on a stream of its own, beside the previous frame's keyframe update) */
void *lom_frontend_done_event(lom_frontend *f);
void *lom_frontend_stream(lom_frontend *f);
/* sequence number of the last lom_frontend_process (words [4] / [5] of lom_frontend_results' d_counts hold the
 * number of the last frame that has to be redone on the host / that hit a grid time-out) */
uint32_t lom_frontend_sequence(const lom_frontend *f);
/* test hook: the device's restatement of glibc's sinf (used by the per-point slerp) on n host values */
int lom_debug_sinf(lom_frontend *f, const float *x, size_t n, float *out);

/* ---- file input: pcl::io::loadPCDFile<pcl::PointXYZ> (test/test.cpp:194) ---------------------------- */
/* PCD v0.7 reader, host code without PCL: `DATA ascii` and `DATA binary`, fields located by name (x y z,
 * optionally normal_x normal_y normal_z), any SIZE / TYPE / COUNT layout -- e.g. the reference's shipped
 * test/test_data/intersection00056.pcd (FIELDS rgb _ x y z _, 32-byte records).  NaN points are kept,
 * like loadPCDFile does.  Returns the number of points in the file (negative: lom_status, text via
 * lom_pcd_last_error()); writes at most `cap` packed xyz triples (and normals, zero when the file has
 * none, if nrm_out != NULL). */
typedef struct {
    uint64_t points;
    uint32_t width, height;
    uint32_t point_step; /* bytes per record in the file */
    int32_t has_normals;
    int32_t data_kind;   /* 0 ascii, 1 binary */
} lom_pcd_info;
int64_t lom_pcd_read(const char *path, float *xyz_out, float *nrm_out, size_t cap, lom_pcd_info *info_or_null);
const char *lom_pcd_last_error(void);

/* ---- message payloads: pcl::fromROSMsg / pcl::toROSMsg of the node (src/lidar_odometry_node.cpp:47-48,61,71) ---- */
/* sensor_msgs/msg/PointField datatypes */
enum {
    LOM_PF_INT8 = 1, LOM_PF_UINT8 = 2, LOM_PF_INT16 = 3, LOM_PF_UINT16 = 4,
    LOM_PF_INT32 = 5, LOM_PF_UINT32 = 6, LOM_PF_FLOAT32 = 7, LOM_PF_FLOAT64 = 8
};
typedef struct {
    const char *name;
    uint32_t offset;
    uint8_t datatype;
    uint32_t count;
} lom_pc2_field;
/* the members of a sensor_msgs/msg/PointCloud2 the conversion reads; nothing is copied or kept */
typedef struct {
    uint32_t height, width;
    const lom_pc2_field *fields;
    uint32_t n_fields;
    uint8_t is_bigendian;
    uint32_t point_step, row_step;
    const uint8_t *data;
    size_t data_bytes;
} lom_pc2_view;
/* fromROSMsg into PointCloud<lidar_point::PointXYZIRT>: x y z intensity (FLOAT32), ring (UINT16), time (FLOAT32)
 * located by name, datatype and count as PCL's FieldMatches does; a field without a match stays zero and sets
 * bit k of *missing_mask (k in the order above).  Returns width * height (writes at most cap records; out may be
 * NULL with cap 0), negative = lom_status with lom_pointcloud2_last_error().  `out` may be the pinned buffer of
 * lom_frontend_stage(). */
int64_t lom_pointcloud2_unpack(const lom_pc2_view *msg, lom_point_xyzirt *out, size_t cap, uint32_t *missing_mask_or_null);
/* toROSMsg: field table and point_step of an outgoing cloud (height 1, width n, row_step n * point_step, little
 * endian); LOM_PC2_XYZ for the keyframe clouds, LOM_PC2_XYZIRT for the deskewed cloud, whose records are
 * lom_point_xyzirt as they are.  Returns the number of fields. */
enum { LOM_PC2_XYZ = 0, LOM_PC2_XYZIRT = 1 };
int lom_pointcloud2_layout(int kind, lom_pc2_field fields_out[6], uint32_t *point_step_out);
/* payload of a PointCloud<pcl::PointXYZ> message from xyz triples (stride 0 = packed): 16-byte records
 * {x, y, z, 1.0f}.  Returns the bytes needed / written. */
int64_t lom_pointcloud2_pack_xyz(const float *xyz, size_t n, size_t stride_bytes, uint8_t *data_out, size_t cap_bytes);
const char *lom_pointcloud2_last_error(void);

/* ---- normal estimation helper: pcl::NormalEstimation, setRadiusSearch(r), viewpoint (0,0,0) (test/test.cpp:196-205) */
/* For every point: covariance of all points within `radius` (itself included), eigenvector of the smallest
 * eigenvalue, flipped towards the origin; NaN where fewer than 3 neighbours exist (test.cpp:219-221 drops those
 * points).  The one plane / covariance accumulation in the reference's data flow; outside the align path, which
 * uses the normals it is given.  Host input and output (packed 12-byte normals; optionally the neighbour counts);
 * returns the number of points with a normal. */
int64_t lom_estimate_normals(const float *xyz, size_t n, size_t stride_bytes, float radius, int device, float *nrm_out,
                             uint32_t *neighbours_out_or_null);

/* LidarOdometry::Params, src/lidar_odometry.h:23-48 */
typedef struct {
    float lidar_min_range, lidar_max_range;
    float keyframe_voxel_size;
    uint32_t keyframe_max_points_cnt;
    float keyframe_matching_voxel_size, keyframe_update_voxel_size;
    float keyframe_cleanup_range, angular_divergence_threshold;
} lom_odometry_params;

typedef struct {
    int64_t planar_points, filtered_points, update_points, matching_points, keyframe_voxels, queries;
    int32_t outer_iterations, initialised_keyframe, unstable_rotation;
    int32_t host_stages; /* 1: the stages before the align ran on the host (LOM_HOST_FRONTEND=1, or a frame the device front end handed back) */
    int64_t queries_total; /* source points x outer iterations of all frames since creation */
} lom_odometry_frame_stats;

typedef struct lom_odometry lom_odometry;
void lom_odometry_default_params(lom_odometry_params *p);          /* lidar_odometry.h:36-48 defaults */
int lom_odometry_create(const lom_odometry_params *p, int device, lom_odometry **out); /* lidar_odometry.cpp:14-20 */
void lom_odometry_destroy(lom_odometry *o);
int lom_odometry_process_cloud(lom_odometry *o, const lom_point_xyzirt *pts, size_t n); /* :22-77 */
/* a caller's frame loop in compiled code (the reference's caller is the C++ node, lidar_odometry_node.cpp:45-76): exactly
 * `count` calls of lom_odometry_process_cloud, frames[i] with n[i] points, nothing else; stops at the first frame that
 * fails and returns its status; *done = frames processed */
int lom_odometry_process_sequence(lom_odometry *o, const lom_point_xyzirt *const *frames, const size_t *n, size_t count,
                                  size_t *done);
/* A caller that already holds the frame that comes after the next lom_odometry_process_cloud (a recorded sequence; a
 * driver that buffers) says so here: while that call's align runs -- its thread would only watch the report -- the hinted
 * frame is copied into the front end's pinned staging buffer, and the process_cloud that then comes with exactly this
 * pointer and size skips that copy (8 us of a C5 frame).  The buffer must stay unchanged until that call has returned.
 * Host work only; poses and counts never differ; a hint that is not followed by its frame costs one copy.
 * lom_odometry_process_sequence hints frame i + 1 before frame i by itself. */
int lom_odometry_hint_next(lom_odometry *o, const lom_point_xyzirt *pts, size_t n);
int lom_odometry_get_pose(const lom_odometry *o, lom_pose *out);   /* getCurrentPose, :87-89 */
/* getTempCloud(), lidar_odometry.h:73-75 (the node publishes it as /deskewed_cloud,
 * lidar_odometry_node.cpp:66-75): the time-normalised, deskewed input cloud of the last processCloud
 * (lidar_odometry.cpp:30-31), all fields kept.  Returns the number of points it holds (0 before the
 * first frame, where the reference returns a null pointer); writes at most `cap` records; out may be
 * NULL with cap 0 (count only). */
int64_t lom_odometry_get_temp_cloud(const lom_odometry *o, lom_point_xyzirt *out, size_t cap);
int lom_odometry_get_stats(const lom_odometry *o, lom_odometry_frame_stats *out);
/* switches of the pipeline (LOM_OPT_TEST_FORCE_HOST_REDO, the LOM_OPT_TEST_GRID_GIVE_UP family) and, for every other
 * option, of its keyframe handle (the align's).  The environment is read once, by lom_odometry_create
 * (LOM_HOST_THREADS, LOM_SYNC_KEYFRAME_UPDATE, LOM_HOST_FRONTEND, LOM_DEBUG_TIMING, and the A/B switches
 * LOM_NO_CLEANUP_BEHIND_ALIGN, LOM_NO_SEND_AHEAD; lom_map_create reads LOM_DENSE_CLEANUP). */
int lom_odometry_set_option(lom_odometry *o, int option, int64_t value);
int64_t lom_odometry_debug_counter(const lom_odometry *o, int which); /* LOM_COUNTER_GRID_REDOS: all its handles + frames redone */
/* test hook (teacher-forced parity tests): overwrite previous_transform_ / current_transform_
 * (lidar_odometry.h:84-85); the keyframe itself can be replaced through lom_odometry_keyframe() */
int lom_odometry_debug_set_state(lom_odometry *o, const lom_pose *previous, const lom_pose *current);
lom_map *lom_odometry_keyframe(lom_odometry *o); /* keyframe_ (getKeyFrameCloud / getFullKeyFrameCloud via lom_map_export) */
const char *lom_odometry_last_error(const lom_odometry *o);

#ifdef __cplusplus
}
#endif
#endif /* LIDAR_ODOMETRY_AMD_H */
