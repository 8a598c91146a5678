/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11, no third-party code) of the scan-matching hot
 * path of vovo-4K/lidar_odometry_demo.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (lidar_odometry_demo_amd/csrc) never links or calls it.
 *
 * Parity status: the reference itself cannot be compiled in this image
 * (Eigen/PCL/Ceres/robin_map absent), so this restatement is
 * pinned by the reference's own unit-test vectors (test/test.cpp:26-189) and
 * by the MatchingTest protocol (test/test.cpp:191-264) run on the one data
 * file the reference ships.  The Ceres-internal trust-region policy and the
 * robin_map iteration order are restated from their published behaviour and
 * are PARITY UNPINNED below the reference test's own tolerance
 * (0.05 m / 1-|q.q| < 0.01).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef LOM_ORACLE_H
#define LOM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SE(3) value type, f32 (src/pose_3d.h:10-59) ----------------------- */
typedef struct {
    float t[3];
    float q[4]; /* w, x, y, z */
} orc_pose;

void orc_pose_identity(orc_pose *p);
/* src/pose_3d.h:29-32 */
void orc_pose_compose(const orc_pose *a, const orc_pose *b, orc_pose *out);
/* src/pose_3d.h:34-39 */
void orc_pose_inverse(const orc_pose *a, orc_pose *out);
/* src/pose_3d.h:23-27 */
void orc_pose_relative_to(const orc_pose *a, const orc_pose *target, orc_pose *out);
/* src/pose_3d.h:41-43 (Eigen Quaternionf::toRotationMatrix), row-major 3x3 */
void orc_pose_rotation_matrix(const orc_pose *a, float R[9]);
/* src/utils/cloud_transform.h:43-66 / :68-97 ; nrm_* may be NULL */
void orc_transform_points(const orc_pose *pose, const float *xyz_in, const float *nrm_in,
                          size_t n, size_t stride_bytes_in, float *xyz_out, float *nrm_out,
                          size_t stride_bytes_out);

/* ---- voxel map (src/voxel_grid.h:17-257, src/voxel_with_planes.h) ------ */
typedef struct orc_map orc_map;

enum {
    ORC_OK = 0,
    ORC_ERR_ARG = -1,
    ORC_ERR_OOM = -2,
    ORC_ERR_RANGE = -3 /* |coordinate / voxel_size| >= 2^20 or non-finite */
};

enum {
    ORC_EXPORT_FULL = 0,           /* getCloud                     :112-130 */
    ORC_EXPORT_FULL_NO_NORMALS = 1,/* getCloudWithoutNormals       :133-147 */
    ORC_EXPORT_FIRST_PER_VOXEL = 2 /* getSparseCloudWithoutNormals :150-162 */
};

orc_map *orc_map_create(float voxel_size, size_t max_points);       /* :48-52 */
void orc_map_destroy(orc_map *m);
int orc_map_clear(orc_map *m, float voxel_size);                     /* :61-66 */
int orc_map_set_max_points(orc_map *m, size_t max_points);           /* :56-59 */
/* addCloud (:77-93) when nrm != NULL, addCloudWithoutNormals (:95-110) when NULL */
int orc_map_add_points(orc_map *m, const float *xyz, const float *nrm, size_t n,
                       size_t stride_bytes);
int orc_map_radius_cleanup(orc_map *m, const float center[3], float radius); /* :236-246 */
size_t orc_map_size(const orc_map *m);                               /* :248-251 */
size_t orc_map_point_count(const orc_map *m);
/* exporters; iteration order = voxel creation order (robin_map order is
 * implementation-defined -> parity unpinned).  Returns number of points the
 * export holds; writes at most cap. */
size_t orc_map_export(const orc_map *m, int mode, float *xyz_out, float *nrm_out, size_t cap);

/* ---- correspondence search (src/voxel_grid.h:164-234) ------------------ */
typedef struct {
    int64_t index;     /* voxel_creation_index * max_points + in_voxel_index, or -1 */
    float origin[3];   /* winner's stored point  (plane_origin)  */
    float normal[3];   /* winner's stored normal (plane_normal)  */
    float sq_dist;     /* f32 squared distance of the winner     */
    uint32_t n_cand;   /* stored points scanned (all occupied neighbours) */
    uint32_t n_occ;    /* occupied neighbour voxels among the 27 */
} orc_corr;

/* findMatchingPairs restated with deterministic (query-order) output.
 * out has n entries; returns number of valid ones (or <0 on error). */
int64_t orc_find_pairs(const orc_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
                       const float t[3], const float q_wxyz[4], float max_dist,
                       orc_corr *out, int nthreads);

/* getCorrespondence (voxel_grid.h:164-204) for one query in the map frame; returns 1 if valid, 0 if not */
int orc_get_correspondence(const orc_map *m, const float query[3], double max_correspondence_distance_sq, orc_corr *out);

/* ---- reduced normal equations for fixed / fresh correspondences --------- */
/* layout shared with the product's C ABI (include/lidar_odometry_amd.h):
 * [0..20] upper triangle of J^T W J (row-major, a<=b), [21..26] J^T W r,
 * [27] sum 0.5*rho(r^2), [28] n_valid, [29] n_cand, [30] n_occ, [31] n_queries.
 * Tangent order: rotation(3) then translation(3).  Prior NOT included. */
#define ORC_NSUMS 32

/* ---- align (src/cloud_matcher.cpp:105-178) ----------------------------- */
typedef struct {
    int outer_iterations;     /* executed outer iterations (<=35)          */
    int lm_iterations;        /* recorded Ceres-style iterations, total    */
    int evaluations;          /* residual evaluations, total               */
    int64_t queries;          /* source points x outer iterations          */
    int64_t valid_last;       /* valid correspondences in last outer it.   */
    int64_t cand_total;       /* sum of n_cand over all queries            */
    int64_t occ_total;        /* sum of n_occ over all queries             */
    double final_cost;        /* cost at the last accepted point           */
    double last_step_norm;    /* summary.iterations.back().step_norm       */
    double search_seconds;    /* wall time in correspondence search        */
    double solve_seconds;     /* wall time in the LM solve                 */
    int points_evaluated;     /* distinct parameter points evaluated (iteration 0 + every LM
                                 candidate, total): Ceres evaluates an accepted candidate twice (cost,
                                 then Jacobian), the product once -- this is the comparable count */
    int pad;
} orc_align_stats;

int orc_align(const orc_map *m, const float *src_xyz, size_t n, size_t stride_bytes,
              const float guess_t[3], const float guess_q_wxyz[4], float out_t[3],
              float out_q_wxyz[4], orc_align_stats *stats, int nthreads);

/* ---- shard evaluators: used by tests to drive the PRODUCT's host-side
 * align driver (lom_align_with_hooks) on CPU with gloo, world_size 2 ------ */
typedef struct orc_shard orc_shard;
orc_shard *orc_shard_create(const orc_map *m, const float *src_xyz, size_t n, size_t stride_bytes);
void orc_shard_destroy(orc_shard *s);
/* new correspondences at the f32 pose, then sums at (q,t) in f64 */
int orc_shard_match_eval(void *shard, const float pose_t[3], const float pose_q[4],
                         const double q[4], const double t[3], double out[ORC_NSUMS]);
/* sums at (q,t) for the correspondences of the last match_eval */
int orc_shard_eval_fixed(void *shard, const double q[4], const double t[3],
                         double out[ORC_NSUMS]);

/* ---- callers of the hot path, oracle/pipeline.c ---------- */
/* lidar_point::PointXYZIRT, src/lidar_point_type.h:13-21 (32 bytes, EIGEN_ALIGN16) */
typedef struct {
    float x, y, z, pad0;
    float intensity;
    uint16_t ring;
    uint16_t pad1;
    float time;
    float pad2;
} orc_point_xyzirt;

void orc_time_normalize(const orc_point_xyzirt *in, size_t n, orc_point_xyzirt *out);
void orc_transform_non_rigid(const orc_point_xyzirt *in, size_t n, const orc_pose *start, const orc_pose *end,
                             orc_point_xyzirt *out);
size_t orc_range_filter(const float *xyz, const float *nrm, size_t n, float min_range, float max_range,
                        float *xyz_out, float *nrm_out);
size_t orc_classify(const orc_point_xyzirt *in, size_t n, float *xyz_out, float *nrm_out, size_t *unclassified_out,
                    size_t grid_out[2]);

/* LidarOdometry::Params, src/lidar_odometry.h:23-48 */
typedef struct {
    float lidar_min_range, lidar_max_range;
    float keyframe_voxel_size;
    uint32_t keyframe_max_points_cnt;
    float keyframe_matching_voxel_size, keyframe_update_voxel_size;
    float keyframe_cleanup_range, angular_divergence_threshold;
} orc_odom_params;

typedef struct {
    int64_t planar_points, filtered_points, update_points, matching_points, keyframe_voxels, queries;
    int32_t outer_iterations, initialised_keyframe, unstable_rotation, pad;
} orc_odom_frame_stats;

typedef struct orc_odom orc_odom;
void orc_odom_default_params(orc_odom_params *p);
orc_odom *orc_odom_create(const orc_odom_params *p);                 /* lidar_odometry.cpp:14-20 */
void orc_odom_destroy(orc_odom *o);
int orc_odom_process(orc_odom *o, const orc_point_xyzirt *pts, size_t n, int nthreads); /* :22-77 */
void orc_odom_get_pose(const orc_odom *o, orc_pose *out);             /* :87-89 */
void orc_odom_get_stats(const orc_odom *o, orc_odom_frame_stats *out);
const orc_map *orc_odom_keyframe(const orc_odom *o);

#ifdef __cplusplus
}
#endif
#endif
