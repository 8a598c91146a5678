/*
 * oracle/pipeline.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * CPU restatement of the callers of the hot path:
 *   utils::pointTimeNormalize          src/utils/point_time_normalize.h:15-39
 *   CloudTransformer::transformNonRigid src/utils/cloud_transform.h:15-40
 *   CloudClassifier::classify           src/utils/cloud_classifier.h:19-168
 *   utils::rangeFilter                  src/utils/range_filter.h:13-28
 *   LidarOdometry::processCloud         src/lidar_odometry.cpp:14-77
 * Citations relative to /root/reference.  Built with -ffp-contract=off.
 * PARITY UNPINNED: the reference holds no test or vector for any of these
 * functions; they are restated from the source text.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define PI_D 3.14159265358979323846

static inline float sum3f(float a0, float a1, float a2) { return a0 + (a1 + a2); }

/* ---- pointTimeNormalize ------------------------------------------------- */
void orc_time_normalize(const orc_point_xyzirt *in, size_t n, orc_point_xyzirt *out)
{
    float min_time = 3.402823466e+38f, max_time = -3.402823466e+38f; /* :18-19 */
    for (size_t i = 0; i < n; i++) {                                 /* :21-25 */
        min_time = in[i].time < min_time ? in[i].time : min_time;
        max_time = in[i].time > max_time ? in[i].time : max_time;
    }
    const float time_range = max_time - min_time; /* :27 */
    for (size_t i = 0; i < n; i++) {               /* :32-37 */
        out[i] = in[i];
        out[i].time = (in[i].time - min_time) / time_range;
    }
}

/* ---- Eigen Quaternionf::slerp ------------------------------------------- */
static void quat_slerp_f(const float a[4], float t, const float b[4], float out[4])
{
    const float one = 1.0f - 1.1920928955078125e-07f; /* Scalar(1) - NumTraits<float>::epsilon() */
    const float d = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    const float absD = fabsf(d);
    float scale0, scale1;
    if (absD >= one) {
        scale0 = 1.0f - t;
        scale1 = t;
    } else {
        const float theta = acosf(absD);
        const float sinTheta = sinf(theta);
        scale0 = sinf((1.0f - t) * theta) / sinTheta;
        scale1 = sinf(t * theta) / sinTheta;
    }
    if (d < 0.0f) scale1 = -scale1;
    for (int i = 0; i < 4; i++) out[i] = scale0 * a[i] + scale1 * b[i];
}

static void quat_rotate_f(const float q[4], const float v[3], float out[3])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    float uv0 = y * v[2] - z * v[1];
    float uv1 = z * v[0] - x * v[2];
    float uv2 = x * v[1] - y * v[0];
    uv0 += uv0;
    uv1 += uv1;
    uv2 += uv2;
    out[0] = (v[0] + w * uv0) + (y * uv2 - z * uv1);
    out[1] = (v[1] + w * uv1) + (z * uv0 - x * uv2);
    out[2] = (v[2] + w * uv2) + (x * uv1 - y * uv0);
}

/* ---- transformNonRigid (deskew) ------------------------------------------ */
void orc_transform_non_rigid(const orc_point_xyzirt *in, size_t n, const orc_pose *start, const orc_pose *end,
                             orc_point_xyzirt *out)
{
    for (size_t i = 0; i < n; i++) {
        const float t = in[i].time;
        float q[4], r[3];
        quat_slerp_f(start->q, t, end->q, q); /* cloud_transform.h:27 start.rotation.slerp(time, end.rotation) */
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        quat_rotate_f(q, p, r);
        const float w1 = (float)(1.0 - (double)t); /* :30 (1.0 - time) is a double expression */
        out[i] = in[i];
        /* :26-30  (q*p) + start.t*time + end.t*(1-time): note the translation weights */
        out[i].x = (r[0] + start->t[0] * t) + end->t[0] * w1;
        out[i].y = (r[1] + start->t[1] * t) + end->t[1] * w1;
        out[i].z = (r[2] + start->t[2] * t) + end->t[2] * w1;
    }
}

/* ---- rangeFilter ----------------------------------------------------------- */
size_t orc_range_filter(const float *xyz, const float *nrm, size_t n, float min_range, float max_range,
                        float *xyz_out, float *nrm_out)
{
    const float lo = min_range * min_range, hi = max_range * max_range; /* range_filter.h:14-15 */
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        const float *p = xyz + 3 * i;
        const float r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2]; /* :21 left to right */
        if (r2 >= lo && r2 <= hi) {                                /* :22 */
            memcpy(xyz_out + 3 * w, p, 12);
            if (nrm && nrm_out) memcpy(nrm_out + 3 * w, nrm + 3 * i, 12);
            w++;
        }
    }
    return w;
}

/* ---- CloudClassifier::classify ---------------------------------------------- */
/* Returns the number of planar points; xyz_out / nrm_out need room for n points.
 * grid_out (optional) receives the organised cloud's height and width. */
size_t orc_classify(const orc_point_xyzirt *in, size_t n, float *xyz_out, float *nrm_out, size_t *unclassified_out,
                    size_t grid_out[2])
{
    /* organize_cloud, cloud_classifier.h:21-68: rings keyed by uint8_t, ascending */
    size_t ring_count[256];
    memset(ring_count, 0, sizeof ring_count);
    for (size_t i = 0; i < n; i++) ring_count[(uint8_t)in[i].ring]++; /* :25-32 */
    size_t W = 0, H = 0;
    int row_of_ring[256];
    for (int r = 0; r < 256; r++) {
        row_of_ring[r] = -1;
        if (ring_count[r]) {
            row_of_ring[r] = (int)H++;
            if (ring_count[r] > W) W = ring_count[r]; /* :35-40 */
        }
    }
    if (grid_out) {
        grid_out[0] = H;
        grid_out[1] = W;
    }
    if (unclassified_out) *unclassified_out = 0;
    const size_t total = H * W;
    if (total == 0) return 0;
    orc_point_xyzirt *cloud = (orc_point_xyzirt *)calloc(total, sizeof(orc_point_xyzirt)); /* PointType() = zeros */
    if (!cloud) return 0;
    for (size_t i = 0; i < n; i++) { /* :48-55, input order inside a ring, last writer wins */
        const orc_point_xyzirt *p = &in[i];
        const float azimuth = (float)(atan2((double)-p->y, (double)p->x) + PI_D);       /* :49 */
        const double idx_d = fabs((double)(azimuth * (float)W) / (2.0 * PI_D));         /* :50 */
        const size_t idx = (size_t)idx_d;
        if (idx < W) cloud[(size_t)row_of_ring[(uint8_t)p->ring] * W + idx] = *p;       /* :52-54 */
    }
    /* curvature over the flattened array, :76-103 */
    const int cw = 4;
    const float intensity_max = 1000.0f;
    if (total > (size_t)(2 * cw)) {
        for (size_t i = (size_t)cw; i < total - (size_t)cw; i++) {
            orc_point_xyzirt *o = &cloud[i];
            const float range = powf(o->x, 2) + powf(o->y, 2) + powf(o->z, 2); /* :83 */
            if ((double)range < 0.1) {                                         /* :84 */
                o->intensity = intensity_max;
                continue;
            }
            float dx = (float)((double)(-o->x) * (cw * 2.0 + 1.0)); /* :89-91 */
            float dy = (float)((double)(-o->y) * (cw * 2.0 + 1.0));
            float dz = (float)((double)(-o->z) * (cw * 2.0 + 1.0));
            for (int w = -cw; w <= cw; w++) { /* :93-97 */
                dx += cloud[i + w].x;
                dy += cloud[i + w].y;
                dz += cloud[i + w].z;
            }
            const float curvature = (float)(sqrt((double)(dx * dx + dy * dy + dz * dz)) / (double)range); /* :99 */
            o->intensity = curvature;
        }
    }
    /* normals, :105-165 */
    const int nw = 4;
    const float flat = 0.05f;
    const double flat10 = (double)flat * 10.0; /* :123 flatness_threshold*10.0 */
    size_t np = 0, nu = 0;
    for (size_t ray = 1; ray < H; ray++) {
        for (long pi = nw; pi < (long)W - nw; pi++) {
            const orc_point_xyzirt *pt = &cloud[ray * W + (size_t)pi];
            if (pt->intensity < flat) { /* :114 */
                const size_t prev = ray - 1;
                int found = 0;
                float L[3] = {0, 0, 0}, R[3] = {0, 0, 0};
                for (long q = pi - nw; q < pi; q++) { /* :120-128 */
                    const orc_point_xyzirt *nb = &cloud[prev * W + (size_t)q];
                    if ((double)nb->intensity < flat10) {
                        L[0] = nb->x, L[1] = nb->y, L[2] = nb->z;
                        found++;
                        break;
                    }
                }
                for (long q = pi + nw; q > pi; q--) { /* :130-138 */
                    const orc_point_xyzirt *nb = &cloud[prev * W + (size_t)q];
                    if ((double)nb->intensity < flat10) {
                        R[0] = nb->x, R[1] = nb->y, R[2] = nb->z;
                        found++;
                        break;
                    }
                }
                if (found == 2) { /* :140-155 */
                    const float a[3] = {L[0] - pt->x, L[1] - pt->y, L[2] - pt->z};
                    const float b[3] = {R[0] - pt->x, R[1] - pt->y, R[2] - pt->z};
                    float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
                    const float z = sum3f(c[0] * c[0], c[1] * c[1], c[2] * c[2]); /* Eigen normalized() */
                    if (z > 0.f) {
                        const float s = sqrtf(z);
                        c[0] /= s, c[1] /= s, c[2] /= s;
                    }
                    xyz_out[3 * np] = pt->x, xyz_out[3 * np + 1] = pt->y, xyz_out[3 * np + 2] = pt->z;
                    nrm_out[3 * np] = c[0], nrm_out[3 * np + 1] = c[1], nrm_out[3 * np + 2] = c[2];
                    np++;
                } else {
                    nu++;
                }
            } else if (pt->intensity < intensity_max) { /* :158-160 */
                nu++;
            }
        }
    }
    if (unclassified_out) *unclassified_out = nu;
    free(cloud);
    return np;
}

/* ---- Eigen Matrix3f::eulerAngles(0,1,2) of q_a * q_b^-1, in degrees ------------ */
static void delta_euler_deg(const float qa[4], const float qb[4], float out[3])
{
    orc_pose pb, inv, pa, prod;
    memset(&pb, 0, sizeof pb);
    memset(&pa, 0, sizeof pa);
    memcpy(pb.q, qb, 16);
    memcpy(pa.q, qa, 16);
    orc_pose_inverse(&pb, &inv);       /* lidar_odometry.cpp:55 current.rotation.inverse() */
    orc_pose_compose(&pa, &inv, &prod); /* rotation product only (translations are zero) */
    float m[9];
    orc_pose_rotation_matrix(&prod, m);
#define M(r, c) m[(r)*3 + (c)]
    /* Eigen EulerAngles.h, a0=0,a1=1,a2=2: odd=0, i=0, j=1, k=2 */
    float res[3];
    res[0] = atan2f(M(1, 2), M(2, 2));
    const float c2 = sqrtf(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
    if (res[0] > 0.f) {
        res[0] -= (float)PI_D;
        res[1] = atan2f(-M(0, 2), -c2);
    } else {
        res[1] = atan2f(-M(0, 2), c2);
    }
    const float s1 = sinf(res[0]), c1 = cosf(res[0]);
    res[2] = atan2f(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
#undef M
    for (int i = 0; i < 3; i++) out[i] = ((-res[i]) * 180.0f) / (float)PI_D; /* :55: Vector3f * 180.0 / pi, scalars taken as f32 */
}

/* ---- LidarOdometry ----------------------------------------------------------------- */
struct orc_odom {
    orc_odom_params cfg;
    orc_map *keyframe;
    orc_pose previous, current;
    orc_odom_frame_stats last;
};

void orc_odom_default_params(orc_odom_params *p)
{
    /* lidar_odometry.h:36-48 / config/params.yaml */
    p->lidar_min_range = 4.0f;
    p->lidar_max_range = 80.0f;
    p->keyframe_voxel_size = 0.2f;
    p->keyframe_max_points_cnt = 20;
    p->keyframe_matching_voxel_size = 0.3f;
    p->keyframe_update_voxel_size = 0.1f;
    p->keyframe_cleanup_range = 80.0f;
    p->angular_divergence_threshold = 5.0f;
}

orc_odom *orc_odom_create(const orc_odom_params *p)
{
    orc_odom *o = (orc_odom *)calloc(1, sizeof(orc_odom));
    if (!o) return NULL;
    o->cfg = *p;
    orc_pose_identity(&o->current); /* lidar_odometry.cpp:15-17 */
    o->previous = o->current;
    o->keyframe = orc_map_create(p->keyframe_voxel_size, p->keyframe_max_points_cnt); /* :18-19 */
    if (!o->keyframe) {
        free(o);
        return NULL;
    }
    return o;
}

void orc_odom_destroy(orc_odom *o)
{
    if (!o) return;
    orc_map_destroy(o->keyframe);
    free(o);
}

const orc_map *orc_odom_keyframe(const orc_odom *o) { return o->keyframe; }
void orc_odom_get_pose(const orc_odom *o, orc_pose *out) { *out = o->current; }
void orc_odom_get_stats(const orc_odom *o, orc_odom_frame_stats *out) { *out = o->last; }

int orc_odom_process(orc_odom *o, const orc_point_xyzirt *pts, size_t n, int nthreads)
{
    if (!o || (!pts && n)) return ORC_ERR_ARG;
    int rc = ORC_OK;
    memset(&o->last, 0, sizeof o->last);
    orc_point_xyzirt *tn = (orc_point_xyzirt *)malloc((n ? n : 1) * sizeof *tn);
    orc_point_xyzirt *dk = (orc_point_xyzirt *)malloc((n ? n : 1) * sizeof *dk);
    float *pl = (float *)malloc((n ? n : 1) * 12), *pn = (float *)malloc((n ? n : 1) * 12);
    float *fl = (float *)malloc((n ? n : 1) * 12), *fn = (float *)malloc((n ? n : 1) * 12);
    orc_map *down = NULL, *matchds = NULL;
    float *dxyz = NULL, *dnrm = NULL, *mxyz = NULL, *uxyz = NULL, *unrm = NULL;
    if (!tn || !dk || !pl || !pn || !fl || !fn) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    orc_time_normalize(pts, n, tn); /* lidar_odometry.cpp:25 */
    orc_pose relative, rel_inv, ident;
    orc_pose_relative_to(&o->previous, &o->current, &relative); /* :27 */
    o->previous = o->current;                                   /* :28 */
    orc_pose_inverse(&relative, &rel_inv);
    orc_pose_identity(&ident);
    orc_transform_non_rigid(tn, n, &rel_inv, &ident, dk); /* :30 */
    size_t nu = 0;
    const size_t np = orc_classify(dk, n, pl, pn, &nu, NULL);                                           /* :33 */
    const size_t nf = orc_range_filter(pl, pn, np, o->cfg.lidar_min_range, o->cfg.lidar_max_range, fl, fn); /* :35 */
    o->last.planar_points = (int64_t)np;
    o->last.filtered_points = (int64_t)nf;
    down = orc_map_create(o->cfg.keyframe_update_voxel_size, 1); /* :37 */
    if (!down) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    if ((rc = orc_map_add_points(down, fl, fn, nf, 12)) != ORC_OK) goto out; /* :38 */
    const size_t nd = orc_map_export(down, ORC_EXPORT_FULL, NULL, NULL, 0);
    dxyz = (float *)malloc((nd ? nd : 1) * 12);
    dnrm = (float *)malloc((nd ? nd : 1) * 12);
    if (!dxyz || !dnrm) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    orc_map_export(down, ORC_EXPORT_FULL, dxyz, dnrm, nd);
    o->last.update_points = (int64_t)nd;
    if (orc_map_size(o->keyframe) == 0) { /* :40-44 init keyframe */
        rc = orc_map_add_points(o->keyframe, dxyz, dnrm, nd, 12);
        o->last.initialised_keyframe = 1;
        o->last.keyframe_voxels = (int64_t)orc_map_size(o->keyframe);
        goto out;
    }
    matchds = orc_map_create(o->cfg.keyframe_matching_voxel_size, 1); /* :46 */
    if (!matchds) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    if ((rc = orc_map_add_points(matchds, fl, fn, nf, 12)) != ORC_OK) goto out; /* :47 */
    const size_t nm = orc_map_export(matchds, ORC_EXPORT_FULL_NO_NORMALS, NULL, NULL, 0);
    mxyz = (float *)malloc((nm ? nm : 1) * 12);
    if (!mxyz) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    orc_map_export(matchds, ORC_EXPORT_FULL_NO_NORMALS, mxyz, NULL, nm);
    o->last.matching_points = (int64_t)nm;
    orc_pose guess, result;
    orc_pose_compose(&o->current, &relative, &guess); /* :51 */
    orc_align_stats ast;
    rc = orc_align(o->keyframe, mxyz, nm, 12, guess.t, guess.q, result.t, result.q, &ast, nthreads); /* :49-51 */
    if (rc != ORC_OK) goto out;
    o->last.outer_iterations = ast.outer_iterations;
    o->last.queries = ast.queries;
    {
        float ang[3]; /* :53-63 divergence guard */
        delta_euler_deg(result.q, o->current.q, ang);
        const float thr = o->cfg.angular_divergence_threshold;
        int ok = 1;
        for (int a = 0; a < 3; a++) ok = ok && (fabsf(ang[a]) < thr || fabsf(ang[a]) > 180 - thr);
        if (!ok) {
            result = guess; /* :61 */
            o->last.unstable_rotation = 1;
        }
    }
    o->current = result;                                                          /* :65 */
    orc_map_radius_cleanup(o->keyframe, o->current.t, o->cfg.keyframe_cleanup_range); /* :67 */
    uxyz = (float *)malloc((nd ? nd : 1) * 12);
    unrm = (float *)malloc((nd ? nd : 1) * 12);
    if (!uxyz || !unrm) {
        rc = ORC_ERR_OOM;
        goto out;
    }
    orc_transform_points(&o->current, dxyz, dnrm, nd, 12, uxyz, unrm, 12); /* :69 */
    rc = orc_map_add_points(o->keyframe, uxyz, unrm, nd, 12);               /* :70 */
    o->last.keyframe_voxels = (int64_t)orc_map_size(o->keyframe);
out:
    free(tn);
    free(dk);
    free(pl);
    free(pn);
    free(fl);
    free(fn);
    free(dxyz);
    free(dnrm);
    free(mxyz);
    free(uxyz);
    free(unrm);
    orc_map_destroy(down);
    orc_map_destroy(matchds);
    return rc;
}
