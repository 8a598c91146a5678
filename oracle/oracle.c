/*
 * oracle/oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * CPU restatement of the reference's hot path.  Citations are relative to
 * /root/reference.  Compile with -ffp-contract=off: the reference is built
 * with plain -O3 on x86-64 (CMakeLists.txt:4-9), i.e. without FMA contraction,
 * and the f32 index/distance arithmetic below must round the same way.
 *
 * Third-party arithmetic restated here (absent from /root/reference):
 *   Eigen 3.4 (Quaternion::_transformVector, toRotationMatrix, inverse,
 *   operator*, 3-vector reductions a0 + (a1 + a2)), Ceres Solver 2.2
 *   (HuberLoss, Corrector, QuaternionManifold, NormalPrior, DENSE_QR,
 *   Levenberg-Marquardt trust-region minimizer).  PARITY UNPINNED at the
 *   last-ulp / LM-policy level: see oracle.h.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                            */
/* ------------------------------------------------------------------------ */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline const float *at(const float *base, size_t i, size_t stride)
{
    return (const float *)((const char *)base + i * stride);
}

/* Eigen fixed-size-3 reduction order: redux_novec_unroller<.,.,0,3> splits
 * as run<0,1> + run<1,2>  ==  a0 + (a1 + a2). */
static inline float sum3f(float a0, float a1, float a2) { return a0 + (a1 + a2); }
static inline double sum3d(double a0, double a1, double a2) { return a0 + (a1 + a2); }

/* ------------------------------------------------------------------------ */
/* Pose3D, f32  (src/pose_3d.h)                                             */
/* ------------------------------------------------------------------------ */

void orc_pose_identity(orc_pose *p)
{
    p->t[0] = p->t[1] = p->t[2] = 0.f; /* pose_3d.h:15-18 */
    p->q[0] = 1.f;
    p->q[1] = p->q[2] = p->q[3] = 0.f;
}

/* Eigen QuaternionBase::_transformVector:  v + w*2(u x v) + u x 2(u x v) */
static void quat_rotate_f(const float q[4], const float v[3], float out[3])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    float uv0 = y * v[2] - z * v[1];
    float uv1 = z * v[0] - x * v[2];
    float uv2 = x * v[1] - y * v[0];
    uv0 += uv0;
    uv1 += uv1;
    uv2 += uv2;
    const float c0 = y * uv2 - z * uv1;
    const float c1 = z * uv0 - x * uv2;
    const float c2 = x * uv1 - y * uv0;
    out[0] = (v[0] + w * uv0) + c0;
    out[1] = (v[1] + w * uv1) + c1;
    out[2] = (v[2] + w * uv2) + c2;
}

static void quat_rotate_d(const double q[4], const double v[3], double out[3])
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    double uv0 = y * v[2] - z * v[1];
    double uv1 = z * v[0] - x * v[2];
    double uv2 = x * v[1] - y * v[0];
    uv0 += uv0;
    uv1 += uv1;
    uv2 += uv2;
    const double c0 = y * uv2 - z * uv1;
    const double c1 = z * uv0 - x * uv2;
    const double c2 = x * uv1 - y * uv0;
    out[0] = (v[0] + w * uv0) + c0;
    out[1] = (v[1] + w * uv1) + c1;
    out[2] = (v[2] + w * uv2) + c2;
}

/* Eigen quat_product (generic form) */
static void quat_mul_f(const float a[4], const float b[4], float out[4])
{
    out[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    out[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    out[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    out[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

void orc_pose_compose(const orc_pose *a, const orc_pose *b, orc_pose *out)
{
    /* pose_3d.h:31  {translation + rotation * another.translation, rotation * another.rotation} */
    orc_pose r;
    float rt[3];
    quat_rotate_f(a->q, b->t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a->t[i] + rt[i];
    quat_mul_f(a->q, b->q, r.q);
    *out = r;
}

void orc_pose_inverse(const orc_pose *a, orc_pose *out)
{
    /* pose_3d.h:36-38 ; Eigen Quaternion::inverse = conjugate / squaredNorm */
    orc_pose r;
    const float n2 = (a->q[0] * a->q[0] + a->q[1] * a->q[1]) + (a->q[2] * a->q[2] + a->q[3] * a->q[3]);
    if (n2 > 0.f) {
        r.q[0] = a->q[0] / n2;
        r.q[1] = -a->q[1] / n2;
        r.q[2] = -a->q[2] / n2;
        r.q[3] = -a->q[3] / n2;
    } else {
        r.q[0] = r.q[1] = r.q[2] = r.q[3] = 0.f;
    }
    const float nt[3] = {-a->t[0], -a->t[1], -a->t[2]};
    quat_rotate_f(r.q, nt, r.t);
    *out = r;
}

void orc_pose_relative_to(const orc_pose *a, const orc_pose *target, orc_pose *out)
{
    /* pose_3d.h:25-26 */
    orc_pose inv;
    orc_pose_inverse(a, &inv);
    orc_pose_compose(&inv, target, out);
}

void orc_pose_rotation_matrix(const orc_pose *a, float R[9])
{
    /* pose_3d.h:43 -> Eigen QuaternionBase::toRotationMatrix, f32 */
    const float w = a->q[0], x = a->q[1], y = a->q[2], z = a->q[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.f - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.f - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.f - (txx + tyy);
}

void orc_transform_points(const orc_pose *pose, const float *xyz_in, const float *nrm_in,
                          size_t n, size_t stride_in, float *xyz_out, float *nrm_out,
                          size_t stride_out)
{
    /* cloud_transform.h:48-49,56 (R * p + t, f32) and :82 (R * normal) */
    float R[9];
    orc_pose_rotation_matrix(pose, R);
    for (size_t i = 0; i < n; i++) {
        const float *p = at(xyz_in, i, stride_in);
        float *o = (float *)((char *)xyz_out + i * stride_out);
        const float p0 = p[0], p1 = p[1], p2 = p[2];
        o[0] = sum3f(R[0] * p0, R[1] * p1, R[2] * p2) + pose->t[0];
        o[1] = sum3f(R[3] * p0, R[4] * p1, R[5] * p2) + pose->t[1];
        o[2] = sum3f(R[6] * p0, R[7] * p1, R[8] * p2) + pose->t[2];
        if (nrm_in && nrm_out) {
            const float *nn = at(nrm_in, i, stride_in);
            float *no = (float *)((char *)nrm_out + i * stride_out);
            const float n0 = nn[0], n1 = nn[1], n2 = nn[2];
            no[0] = sum3f(R[0] * n0, R[1] * n1, R[2] * n2);
            no[1] = sum3f(R[3] * n0, R[4] * n1, R[5] * n2);
            no[2] = sum3f(R[6] * n0, R[7] * n1, R[8] * n2);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Voxel map  (src/voxel_grid.h, src/voxel_with_planes.h)                   */
/* ------------------------------------------------------------------------ */

#define ORC_IDX_LIMIT 1048576.0f /* 2^20: range shared with the product's packed key */

struct orc_map {
    float voxel_size; /* voxel_grid.h:254 */
    size_t K;         /* row stride of the payload arrays: the largest max_points_ a stored voxel has seen */
    size_t max_points; /* max_points_, voxel_grid.h:253 (<= K) */
    size_t n_vox, cap_vox;
    int64_t *keys;   /* 3 per voxel: Indices{ix,iy,iz}, voxel_grid.h:20-23 */
    uint32_t *count; /* points_with_normals.size() */
    float *pts;      /* [cap_vox][K][3]  PointWithNormal::point  */
    float *nrm;      /* [cap_vox][K][3]  PointWithNormal::normal */
    size_t hcap;     /* power of two */
    int64_t *hslot;  /* voxel id or -1 */
};

/* voxel_grid.h:31-38 IndicesHash (Niessner primes); the 22-bit mask of the
 * reference is replaced by the table mask -- the hash value never influences
 * results, only robin_map's (unpinned) iteration order. */
static inline uint64_t idx_hash(int64_t ix, int64_t iy, int64_t iz)
{
    uint64_t h = (uint64_t)(ix * 73856093LL) ^ (uint64_t)(iy * 19349669LL) ^ (uint64_t)(iz * 83492791LL);
    h ^= h >> 17; /* spread into the low bits used by the power-of-two mask */
    return h;
}

/* voxel_grid.h:68-75 getIndices: static_cast<int64_t>(x / voxel_size_), f32
 * division, truncation toward zero.  Returns 0 if out of the supported range. */
static inline int vox_index(float x, float voxel_size, int64_t *out)
{
    const float f = x / voxel_size;
    if (!(f > -ORC_IDX_LIMIT && f < ORC_IDX_LIMIT)) return 0; /* also rejects NaN */
    *out = (int64_t)f;
    return 1;
}

static int64_t map_find(const orc_map *m, int64_t ix, int64_t iy, int64_t iz)
{
    if (m->hcap == 0) return -1;
    const size_t mask = m->hcap - 1;
    size_t h = (size_t)idx_hash(ix, iy, iz) & mask;
    for (;;) {
        const int64_t v = m->hslot[h];
        if (v < 0) return -1;
        const int64_t *k = m->keys + 3 * v;
        if (k[0] == ix && k[1] == iy && k[2] == iz) return v;
        h = (h + 1) & mask;
    }
}

static void map_index_put(orc_map *m, int64_t v)
{
    const size_t mask = m->hcap - 1;
    const int64_t *k = m->keys + 3 * v;
    size_t h = (size_t)idx_hash(k[0], k[1], k[2]) & mask;
    while (m->hslot[h] >= 0) h = (h + 1) & mask;
    m->hslot[h] = v;
}

static int map_rehash(orc_map *m, size_t hcap)
{
    int64_t *ns = (int64_t *)malloc(hcap * sizeof(int64_t));
    if (!ns) return ORC_ERR_OOM;
    for (size_t i = 0; i < hcap; i++) ns[i] = -1;
    free(m->hslot);
    m->hslot = ns;
    m->hcap = hcap;
    for (size_t v = 0; v < m->n_vox; v++) map_index_put(m, (int64_t)v);
    return ORC_OK;
}

static int map_reserve_voxels(orc_map *m, size_t want)
{
    if (want <= m->cap_vox) return ORC_OK;
    size_t nc = m->cap_vox ? m->cap_vox : 1024;
    while (nc < want) nc *= 2;
    int64_t *k = (int64_t *)realloc(m->keys, nc * 3 * sizeof(int64_t));
    if (!k) return ORC_ERR_OOM;
    m->keys = k;
    uint32_t *c = (uint32_t *)realloc(m->count, nc * sizeof(uint32_t));
    if (!c) return ORC_ERR_OOM;
    m->count = c;
    float *p = (float *)realloc(m->pts, nc * m->K * 3 * sizeof(float));
    if (!p) return ORC_ERR_OOM;
    m->pts = p;
    float *nn = (float *)realloc(m->nrm, nc * m->K * 3 * sizeof(float));
    if (!nn) return ORC_ERR_OOM;
    m->nrm = nn;
    m->cap_vox = nc;
    return ORC_OK;
}

orc_map *orc_map_create(float voxel_size, size_t max_points)
{
    if (!(voxel_size > 0.f) || max_points == 0) return NULL;
    orc_map *m = (orc_map *)calloc(1, sizeof(orc_map));
    if (!m) return NULL;
    m->voxel_size = voxel_size;
    m->K = max_points;
    m->max_points = max_points;
    if (map_rehash(m, 4096) != ORC_OK) {
        free(m);
        return NULL;
    }
    return m;
}

void orc_map_destroy(orc_map *m)
{
    if (!m) return;
    free(m->keys);
    free(m->count);
    free(m->pts);
    free(m->nrm);
    free(m->hslot);
    free(m);
}

int orc_map_clear(orc_map *m, float voxel_size)
{
    /* voxel_grid.h:61-66 setVoxelSize clears the map */
    if (!m || !(voxel_size > 0.f)) return ORC_ERR_ARG;
    m->voxel_size = voxel_size;
    m->n_vox = 0;
    for (size_t i = 0; i < m->hcap; i++) m->hslot[i] = -1;
    return ORC_OK;
}

int orc_map_set_max_points(orc_map *m, size_t max_points)
{
    /* voxel_grid.h:56-59: max_points_ = max_points, nothing else -- stored voxels keep what they hold, and
     * :86-90 appends to a voxel only while size() < max_points_.  The payload stride K follows the largest
     * value seen while voxels exist (a raise re-strides the arrays); an empty map starts over. */
    if (!m || max_points == 0) return ORC_ERR_ARG;
    if (m->n_vox == 0) {
        if (max_points != m->K) {
            free(m->pts);
            free(m->nrm);
            free(m->keys);
            free(m->count);
            m->pts = m->nrm = NULL;
            m->keys = NULL;
            m->count = NULL;
            m->cap_vox = 0;
            m->K = max_points;
        }
        m->max_points = max_points;
        return ORC_OK;
    }
    if (max_points > m->K) {
        const size_t oldK = m->K, newK = max_points;
        float *p = (float *)malloc(m->cap_vox * newK * 3 * sizeof(float));
        float *nn = (float *)malloc(m->cap_vox * newK * 3 * sizeof(float));
        if (!p || !nn) {
            free(p);
            free(nn);
            return ORC_ERR_OOM;
        }
        for (size_t v = 0; v < m->n_vox; v++) {
            memcpy(p + v * newK * 3, m->pts + v * oldK * 3, (size_t)m->count[v] * 3 * sizeof(float));
            memcpy(nn + v * newK * 3, m->nrm + v * oldK * 3, (size_t)m->count[v] * 3 * sizeof(float));
        }
        free(m->pts);
        free(m->nrm);
        m->pts = p;
        m->nrm = nn;
        m->K = newK;
    }
    m->max_points = max_points;
    return ORC_OK;
}

int orc_map_add_points(orc_map *m, const float *xyz, const float *nrm, size_t n, size_t stride)
{
    if (!m || (!xyz && n)) return ORC_ERR_ARG;
    if (stride < 3 * sizeof(float)) return ORC_ERR_ARG;
    /* validate first so a failing call inserts nothing */
    for (size_t i = 0; i < n; i++) {
        const float *p = at(xyz, i, stride);
        int64_t d;
        if (!vox_index(p[0], m->voxel_size, &d) || !vox_index(p[1], m->voxel_size, &d) ||
            !vox_index(p[2], m->voxel_size, &d))
            return ORC_ERR_RANGE;
    }
    const size_t K = m->K;
    for (size_t i = 0; i < n; i++) { /* voxel_grid.h:79 serial, input order */
        const float *p = at(xyz, i, stride);
        int64_t ix = 0, iy = 0, iz = 0;
        vox_index(p[0], m->voxel_size, &ix); /* :80 */
        vox_index(p[1], m->voxel_size, &iy);
        vox_index(p[2], m->voxel_size, &iz);
        int64_t v = map_find(m, ix, iy, iz); /* :82 */
        if (v < 0) {                         /* :83-87 new voxel, first point always stored */
            int rc = map_reserve_voxels(m, m->n_vox + 1);
            if (rc != ORC_OK) return rc;
            if ((m->n_vox + 1) * 2 > m->hcap) {
                rc = map_rehash(m, m->hcap * 2);
                if (rc != ORC_OK) return rc;
            }
            v = (int64_t)m->n_vox++;
            m->keys[3 * v + 0] = ix;
            m->keys[3 * v + 1] = iy;
            m->keys[3 * v + 2] = iz;
            m->count[v] = 0;
            map_index_put(m, v);
        } else if (m->count[v] >= m->max_points) { /* :89 size() < max_points_ */
            continue;
        }
        const size_t j = m->count[v]++;
        float *dp = m->pts + ((size_t)v * K + j) * 3;
        float *dn = m->nrm + ((size_t)v * K + j) * 3;
        dp[0] = p[0];
        dp[1] = p[1];
        dp[2] = p[2];
        if (nrm) {
            const float *q = at(nrm, i, stride);
            dn[0] = q[0];
            dn[1] = q[1];
            dn[2] = q[2];
        } else { /* :103,107 normal (0,0,0) */
            dn[0] = dn[1] = dn[2] = 0.f;
        }
    }
    return ORC_OK;
}

int orc_map_radius_cleanup(orc_map *m, const float center[3], float radius)
{
    /* voxel_grid.h:236-246: erase iff (getOrigin() - point).squaredNorm() > radius*radius,
     * f32, strict.  Survivors keep their relative (creation) order. */
    if (!m || !center) return ORC_ERR_ARG;
    const float r2 = radius * radius;
    const size_t K = m->K;
    size_t w = 0;
    for (size_t v = 0; v < m->n_vox; v++) {
        const float *o = m->pts + v * K * 3; /* voxel_with_planes.h:32-35 front() */
        const float dx = o[0] - center[0], dy = o[1] - center[1], dz = o[2] - center[2];
        const float d2 = sum3f(dx * dx, dy * dy, dz * dz);
        if (d2 > r2) continue;
        if (w != v) {
            memcpy(m->keys + 3 * w, m->keys + 3 * v, 3 * sizeof(int64_t));
            m->count[w] = m->count[v];
            memcpy(m->pts + w * K * 3, m->pts + v * K * 3, (size_t)m->count[v] * 3 * sizeof(float));
            memcpy(m->nrm + w * K * 3, m->nrm + v * K * 3, (size_t)m->count[v] * 3 * sizeof(float));
        }
        w++;
    }
    if (w != m->n_vox) {
        m->n_vox = w;
        for (size_t i = 0; i < m->hcap; i++) m->hslot[i] = -1;
        for (size_t v = 0; v < m->n_vox; v++) map_index_put(m, (int64_t)v);
    }
    return ORC_OK;
}

size_t orc_map_size(const orc_map *m) { return m ? m->n_vox : 0; }

size_t orc_map_point_count(const orc_map *m)
{
    size_t s = 0;
    if (!m) return 0;
    for (size_t v = 0; v < m->n_vox; v++) s += m->count[v];
    return s;
}

size_t orc_map_export(const orc_map *m, int mode, float *xyz_out, float *nrm_out, size_t cap)
{
    if (!m) return 0;
    const size_t K = m->K;
    size_t w = 0;
    for (size_t v = 0; v < m->n_vox; v++) {
        const size_t cnt = (mode == ORC_EXPORT_FIRST_PER_VOXEL) ? 1 : m->count[v];
        for (size_t j = 0; j < cnt; j++) {
            if (w < cap) {
                if (xyz_out) memcpy(xyz_out + 3 * w, m->pts + (v * K + j) * 3, 3 * sizeof(float));
                if (nrm_out && mode == ORC_EXPORT_FULL)
                    memcpy(nrm_out + 3 * w, m->nrm + (v * K + j) * 3, 3 * sizeof(float));
            }
            w++;
        }
    }
    return w;
}

/* ------------------------------------------------------------------------ */
/* Correspondence search                                                    */
/* ------------------------------------------------------------------------ */

/* voxel_grid.h:164-204 getCorrespondence */
static void query_one(const orc_map *m, const float query[3], double max_sq, orc_corr *c)
{
    c->index = -1;
    c->n_cand = 0;
    c->n_occ = 0;
    c->sq_dist = 0.f;
    memset(c->origin, 0, sizeof c->origin);
    memset(c->normal, 0, sizeof c->normal);
    int64_t ox, oy, oz;
    if (!vox_index(query[0], m->voxel_size, &ox) || !vox_index(query[1], m->voxel_size, &oy) ||
        !vox_index(query[2], m->voxel_size, &oz))
        return; /* outside the supported index range: nothing can be stored there */
    const size_t K = m->K;
    double min_dist = DBL_MAX; /* :172 */
    for (int64_t ix = ox - 1; ix <= ox + 1; ix++)         /* :175 */
        for (int64_t iy = oy - 1; iy <= oy + 1; iy++)     /* :176 */
            for (int64_t iz = oz - 1; iz <= oz + 1; iz++) { /* :177 */
                const int64_t v = map_find(m, ix, iy, iz);
                if (v < 0) continue;
                c->n_occ++;
                const uint32_t cnt = m->count[v];
                const float *p = m->pts + (size_t)v * K * 3;
                for (uint32_t j = 0; j < cnt; j++, p += 3) { /* :183 insertion order */
                    const float dx = query[0] - p[0], dy = query[1] - p[1], dz = query[2] - p[2];
                    const float d2f = sum3f(dx * dx, dy * dy, dz * dz); /* :184 f32 squaredNorm */
                    const double d2 = (double)d2f;
                    c->n_cand++;
                    if (d2 < max_sq && d2 < min_dist) { /* :186-187 strict */
                        min_dist = d2;
                        c->index = v * (int64_t)K + j;
                        c->sq_dist = d2f;
                    }
                }
            }
    if (c->index >= 0) { /* :195-199 */
        memcpy(c->origin, m->pts + c->index * 3, 3 * sizeof(float));
        memcpy(c->normal, m->nrm + c->index * 3, 3 * sizeof(float));
    }
}

/* voxel_grid.h:164 as the reference declares it: one f32 query already in the map frame, the SQUARED threshold a double */
int orc_get_correspondence(const orc_map *m, const float query[3], double max_correspondence_distance_sq, orc_corr *out)
{
    if (!m || !query || !out) return ORC_ERR_ARG;
    query_one(m, query, max_correspondence_distance_sq, out);
    return out->index >= 0 ? 1 : 0;
}

typedef struct {
    const orc_map *m;
    const float *src;
    size_t stride, begin, end;
    double R[9], t[3];
    double max_sq;
    orc_corr *out;
} search_job;

static void *search_worker(void *arg)
{
    search_job *j = (search_job *)arg;
    for (size_t i = j->begin; i < j->end; i++) {
        const float *p = at(j->src, i, j->stride);
        /* voxel_grid.h:220-223: f64 transform of the f32 point, cast to f32 */
        const double p0 = (double)p[0], p1 = (double)p[1], p2 = (double)p[2];
        const double w0 = sum3d(j->R[0] * p0, j->R[1] * p1, j->R[2] * p2) + j->t[0];
        const double w1 = sum3d(j->R[3] * p0, j->R[4] * p1, j->R[5] * p2) + j->t[1];
        const double w2 = sum3d(j->R[6] * p0, j->R[7] * p1, j->R[8] * p2) + j->t[2];
        const float qf[3] = {(float)w0, (float)w1, (float)w2};
        query_one(j->m, qf, j->max_sq, &j->out[i]);
    }
    return NULL;
}

int64_t orc_find_pairs(const orc_map *m, const float *src, size_t n, size_t stride,
                       const float t[3], const float q[4], float max_dist, orc_corr *out,
                       int nthreads)
{
    /* voxel_grid.h:206-234 findMatchingPairs; output kept in query order
     * (the reference's push order under the mutex is nondeterministic). */
    if (!m || (!src && n) || !out || stride < 12) return ORC_ERR_ARG;
    orc_pose pose;
    memcpy(pose.t, t, sizeof pose.t);
    memcpy(pose.q, q, sizeof pose.q);
    float Rf[9];
    orc_pose_rotation_matrix(&pose, Rf); /* :212 rotationMatrix().cast<double>() */
    const float max_sq_f = max_dist * max_dist; /* :215 f32 product */
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    search_job *jobs = (search_job *)calloc((size_t)nthreads, sizeof(search_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!jobs || !th) {
        free(jobs);
        free(th);
        return ORC_ERR_OOM;
    }
    for (int k = 0; k < nthreads; k++) {
        search_job *j = &jobs[k];
        j->m = m;
        j->src = src;
        j->stride = stride;
        j->begin = n * (size_t)k / (size_t)nthreads;
        j->end = n * (size_t)(k + 1) / (size_t)nthreads;
        for (int a = 0; a < 9; a++) j->R[a] = (double)Rf[a];
        for (int a = 0; a < 3; a++) j->t[a] = (double)t[a]; /* :213 */
        j->max_sq = (double)max_sq_f;
        j->out = out;
    }
    if (nthreads == 1) {
        search_worker(&jobs[0]);
    } else {
        for (int k = 0; k < nthreads; k++) pthread_create(&th[k], NULL, search_worker, &jobs[k]);
        for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    }
    free(jobs);
    free(th);
    int64_t valid = 0;
    for (size_t i = 0; i < n; i++) valid += out[i].index >= 0;
    return valid;
}

/* ------------------------------------------------------------------------ */
/* Residual / Jacobian  (src/cloud_matcher.cpp:38-103) + Ceres pieces       */
/* ------------------------------------------------------------------------ */

typedef struct {
    double p[3], o[3], n[3]; /* source_point_local, plane_origin, plane_normal */
} match_t;

#define HUBER_A 0.15          /* cloud_matcher.cpp:134 */
#define PRIOR_W 10.0          /* cloud_matcher.cpp:153  diag(0.1).inverse() */

/* ceres::HuberLoss::Evaluate */
static inline void huber(double s, double *rho0, double *rho1)
{
    const double b = HUBER_A * HUBER_A;
    if (s > b) {
        const double r = sqrt(s);
        *rho0 = 2.0 * HUBER_A * r - b;
        const double v = HUBER_A / r;
        *rho1 = v > DBL_MIN ? v : DBL_MIN;
    } else {
        *rho0 = s;
        *rho1 = 1.0;
    }
}

/* cloud_matcher.cpp:48-102 Evaluate + ceres QuaternionManifold::PlusJacobian.
 * Returns raw residual; jt (6) = tangent Jacobian row [rot(3), trans(3)]. */
static inline double p2pl_residual(const match_t *m, const double x[7], double *jt)
{
    const double *q = x, *t = x + 4;
    double rp[3];
    quat_rotate_d(q, m->p, rp); /* :54 rot*local_point */
    const double e0 = rp[0] + t[0] - m->o[0];
    const double e1 = rp[1] + t[1] - m->o[1];
    const double e2 = rp[2] + t[2] - m->o[2];
    const double r = sum3d(e0 * m->n[0], e1 * m->n[1], e2 * m->n[2]);
    if (jt) {
        const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        const double *p = m->p, *n = m->n;
        /* :64-86 four dR/dq_i matrices (each 2*M), times p, dot n */
        double v[3], ja[4];
        v[0] = 2.0 * q0 * p[0] + 2.0 * -q3 * p[1] + 2.0 * q2 * p[2];
        v[1] = 2.0 * q3 * p[0] + 2.0 * q0 * p[1] + 2.0 * -q1 * p[2];
        v[2] = 2.0 * -q2 * p[0] + 2.0 * q1 * p[1] + 2.0 * q0 * p[2];
        ja[0] = sum3d(v[0] * n[0], v[1] * n[1], v[2] * n[2]);
        v[0] = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        v[1] = 2.0 * q2 * p[0] + 2.0 * -q1 * p[1] + 2.0 * -q0 * p[2];
        v[2] = 2.0 * q3 * p[0] + 2.0 * q0 * p[1] + 2.0 * -q1 * p[2];
        ja[1] = sum3d(v[0] * n[0], v[1] * n[1], v[2] * n[2]);
        v[0] = 2.0 * -q2 * p[0] + 2.0 * q1 * p[1] + 2.0 * q0 * p[2];
        v[1] = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        v[2] = 2.0 * -q0 * p[0] + 2.0 * q3 * p[1] + 2.0 * -q2 * p[2];
        ja[2] = sum3d(v[0] * n[0], v[1] * n[1], v[2] * n[2]);
        v[0] = 2.0 * -q3 * p[0] + 2.0 * -q0 * p[1] + 2.0 * q1 * p[2];
        v[1] = 2.0 * q0 * p[0] + 2.0 * -q3 * p[1] + 2.0 * q2 * p[2];
        v[2] = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        ja[3] = sum3d(v[0] * n[0], v[1] * n[1], v[2] * n[2]);
        /* Ceres QuaternionPlusJacobian (4x3):
         *  [-x -y -z;  w  z -y;  -z  w  x;  y -x  w] */
        jt[0] = ja[0] * -q1 + ja[1] * q0 + ja[2] * -q3 + ja[3] * q2;
        jt[1] = ja[0] * -q2 + ja[1] * q3 + ja[2] * q0 + ja[3] * -q1;
        jt[2] = ja[0] * -q3 + ja[1] * -q2 + ja[2] * q1 + ja[3] * q0;
        jt[3] = n[0]; /* :96-98 */
        jt[4] = n[1];
        jt[5] = n[2];
    }
    return r;
}

/* Ceres ResidualBlock::Evaluate with Corrector (rho'' <= 0 branch): residual
 * and Jacobian row scaled by sqrt(rho').  res has nm+3 rows; J (row-major,
 * 6 columns) may be NULL.  Returns total cost 0.5*sum rho. */
static double evaluate(const match_t *M, size_t nm, const double x[7], const double prior_b[3],
                       double *res, double *J)
{
    double cost = 0.0;
    for (size_t i = 0; i < nm; i++) {
        double jt[6];
        const double r = p2pl_residual(&M[i], x, J ? jt : NULL);
        double rho0, rho1;
        huber(r * r, &rho0, &rho1);
        cost += 0.5 * rho0;
        const double sq = sqrt(rho1);
        if (res) res[i] = sq * r;
        if (J)
            for (int c = 0; c < 6; c++) J[i * 6 + c] = sq * jt[c];
    }
    /* ceres::NormalPrior: A (x - b), Jacobian A; TrivialLoss (cloud_matcher.cpp:135,153-154) */
    for (int a = 0; a < 3; a++) {
        const double rr = PRIOR_W * (x[4 + a] - prior_b[a]);
        cost += 0.5 * rr * rr;
        if (res) res[nm + a] = rr;
        if (J) {
            for (int c = 0; c < 6; c++) J[(nm + a) * 6 + c] = 0.0;
            J[(nm + a) * 6 + 3 + a] = PRIOR_W;
        }
    }
    return cost;
}

/* ceres QuaternionManifold::Plus for [w,x,y,z] + translation add */
static void manifold_plus(const double x[7], const double d[6], double out[7])
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) {
        for (int i = 0; i < 4; i++) out[i] = x[i];
    } else {
        const double s = sin(nd) / nd;
        const double z[4] = {cos(nd), s * d[0], s * d[1], s * d[2]};
        const double *w = x;
        out[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
        out[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
        out[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
        out[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
    }
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + d[3 + i];
}

/* DENSE_QR: min || [J; diag(D)] y - [r; 0] ||  by Householder QR (f64).
 * A is (rows x 6) row-major and is overwritten. */
static int qr_solve6(double *A, double *b, size_t rows, double y[6])
{
    for (int k = 0; k < 6; k++) {
        double nrm = 0.0;
        for (size_t i = (size_t)k; i < rows; i++) nrm += A[i * 6 + k] * A[i * 6 + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) return 0;
        const double alpha = A[k * 6 + k] > 0 ? -nrm : nrm;
        const double v0 = A[k * 6 + k] - alpha;
        /* v = [v0, A[k+1..,k]]; beta = 2 / (v.v) */
        double vv = v0 * v0;
        for (size_t i = (size_t)k + 1; i < rows; i++) vv += A[i * 6 + k] * A[i * 6 + k];
        if (vv == 0.0) return 0;
        const double beta = 2.0 / vv;
        for (int c = k + 1; c < 6; c++) {
            double s = v0 * A[k * 6 + c];
            for (size_t i = (size_t)k + 1; i < rows; i++) s += A[i * 6 + k] * A[i * 6 + c];
            s *= beta;
            A[k * 6 + c] -= s * v0;
            for (size_t i = (size_t)k + 1; i < rows; i++) A[i * 6 + c] -= s * A[i * 6 + k];
        }
        {
            double s = v0 * b[k];
            for (size_t i = (size_t)k + 1; i < rows; i++) s += A[i * 6 + k] * b[i];
            s *= beta;
            b[k] -= s * v0;
            for (size_t i = (size_t)k + 1; i < rows; i++) b[i] -= s * A[i * 6 + k];
        }
        A[k * 6 + k] = alpha;
    }
    for (int k = 5; k >= 0; k--) {
        double s = b[k];
        for (int c = k + 1; c < 6; c++) s -= A[k * 6 + c] * y[c];
        y[k] = s / A[k * 6 + k];
    }
    for (int k = 0; k < 6; k++)
        if (!isfinite(y[k])) return 0;
    return 1;
}

typedef struct {
    int recorded_iterations; /* summary.iterations.size() */
    int evaluations;
    int points;            /* distinct parameter points evaluated: iteration 0 + every candidate */
    double last_step_norm; /* summary.iterations.back().step_norm */
    double cost;
} lm_result;

/* Ceres 2.2 TrustRegionMinimizer + LevenbergMarquardtStrategy + DenseQRSolver
 * with the options of cloud_matcher.cpp:109-112 (max_num_iterations 4,
 * function_tolerance 1e-5, DENSE_QR) and library defaults otherwise. */
static int lm_solve(const match_t *M, size_t nm, double x[7], const double prior_b[3], lm_result *out)
{
    const size_t R = nm + 3;
    double *res = (double *)malloc(R * sizeof(double));
    double *J = (double *)malloc(R * 6 * sizeof(double));
    double *A = (double *)malloc((R + 6) * 6 * sizeof(double));
    double *bb = (double *)malloc((R + 6) * sizeof(double));
    if (!res || !J || !A || !bb) {
        free(res);
        free(J);
        free(A);
        free(bb);
        return ORC_ERR_OOM;
    }
    const int max_iter = 4;
    const double ftol = 1e-5, gtol = 1e-10, ptol = 1e-8;
    const double min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16;
    double radius = 1e4, decrease_factor = 2.0;
    int reuse_diag = 0;
    double scale[6], diag[6], g[6];

    out->recorded_iterations = 1; /* iteration 0 */
    out->evaluations = 1;
    out->points = 1;
    out->last_step_norm = 0.0;

    /* IterationZero: cost, residuals, Jacobian, gradient; Jacobi scaling once */
    double cost = evaluate(M, nm, x, prior_b, res, J);
    for (int c = 0; c < 6; c++) {
        double s = 0.0, gg = 0.0;
        for (size_t i = 0; i < R; i++) {
            s += J[i * 6 + c] * J[i * 6 + c];
            gg += J[i * 6 + c] * res[i];
        }
        scale[c] = 1.0 / (1.0 + sqrt(s));
        g[c] = gg;
    }
    for (size_t i = 0; i < R; i++)
        for (int c = 0; c < 6; c++) J[i * 6 + c] *= scale[c];
    double gmax = 0.0;
    for (int c = 0; c < 6; c++) gmax = fmax(gmax, fabs(g[c]));
    double x_norm = 0.0;
    for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
    x_norm = sqrt(x_norm);
    int invalid_run = 0;
    int rc = ORC_OK;

    if (gmax <= gtol) goto done;

    for (int iter = 1; iter <= max_iter; iter++) {
        /* LevenbergMarquardtStrategy::ComputeStep */
        if (!reuse_diag) {
            for (int c = 0; c < 6; c++) {
                double s = 0.0;
                for (size_t i = 0; i < R; i++) s += J[i * 6 + c] * J[i * 6 + c];
                diag[c] = fmin(fmax(s, min_diag), max_diag);
            }
        }
        memcpy(A, J, R * 6 * sizeof(double));
        memcpy(bb, res, R * sizeof(double));
        for (int c = 0; c < 6; c++) {
            for (int d = 0; d < 6; d++) A[(R + (size_t)c) * 6 + d] = 0.0;
            A[(R + (size_t)c) * 6 + c] = sqrt(diag[c] / radius);
            bb[R + (size_t)c] = 0.0;
        }
        double step[6];
        int ok = qr_solve6(A, bb, R + 6, step);
        reuse_diag = 1;
        double model_change = 0.0;
        if (ok) {
            for (int c = 0; c < 6; c++) step[c] = -step[c];
            /* model_cost_change = -(J s).(r + J s / 2) */
            for (size_t i = 0; i < R; i++) {
                double ms = 0.0;
                for (int c = 0; c < 6; c++) ms += J[i * 6 + c] * step[c];
                model_change -= ms * (res[i] + ms / 2.0);
            }
        }
        if (!ok || !(model_change > 0.0)) {
            /* HandleInvalidStep: recorded with step_norm = 0 */
            if (++invalid_run >= 5) break;
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            out->recorded_iterations++;
            out->last_step_norm = 0.0;
            continue;
        }
        invalid_run = 0;
        double delta[6], cand[7];
        for (int c = 0; c < 6; c++) delta[c] = step[c] * scale[c];
        manifold_plus(x, delta, cand);
        const double cand_cost = evaluate(M, nm, cand, prior_b, NULL, NULL);
        out->evaluations++;
        out->points++;
        double sn = 0.0;
        for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
        sn = sqrt(sn);
        /* ParameterToleranceReached / FunctionToleranceReached: return before
         * the iteration is recorded and without applying the candidate */
        if (sn <= ptol * (x_norm + ptol)) break;
        const double cost_change = cost - cand_cost;
        if (fabs(cost_change) <= ftol * cost) break;
        const double rel_dec = cost_change / model_change;
        if (rel_dec > min_rel_dec) {
            /* HandleSuccessfulStep */
            memcpy(x, cand, sizeof cand);
            x_norm = 0.0;
            for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
            x_norm = sqrt(x_norm);
            cost = evaluate(M, nm, x, prior_b, res, J);
            out->evaluations++;
            for (int c = 0; c < 6; c++) {
                double gg = 0.0;
                for (size_t i = 0; i < R; i++) gg += J[i * 6 + c] * res[i];
                g[c] = gg;
            }
            for (size_t i = 0; i < R; i++)
                for (int c = 0; c < 6; c++) J[i * 6 + c] *= scale[c];
            gmax = 0.0;
            for (int c = 0; c < 6; c++) gmax = fmax(gmax, fabs(g[c]));
            const double d3 = 2.0 * rel_dec - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - d3 * d3 * d3);
            radius = fmin(max_radius, radius);
            decrease_factor = 2.0;
            reuse_diag = 0;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            reuse_diag = 1;
        }
        out->recorded_iterations++;
        out->last_step_norm = sn;
        if (gmax <= gtol) break;
    }
done:
    out->cost = cost;
    free(res);
    free(J);
    free(A);
    free(bb);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* align  (src/cloud_matcher.cpp:105-178)                                   */
/* ------------------------------------------------------------------------ */

int orc_align(const orc_map *m, const float *src, size_t n, size_t stride, const float guess_t[3],
              const float guess_q[4], float out_t[3], float out_q[4], orc_align_stats *st,
              int nthreads)
{
    if (!m || (!src && n) || !guess_t || !guess_q || !out_t || !out_q || stride < 12)
        return ORC_ERR_ARG;
    orc_corr *corr = (orc_corr *)malloc((n ? n : 1) * sizeof(orc_corr));
    match_t *M = (match_t *)malloc((n ? n : 1) * sizeof(match_t));
    if (!corr || !M) {
        free(corr);
        free(M);
        return ORC_ERR_OOM;
    }
    orc_align_stats s;
    memset(&s, 0, sizeof s);
    orc_pose pose; /* :107 current_pose = position_guess */
    memcpy(pose.t, guess_t, sizeof pose.t);
    memcpy(pose.q, guess_q, sizeof pose.q);
    const double prior_b[3] = {(double)guess_t[0], (double)guess_t[1], (double)guess_t[2]}; /* :153 */
    int rc = ORC_OK;
    for (int i = 0; i < 35; i++) { /* :117 */
        double x[7] = {(double)pose.q[0], (double)pose.q[1], (double)pose.q[2], (double)pose.q[3], /* :122-126 */
                       (double)pose.t[0], (double)pose.t[1], (double)pose.t[2]};                   /* :129-131 */
        double t0 = now_s();
        const int64_t nv = orc_find_pairs(m, src, n, stride, pose.t, pose.q, 0.3f, corr, nthreads); /* :138-139 */
        if (nv < 0) {
            rc = (int)nv;
            break;
        }
        size_t nm = 0;
        for (size_t k = 0; k < n; k++) {
            s.cand_total += corr[k].n_cand;
            s.occ_total += corr[k].n_occ;
            if (corr[k].index < 0) continue;
            const float *p = at(src, k, stride);
            for (int a = 0; a < 3; a++) {
                M[nm].p[a] = (double)p[a]; /* voxel_grid.h:220,226 */
                M[nm].o[a] = (double)corr[k].origin[a];
                M[nm].n[a] = (double)corr[k].normal[a];
            }
            nm++;
        }
        double t1 = now_s();
        s.search_seconds += t1 - t0;
        lm_result lr;
        rc = lm_solve(M, nm, x, prior_b, &lr); /* :157-158 */
        s.solve_seconds += now_s() - t1;
        if (rc != ORC_OK) break;
        s.outer_iterations = i + 1;
        s.lm_iterations += lr.recorded_iterations;
        s.evaluations += lr.evaluations;
        s.points_evaluated += lr.points;
        s.queries += (int64_t)n;
        s.valid_last = (int64_t)nm;
        s.final_cost = lr.cost;
        s.last_step_norm = lr.last_step_norm;
        for (int a = 0; a < 4; a++) pose.q[a] = (float)x[a];     /* :161-164 */
        for (int a = 0; a < 3; a++) pose.t[a] = (float)x[4 + a]; /* :165-167 */
        if (lr.last_step_norm < 1e-4 && i > 3) break;            /* :169-172 */
    }
    /* :175 rotation.normalize(), f32 */
    {
        const float n2 = (pose.q[0] * pose.q[0] + pose.q[1] * pose.q[1]) +
                         (pose.q[2] * pose.q[2] + pose.q[3] * pose.q[3]);
        const float nn = sqrtf(n2);
        for (int a = 0; a < 4; a++) pose.q[a] = pose.q[a] / nn;
    }
    memcpy(out_t, pose.t, sizeof pose.t);
    memcpy(out_q, pose.q, sizeof pose.q);
    if (st) *st = s;
    free(corr);
    free(M);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Shard evaluators (reduced normal equations, layout of ORC_NSUMS)         */
/* ------------------------------------------------------------------------ */

struct orc_shard {
    const orc_map *m;
    const float *src;
    size_t n, stride;
    orc_corr *corr;
    match_t *M;
    size_t nm;
    double cand, occ;
};

orc_shard *orc_shard_create(const orc_map *m, const float *src, size_t n, size_t stride)
{
    orc_shard *s = (orc_shard *)calloc(1, sizeof(orc_shard));
    if (!s) return NULL;
    s->m = m;
    s->src = src;
    s->n = n;
    s->stride = stride;
    s->corr = (orc_corr *)malloc((n ? n : 1) * sizeof(orc_corr));
    s->M = (match_t *)malloc((n ? n : 1) * sizeof(match_t));
    if (!s->corr || !s->M) {
        orc_shard_destroy(s);
        return NULL;
    }
    return s;
}

void orc_shard_destroy(orc_shard *s)
{
    if (!s) return;
    free(s->corr);
    free(s->M);
    free(s);
}

static void shard_sums(const orc_shard *s, const double q[4], const double t[3], double out[ORC_NSUMS])
{
    const double x[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
    memset(out, 0, ORC_NSUMS * sizeof(double));
    for (size_t i = 0; i < s->nm; i++) {
        double jt[6];
        const double r = p2pl_residual(&s->M[i], x, jt);
        double rho0, rho1;
        huber(r * r, &rho0, &rho1);
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) out[k++] += rho1 * jt[a] * jt[b];
        for (int a = 0; a < 6; a++) out[21 + a] += rho1 * jt[a] * r;
        out[27] += 0.5 * rho0;
    }
    out[28] = (double)s->nm;
    out[29] = s->cand;
    out[30] = s->occ;
    out[31] = (double)s->n;
}

int orc_shard_match_eval(void *shard, const float pose_t[3], const float pose_q[4],
                         const double q[4], const double t[3], double out[ORC_NSUMS])
{
    orc_shard *s = (orc_shard *)shard;
    const int64_t nv = orc_find_pairs(s->m, s->src, s->n, s->stride, pose_t, pose_q, 0.3f, s->corr, 1);
    if (nv < 0) return (int)nv;
    s->nm = 0;
    s->cand = s->occ = 0.0;
    for (size_t k = 0; k < s->n; k++) {
        s->cand += s->corr[k].n_cand;
        s->occ += s->corr[k].n_occ;
        if (s->corr[k].index < 0) continue;
        const float *p = at(s->src, k, s->stride);
        for (int a = 0; a < 3; a++) {
            s->M[s->nm].p[a] = (double)p[a];
            s->M[s->nm].o[a] = (double)s->corr[k].origin[a];
            s->M[s->nm].n[a] = (double)s->corr[k].normal[a];
        }
        s->nm++;
    }
    shard_sums(s, q, t, out);
    return ORC_OK;
}

int orc_shard_eval_fixed(void *shard, const double q[4], const double t[3], double out[ORC_NSUMS])
{
    shard_sums((const orc_shard *)shard, q, t, out);
    return ORC_OK;
}
