// f32 SE(3) algebra of the boundary type Pose3D (reference src/pose_3d.h:10-59)
// and the f64 manifold step used by the solver.  Host and device code (the device-resident
// Levenberg-Marquardt loop uses the same functions); compiled with -ffp-contract=off so the
// f32 rotation matrix that feeds the device search is rounded exactly like the reference's
// Eigen expressions on x86-64.
#pragma once
#include <cmath>

#include "../../include/lidar_odometry_amd.h"

#if defined(__HIPCC__)
#define LOM_HD __host__ __device__ inline
#else
#define LOM_HD inline
#endif

namespace lom {

// Eigen reduces fixed-size-3 expressions as a0 + (a1 + a2).
LOM_HD float sum3(float a, float b, float c) { return a + (b + c); }
LOM_HD double sum3(double a, double b, double c) { return a + (b + c); }

// Quaternion * vector as Eigen evaluates it: v + w*2(u x v) + u x 2(u x v).
template <typename T>
LOM_HD void quat_rotate(const T q[4], const T v[3], T out[3])
{
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    T a0 = y * v[2] - z * v[1];
    T a1 = z * v[0] - x * v[2];
    T a2 = x * v[1] - y * v[0];
    a0 += a0;
    a1 += a1;
    a2 += a2;
    const T c0 = y * a2 - z * a1;
    const T c1 = z * a0 - x * a2;
    const T c2 = x * a1 - y * a0;
    out[0] = (v[0] + w * a0) + c0;
    out[1] = (v[1] + w * a1) + c1;
    out[2] = (v[2] + w * a2) + c2;
}

template <typename T>
LOM_HD void quat_mul(const T a[4], const T b[4], T out[4])
{
    const T r0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const T r1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const T r2 = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    const T r3 = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    out[3] = r3;
}

// Quaternionf::toRotationMatrix (pose_3d.h:43), row-major.
LOM_HD void rotation_matrix(const float q[4], float R[9])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
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

inline void pose_compose(const lom_pose &a, const lom_pose &b, lom_pose &out)
{
    lom_pose r;
    float rt[3];
    quat_rotate<float>(a.q, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
    quat_mul<float>(a.q, b.q, r.q);
    out = r;
}

inline void pose_inverse(const lom_pose &a, lom_pose &out)
{
    lom_pose r;
    const float n2 = (a.q[0] * a.q[0] + a.q[1] * a.q[1]) + (a.q[2] * a.q[2] + a.q[3] * a.q[3]);
    if (n2 > 0.f) {
        r.q[0] = a.q[0] / n2;
        r.q[1] = -a.q[1] / n2;
        r.q[2] = -a.q[2] / n2;
        r.q[3] = -a.q[3] / n2;
    } else {
        r.q[0] = r.q[1] = r.q[2] = r.q[3] = 0.f;
    }
    const float nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
    quat_rotate<float>(r.q, nt, r.t);
    out = r;
}

// sin(a)/a and cos(a) for the rotation step of the manifold update.  LM steps are small angles:
// below 0.5 rad both come from their Taylor series (terms to a^20: truncation below 1e-22
// relative), which keeps the device-resident solve free of the math library's argument reduction
// (and its scratch memory); larger angles use sin / cos.  Host and device run this same code.
LOM_HD void sinc_cos(double a, double &sinc, double &c)
{
    if (a < 0.5) {
        const double z = a * a;
        double s = 1.0 / 51090942171709440000.0;  // 1/21!
        s = 1.0 / 121645100408832000.0 - z * s;   // 1/19!
        s = 1.0 / 355687428096000.0 - z * s;      // 1/17!
        s = 1.0 / 1307674368000.0 - z * s;        // 1/15!
        s = 1.0 / 6227020800.0 - z * s;           // 1/13!
        s = 1.0 / 39916800.0 - z * s;             // 1/11!
        s = 1.0 / 362880.0 - z * s;               // 1/9!
        s = 1.0 / 5040.0 - z * s;                 // 1/7!
        s = 1.0 / 120.0 - z * s;                  // 1/5!
        s = 1.0 / 6.0 - z * s;                    // 1/3!
        sinc = 1.0 - z * s;
        double k = 1.0 / 2432902008176640000.0;   // 1/20!
        k = 1.0 / 6402373705728000.0 - z * k;     // 1/18!
        k = 1.0 / 20922789888000.0 - z * k;       // 1/16!
        k = 1.0 / 87178291200.0 - z * k;          // 1/14!
        k = 1.0 / 479001600.0 - z * k;            // 1/12!
        k = 1.0 / 3628800.0 - z * k;              // 1/10!
        k = 1.0 / 40320.0 - z * k;                // 1/8!
        k = 1.0 / 720.0 - z * k;                  // 1/6!
        k = 1.0 / 24.0 - z * k;                   // 1/4!
        k = 1.0 / 2.0 - z * k;                    // 1/2!
        c = 1.0 - z * k;
    } else {
        sinc = sin(a) / a;
        c = cos(a);
    }
}

// Ceres QuaternionManifold::Plus on [w,x,y,z] (delta = half-angle vector applied
// on the left) followed by the Euclidean translation update.
LOM_HD void manifold_plus(const double x[7], const double d[6], double out[7])
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) {
        for (int i = 0; i < 4; i++) out[i] = x[i];
    } else {
        double s, c;
        sinc_cos(nd, s, c);
        const double z[4] = {c, s * d[0], s * d[1], s * d[2]};
        out[0] = z[0] * x[0] - z[1] * x[1] - z[2] * x[2] - z[3] * x[3];
        out[1] = z[0] * x[1] + z[1] * x[0] + z[2] * x[3] - z[3] * x[2];
        out[2] = z[0] * x[2] - z[1] * x[3] + z[2] * x[0] + z[3] * x[1];
        out[3] = z[0] * x[3] + z[1] * x[2] - z[2] * x[1] + z[3] * x[0];
    }
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + d[3 + i];
}

}  // namespace lom
