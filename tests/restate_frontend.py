"""Independent restatement, in numpy, of the stages LidarOdometry::processCloud runs before the align -- written
straight from the reference text (src/utils/point_time_normalize.h:15-39, src/utils/cloud_transform.h:15-40,
src/utils/cloud_classifier.h:19-168, src/utils/range_filter.h:13-28), NOT from oracle/pipeline.c or the product's
csrc/odometry.cpp (those two share their text; a misreading common to both passes every oracle-vs-product test).
Test infrastructure: tests/golden/make_fixtures.py runs it in the build container to write the golden planar-point
indices and normals of three seeded frames; tests/test_frontend_restatement.py compares oracle, host stages and
device front end with those.

Every operation is single-precision where the reference's is (numpy float32 arithmetic is correctly rounded and never
fused), and every place where the C++ text leaves the arithmetic to overload resolution or to Eigen's evaluation order
is a named READING with the alternative implemented beside it, so that the tests can say how many points of a frame
the choice moves:

  atan2   cloud_classifier.h:49  `float azimuth = atan2(-point.y, point.x) + std::numbers::pi;`  unqualified call with
          float arguments: ::atan2(double, double) when only <cmath> is in play (libstdc++ puts the float overload in
          std:: only), atan2f when a header pulled in libstdc++'s <math.h> wrapper (which adds `using std::atan2`).
          Either way the sum with the double pi is rounded to float.  READING "double" (what oracle and product assume).
  sqrt    :97  `float curvature = sqrt(dx*dx + dy*dy + dz*dz) / range;`  same question: double sqrt and a double
          division, or sqrtf and a float division.  READING "double".
  abs     :50  `std::abs(azimuth * max_row_width / (2.0 * std::numbers::pi))`: float * size_t is a FLOAT product (usual
          arithmetic conversions), divided in double, |.| in double, truncated to size_t.  No alternative.
  dot     Eigen Quaternionf::slerp's dot product: (a0 b0 + a1 b1) + (a2 b2 + a3 b3) without vectorisation or with SSE3
          hadd, (a0 b0 + a2 b2) + (a1 b1 + a3 b3) with plain SSE2.  READING "pairs".
"""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "acosf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def _sinf(x):
    """std::sin(float) of the platform's libm, element by element (numpy's own float32 sine differs in the last bit)"""
    a = np.asarray(x, F)
    return np.array([_libm.sinf(float(v)) for v in a.ravel()], F).reshape(a.shape)


# ---- utils::pointTimeNormalize, point_time_normalize.h:15-39 ----------------------------------------------------
def time_normalize(points):
    t = points["time"].astype(F)
    mn, mx = F(t.min()), F(t.max())          # :18-25 sequential min / max
    rng = F(mx - mn)                         # :27
    out = points.copy()
    out["time"] = (t - mn) / rng             # :35 (input.time - min_time) / time_range, f32
    return out


# ---- Eigen::Quaternionf pieces used by cloud_transform.h:27 -----------------------------------------------------
def _slerp(qa, t, qb, dot="pairs"):
    """Eigen 3.4 QuaternionBase::slerp, Scalar = float: t (n,) -> (n, 4) coefficients [w, x, y, z]"""
    a, b = np.asarray(qa, F), np.asarray(qb, F)
    p = a * b
    d = (p[0] + p[1]) + (p[2] + p[3]) if dot == "pairs" else (p[0] + p[2]) + (p[1] + p[3])
    one = F(1) - np.finfo(F).eps
    ad = F(abs(d))
    t = np.asarray(t, F)
    if ad >= one:
        s0, s1 = F(1) - t, t.copy()
    else:
        theta = F(_libm.acosf(float(ad)))
        sin_theta = F(_libm.sinf(float(theta)))
        s0 = _sinf((F(1) - t) * theta) / sin_theta
        s1 = _sinf(t * theta) / sin_theta
    if d < 0:
        s1 = -s1
    return s0[:, None] * a[None, :] + s1[:, None] * b[None, :]


def _rotate(q, v):
    """Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w uv + q.vec x uv (f32)"""
    w, qv = q[:, 0], q[:, 1:4]

    def cross(a, b):
        return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                         a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)

    uv = cross(qv, v)
    uv = uv + uv
    return (v + w[:, None] * uv) + cross(qv, uv)


# ---- CloudTransformer::transformNonRigid, cloud_transform.h:15-40 ------------------------------------------------
def deskew(points, start_t, start_q, end_t, end_q, dot="pairs"):
    """start_pose.rotation.slerp(time, end_pose.rotation) * p + start_pose.translation * time +
    end_pose.translation * (1.0 - time)  -- the translation weights are the reverse of the rotation's (:27-30).
    `(1.0 - time)` is a double; Eigen casts a scalar factor to the vector's Scalar (float) before multiplying."""
    t = points["time"].astype(F)
    v = np.stack([points["x"], points["y"], points["z"]], 1).astype(F)
    q = _slerp(start_q, t, end_q, dot)
    st, et = np.asarray(start_t, F), np.asarray(end_t, F)
    w_end = (1.0 - t.astype(np.float64)).astype(F)
    moved = (_rotate(q, v) + st[None, :] * t[:, None]) + et[None, :] * w_end[:, None]
    out = points.copy()
    out["x"], out["y"], out["z"] = moved[:, 0], moved[:, 1], moved[:, 2]
    return out


# ---- CloudClassifier::classify, cloud_classifier.h:19-168 ---------------------------------------------------------
def classify(points, atan2="double", sqrt="double"):
    """Returns (planar xyz (m,3) f32, normals (m,3) f32, cell index of every planar point in the organised cloud,
    (height, width)).  The unclassified cloud is discarded by the caller (lidar_odometry.cpp:33) and not built."""
    n = len(points)
    ring = points["ring"].astype(np.uint8)                     # :25 std::map<uint8_t, ...> keyed by point.ring
    ids = np.unique(ring)                                      # rows in ascending key order (:61)
    width = int(max((ring == r).sum() for r in ids)) if n else 0   # :35-40 the largest ring
    height = len(ids)
    x, y, z = points["x"].astype(F), points["y"].astype(F), points["z"].astype(F)
    # :49 azimuth
    if atan2 == "double":
        az = (np.arctan2(-y.astype(np.float64), x.astype(np.float64)) + np.pi).astype(F)
    else:
        az = (np.array([_libm.atan2f(float(-a), float(b)) for a, b in zip(y, x)], F).astype(np.float64) + np.pi).astype(F)
    # :50 std::abs(azimuth * max_row_width / (2.0 * pi)) -> size_t
    col = np.abs((az * F(width)).astype(np.float64) / (2.0 * np.pi)).astype(np.int64)
    org = np.zeros((height, width, 4), F)                      # :45 PointType(): x y z intensity = 0
    row_of = {int(r): k for k, r in enumerate(ids)}
    rows = np.array([row_of[int(r)] for r in ring], np.int64)
    ok = col < width                                           # :52
    # :53 indexed_row[i] = point -- the LAST point of a ring that lands in a cell stays
    cell = rows * max(width, 1) + col
    order = np.flatnonzero(ok)
    last = {}
    for i in order:
        last[int(cell[i])] = int(i)
    flat = org.reshape(-1, 4)
    if last:
        cells = np.fromiter(last.keys(), np.int64)
        src = np.fromiter(last.values(), np.int64)
        flat[cells, 0], flat[cells, 1], flat[cells, 2] = x[src], y[src], z[src]
        flat[cells, 3] = points["intensity"].astype(F)[src]
    total = height * width
    # :81-106 curvature over the FLATTENED cloud, window +-4 (crosses ring boundaries; the first / last 4 cells keep
    # their intensity)
    X, Y, Z = flat[:, 0].copy(), flat[:, 1].copy(), flat[:, 2].copy()
    if total > 8:
        i = np.arange(4, total - 4)
        rng = (X[i] * X[i] + Y[i] * Y[i]) + Z[i] * Z[i]        # :86 powf(x,2) + powf(y,2) + powf(z,2), float
        empty = rng.astype(np.float64) < 0.1                   # :87 float < double literal
        nine = 4 * 2.0 + 1.0
        acc = []
        for c in (X, Y, Z):
            d = ((-c[i]).astype(np.float64) * nine).astype(F)  # :92-94 -x * (double) -> float
            for w in range(-4, 5):                             # :96-100 in this order, w = 0 included
                d = d + c[i + w]
            acc.append(d)
        ss = (acc[0] * acc[0] + acc[1] * acc[1]) + acc[2] * acc[2]
        with np.errstate(divide="ignore", invalid="ignore"):   # empty cells (range 0) are overwritten below
            if sqrt == "double":
                curv = (np.sqrt(ss.astype(np.float64)) / rng.astype(np.float64)).astype(F)   # :102
            else:
                curv = np.sqrt(ss) / rng
        flat[i, 3] = np.where(empty, F(1000.0), curv)          # :88 intensity_max, :104
    # :108-164 normals from the previous ring
    inten = flat[:, 3].reshape(height, width)
    P = flat[:, :3].reshape(height, width, 3)
    flat_thr = F(0.05)
    near_thr = np.float64(flat_thr) * 10.0                     # :121 float * double literal
    out_xyz, out_nrm, out_cell = [], [], []
    for ray in range(1, height):                               # :113
        for pi in range(4, width - 4):                         # :114
            if not (inten[ray, pi] < flat_thr):                # :116
                continue
            prev = ray - 1
            left = right = None
            for q in range(pi - 4, pi):                        # :120-127 first from the left
                if np.float64(inten[prev, q]) < near_thr:
                    left = P[prev, q]
                    break
            for q in range(pi + 4, pi, -1):                    # :129-136 first from the right
                if np.float64(inten[prev, q]) < near_thr:
                    right = P[prev, q]
                    break
            if left is None or right is None:                  # :138
                continue
            o = P[ray, pi]
            a, b = left - o, right - o
            c = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)   # :140 cross
            zz = F(c[0] * c[0]) + (F(c[1] * c[1]) + F(c[2] * c[2]))   # Eigen's 3-vector sum: a0 + (a1 + a2)
            if zz > 0:                                         # normalized(): divide by the norm unless it is zero
                c = c / np.sqrt(zz)
            out_xyz.append(o.copy())
            out_nrm.append(c)
            out_cell.append(ray * width + pi)
    m = len(out_xyz)
    return (np.array(out_xyz, F).reshape(m, 3), np.array(out_nrm, F).reshape(m, 3), np.array(out_cell, np.int64),
            (height, width))


# ---- utils::rangeFilter, range_filter.h:13-28 -----------------------------------------------------------------------
def range_filter(xyz, min_range, max_range):
    """keep mask: min^2 <= x*x + y*y + z*z <= max^2, all float, left to right"""
    lo, hi = F(min_range) * F(min_range), F(max_range) * F(max_range)
    r = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]
    return (r >= lo) & (r <= hi)


def front_end(points, start_t, start_q, end_t, end_q, min_range=4.0, max_range=80.0, **readings):
    """lidar_odometry.cpp:25-35 for one frame: time-normalise, deskew (start = relative.inverse(), end = identity),
    classify, range filter.  Returns the dict the fixtures hold."""
    dot = readings.pop("dot", "pairs")
    desk = deskew(time_normalize(points), start_t, start_q, end_t, end_q, dot)
    xyz, nrm, cell, shape = classify(desk, **readings)
    keep = range_filter(xyz, min_range, max_range) if len(xyz) else np.zeros(0, bool)
    return {"deskewed_xyz": np.stack([desk["x"], desk["y"], desk["z"]], 1), "planar_cell": cell, "planar_xyz": xyz,
            "planar_nrm": nrm, "kept": keep, "shape": np.array(shape, np.int64)}
