"""Seeded synthetic piecewise-planar street scene.

Benchmark/test input generator only -- nothing here is on the measured path.
The reference ships no VLP16 data, so all VLP16 / 64- / 128-beam inputs are
synthetic: an analytic world (ground plane, two canyon walls, 40 axis-aligned
boxes) ray-cast from a sensor pose, plus an area-uniform map sampling with
exact plane normals.
"""
import numpy as np

SEED_GEOMETRY = 0x5EED0001
SEED_NOISE = 0x5EED0002
SEED_MAP = 0x5EED0003

GROUND_Z = -1.8
WALL_Y = 12.0
WALL_TOP = GROUND_Z + 10.0
N_BOXES = 40
MAX_RANGE = 100.0


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_boxes(seed=SEED_GEOMETRY):
    """(40, 6) array [xmin, ymin, zmin, xmax, ymax, zmax]; |y| < 3 m corridor kept free."""
    r = _rng(seed)
    boxes = []
    while len(boxes) < N_BOXES:
        cx, cy = r.uniform(-90, 90, 2)
        sx, sy = r.uniform(1, 6, 2)
        h = r.uniform(1, 8)
        if abs(cy) - sy / 2 < 3.0:
            continue
        boxes.append([cx - sx / 2, cy - sy / 2, GROUND_Z, cx + sx / 2, cy + sy / 2, GROUND_Z + h])
    return np.asarray(boxes, dtype=np.float64)


def quat_from_ypr(yaw_deg=0.0, pitch_deg=0.0, roll_deg=0.0):
    """wxyz quaternion, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    y, p, r = np.deg2rad([yaw_deg, pitch_deg, roll_deg]) / 2
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], dtype=np.float64)


def quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def beam_elevations(n_beams):
    if n_beams == 16:  # VLP16: -15..+15 step 2 deg
        return np.arange(-15.0, 15.1, 2.0)
    return np.linspace(-25.0, 15.0, n_beams)  # 64 / 128 beams


def raycast(origin, dirs, boxes):
    """Nearest hit distance of rays origin + s*dirs against the scene; inf if none."""
    o = np.asarray(origin, np.float64)
    d = np.asarray(dirs, np.float64)
    n = len(d)
    best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground
        s = (GROUND_Z - o[2]) / d[:, 2]
        s[~(s > 1e-6)] = np.inf
        best = np.minimum(best, s)
        # walls y = +-WALL_Y, GROUND_Z <= z <= WALL_TOP
        for wy in (-WALL_Y, WALL_Y):
            s = (wy - o[1]) / d[:, 1]
            z = o[2] + s * d[:, 2]
            s[~((s > 1e-6) & (z >= GROUND_Z) & (z <= WALL_TOP))] = np.inf
            best = np.minimum(best, s)
        # boxes (slab test), chunked to bound memory
        inv = 1.0 / d
        for b in boxes:
            t1 = (b[:3] - o) * inv
            t2 = (b[3:] - o) * inv
            tn = np.nanmax(np.minimum(t1, t2), axis=1)
            tf = np.nanmin(np.maximum(t1, t2), axis=1)
            hit = (tn <= tf) & (tn > 1e-6)
            best = np.where(hit & (tn < best), tn, best)
    return best


def make_scan(n_beams=16, n_az=1800, true_t=(0.10, -0.05, 0.02), true_ypr=(1.0, 0.2, 0.0),
              noise_sigma=0.02, seed_noise=SEED_NOISE, boxes=None, beam_range=None):
    """Scan in the SENSOR frame taken from pose (true_t, true_ypr) in the map frame.

    Returns (xyz float32 (n,3), ring uint16, az_step int32, true_q wxyz float64).
    Order: azimuth-major within each beam (beam-major), i.e. contiguous index
    ranges are whole beams -- the sharding unit of the multi-GPU configs.
    beam_range=(b0, b1) generates only beams [b0, b1).
    """
    boxes = make_boxes() if boxes is None else boxes
    el = np.deg2rad(beam_elevations(n_beams))
    b0, b1 = (0, n_beams) if beam_range is None else beam_range
    az = np.arange(n_az) * (2 * np.pi / n_az)
    ring = np.repeat(np.arange(b0, b1), n_az)
    azs = np.tile(np.arange(n_az), b1 - b0)
    ce, se = np.cos(el[ring]), np.sin(el[ring])
    d_s = np.stack([ce * np.cos(az[azs]), ce * np.sin(az[azs]), se], axis=1)
    q = quat_from_ypr(*true_ypr)
    R = quat_to_matrix(q)
    d_w = d_s @ R.T
    rng = raycast(np.asarray(true_t, np.float64), d_w, boxes)
    # noise stream is indexed by (ring, az) so a beam sub-range reproduces the full scan
    noise_all = _rng(seed_noise).normal(0.0, noise_sigma, size=(n_beams, n_az))
    rng = rng + noise_all[ring, azs]
    keep = np.isfinite(rng) & (rng < MAX_RANGE) & (rng > 0.5)
    xyz = (d_s[keep] * rng[keep, None]).astype(np.float32)
    return xyz, ring[keep].astype(np.uint16), azs[keep].astype(np.int32), q


def make_map_points(n_points, radius=80.0, seed=SEED_MAP, boxes=None):
    """Area-uniform noise-free samples on the scene surfaces within `radius` of the
    origin, with analytic unit normals.  Returns (xyz f32 (n,3), normals f32 (n,3))."""
    boxes = make_boxes() if boxes is None else boxes
    r = _rng(seed)
    # surface list: (area, sampler)
    surfs = []
    surfs.append(("ground", (2 * radius) ** 2))
    surfs.append(("wall-", 2 * radius * (WALL_TOP - GROUND_Z)))
    surfs.append(("wall+", 2 * radius * (WALL_TOP - GROUND_Z)))
    for i, b in enumerate(boxes):
        sx, sy, sz = b[3] - b[0], b[4] - b[1], b[5] - b[2]
        surfs += [((i, "x-"), sy * sz), ((i, "x+"), sy * sz), ((i, "y-"), sx * sz),
                  ((i, "y+"), sx * sz), ((i, "top"), sx * sy)]
    areas = np.array([a for _, a in surfs])
    prob = areas / areas.sum()
    out_p = np.empty((0, 3))
    out_n = np.empty((0, 3))
    while len(out_p) < n_points:
        m = int((n_points - len(out_p)) * 1.4) + 1024
        which = r.choice(len(surfs), size=m, p=prob)
        u, v = r.random(m), r.random(m)
        p = np.zeros((m, 3))
        nn = np.zeros((m, 3))
        g = which == 0
        p[g] = np.stack([(u[g] * 2 - 1) * radius, (v[g] * 2 - 1) * radius, np.full(g.sum(), GROUND_Z)], 1)
        nn[g] = (0, 0, 1)
        for k, (wy, ny) in enumerate(((-WALL_Y, 1.0), (WALL_Y, -1.0))):
            g = which == 1 + k
            p[g] = np.stack([(u[g] * 2 - 1) * radius, np.full(g.sum(), wy),
                             GROUND_Z + v[g] * (WALL_TOP - GROUND_Z)], 1)
            nn[g] = (0, ny, 0)
        bi = (which - 3) // 5
        fi = (which - 3) % 5
        for f in range(5):
            g = (which >= 3) & (fi == f)
            if not g.any():
                continue
            b = boxes[bi[g]]
            uu, vv = u[g], v[g]
            if f in (0, 1):
                x = b[:, 0] if f == 0 else b[:, 3]
                p[g] = np.stack([x, b[:, 1] + uu * (b[:, 4] - b[:, 1]), b[:, 2] + vv * (b[:, 5] - b[:, 2])], 1)
                nn[g] = (-1 if f == 0 else 1, 0, 0)
            elif f in (2, 3):
                y = b[:, 1] if f == 2 else b[:, 4]
                p[g] = np.stack([b[:, 0] + uu * (b[:, 3] - b[:, 0]), y, b[:, 2] + vv * (b[:, 5] - b[:, 2])], 1)
                nn[g] = (0, -1 if f == 2 else 1, 0)
            else:
                p[g] = np.stack([b[:, 0] + uu * (b[:, 3] - b[:, 0]), b[:, 1] + vv * (b[:, 4] - b[:, 1]), b[:, 5]], 1)
                nn[g] = (0, 0, 1)
        keep = (p[:, 0] ** 2 + p[:, 1] ** 2 + p[:, 2] ** 2) <= radius * radius
        out_p = np.concatenate([out_p, p[keep]])
        out_n = np.concatenate([out_n, nn[keep]])
    return out_p[:n_points].astype(np.float32), out_n[:n_points].astype(np.float32)


# BASELINE.json configs (C2..C4): (n_beams, n_az, map points)
CONFIGS = {
    "C2": dict(n_beams=16, n_az=1800, map_points=500_000),
    "C3": dict(n_beams=64, n_az=2048, map_points=2_000_000),
    "C4": dict(n_beams=128, n_az=2048, map_points=2_000_000),
}


# ---- streaming sequence (BASELINE.json configs[4] / SURVEY.md 8d "C5") ---------------------------

POINT_XYZIRT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"),
                         ("ring", "<u2"), ("pad1", "<u2"), ("time", "<f4"), ("pad2", "<f4")])
FRAME_PERIOD = 0.1      # 10 Hz
SPEED = 5.0             # m/s along +x, reached after RAMP seconds from rest
YAW_RATE_DEG = 5.0      # deg/s, likewise
RAMP = 2.5              # s of constant acceleration (2 m/s^2, 2 deg/s^2)


def _ramp_integral(t):
    """integral of min(1, tau / RAMP) d tau over [0, t]"""
    t = float(t)
    return t * t / (2 * RAMP) if t <= RAMP else RAMP / 2 + (t - RAMP)


def sequence_pose(t):
    """Ground-truth sensor pose (position, wxyz quaternion) at time t [s].  The vehicle starts at
    rest and accelerates to 5 m/s and 5 deg/s within 2.5 s: the reference's constant-velocity
    guess starts from zero velocity and its 0.3 m correspondence gate cannot acquire a 0.5 m jump
    between the first two frames."""
    s = _ramp_integral(t)
    return np.array([SPEED * s, 0.0, 0.0]), quat_from_ypr(YAW_RATE_DEG * s, 0.0, 0.0)


def make_sequence_frame(k, n_beams=16, n_az=1800, noise_sigma=0.02, boxes=None):
    """Frame k of the streaming sequence as lidar_point::PointXYZIRT records in firing order
    (azimuth-major).  Every return is taken from the sensor pose at ITS OWN firing time
    t_k + az/n_az * 0.1 s and reported in that instant's sensor frame (motion distortion, which
    the pipeline's deskew step removes); `time` = seconds since the frame start, `ring` = beam."""
    boxes = make_boxes() if boxes is None else boxes
    el = np.deg2rad(beam_elevations(n_beams))
    az = np.arange(n_az) * (2 * np.pi / n_az)
    azs = np.repeat(np.arange(n_az), n_beams)
    ring = np.tile(np.arange(n_beams), n_az)
    t_pt = azs / n_az * FRAME_PERIOD
    ce, se = np.cos(el[ring]), np.sin(el[ring])
    d_s = np.stack([ce * np.cos(az[azs]), ce * np.sin(az[azs]), se], axis=1)
    rng = np.full(len(d_s), np.inf)
    t0 = k * FRAME_PERIOD
    # one ray-cast per azimuth step would be slow in numpy: group steps into 36 chunks whose pose
    # is evaluated at the chunk centre (50 steps = 2.8 ms = 1.4 cm of travel)
    for c in range(0, n_az, 50):
        sel = (azs >= c) & (azs < c + 50)
        pos, q = sequence_pose(t0 + (c + 25) / n_az * FRAME_PERIOD)
        rng[sel] = raycast(pos, d_s[sel] @ quat_to_matrix(q).T, boxes)
    noise = _rng(SEED_NOISE + 1000 + k).normal(0.0, noise_sigma, size=len(d_s))
    rng = rng + noise
    keep = np.isfinite(rng) & (rng < MAX_RANGE) & (rng > 0.5)
    out = np.zeros(int(keep.sum()), POINT_XYZIRT)
    xyz = (d_s[keep] * rng[keep, None]).astype(np.float32)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["ring"] = ring[keep]
    out["time"] = t_pt[keep].astype(np.float32)
    return out
