// Host-side callers of the scan-matching core, ROS-free:
//   utils::pointTimeNormalize            reference src/utils/point_time_normalize.h:15-39
//   CloudTransformer::transformNonRigid  reference src/utils/cloud_transform.h:15-40
//   CloudClassifier::classify            reference src/utils/cloud_classifier.h:19-168
//   utils::rangeFilter                   reference src/utils/range_filter.h:13-28
//   LidarOdometry                        reference src/lidar_odometry.{h,cpp}
// In the reference these stay C++ on the host (north_star); they are restated here so the
// streaming configuration (BASELINE.json configs[4]) can run end to end without ROS2/PCL/Eigen.
// Everything that touches the voxel maps or the matcher goes through the C ABI, i.e. the GPU.
// Built with -ffp-contract=off; f32 expression shapes follow the reference's.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lidar_odometry_amd.h"
#include "pose_math.hpp"

namespace {

constexpr double kPi = 3.14159265358979323846;

// LOM_DEBUG_TIMING=1 (read once, by lom_odometry_create): per-stage wall times of processCloud on stderr
struct StageTimer {
    bool on;
    double t0 = now(), last = t0;
    explicit StageTimer(bool enabled) : on(enabled) {}
    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    void lap(const char *what)
    {
        if (!on) return;
        const double t = now();
        std::fprintf(stderr, "  %-14s %8.1f us\n", what, (t - last) * 1e6);
        last = t;
    }
    void total()
    {
        if (on) std::fprintf(stderr, "processCloud total %8.1f us\n", (now() - t0) * 1e6);
    }
};

// ---- host worker pool -------------------------------------------------------------
// The reference runs its per-point transforms under std::execution::par
// (point_time_normalize.h:31, cloud_transform.h:21); this is the same idea without TBB.
// Work is split into contiguous index ranges, so results do not depend on the thread count.
class Pool {
public:
    explicit Pool(unsigned n_threads)
    {
        for (unsigned i = 1; i < n_threads; i++) workers_.emplace_back([this, i] { run(i); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }

    // fn(begin, end, part) over [0, n) in size() contiguous parts; the caller takes part 0,
    // worker w always takes part w.  Workers spin for a while after each job (a frame issues
    // five of these within a millisecond) and park on a condition variable when idle longer.
    template <typename F>
    void parallel_for(size_t n, F &&fn, size_t serial_below = 2048)
    {
        const unsigned parts = size();
        if (parts == 1 || n < serial_below) {
            fn(size_t(0), n, 0u);
            return;
        }
        std::function<void(unsigned)> job = [&](unsigned p) { fn(n * p / parts, n * (p + 1) / parts, p); };
        job_ = &job;
        pending_.store(parts - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> l(m_);  // pairs with the parked workers' predicate check
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        job(0);
        while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
        job_ = nullptr;
    }

private:
    void run(unsigned part)
    {
        unsigned long seen = 0;
        for (;;) {
            // spin ~100 us for the next job, then park
            unsigned long g = seen;
            for (int spin = 0; spin < 40000 && (g = generation_.load(std::memory_order_acquire)) == seen; spin++)
                __builtin_ia32_pause();
            if (g == seen) {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                g = generation_.load(std::memory_order_acquire);
            }
            seen = g;
            if (stop_) return;
            (*job_)(part);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(unsigned)> *job_ = nullptr;
    std::atomic<unsigned> pending_{0};
    std::atomic<unsigned long> generation_{0};
    bool stop_ = false;
};

template <typename F>
void run_parts(Pool *pool, size_t n, F &&fn, size_t serial_below = 2048)
{
    if (pool)
        pool->parallel_for(n, fn, serial_below);
    else
        fn(size_t(0), n, 0u);
}

// ---- utils::pointTimeNormalize ---------------------------------------------------
void time_normalize(const lom_point_xyzirt *in, size_t n, lom_point_xyzirt *out, Pool *pool = nullptr)
{
    // min / max of the stamps (:21, sequential in the reference; exact, so parts may be combined)
    float part_lo[64], part_hi[64];
    for (int p = 0; p < 64; p++) part_lo[p] = 3.402823466e+38f, part_hi[p] = -3.402823466e+38f;
    run_parts(pool, n, [&](size_t b, size_t e, unsigned part) {
        float l = 3.402823466e+38f, h = -3.402823466e+38f;
        for (size_t i = b; i < e; i++) {
            l = in[i].time < l ? in[i].time : l;
            h = in[i].time > h ? in[i].time : h;
        }
        part_lo[part & 63] = l;
        part_hi[part & 63] = h;
    });
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    for (int p = 0; p < 64; p++) {
        lo = part_lo[p] < lo ? part_lo[p] : lo;
        hi = part_hi[p] > hi ? part_hi[p] : hi;
    }
    const float range = hi - lo;  // point_time_normalize.h:27 (0/0 when all stamps are equal, as there)
    run_parts(pool, n, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; i++) {
            out[i] = in[i];
            out[i].time = (in[i].time - lo) / range;
        }
    });
}

// ---- Eigen Quaternionf::slerp (used by transformNonRigid) ------------------------------
void slerp(const float a[4], float t, const float b[4], float out[4])
{
    const float one = 1.0f - 1.1920928955078125e-07f;
    const float d = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    const float ad = std::fabs(d);
    float s0, s1;
    if (ad >= one) {
        s0 = 1.0f - t;
        s1 = t;
    } else {
        const float theta = std::acos(ad);
        const float st = std::sin(theta);
        s0 = std::sin((1.0f - t) * theta) / st;
        s1 = std::sin(t * theta) / st;
    }
    if (d < 0.0f) s1 = -s1;
    for (int i = 0; i < 4; i++) out[i] = s0 * a[i] + s1 * b[i];
}

// ---- CloudTransformer::transformNonRigid ---------------------------------------------
void transform_non_rigid(const lom_point_xyzirt *in, size_t n, const lom_pose &start, const lom_pose &end,
                         lom_point_xyzirt *out, Pool *pool = nullptr)
{
    run_parts(pool, n, [&](size_t pb, size_t pe, unsigned) {
    for (size_t i = pb; i < pe; i++) {
        const float t = in[i].time;
        float q[4], r[3];
        slerp(start.q, t, end.q, q);  // cloud_transform.h:27
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        lom::quat_rotate<float>(q, p, r);
        const float w1 = (float)(1.0 - (double)t);  // :30
        out[i] = in[i];
        // the reference weights start.translation by time and end.translation by (1 - time)
        out[i].x = (r[0] + start.t[0] * t) + end.t[0] * w1;
        out[i].y = (r[1] + start.t[1] * t) + end.t[1] * w1;
        out[i].z = (r[2] + start.t[2] * t) + end.t[2] * w1;
    }
    });
}

// ---- utils::rangeFilter ------------------------------------------------------------
size_t range_filter(const float *xyz, const float *nrm, size_t n, float min_range, float max_range, float *xyz_out,
                    float *nrm_out, Pool *pool = nullptr)
{
    const float lo = min_range * min_range, hi = max_range * max_range;
    auto keep = [&](size_t i) {
        const float *p = xyz + 3 * i;
        const float r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        return r2 >= lo && r2 <= hi;
    };
    // contiguous parts: count, then copy each part to its offset -- the output keeps the input order
    size_t count[65] = {};
    run_parts(pool, n, [&](size_t b, size_t e, unsigned part) {
        size_t c = 0;
        for (size_t i = b; i < e; i++) c += keep(i) ? 1 : 0;
        count[part & 63] = c;
    });
    size_t offset[65];
    offset[0] = 0;
    for (int p = 0; p < 64; p++) offset[p + 1] = offset[p] + count[p];
    run_parts(pool, n, [&](size_t b, size_t e, unsigned part) {
        size_t w = offset[part & 63];
        for (size_t i = b; i < e; i++) {
            if (!keep(i)) continue;
            std::memcpy(xyz_out + 3 * w, xyz + 3 * i, 12);
            if (nrm && nrm_out) std::memcpy(nrm_out + 3 * w, nrm + 3 * i, 12);
            w++;
        }
    });
    return offset[64];
}

// ---- CloudClassifier::classify ---------------------------------------------------------
// planar points + normals (the unclassified cloud is discarded by the only caller,
// lidar_odometry.cpp:33, so only its size is reported)
struct ClassifyScratch {  // reused across frames: no allocation or zero-fill beyond what the algorithm needs
    std::vector<lom_point_xyzirt> cloud;
    std::vector<uint32_t> cell, hist;
    std::vector<float> tmp_xyz, tmp_nrm;
    std::vector<size_t> cnt_p, cnt_u, off_p;
};

size_t classify(const lom_point_xyzirt *in, size_t n, float *xyz_out, float *nrm_out, size_t *unclassified,
                size_t grid[2], ClassifyScratch &sc, Pool *pool = nullptr)
{
    std::vector<lom_point_xyzirt> &cloud = sc.cloud;
    // organise by ring (map key is uint8_t in the reference, :23) and azimuth bin
    size_t ring_count[256] = {};
    {
        std::vector<uint32_t> &hist = sc.hist;
        const unsigned parts = pool ? pool->size() : 1u;
        hist.assign((size_t)parts * 256, 0u);
        run_parts(pool, n, [&](size_t b, size_t e, unsigned part) {
            uint32_t *h = hist.data() + (size_t)part * 256;
            for (size_t i = b; i < e; i++) h[(uint8_t)in[i].ring]++;
        });
        for (unsigned p = 0; p < parts; p++)
            for (int r = 0; r < 256; r++) ring_count[r] += hist[(size_t)p * 256 + r];
    }
    int row_of[256];
    size_t H = 0, W = 0;
    for (int r = 0; r < 256; r++) {
        row_of[r] = -1;
        if (ring_count[r]) {
            row_of[r] = (int)H++;
            W = ring_count[r] > W ? ring_count[r] : W;
        }
    }
    if (grid) grid[0] = H, grid[1] = W;
    if (unclassified) *unclassified = 0;
    const size_t total = H * W;
    if (!total) return 0;
    if (cloud.size() < total) cloud.resize(total);
    run_parts(pool, total, [&](size_t b, size_t e, unsigned) {  // empty cells are zero points (:41-46)
        std::memset(static_cast<void *>(cloud.data() + b), 0, (e - b) * sizeof(lom_point_xyzirt));
    });
    // cell of every point in parallel, then the scatter in input order (last writer wins, :52-54)
    std::vector<uint32_t> &cell = sc.cell;
    if (cell.size() < n) cell.resize(n);
    run_parts(pool, n, [&](size_t pb, size_t pe, unsigned) {
        for (size_t i = pb; i < pe; i++) {
            const lom_point_xyzirt &p = in[i];
            const float azimuth = (float)(std::atan2((double)-p.y, (double)p.x) + kPi);       // :49 (double atan2)
            const size_t idx = (size_t)std::fabs((double)(azimuth * (float)W) / (2.0 * kPi));  // :50
            cell[i] = idx < W ? (uint32_t)((size_t)row_of[(uint8_t)p.ring] * W + idx) : 0xFFFFFFFFu;
        }
    });
    // every part owns a contiguous range of cells and walks the points in input order, so the last
    // writer of a cell is the same as in the sequential loop
    run_parts(pool, total, [&](size_t cb, size_t ce, unsigned) {
        for (size_t i = 0; i < n; i++) {
            const uint32_t c = cell[i];
            if (c >= cb && c < ce) cloud[c] = in[i];
        }
    });
    // curvature over the flattened array (+-4 window crosses ring boundaries), :76-103
    const int cw = 4;
    const float intensity_max = 1000.0f;
    if (total > (size_t)(2 * cw)) {
        // each cell reads its neighbours' coordinates only and writes its own intensity
        run_parts(pool, total - 2 * (size_t)cw, [&](size_t pb, size_t pe, unsigned) {
        for (size_t i = pb + (size_t)cw; i < pe + (size_t)cw; i++) {
            lom_point_xyzirt &o = cloud[i];
            const float range = powf(o.x, 2) + powf(o.y, 2) + powf(o.z, 2);
            if ((double)range < 0.1) {
                o.intensity = intensity_max;
                continue;
            }
            float dx = (float)((double)(-o.x) * (cw * 2.0 + 1.0));
            float dy = (float)((double)(-o.y) * (cw * 2.0 + 1.0));
            float dz = (float)((double)(-o.z) * (cw * 2.0 + 1.0));
            for (int w = -cw; w <= cw; w++) {
                dx += cloud[i + w].x;
                dy += cloud[i + w].y;
                dz += cloud[i + w].z;
            }
            o.intensity = (float)(std::sqrt((double)(dx * dx + dy * dy + dz * dz)) / (double)range);
        }
        });
    }
    // normals from the previous ring, :105-165
    const int nw = 4;
    const float flat = 0.05f;
    const double flat10 = (double)flat * 10.0;
    // rays are independent: each one fills its own slice, slices are concatenated in ray order
    std::vector<float> &tmp_xyz = sc.tmp_xyz, &tmp_nrm = sc.tmp_nrm;
    if (tmp_xyz.size() < total * 3) tmp_xyz.resize(total * 3), tmp_nrm.resize(total * 3);
    std::vector<size_t> &cnt_p = sc.cnt_p, &cnt_u = sc.cnt_u;
    cnt_p.assign(H, 0);
    cnt_u.assign(H, 0);
    run_parts(pool, H - 1, [&](size_t rb, size_t re, unsigned) {
    for (size_t ray = rb + 1; ray < re + 1; ray++) {
        size_t np = 0, nu = 0;
        float *oxyz = tmp_xyz.data() + ray * W * 3, *onrm = tmp_nrm.data() + ray * W * 3;
        for (long pi = nw; pi < (long)W - nw; pi++) {
            const lom_point_xyzirt &pt = cloud[ray * W + (size_t)pi];
            if (pt.intensity < flat) {
                const lom_point_xyzirt *row = &cloud[(ray - 1) * W];
                int found = 0;
                float L[3] = {0, 0, 0}, R[3] = {0, 0, 0};
                for (long q = pi - nw; q < pi; q++)
                    if ((double)row[q].intensity < flat10) {
                        L[0] = row[q].x, L[1] = row[q].y, L[2] = row[q].z;
                        found++;
                        break;
                    }
                for (long q = pi + nw; q > pi; q--)
                    if ((double)row[q].intensity < flat10) {
                        R[0] = row[q].x, R[1] = row[q].y, R[2] = row[q].z;
                        found++;
                        break;
                    }
                if (found == 2) {
                    const float a[3] = {L[0] - pt.x, L[1] - pt.y, L[2] - pt.z};
                    const float b[3] = {R[0] - pt.x, R[1] - pt.y, R[2] - pt.z};
                    float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
                    const float z = lom::sum3(c[0] * c[0], c[1] * c[1], c[2] * c[2]);
                    if (z > 0.f) {
                        const float s = std::sqrt(z);
                        c[0] /= s, c[1] /= s, c[2] /= s;
                    }
                    oxyz[3 * np] = pt.x, oxyz[3 * np + 1] = pt.y, oxyz[3 * np + 2] = pt.z;
                    onrm[3 * np] = c[0], onrm[3 * np + 1] = c[1], onrm[3 * np + 2] = c[2];
                    np++;
                } else {
                    nu++;
                }
            } else if (pt.intensity < intensity_max) {
                nu++;
            }
        }
        cnt_p[ray] = np;
        cnt_u[ray] = nu;
    }
    }, 2);
    size_t np = 0, nu = 0;
    std::vector<size_t> &off_p = sc.off_p;
    off_p.assign(H + 1, 0);
    for (size_t ray = 1; ray < H; ray++) {
        off_p[ray] = np;
        np += cnt_p[ray];
        nu += cnt_u[ray];
    }
    run_parts(pool, H - 1, [&](size_t rb, size_t re, unsigned) {
        for (size_t ray = rb + 1; ray < re + 1; ray++) {
            std::memcpy(xyz_out + 3 * off_p[ray], tmp_xyz.data() + ray * W * 3, cnt_p[ray] * 12);
            std::memcpy(nrm_out + 3 * off_p[ray], tmp_nrm.data() + ray * W * 3, cnt_p[ray] * 12);
        }
    }, 2);
    if (unclassified) *unclassified = nu;
    return np;
}

// Eigen eulerAngles(0,1,2) of (qa * qb^-1).toRotationMatrix(), degrees (lidar_odometry.cpp:54-55)
void delta_euler_deg(const float qa[4], const float qb[4], float out[3])
{
    lom_pose a{}, b{}, inv, prod;
    std::memcpy(a.q, qa, 16);
    std::memcpy(b.q, qb, 16);
    lom::pose_inverse(b, inv);
    lom::pose_compose(a, inv, prod);
    float m[9];
    lom::rotation_matrix(prod.q, m);
    auto M = [&m](int r, int c) { return m[r * 3 + c]; };
    float res[3];
    res[0] = std::atan2(M(1, 2), M(2, 2));
    const float c2 = std::sqrt(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
    if (res[0] > 0.f) {
        res[0] -= (float)kPi;
        res[1] = std::atan2(-M(0, 2), -c2);
    } else {
        res[1] = std::atan2(-M(0, 2), c2);
    }
    const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
    res[2] = std::atan2(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
    for (int i = 0; i < 3; i++) out[i] = ((-res[i]) * 180.0f) / (float)kPi;
}

}  // namespace

// ---- LidarOdometry (src/lidar_odometry.{h,cpp}) ---------------------------------------------
// One helper thread per odometry: the keyframe update of frame k (radiusCleanup, rigid transform,
// insert: lidar_odometry.cpp:67-70) does not influence frame k's pose, and frame k+1 touches the
// GPU handles only after its host stages (time normalisation, deskew, classifier, range filter).
// So processCloud returns the pose and lets the update run here; the next call (or any accessor)
// joins it before it uses a handle.  Same operations in the same order on the same stream: results
// do not change.
class Deferred {
public:
    Deferred() : th_([this] { loop(); }) {}
    ~Deferred()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void submit(std::function<int()> f)
    {
        bool asleep;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = std::move(f);
            busy_ = true;
            asleep = asleep_;
        }
        running_.store(true, std::memory_order_release);
        posted_.store(true, std::memory_order_release);
        if (asleep) cv_.notify_all();
    }
    int join()  // status of the last job (LOM_OK if none is pending)
    {
        // a job is a few tens of microseconds of enqueues and two looks at the device: watch for its end
        // before going to sleep on it (a futex wake-up costs as much as the job)
        for (int i = 0; i < kSpins && running_.load(std::memory_order_acquire); i++) __builtin_ia32_pause();
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !busy_; });
        const int rc = rc_;
        rc_ = LOM_OK;
        return rc;
    }

private:
    void loop()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            // frames that follow each other closely find the worker awake: it watches for the next job for
            // about half a millisecond before it sleeps on the condition variable (10 Hz input: asleep 99 %)
            g.unlock();
            for (int i = 0; i < kSpins && !posted_.load(std::memory_order_acquire); i++) __builtin_ia32_pause();
            g.lock();
            asleep_ = true;
            cv_.wait(g, [this] { return stop_ || (busy_ && job_); });
            asleep_ = false;
            if (stop_) return;
            posted_.store(false, std::memory_order_relaxed);
            std::function<int()> f = std::move(job_);
            job_ = nullptr;
            g.unlock();
            const int rc = f();
            g.lock();
            rc_ = rc;
            busy_ = false;
            running_.store(false, std::memory_order_release);
            cv_.notify_all();
        }
    }
    static constexpr int kSpins = 20000;  // x one `pause` (about 25 ns)
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    std::atomic<bool> posted_{false}, running_{false};  // a job waits for the worker / is not finished yet
    bool busy_ = false, stop_ = false, asleep_ = false;
    int rc_ = LOM_OK;
    std::thread th_;
};

struct lom_odometry {
    lom_odometry_params cfg;
    lom_map *keyframe = nullptr;       // keyframe_           lidar_odometry.h:82
    // keyframe_downsampler lidar_odometry.cpp:37: two workspaces, alternating per frame -- its output feeds the
    // keyframe update of the frame, which runs on the keyframe's stream beside the NEXT frame's stages
    lom_map *update_ds2[2] = {nullptr, nullptr};
    lom_map *update_ds = nullptr;      // the one of the current frame
    int parity = 0;
    lom_map *matching_ds = nullptr;    // matching_downsampler lidar_odometry.cpp:46 (reused per frame)
    lom_pose previous, current;        // lidar_odometry.h:84-85
    lom_odometry_frame_stats last{};
    std::vector<lom_point_xyzirt> normalized, deskewed;
    lom_frontend *frontend = nullptr;  // :25-35 on the device (csrc/frontend.hip); LOM_HOST_FRONTEND=1 keeps them on the host
    bool temp_on_device = false;       // temp_cloud_ lives in the front end's HBM buffer
    // keyframe_.size() != 0 (lidar_odometry.cpp:40), tracked on the host.  Atomic: the deferred keyframe update of
    // frame k writes it on the helper thread while frame k+1's stages read it.  A stale `true` is harmless -- the
    // frame prepares a matching cloud it does not use, and the init branch collects the update cloud's count --
    // and it never goes from false to true on the helper thread.
    std::atomic<bool> keyframe_has_voxels{false};
    bool test_force_host_redo = false;  // LOM_OPT_TEST_FORCE_HOST_REDO
    bool debug_timing = false;          // LOM_DEBUG_TIMING=1 at create / LOM_OPT_DEBUG_TIMING
    bool no_cleanup_behind_align = false;  // LOM_NO_CLEANUP_BEHIND_ALIGN=1 at create: the cleanup's scan waits for the host (A/B)
    bool no_send_ahead = false;            // LOM_NO_SEND_AHEAD=1 at create: hints are ignored (A/B)
    // lom_odometry_hint_next: the frame the caller will bring next; `ahead_*`: what the align's idle time has sent ahead
    const lom_point_xyzirt *hint_pts = nullptr, *hint_now = nullptr;  // (hint_now: the hint the running processCloud may use)
    size_t hint_n = 0;
    const lom_point_xyzirt *ahead_pts = nullptr;
    const lom_point_xyzirt *ahead_stage = nullptr;  // where in pinned memory it went
    size_t ahead_n = 0;
    uint64_t frames_sent_ahead = 0;
    int64_t grid_redos = 0;             // frames sent to the host stages because an in-kernel scan gave up
    size_t temp_points = 0;  // temp_cloud_ (lidar_odometry.h:73-77) = the first temp_points records of `deskewed`
    ClassifyScratch classify_scratch;
    std::vector<float> planar, planar_n, filtered, filtered_n, down, down_n, match, upd, upd_n;
    std::string error;
    std::unique_ptr<Pool> pool;  // host workers for the per-point stages (std::execution::par in the reference)
    int64_t queries_total = 0;
    std::unique_ptr<Deferred> deferred;  // keyframe update of the previous frame
    std::string deferred_error;
    // finish the previous frame's keyframe update; its failure is this call's failure
    int settle()
    {
        if (!deferred) return LOM_OK;
        const int rc = deferred->join();
        if (rc != LOM_OK) error = deferred_error;
        return rc;
    }
};

extern "C" {

void lom_point_time_normalize(const lom_point_xyzirt *in, size_t n, lom_point_xyzirt *out) { time_normalize(in, n, out); }

void lom_transform_non_rigid(const lom_point_xyzirt *in, size_t n, const lom_pose *start, const lom_pose *end,
                             lom_point_xyzirt *out)
{
    transform_non_rigid(in, n, *start, *end, out);
}

size_t lom_range_filter(const float *xyz, const float *nrm, size_t n, float min_range, float max_range, float *xyz_out,
                        float *nrm_out)
{
    return range_filter(xyz, nrm, n, min_range, max_range, xyz_out, nrm_out);
}

size_t lom_cloud_classify(const lom_point_xyzirt *in, size_t n, float *xyz_out, float *nrm_out,
                          size_t *unclassified_out, size_t grid_out[2])
{
    ClassifyScratch scratch;
    return classify(in, n, xyz_out, nrm_out, unclassified_out, grid_out, scratch);
}

void lom_odometry_default_params(lom_odometry_params *p)
{
    // lidar_odometry.h:36-48 and config/params.yaml
    p->lidar_min_range = 4.0f;
    p->lidar_max_range = 80.0f;
    p->keyframe_voxel_size = 0.2f;
    p->keyframe_max_points_cnt = 20;
    p->keyframe_matching_voxel_size = 0.3f;
    p->keyframe_update_voxel_size = 0.1f;
    p->keyframe_cleanup_range = 80.0f;
    p->angular_divergence_threshold = 5.0f;
}

int lom_odometry_create(const lom_odometry_params *params, int device, lom_odometry **out)
{
    if (!params || !out) return LOM_ERR_ARG;
    *out = nullptr;
    lom_odometry *o = new (std::nothrow) lom_odometry();
    if (!o) return LOM_ERR_OOM;
    o->cfg = *params;
    o->debug_timing = getenv("LOM_DEBUG_TIMING") != nullptr;
    o->no_cleanup_behind_align = getenv("LOM_NO_CLEANUP_BEHIND_ALIGN") != nullptr;
    o->no_send_ahead = getenv("LOM_NO_SEND_AHEAD") != nullptr;
    {
        unsigned hw = std::thread::hardware_concurrency();
        if (const char *e = getenv("LOM_HOST_THREADS")) hw = (unsigned)std::max(1, atoi(e));
        o->pool.reset(new Pool(std::max(1u, std::min(hw, 16u))));
    }
    if (!getenv("LOM_SYNC_KEYFRAME_UPDATE")) o->deferred.reset(new Deferred());
    lom_pose_identity(&o->current);  // lidar_odometry.cpp:15-17
    o->previous = o->current;
    int rc = lom_map_create(params->keyframe_voxel_size, params->keyframe_max_points_cnt, 1 << 16, device,
                            &o->keyframe);  // :18-19
    if (rc == LOM_OK) rc = lom_map_create(params->keyframe_update_voxel_size, 1, 1 << 15, device, &o->update_ds2[0]);
    if (rc == LOM_OK) rc = lom_map_create(params->keyframe_update_voxel_size, 1, 1 << 15, device, &o->update_ds2[1]);
    o->update_ds = o->update_ds2[0];
    if (rc == LOM_OK) rc = lom_map_create(params->keyframe_matching_voxel_size, 1, 1 << 14, device, &o->matching_ds);
    // Two streams: the keyframe's (align, keyframe update) and the front end's (upload, front-end kernels, both
    // down-samplers).  The stages of frame k+1 run beside the keyframe update of frame k; the host reads the
    // stage results back (a synchronisation with the front end's stream) before it launches the align.
    if (rc == LOM_OK && !getenv("LOM_HOST_FRONTEND")) rc = lom_frontend_create(device, nullptr, &o->frontend);
    void *stage_stream = o->frontend ? lom_frontend_stream(o->frontend) : (o->keyframe ? lom_map_get_stream(o->keyframe) : nullptr);
    for (lom_map *h : {o->update_ds2[0], o->update_ds2[1], o->matching_ds})
        if (rc == LOM_OK) rc = lom_map_set_stream(h, stage_stream);
    if (rc != LOM_OK) {
        lom_odometry_destroy(o);
        return rc;
    }
    *out = o;
    return LOM_OK;
}

void lom_odometry_destroy(lom_odometry *o)
{
    if (!o) return;
    (void)o->settle();
    o->deferred.reset();
    // the down-samplers run on the front end's stream (or the keyframe's): they go before the owner of that stream
    lom_map_destroy(o->update_ds2[0]);
    lom_map_destroy(o->update_ds2[1]);
    lom_map_destroy(o->matching_ds);
    lom_frontend_destroy(o->frontend);
    lom_map_destroy(o->keyframe);
    delete o;
}

const char *lom_odometry_last_error(const lom_odometry *o) { return o ? o->error.c_str() : ""; }
lom_map *lom_odometry_keyframe(lom_odometry *o)
{
    if (!o) return nullptr;
    (void)o->settle();
    return o->keyframe;
}

int lom_odometry_get_pose(const lom_odometry *o, lom_pose *out)
{
    if (!o || !out) return LOM_ERR_ARG;
    *out = o->current;  // lidar_odometry.cpp:87-89
    return LOM_OK;
}

// getTempCloud(), lidar_odometry.h:73-75: the deskewed input cloud of the last processCloud (:31)
int64_t lom_odometry_get_temp_cloud(const lom_odometry *o, lom_point_xyzirt *out, size_t cap)
{
    if (!o || (cap && !out)) return LOM_ERR_ARG;
    const size_t n = o->temp_points;
    if (o->temp_on_device) return lom_frontend_fetch(o->frontend, 0, out, nullptr, cap);
    if (out && cap) std::memcpy(static_cast<void *>(out), o->deskewed.data(), std::min(n, cap) * sizeof(lom_point_xyzirt));
    return (int64_t)n;
}

// test hook: overwrite previous_transform_ / current_transform_ (lidar_odometry.h:84-85); together with
// clear + add on lom_odometry_keyframe() this lets a test put the pipeline into a given state before a frame
int lom_odometry_debug_set_state(lom_odometry *o, const lom_pose *previous, const lom_pose *current)
{
    if (!o || !previous || !current) return LOM_ERR_ARG;
    const int rc = o->settle();
    o->previous = *previous;
    o->current = *current;
    o->keyframe_has_voxels = lom_map_size(o->keyframe) > 0;  // the test may have replaced the keyframe
    return rc;
}

int lom_odometry_set_option(lom_odometry *o, int option, int64_t value)
{
    if (!o) return LOM_ERR_ARG;
    int rc = o->settle();
    if (rc != LOM_OK) return rc;
    switch (option) {
    case LOM_OPT_TEST_FORCE_HOST_REDO: o->test_force_host_redo = value != 0; return LOM_OK;
    case LOM_OPT_DEBUG_TIMING: o->debug_timing = value != 0; return lom_map_set_option(o->keyframe, option, value);
    case LOM_OPT_TEST_GRID_GIVE_UP:  // the front end's scan of the next frame
        return o->frontend ? lom_frontend_set_option(o->frontend, option, value) : LOM_ERR_STATE;
    case LOM_OPT_TEST_GRID_GIVE_UP_MATCHING_DS: return lom_map_set_option(o->matching_ds, LOM_OPT_TEST_GRID_GIVE_UP, value);
    case LOM_OPT_TEST_GRID_GIVE_UP_UPDATE_DS:  // the workspace of the NEXT frame
        return lom_map_set_option(o->update_ds2[o->parity ^ 1], LOM_OPT_TEST_GRID_GIVE_UP, value);
    case LOM_OPT_TEST_GRID_GIVE_UP_KEYFRAME: return lom_map_set_option(o->keyframe, LOM_OPT_TEST_GRID_GIVE_UP, value);
    default: return lom_map_set_option(o->keyframe, option, value);  // the align's switches live on the keyframe handle
    }
}

int64_t lom_odometry_debug_counter(const lom_odometry *o, int which)
{
    if (!o) return LOM_ERR_ARG;
    if (which == LOM_COUNTER_GRID_REDOS) {
        int64_t v = o->grid_redos;
        for (lom_map *m : {o->keyframe, o->update_ds2[0], o->update_ds2[1], o->matching_ds}) v += lom_map_debug_counter(m, which);
        return v;
    }
    if (which == LOM_COUNTER_FRAMES_SENT_AHEAD) return (int64_t)o->frames_sent_ahead;
    if (which == LOM_COUNTER_CLEANUPS_BEHIND_ALIGN) {
        const int rc = const_cast<lom_odometry *>(o)->settle();
        return rc != LOM_OK ? rc : lom_map_debug_counter(o->keyframe, which);
    }
    return LOM_ERR_ARG;
}

int lom_odometry_get_stats(const lom_odometry *o, lom_odometry_frame_stats *out)
{
    if (!o || !out) return LOM_ERR_ARG;
    const int rc = const_cast<lom_odometry *>(o)->settle();  // keyframe_voxels comes from the keyframe update
    *out = o->last;
    return rc;
}

}  // extern "C" (the C ABI continues below)

// ---- LidarOdometry::processCloud, lidar_odometry.cpp:22-77 ---------------------------------------
namespace {

struct FrameInputs {  // what the stages before the align leave in HBM for it and for the keyframe update
    const float *d_down = nullptr, *d_down_n = nullptr;  // keyframe_downsampler.getCloud()            :37-38,42,69
    const float *d_match = nullptr;                      // matching_downsampler.getCloudWithoutNormals() :46-47,50
    int64_t nd = 0, nm = 0;
    // device stages: the update cloud's size and verdict are still on their way (a read-back is enqueued on this
    // workspace); collect_update() waits for them.  Returns LOM_OK / LOM_ERR_RANGE / LOM_ERR_HIP.
    lom_map *pending_update = nullptr;
    uint32_t pending_seq = 0;
    // the front end's filtered cloud (input of both down-samplers), for the redo of an update down-sampling whose
    // in-kernel scan gave up
    const float *d_fx = nullptr, *d_fn = nullptr;
    int64_t nf = 0;
    static constexpr int kScanGaveUp = 2;
    int collect_update(const char **error_out)
    {
        if (!pending_update) return LOM_OK;
        lom_map *m = pending_update;
        pending_update = nullptr;
        uint32_t w[3] = {0, 0, 0};
        const int rc = lom_map_read_device_words_end(m, w);
        if (rc != LOM_OK) {
            if (error_out) *error_out = lom_last_error(m);
            return rc;
        }
        if (w[2] == pending_seq) return kScanGaveUp;  // nothing written, workspace at rest: the caller redoes it
        if (w[1] == pending_seq) {
            if (error_out) *error_out = "coordinate / voxel_size out of range or not finite";
            return LOM_ERR_RANGE;
        }
        nd = w[0];
        return LOM_OK;
    }
};

int fail_map(lom_odometry *o, int rc, lom_map *m)
{
    o->error = lom_last_error(m);
    return rc;
}

// :25-47 on the host (worker pool), then one upload: the path of frames the device front end hands back
int stages_on_host(lom_odometry *o, const lom_point_xyzirt *pts, size_t n, const lom_pose &rel_inv, const lom_pose &ident,
                   lom_odometry_frame_stats &cur, FrameInputs &in, StageTimer &tm)
{
    const size_t cap = n ? n : 1;
    o->normalized.resize(cap);
    o->deskewed.resize(cap);
    for (auto *v : {&o->planar, &o->planar_n, &o->filtered, &o->filtered_n}) v->resize(cap * 3);
    time_normalize(pts, n, o->normalized.data(), o->pool.get());                                      // :25
    transform_non_rigid(o->normalized.data(), n, rel_inv, ident, o->deskewed.data(), o->pool.get());  // :30
    o->temp_points = n;  // :31 temp_cloud_ = deskewed_input_cloud
    o->temp_on_device = false;
    tm.lap("norm+deskew");
    size_t nu = 0;
    const size_t np = classify(o->deskewed.data(), n, o->planar.data(), o->planar_n.data(), &nu, nullptr,
                               o->classify_scratch, o->pool.get());  // :33
    const size_t nf = range_filter(o->planar.data(), o->planar_n.data(), np, o->cfg.lidar_min_range,
                                   o->cfg.lidar_max_range, o->filtered.data(), o->filtered_n.data(), o->pool.get());  // :35
    cur.planar_points = (int64_t)np;
    cur.filtered_points = (int64_t)nf;
    tm.lap("classify+filter");
    int rc;
    // the previous frame's keyframe update ran beside the host stages above; it must be through before this
    // frame touches a handle.  Its failure is reported here, by the call after the one it belongs to; this
    // frame is then not processed and poses / keyframe stay as they were.
    if ((rc = o->settle()) != LOM_OK) return rc;
    const float *d_fx = nullptr, *d_fn = nullptr;
    if ((rc = lom_upload_points(o->update_ds, o->filtered.data(), o->filtered_n.data(), nf, 12, &d_fx, &d_fn)) != LOM_OK)
        return fail_map(o, rc, o->update_ds);
    in.nd = lom_voxel_downsample_device(o->update_ds, o->cfg.keyframe_update_voxel_size, d_fx, d_fn, nf, 12, &in.d_down,
                                        &in.d_down_n);
    if (in.nd < 0) return fail_map(o, (int)in.nd, o->update_ds);
    if (o->keyframe_has_voxels) {
        in.nm = lom_voxel_downsample_device(o->matching_ds, o->cfg.keyframe_matching_voxel_size, d_fx, nullptr, nf, 12,
                                            &in.d_match, nullptr);
        if (in.nm < 0) return fail_map(o, (int)in.nm, o->matching_ds);
    }
    tm.lap("down-samplers");
    return LOM_OK;
}

// :25-47 on the device: the frame stays in HBM from its upload to its pose.  Front end (4 kernels), both
// down-samplers (2 kernels each) fed with device-side counts, then ONE look at the host for the sizes the
// align and the keyframe update are launched with.  Returns 1 when the front end hands the frame back.
int stages_on_device(lom_odometry *o, const lom_point_xyzirt *pts, size_t n, const lom_pose &rel_inv, const lom_pose &ident,
                     lom_odometry_frame_stats &cur, FrameInputs &in, StageTimer &tm)
{
    int rc;
    // front end and down-samplers take frames of up to ~170k points (their in-kernel scans cover 262144 cells /
    // points); larger ones go through the host stages
    if (n > 170000) return 1;
    // the frame goes into the front end's pinned buffer by the worker pool (one pass over ~1 MB), then to HBM -- unless
    // it went there while the previous frame's align ran (lom_odometry_hint_next)
    const bool staged = o->ahead_pts != nullptr && o->ahead_pts == pts && o->ahead_n == n;
    o->ahead_pts = nullptr;
    lom_point_xyzirt *stage = nullptr;
    if ((rc = lom_frontend_stage(o->frontend, n, &stage)) != LOM_OK) {
        o->error = lom_frontend_last_error(o->frontend);
        return rc;
    }
    if (staged && stage == o->ahead_stage) {
        o->frames_sent_ahead++;
    } else {
        run_parts(o->pool.get(), n, [&](size_t b, size_t e, unsigned) {
            std::memcpy(static_cast<void *>(stage + b), pts + b, (e - b) * sizeof(lom_point_xyzirt));
        }, 8192);
    }
    if ((rc = lom_frontend_process(o->frontend, stage, n, &rel_inv, &ident, o->cfg.lidar_min_range, o->cfg.lidar_max_range)) !=
        LOM_OK) {
        if (rc == LOM_ERR_ARG) return 1;  // a frame beyond the front end's size limit
        o->error = lom_frontend_last_error(o->frontend);
        return rc;
    }
    o->temp_points = n;  // :31 temp_cloud_ = deskewed_input_cloud (fetched from HBM on demand)
    o->temp_on_device = true;
    tm.lap("front end enq.");
    const float *d_fx = nullptr, *d_fn = nullptr;
    const uint32_t *d_fe = nullptr, *d_nd = nullptr, *d_nm = nullptr;
    uint32_t bound = 0;
    lom_frontend_results(o->frontend, &d_fx, &d_fn, &d_fe, &bound);
    // front end and down-samplers share a stream of their own: all of this runs beside the previous frame's
    // keyframe update (whose input is the OTHER update workspace).  Only the matching cloud is on the way to the
    // align; the keyframe-update cloud is enqueued behind the read-back the align waits for, runs beside the
    // align's first kernels, and its count and verdict are collected after the align (pending_update).
    const uint32_t *ptrs[12];
    uint32_t seq_u = 0, seq_m = 0, got[12] = {0};
    const uint32_t *u_range = nullptr, *u_grid = nullptr, *m_range = nullptr, *m_grid = nullptr;
    int k = 0;
    ptrs[k++] = d_fe;      // 0 planar
    ptrs[k++] = d_fe + 1;  // 1 filtered
    ptrs[k++] = d_fe + 4;  // 2 front end: redo on the host (sequence number of the frame)
    ptrs[k++] = d_fe + 5;  // 3 front end: grid error
    auto update_downsample = [&]() -> int {
        const int r = lom_voxel_downsample_device_nowait(o->update_ds, o->cfg.keyframe_update_voxel_size, d_fx, d_fn, bound,
                                                         d_fe + 1, 12, &in.d_down, &in.d_down_n, &d_nd);
        if (r != LOM_OK) return fail_map(o, r, o->update_ds);
        lom_map_status_words(o->update_ds, &u_range, &u_grid, &seq_u);
        return LOM_OK;
    };
    lom_map *reader = o->update_ds;
    const bool has_keyframe = o->keyframe_has_voxels.load();  // one look; possibly a stale `true` (see the member)
    if (has_keyframe) {
        if ((rc = lom_voxel_downsample_device_nowait(o->matching_ds, o->cfg.keyframe_matching_voxel_size, d_fx, nullptr, bound,
                                                     d_fe + 1, 12, &in.d_match, nullptr, &d_nm)) != LOM_OK)
            return fail_map(o, rc, o->matching_ds);
        lom_map_status_words(o->matching_ds, &m_range, &m_grid, &seq_m);
        ptrs[k++] = d_nm;     // 4
        ptrs[k++] = m_range;  // 5
        ptrs[k++] = m_grid;   // 6
        reader = o->matching_ds;
        if ((rc = lom_map_read_device_words_begin(reader, ptrs, k)) != LOM_OK) return fail_map(o, rc, reader);
        if ((rc = update_downsample()) != LOM_OK) return rc;
        const uint32_t *late[3] = {d_nd, u_range, u_grid};
        if ((rc = lom_map_read_device_words_begin(o->update_ds, late, 3)) != LOM_OK) return fail_map(o, rc, o->update_ds);
        in.pending_update = o->update_ds;
        in.pending_seq = seq_u;
    } else {  // first frame: the keyframe is initialised from the update cloud, there is no align
        if ((rc = update_downsample()) != LOM_OK) return rc;
        ptrs[k++] = d_nd;     // 4
        ptrs[k++] = u_range;  // 5
        ptrs[k++] = u_grid;   // 6
        if ((rc = lom_map_read_device_words_begin(reader, ptrs, k)) != LOM_OK) return fail_map(o, rc, reader);
    }
    // the one wait before the align: counts and verdicts of what it needs
    if ((rc = lom_map_read_device_words_end(reader, got)) != LOM_OK) return fail_map(o, rc, reader);
    tm.lap("stages (device)");
    // the previous frame's keyframe update must be through before this frame touches the keyframe handle.  Its
    // failure is reported here, by the call after the one it belongs to; poses / keyframe stay as they were.
    if ((rc = o->settle()) != LOM_OK) {
        (void)in.collect_update(nullptr);
        return rc;
    }
    tm.lap("settle");
    const uint32_t fe_seq = lom_frontend_sequence(o->frontend);
    const uint32_t seq_ds = has_keyframe ? seq_m : seq_u;
    // An azimuth on a bin boundary, an organised cloud beyond the buffers -- or an in-kernel scan of the front end
    // or of the down-sampler that gave up waiting: such a grid has written nothing and left its tables at rest
    // (grid_scan.hpp), so the frame simply takes the host stages, whose kernels wait for nobody.
    // (LOM_OPT_TEST_FORCE_HOST_REDO: tests take this path on every frame.)
    if (got[2] == fe_seq || got[3] == fe_seq || got[6] == seq_ds || o->test_force_host_redo) {
        if (got[3] == fe_seq || got[6] == seq_ds) o->grid_redos++;
        (void)in.collect_update(nullptr);
        return 1;
    }
    if (got[5] == seq_ds) {
        (void)in.collect_update(nullptr);
        o->error = "coordinate / voxel_size out of range or not finite";
        return LOM_ERR_RANGE;
    }
    cur.planar_points = got[0];
    cur.filtered_points = got[1];
    in.d_fx = d_fx;
    in.d_fn = d_fn;
    in.nf = got[1];
    if (has_keyframe) {
        in.nm = got[4];
    } else {
        in.nd = got[4];
    }
    return LOM_OK;
}

// the update cloud's size and verdict; a down-sampling whose in-kernel scan gave up is redone here from the
// filtered cloud still in HBM (lom_voxel_downsample_device waits for its own verdict and falls back to the
// multi-launch scan by itself)
int collect_or_redo_update(lom_odometry *o, FrameInputs &in, const char **why)
{
    lom_map *ws = in.pending_update;
    int rc = in.collect_update(why);
    if (rc != FrameInputs::kScanGaveUp) return rc;
    o->grid_redos++;
    in.nd = lom_voxel_downsample_device(ws, o->cfg.keyframe_update_voxel_size, in.d_fx, in.d_fn, (size_t)in.nf, 12,
                                        &in.d_down, &in.d_down_n);
    if (in.nd < 0) {
        *why = lom_last_error(ws);
        return (int)in.nd;
    }
    return LOM_OK;
}

// lom_map_set_align_idle_hook: runs on the caller's thread while the align's kernels work -- the hinted next frame goes
// into the front end's pinned buffer (the current frame's copy there has long been read by the device).  Host work only:
// the frame's first kernel (upload + statistics) sent ahead as well was measured and is not (a
// second queue's kernel is not started while the align's queue holds packets, and a copy-engine upload made frames slower)
void send_next_frame_ahead(void *user)
{
    lom_odometry *o = static_cast<lom_odometry *>(user);
    const lom_point_xyzirt *pts = o->hint_now;
    const size_t n = o->hint_n;
    o->hint_now = nullptr;
    if (!pts || !n || n > 170000 || !o->frontend) return;
    StageTimer tm(o->debug_timing);  // ("ahead ..." line: inside the align's lap)
    lom_point_xyzirt *stage = nullptr;
    if (lom_frontend_stage(o->frontend, n, &stage) != LOM_OK) return;
    run_parts(o->pool.get(), n, [&](size_t b, size_t e, unsigned) {
        std::memcpy(static_cast<void *>(stage + b), pts + b, (e - b) * sizeof(lom_point_xyzirt));
    }, 8192);
    tm.lap("ahead copy");
    o->ahead_pts = pts;
    o->ahead_n = n;
    o->ahead_stage = stage;
}

}  // namespace

extern "C" {

int lom_odometry_hint_next(lom_odometry *o, const lom_point_xyzirt *pts, size_t n)
{
    if (!o || (!pts && n)) return LOM_ERR_ARG;
    o->hint_pts = (o->no_send_ahead || !n) ? nullptr : pts;
    o->hint_n = n;
    return LOM_OK;
}

int lom_odometry_process_cloud(lom_odometry *o, const lom_point_xyzirt *pts, size_t n)
{
    if (!o || (!pts && n)) return LOM_ERR_ARG;
    try {
        lom_odometry_frame_stats cur{};  // becomes o->last when the frame is through
        StageTimer tm(o->debug_timing);
        lom_pose relative, rel_inv, ident, guess, result;
        lom_pose_relative_to(&o->previous, &o->current, &relative);  // :27
        // :28 previous_transform_ = current_transform_ -- committed where the frame succeeds (the
        // reference has no error channel; here a frame that fails must leave the state as it found it,
        // or the next frame's constant-velocity guess and deskew would start from a zero motion)
        const lom_pose previous_next = o->current;
        lom::pose_inverse(relative, rel_inv);
        lom_pose_identity(&ident);
        FrameInputs in;
        int rc = 1;
        // a hint is for the call that follows it, what was sent ahead for the call after that: neither outlives its call
        o->hint_now = o->hint_pts;
        o->hint_pts = nullptr;
        struct DropHints {
            lom_odometry *o;
            const lom_point_xyzirt *sent_before;
            ~DropHints()
            {
                o->hint_now = nullptr;
                if (o->ahead_pts == sent_before) o->ahead_pts = nullptr;  // (this call did not use it: the front end drops it)
            }
        } drop_hints{o, o->ahead_pts};
        o->parity ^= 1;
        o->update_ds = o->update_ds2[o->parity];
        if (o->frontend) rc = stages_on_device(o, pts, n, rel_inv, ident, cur, in, tm);
        if (rc == 1) {
            in = FrameInputs();
            rc = stages_on_host(o, pts, n, rel_inv, ident, cur, in, tm);
            cur.host_stages = 1;
        }
        if (rc != LOM_OK) return rc;
        // :40 keyframe_.size() == 0 -- known on the host: the keyframe is empty until a frame has put voxels
        // into it (nd > 0 points always create at least one), and stays non-empty unless a cleanup empties it
        if (!o->keyframe_has_voxels) {  // :40-44 init keyframe
            {   // the stages ran on a stale "has voxels" (the previous update emptied the keyframe meanwhile): the
                // update cloud's count is still on its way
                const char *why = nullptr;
                const int rcu = collect_or_redo_update(o, in, &why);
                if (rcu != LOM_OK) {
                    o->error = why ? why : "keyframe-update down-sampling failed";
                    return rcu;
                }
            }
            if ((rc = lom_map_add_points_device(o->keyframe, in.d_down, in.d_down_n, (size_t)in.nd, 12)) != LOM_OK)
                return fail_map(o, rc, o->keyframe);
            cur.initialised_keyframe = 1;
            cur.update_points = in.nd;
            cur.keyframe_voxels = lom_map_size(o->keyframe);
            o->keyframe_has_voxels = cur.keyframe_voxels > 0;
            o->last = cur;
            o->previous = previous_next;  // :28
            return LOM_OK;
        }
        cur.matching_points = in.nm;
        lom_pose_compose(&o->current, &relative, &guess);  // :51
        lom_align_stats ast;
        // :65-67: the keyframe update below starts with radiusCleanup(current_transform_.translation): its scan may run
        // right behind the align, on the align's own result
        if (!o->no_cleanup_behind_align) (void)lom_map_radius_cleanup_after_align(o->keyframe, o->cfg.keyframe_cleanup_range);
        // ... and the frame the caller has announced (lom_odometry_hint_next) is sent ahead while this thread would only
        // watch the align's report
        if (o->hint_now && o->frontend && o->temp_on_device) (void)lom_map_set_align_idle_hook(o->keyframe, send_next_frame_ahead, o);
        if ((rc = lom_match_align_device(o->keyframe, in.d_match, (size_t)in.nm, 12, guess.t, guess.q, result.t, result.q,
                                         &ast)) != LOM_OK) {  // :49-51
            (void)in.collect_update(nullptr);
            return fail_map(o, rc, o->keyframe);
        }
        {   // the update cloud was down-sampled beside the align: its size and verdict (long since on the host)
            const char *why = nullptr;
            const int rcu = collect_or_redo_update(o, in, &why);
            if (rcu != LOM_OK) {
                o->error = why ? why : "keyframe-update down-sampling failed";
                return rcu;
            }
        }
        cur.update_points = in.nd;
        cur.outer_iterations = ast.outer_iterations;
        cur.queries = ast.queries;
        o->queries_total += ast.queries;
        cur.queries_total = o->queries_total;
        tm.lap("align");
        {  // :53-63 divergence guard
            float ang[3];
            delta_euler_deg(result.q, o->current.q, ang);
            const float thr = o->cfg.angular_divergence_threshold;
            bool ok = true;
            for (int a = 0; a < 3; a++) ok = ok && (std::fabs(ang[a]) < thr || std::fabs(ang[a]) > 180 - thr);
            if (!ok) {
                result = guess;  // :61
                cur.unstable_rotation = 1;
            }
        }
        o->previous = previous_next;                                                                  // :28
        o->current = result;                                                                          // :65
        o->last = cur;
        // keyframe update (:67-70): same calls in the same order, on the helper thread when there is one
        const lom_pose pose_now = o->current;
        const size_t n_down = (size_t)in.nd;
        const float *d_down = in.d_down, *d_down_n = in.d_down_n;
        const double t_submit = o->debug_timing ? StageTimer::now() : 0.0;
        auto update = [o, pose_now, d_down, d_down_n, n_down, t_submit]() -> int {
            auto bad = [o](int rc, lom_map *m) {
                o->deferred_error = lom_last_error(m);
                return rc;
            };
            StageTimer ut(o->debug_timing);  // (the helper thread's own laps: "upd ..." lines)
            if (o->debug_timing) std::fprintf(stderr, "  %-14s %8.1f us\n", "upd hand-off", (ut.t0 - t_submit) * 1e6);
            int rc;
            // :69 first: the rigid transform of the update cloud reads neither the map nor what the cleanup leaves, and its
            // launch fills the time the cleanup spends waiting for its scan (enqueued behind the align) to report
            const float *d_upd = nullptr, *d_upd_n = nullptr;
            if ((rc = lom_transform_points_device(o->keyframe, &pose_now, d_down, d_down_n, n_down, 12, &d_upd,
                                                  &d_upd_n)) != LOM_OK)
                return bad(rc, o->keyframe);
            if ((rc = lom_map_radius_cleanup(o->keyframe, pose_now.t, o->cfg.keyframe_cleanup_range)) != LOM_OK)  // :67
                return bad(rc, o->keyframe);
            ut.lap("upd cleanup");
            if ((rc = lom_map_add_points_device_nowait(o->keyframe, d_upd, d_upd_n, n_down, 12)) != LOM_OK)  // :70
                return bad(rc, o->keyframe);
            ut.lap("upd enqueue");
            // one look at the host per update: the deferred verdict of the insert and the voxel count
            if ((rc = lom_map_status(o->keyframe)) != LOM_OK) return bad(rc, o->keyframe);
            ut.lap("upd status");
            o->last.keyframe_voxels = lom_map_size(o->keyframe);
            o->keyframe_has_voxels = o->last.keyframe_voxels > 0;
            return LOM_OK;
        };
        if (o->deferred) {
            o->deferred->submit(update);
        } else if ((rc = update()) != LOM_OK) {
            o->error = o->deferred_error;
            return rc;
        }
        tm.lap("keyframe update");
        tm.total();
        return LOM_OK;
    } catch (const std::bad_alloc &) {
        o->error = "host allocation failed";
        return LOM_ERR_OOM;
    }
}

int lom_odometry_process_sequence(lom_odometry *o, const lom_point_xyzirt *const *frames, const size_t *n, size_t count,
                                  size_t *done)
{
    if (done) *done = 0;
    if (!o || (count && (!frames || !n))) return LOM_ERR_ARG;
    for (size_t i = 0; i < count; i++) {
        if (i + 1 < count) (void)lom_odometry_hint_next(o, frames[i + 1], n[i + 1]);
        const int rc = lom_odometry_process_cloud(o, frames[i], n[i]);
        if (rc != LOM_OK) return rc;
        if (done) *done = i + 1;
    }
    return LOM_OK;
}

}  // extern "C"
