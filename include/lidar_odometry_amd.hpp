// lidar_odometry_amd.hpp -- header-only C++ mirror of the reference's hot-path
// classes over the C ABI of lidar_odometry_amd.h.
//
// Same class and method names, argument meaning and return shapes as
//   Pose3D        reference src/pose_3d.h:10-59
//   VoxelGrid     reference src/voxel_grid.h:17-257
//   CloudMatcher  reference src/cloud_matcher.h:13-17
//   CloudTransformer, CloudClassifier, utils::pointTimeNormalize, utils::rangeFilter  reference src/utils/*.h
//   LidarOdometry reference src/lidar_odometry.h:20-85
// so that reference src/lidar_odometry.cpp compiles against it with a type
// alias or two.  No Eigen / PCL / Ceres / robin_map needed:
// the point structs below have the memory layout of pcl::PointXYZ (16 bytes)
// and pcl::PointNormal (48 bytes, normal at byte 16), and any cloud type whose
// `.points` is a contiguous array of such structs can be passed as is.
//
// Errors: the reference has no error channel; here a failing call throws
// lom::Error (status code + lom_last_error text).  Without a gfx950 device the
// VoxelGrid constructor throws -- there is no CPU fallback.
#pragma once
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "lidar_odometry_amd.h"

namespace lom {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

struct Vector3f {
    float v[3] = {0.f, 0.f, 0.f};
    Vector3f() = default;
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float operator()(int i) const { return v[i]; }
    float norm() const { return std::sqrt(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])); }
};

struct Quaternionf {
    float q[4] = {1.f, 0.f, 0.f, 0.f};  // w, x, y, z
    Quaternionf() = default;
    Quaternionf(float w, float x, float y, float z) : q{w, x, y, z} {}
    float w() const { return q[0]; }
    float x() const { return q[1]; }
    float y() const { return q[2]; }
    float z() const { return q[3]; }
    float dot(const Quaternionf &o) const { return q[0] * o.q[0] + q[1] * o.q[1] + q[2] * o.q[2] + q[3] * o.q[3]; }
    static Quaternionf Identity() { return {}; }
};

// pcl::PointXYZ layout (16 bytes)
struct alignas(16) PointXYZ {
    float x = 0.f, y = 0.f, z = 0.f, pad = 1.f;
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
// pcl::PointNormal layout (48 bytes): xyz+pad, normal+pad, curvature+pad
struct alignas(16) PointNormal {
    float x = 0.f, y = 0.f, z = 0.f, pad0 = 1.f;
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, pad1 = 0.f;
    float curvature = 0.f, pad2[3] = {0.f, 0.f, 0.f};
    PointNormal() = default;
    PointNormal(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointNormal) == 48, "PCL layouts");

template <typename PointT>
struct PointCloud {
    std::vector<PointT> points;
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    size_t size() const { return points.size(); }
    const PointT &at(size_t i) const { return points.at(i); }
};

// ---- Pose3D (src/pose_3d.h) ---------------------------------------------------
class Pose3D {
public:
    Vector3f translation;
    Quaternionf rotation;

    Pose3D() = default;
    Pose3D(const Vector3f &t, const Quaternionf &q) : translation(t), rotation(q) {}

    Pose3D relativeTo(const Pose3D &target) const
    {
        lom_pose a = c(), b = target.c(), o;
        lom_pose_relative_to(&a, &b, &o);
        return from(o);
    }
    Pose3D compose(const Pose3D &another) const
    {
        lom_pose a = c(), b = another.c(), o;
        lom_pose_compose(&a, &b, &o);
        return from(o);
    }
    Pose3D inverse() const
    {
        lom_pose a = c(), o;
        lom_pose_inverse(&a, &o);
        return from(o);
    }
    // row-major 3x3
    void rotationMatrix(float R[9]) const
    {
        lom_pose a = c();
        lom_pose_rotation_matrix(&a, R);
    }

    lom_pose c() const
    {
        lom_pose p;
        for (int i = 0; i < 3; i++) p.t[i] = translation.v[i];
        for (int i = 0; i < 4; i++) p.q[i] = rotation.q[i];
        return p;
    }
    static Pose3D from(const lom_pose &p)
    {
        return {Vector3f(p.t[0], p.t[1], p.t[2]), Quaternionf(p.q[0], p.q[1], p.q[2], p.q[3])};
    }
};

// ---- VoxelGrid (src/voxel_grid.h) -----------------------------------------------
class VoxelGrid {
public:
    struct Correspondence {  // voxel_grid.h:40-46 (f64 like the reference)
        double source_point_local[3] = {0, 0, 0};
        double plane_origin[3] = {0, 0, 0};
        double plane_normal[3] = {0, 0, 0};
        bool valid = false;
    };

    VoxelGrid() : VoxelGrid(0.5f, 10) {}  // voxel_grid.h:253-254 defaults
    VoxelGrid(float voxel_size, size_t max_points, int device = 0)
    {
        const int rc = lom_map_create(voxel_size, max_points, 0, device, &h_);
        if (rc != LOM_OK) throw Error(rc, std::string("lom_map_create: ") + lom_last_error(nullptr));
    }
    ~VoxelGrid()
    {
        for (lom_scan *s : scans_) lom_scan_destroy(s);  // contexts go before their map
        lom_map_destroy(h_);
    }
    VoxelGrid(const VoxelGrid &) = delete;
    VoxelGrid &operator=(const VoxelGrid &) = delete;
    VoxelGrid(VoxelGrid &&o) noexcept : h_(o.h_), id_(o.id_), scans_(std::move(o.scans_))
    {
        o.h_ = nullptr;
        o.scans_.clear();
    }

    // The const members of the reference (getCorrespondence, findMatchingPairs, and CloudMatcher::align, which takes
    // `const VoxelGrid&`) may be called from several threads at once.  A map handle is single-caller, so every thread
    // gets a scan context of its own for this grid (lom_scan_create: stream, per-scan buffers, solve state), created
    // on its first such call and owned by the grid.  As in the reference, nobody may change the grid meanwhile.
    lom_scan *scan_context() const
    {
        thread_local std::unordered_map<uint64_t, lom_scan *> mine;  // this thread's contexts, by grid id
        auto it = mine.find(id_);
        if (it != mine.end()) return it->second;
        lom_scan *s = nullptr;
        const int rc = lom_scan_create(h_, &s);
        if (rc != LOM_OK) throw Error(rc, lom_last_error(h_));
        {
            std::lock_guard<std::mutex> lock(scans_mutex_);
            scans_.push_back(s);
        }
        mine.emplace(id_, s);
        return s;
    }

    void setMaxPoints(size_t max_points) { check(lom_map_set_max_points(h_, max_points)); }
    void setVoxelSize(float voxel_size) { check(lom_map_clear(h_, voxel_size)); }

    void addCloud(const PointCloud<PointNormal> &cloud)
    {
        if (cloud.points.empty()) return;
        const PointNormal *p = cloud.points.data();
        check(lom_map_add_points(h_, &p->x, &p->normal_x, cloud.points.size(), sizeof(PointNormal)));
    }
    void addCloudWithoutNormals(const PointCloud<PointXYZ> &cloud)
    {
        if (cloud.points.empty()) return;
        check(lom_map_add_points(h_, &cloud.points.data()->x, nullptr, cloud.points.size(), sizeof(PointXYZ)));
    }

    PointCloud<PointNormal>::Ptr getCloud() const
    {
        auto out = std::make_shared<PointCloud<PointNormal>>();
        std::vector<float> xyz, nrm;
        const size_t n = fetch(LOM_EXPORT_FULL, xyz, &nrm);
        out->points.resize(n);
        for (size_t i = 0; i < n; i++) {
            PointNormal &p = out->points[i];
            p.x = xyz[3 * i], p.y = xyz[3 * i + 1], p.z = xyz[3 * i + 2];
            p.normal_x = nrm[3 * i], p.normal_y = nrm[3 * i + 1], p.normal_z = nrm[3 * i + 2];
        }
        return out;
    }
    PointCloud<PointXYZ>::Ptr getCloudWithoutNormals() const { return xyz_cloud(LOM_EXPORT_FULL_NO_NORMALS); }
    PointCloud<PointXYZ>::Ptr getSparseCloudWithoutNormals() const { return xyz_cloud(LOM_EXPORT_FIRST_PER_VOXEL); }

    // voxel_grid.h:164-204, query already in the map frame
    Correspondence getCorrespondence(const Vector3f &query, double max_correspondence_distance_sq) const
    {
        PointCloud<PointXYZ> one;
        one.points.emplace_back(query.x(), query.y(), query.z());
        auto v = findAll(one, Pose3D(), 0.f, &max_correspondence_distance_sq);  // the squared threshold as it is (:164)
        Correspondence c = v[0];
        for (double &d : c.source_point_local) d = 0.0;  // the reference leaves it unset here
        return c;
    }

    // voxel_grid.h:206-234: valid correspondences only, in source order (the reference's
    // order under its mutex is unspecified)
    std::vector<Correspondence> findMatchingPairs(const PointCloud<PointXYZ> &cloud, const Pose3D &transform,
                                                  float max_correspondence_distance) const
    {
        std::vector<Correspondence> all = findAll(cloud, transform, max_correspondence_distance), out;
        out.reserve(all.size());
        for (const auto &c : all)
            if (c.valid) out.push_back(c);
        return out;
    }

    void radiusCleanup(const Vector3f &point, float radius) { check(lom_map_radius_cleanup(h_, point.v, radius)); }
    // not in the reference: says that the next align on this grid is followed by radiusCleanup(<its result translation>,
    // radius), as lidar_odometry.cpp:65-67 does -- the cleanup's scan then runs right behind the align (results never differ)
    void radiusCleanupAfterAlign(float radius) { check(lom_map_radius_cleanup_after_align(h_, radius)); }

    size_t size() const
    {
        const int64_t n = lom_map_size(h_);
        if (n < 0) throw Error((int)n, "lom_map_size");
        return (size_t)n;
    }

    lom_map *handle() const { return h_; }

private:
    void check(int rc) const
    {
        if (rc < 0) throw Error(rc, lom_last_error(h_));
    }
    size_t fetch(int mode, std::vector<float> &xyz, std::vector<float> *nrm) const
    {
        const int64_t n = lom_map_export(h_, mode, nullptr, nullptr, 0);
        if (n < 0) throw Error((int)n, lom_last_error(h_));
        xyz.resize((size_t)n * 3);
        if (nrm) nrm->resize((size_t)n * 3);
        if (n) {
            const int64_t m = lom_map_export(h_, mode, xyz.data(), nrm ? nrm->data() : nullptr, (size_t)n);
            if (m < 0) throw Error((int)m, lom_last_error(h_));
        }
        return (size_t)n;
    }
    PointCloud<PointXYZ>::Ptr xyz_cloud(int mode) const
    {
        auto out = std::make_shared<PointCloud<PointXYZ>>();
        std::vector<float> xyz;
        const size_t n = fetch(mode, xyz, nullptr);
        out->points.resize(n);
        for (size_t i = 0; i < n; i++) out->points[i] = PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        return out;
    }
    std::vector<Correspondence> findAll(const PointCloud<PointXYZ> &cloud, const Pose3D &transform, float max_dist,
                                        const double *max_dist_sq = nullptr) const
    {
        std::vector<lom_correspondence> raw(cloud.points.size());
        std::vector<Correspondence> out(cloud.points.size());
        if (cloud.points.empty()) return out;
        const lom_pose p = transform.c();
        lom_scan *ctx = scan_context();
        const int64_t rc = max_dist_sq ? lom_scan_find_pairs_sq(ctx, &cloud.points.data()->x, cloud.points.size(),
                                                                sizeof(PointXYZ), p.t, p.q, *max_dist_sq, raw.data())
                                       : lom_scan_find_pairs(ctx, &cloud.points.data()->x, cloud.points.size(),
                                                             sizeof(PointXYZ), p.t, p.q, max_dist, raw.data());
        if (rc < 0) throw Error((int)rc, lom_scan_last_error(ctx));
        for (size_t i = 0; i < raw.size(); i++) {
            Correspondence &c = out[i];
            c.valid = raw[i].index >= 0;
            c.source_point_local[0] = cloud.points[i].x;
            c.source_point_local[1] = cloud.points[i].y;
            c.source_point_local[2] = cloud.points[i].z;
            for (int a = 0; a < 3; a++) {
                c.plane_origin[a] = raw[i].origin[a];
                c.plane_normal[a] = raw[i].normal[a];
            }
        }
        return out;
    }

    static uint64_t next_id()
    {
        static std::atomic<uint64_t> n{1};
        return n.fetch_add(1);
    }
    lom_map *h_ = nullptr;
    uint64_t id_ = next_id();  // never reused: a thread's cached context of a destroyed grid is never looked up again
    mutable std::mutex scans_mutex_;
    mutable std::vector<lom_scan *> scans_;
};

// ---- CloudMatcher (src/cloud_matcher.h) --------------------------------------------
class CloudMatcher {
public:
    Pose3D align(const VoxelGrid &keyframe, const PointCloud<PointXYZ> &planar_cloud, const Pose3D &position_guess)
    {
        const lom_pose g = position_guess.c();
        lom_pose o;
        const float *src = planar_cloud.points.empty() ? nullptr : &planar_cloud.points.data()->x;
        // stateless like the reference's (lidar_odometry.cpp:49 constructs one per frame): the solve state lives in the
        // calling thread's scan context of this keyframe, so several threads may align against one grid at a time
        lom_scan *ctx = keyframe.scan_context();
        const int rc = lom_scan_align(ctx, src, planar_cloud.points.size(), sizeof(PointXYZ), g.t, g.q, o.t, o.q, &last_stats);
        if (rc != LOM_OK) throw Error(rc, lom_scan_last_error(ctx));
        return Pose3D::from(o);
    }
    lom_align_stats last_stats{};
};

// ---- CloudTransformer::transform / transformWithNormals (src/utils/cloud_transform.h:43-97)
struct CloudTransformer {
    static PointCloud<PointXYZ>::Ptr transform(const PointCloud<PointXYZ> &input, const Pose3D &pose)
    {
        auto out = std::make_shared<PointCloud<PointXYZ>>();
        out->points = input.points;
        if (input.points.empty()) return out;
        const lom_pose p = pose.c();
        lom_transform_points(&p, &input.points.data()->x, nullptr, input.points.size(), sizeof(PointXYZ),
                             &out->points.data()->x, nullptr, sizeof(PointXYZ));
        return out;
    }
    // transformNonRigid (cloud_transform.h:15-40): the deskew, on the host (lom::LidarOdometry runs it on the device)
    static PointCloud<lom_point_xyzirt>::Ptr transformNonRigid(const PointCloud<lom_point_xyzirt> &input,
                                                              const Pose3D &start_pose, const Pose3D &end_pose)
    {
        auto out = std::make_shared<PointCloud<lom_point_xyzirt>>();
        out->points.resize(input.points.size());
        const lom_pose s = start_pose.c(), e = end_pose.c();
        lom_transform_non_rigid(input.points.data(), input.points.size(), &s, &e, out->points.data());
        return out;
    }
    static PointCloud<PointNormal>::Ptr transformWithNormals(const PointCloud<PointNormal> &input, const Pose3D &pose)
    {
        auto out = std::make_shared<PointCloud<PointNormal>>();
        out->points = input.points;
        if (input.points.empty()) return out;
        const lom_pose p = pose.c();
        lom_transform_points(&p, &input.points.data()->x, &input.points.data()->normal_x, input.points.size(),
                             sizeof(PointNormal), &out->points.data()->x, &out->points.data()->normal_x,
                             sizeof(PointNormal));
        return out;
    }
};

// ---- the stages processCloud runs before the align (src/lidar_odometry.cpp:25-35), host functions of the library
// with the reference's names: a caller that keeps the reference's own orchestration but not PCL uses these;
// lom::LidarOdometry below runs the same stages on the device.
using PointXYZIRT = lom_point_xyzirt;  // lidar_point::PointXYZIRT (src/lidar_point_type.h:13-31), same 32-byte layout

namespace utils {
// utils::pointTimeNormalize (src/utils/point_time_normalize.h:15-39)
inline PointCloud<PointXYZIRT>::Ptr pointTimeNormalize(const PointCloud<PointXYZIRT> &input)
{
    auto out = std::make_shared<PointCloud<PointXYZIRT>>();
    out->points.resize(input.points.size());
    lom_point_time_normalize(input.points.data(), input.points.size(), out->points.data());
    return out;
}
// utils::rangeFilter (src/utils/range_filter.h:13-28) for the two point types the reference's data flow holds
inline PointCloud<PointNormal>::Ptr rangeFilter(const PointCloud<PointNormal> &input, float min_range, float max_range)
{
    const size_t n = input.points.size();
    std::vector<float> xyz(3 * n + 3), nrm(3 * n + 3), fx(3 * n + 3), fn(3 * n + 3);
    for (size_t i = 0; i < n; i++) {
        const PointNormal &p = input.points[i];
        xyz[3 * i] = p.x, xyz[3 * i + 1] = p.y, xyz[3 * i + 2] = p.z;
        nrm[3 * i] = p.normal_x, nrm[3 * i + 1] = p.normal_y, nrm[3 * i + 2] = p.normal_z;
    }
    const size_t m = lom_range_filter(xyz.data(), nrm.data(), n, min_range, max_range, fx.data(), fn.data());
    auto out = std::make_shared<PointCloud<PointNormal>>();
    out->points.resize(m);
    for (size_t i = 0; i < m; i++) {
        PointNormal &p = out->points[i];
        p.x = fx[3 * i], p.y = fx[3 * i + 1], p.z = fx[3 * i + 2];
        p.normal_x = fn[3 * i], p.normal_y = fn[3 * i + 1], p.normal_z = fn[3 * i + 2];
    }
    return out;
}
inline PointCloud<PointXYZ>::Ptr rangeFilter(const PointCloud<PointXYZ> &input, float min_range, float max_range)
{
    const size_t n = input.points.size();
    std::vector<float> xyz(3 * n + 3), fx(3 * n + 3);
    for (size_t i = 0; i < n; i++) xyz[3 * i] = input.points[i].x, xyz[3 * i + 1] = input.points[i].y, xyz[3 * i + 2] = input.points[i].z;
    const size_t m = lom_range_filter(xyz.data(), nullptr, n, min_range, max_range, fx.data(), nullptr);
    auto out = std::make_shared<PointCloud<PointXYZ>>();
    out->points.reserve(m);
    for (size_t i = 0; i < m; i++) out->points.emplace_back(fx[3 * i], fx[3 * i + 1], fx[3 * i + 2]);
    return out;
}
}  // namespace utils

// CloudClassifier::classify (src/utils/cloud_classifier.h:19-168): {planar points with normals, unclassified points}.
// The second cloud is returned with the reference's SIZE only (default points): its one caller drops it
// (src/lidar_odometry.cpp:33), and the library does not build it.
struct CloudClassifier {
    static std::pair<PointCloud<PointNormal>::Ptr, PointCloud<PointXYZIRT>::Ptr> classify(const PointCloud<PointXYZIRT> &input)
    {
        const size_t n = input.points.size();
        std::vector<float> xyz(3 * n + 3), nrm(3 * n + 3);
        size_t unclassified = 0;
        const size_t m = lom_cloud_classify(input.points.data(), n, xyz.data(), nrm.data(), &unclassified, nullptr);
        auto planar = std::make_shared<PointCloud<PointNormal>>();
        planar->points.resize(m);
        for (size_t i = 0; i < m; i++) {
            PointNormal &p = planar->points[i];
            p.x = xyz[3 * i], p.y = xyz[3 * i + 1], p.z = xyz[3 * i + 2];
            p.normal_x = nrm[3 * i], p.normal_y = nrm[3 * i + 1], p.normal_z = nrm[3 * i + 2];
        }
        auto rest = std::make_shared<PointCloud<PointXYZIRT>>();
        rest->points.resize(unclassified);
        return {planar, rest};
    }
};

// ---- LidarOdometry (src/lidar_odometry.h:20-85) --------------------------------------
// For callers that do not keep the reference's own orchestration: processCloud, getCurrentPose and the
// two key-frame exporters over lom_odometry_*.  lidar_point::PointXYZIRT (src/lidar_point_type.h:13-31)
// has the layout of lom_point_xyzirt, so a PCL cloud's points can be passed as they are.
class LidarOdometry {
public:
    struct Params {  // lidar_odometry.h:23-48, defaults of GetROSDeclaration()
        float lidar_min_range = 4.0f;
        float lidar_max_range = 80.0f;
        float keyframe_voxel_size = 0.2f;
        size_t keyframe_max_points_cnt = 20;
        float keyframe_matching_voxel_size = 0.3f;
        float keyframe_update_voxel_size = 0.1f;
        float keyframe_cleanup_range = 80.0f;
        float angular_divergence_threshold = 5.0f;
    };
    using CloudType = PointCloud<lom_point_xyzirt>;

    explicit LidarOdometry(const Params &config, int device = 0)
    {
        lom_odometry_params p;
        p.lidar_min_range = config.lidar_min_range;
        p.lidar_max_range = config.lidar_max_range;
        p.keyframe_voxel_size = config.keyframe_voxel_size;
        p.keyframe_max_points_cnt = (uint32_t)config.keyframe_max_points_cnt;
        p.keyframe_matching_voxel_size = config.keyframe_matching_voxel_size;
        p.keyframe_update_voxel_size = config.keyframe_update_voxel_size;
        p.keyframe_cleanup_range = config.keyframe_cleanup_range;
        p.angular_divergence_threshold = config.angular_divergence_threshold;
        const int rc = lom_odometry_create(&p, device, &h_);
        if (rc != LOM_OK) throw Error(rc, lom_last_error(nullptr));
    }
    ~LidarOdometry() { lom_odometry_destroy(h_); }
    LidarOdometry(const LidarOdometry &) = delete;
    LidarOdometry &operator=(const LidarOdometry &) = delete;

    void processCloud(const CloudType &input_cloud)  // lidar_odometry.cpp:22-77
    {
        const int rc = lom_odometry_process_cloud(h_, input_cloud.points.data(), input_cloud.points.size());
        if (rc != LOM_OK) throw Error(rc, lom_odometry_last_error(h_));
    }
    // not in the reference: the cloud that will come after the next processCloud, for callers that hold it already; it must
    // stay unchanged until it has been processed (it is copied to pinned memory while that processCloud's align runs)
    void hintNextCloud(const CloudType &next_cloud)
    {
        const int rc = lom_odometry_hint_next(h_, next_cloud.points.data(), next_cloud.points.size());
        if (rc != LOM_OK) throw Error(rc, lom_odometry_last_error(h_));
    }
    Pose3D getCurrentPose() const  // :87-89
    {
        lom_pose p;
        lom_odometry_get_pose(h_, &p);
        return Pose3D::from(p);
    }
    PointCloud<PointXYZ>::Ptr getKeyFrameCloud() const { return export_(LOM_EXPORT_FIRST_PER_VOXEL); }      // :79-81
    PointCloud<PointXYZ>::Ptr getFullKeyFrameCloud() const { return export_(LOM_EXPORT_FULL_NO_NORMALS); }  // :83-85
    // lidar_odometry.h:73-75; null before the first frame, like the reference's unset shared_ptr
    CloudType::Ptr getTempCloud() const
    {
        const int64_t n = lom_odometry_get_temp_cloud(h_, nullptr, 0);
        if (n < 0) throw Error((int)n, lom_odometry_last_error(h_));
        if (n == 0) return nullptr;
        auto out = std::make_shared<CloudType>();
        out->points.resize((size_t)n);
        lom_odometry_get_temp_cloud(h_, out->points.data(), (size_t)n);
        return out;
    }
    lom_odometry_frame_stats lastFrameStats() const
    {
        lom_odometry_frame_stats s;
        lom_odometry_get_stats(h_, &s);
        return s;
    }

private:
    PointCloud<PointXYZ>::Ptr export_(int mode) const
    {
        lom_map *kf = lom_odometry_keyframe(h_);
        auto out = std::make_shared<PointCloud<PointXYZ>>();
        const int64_t n = lom_map_export(kf, mode, nullptr, nullptr, 0);
        if (n < 0) throw Error((int)n, lom_last_error(kf));
        std::vector<float> xyz((size_t)n * 3 + 3);
        const int64_t m = lom_map_export(kf, mode, xyz.data(), nullptr, (size_t)n);
        if (m < 0) throw Error((int)m, lom_last_error(kf));
        out->points.resize((size_t)n);
        for (size_t i = 0; i < (size_t)n; i++) {
            out->points[i].x = xyz[3 * i];
            out->points[i].y = xyz[3 * i + 1];
            out->points[i].z = xyz[3 * i + 2];
        }
        return out;
    }
    lom_odometry *h_ = nullptr;
};

}  // namespace lom
