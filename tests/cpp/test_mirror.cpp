// The reference's gtest cases (test/test.cpp) restated against the header-only mirror
// include/lidar_odometry_amd.hpp -- same class and method names as the reference, so
// this file reads like test/test.cpp with `lom::` types in place of Eigen/PCL ones.
// Built and run by tests/test_cpp_mirror.py on the GPU box (g++, links the C-ABI .so).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <thread>

#include "lidar_odometry_amd.hpp"

using namespace lom;

static int g_fail = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            g_fail++;                                                      \
        }                                                                  \
    } while (0)

static Quaternionf angleAxis(float angle, float ax, float ay, float az)
{
    const float ha = 0.5f * angle, s = std::sin(ha);
    return {std::cos(ha), s * ax, s * ay, s * az};
}

static void VoxelGrid_UniquePoints()  // test.cpp:26-55
{
    PointCloud<PointNormal> input_cloud;
    const float pts[7][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
    for (auto &p : pts) input_cloud.points.emplace_back(p[0], p[1], p[2]);
    VoxelGrid voxel_grid(0.5, 1);
    voxel_grid.addCloud(input_cloud);
    EXPECT(voxel_grid.size() == input_cloud.size());
    auto output_cloud = voxel_grid.getCloud();
    EXPECT(input_cloud.size() == output_cloud->size());
    for (const auto &o : output_cloud->points) {
        auto it = std::find_if(input_cloud.points.begin(), input_cloud.points.end(),
                               [&o](const PointNormal &i) { return o.x == i.x && o.y == i.y && o.z == i.z; });
        EXPECT(it != input_cloud.points.end());
        if (it != input_cloud.points.end()) input_cloud.points.erase(it);
    }
}

static void VoxelGrid_DuplicatePoints()  // test.cpp:57-75
{
    PointCloud<PointNormal> input_cloud;
    input_cloud.points.emplace_back(0, 0, 0);
    input_cloud.points.emplace_back(1, 0, 0);
    input_cloud.points.emplace_back(0, 0, 0);
    input_cloud.points.emplace_back(1, 0, 0);
    VoxelGrid voxel_grid(0.5, 1);
    voxel_grid.addCloud(input_cloud);
    EXPECT(voxel_grid.size() == size_t(2));
    auto output_cloud = voxel_grid.getCloud();
    EXPECT(output_cloud->size() == size_t(2));
    EXPECT(!(output_cloud->points.at(0).x == output_cloud->points.at(1).x &&
             output_cloud->points.at(0).y == output_cloud->points.at(1).y &&
             output_cloud->points.at(0).z == output_cloud->points.at(1).z));
}

static void Pose3D_ComposeRelativeInverse()  // test.cpp:77-149 (group identities)
{
    const Pose3D a({1, 0.5f, -0.5f}, angleAxis(0.456f, 0.0976f, 0.1952f, 0.9759f));
    const Pose3D b({-1, -0.6f, 0}, angleAxis(-0.245f, -1, 0, 0));
    const Pose3D id = a.compose(a.inverse());
    EXPECT(std::fabs(id.translation.norm()) < 1e-5f);
    EXPECT(std::fabs(std::fabs(id.rotation.dot(Quaternionf::Identity())) - 1.f) < 1e-6f);
    const Pose3D rel = a.relativeTo(b);  // a^-1 * b
    const Pose3D back = a.compose(rel);
    EXPECT(std::fabs(back.translation.x() - b.translation.x()) < 1e-5f);
    EXPECT(std::fabs(back.translation.y() - b.translation.y()) < 1e-5f);
    EXPECT(std::fabs(back.translation.z() - b.translation.z()) < 1e-5f);
    EXPECT(std::fabs(std::fabs(back.rotation.dot(b.rotation)) - 1.f) < 1e-6f);
}

static void CloudTransformer_RigidTransform()  // test.cpp:151-189
{
    PointCloud<PointXYZ> sample_cloud;
    const float pts[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    for (auto &p : pts) sample_cloud.points.emplace_back(p[0], p[1], p[2]);
    const float d = 3.14159265358979f / 180.f;
    const Pose3D pose({1, 1, 1}, angleAxis(45.f * d, 0, 0.70710678f, 0.70710678f));
    auto transformed = CloudTransformer::transform(sample_cloud, pose);
    float R[9];
    pose.rotationMatrix(R);
    for (size_t i = 0; i < sample_cloud.size(); i++) {
        const PointXYZ &p = sample_cloud.points[i];
        const float ex = R[0] * p.x + R[1] * p.y + R[2] * p.z + 1.f;
        const float ey = R[3] * p.x + R[4] * p.y + R[5] * p.z + 1.f;
        const float ez = R[6] * p.x + R[7] * p.y + R[8] * p.z + 1.f;
        EXPECT(std::fabs(transformed->at(i).x - ex) < 1e-6f && std::fabs(transformed->at(i).y - ey) < 1e-6f &&
               std::fabs(transformed->at(i).z - ez) < 1e-6f);
    }
}

static void CloudMatcher_MatchingTest()  // protocol of test.cpp:226-262 on a synthetic corner scene
{
    // three mutually orthogonal 25 m walls, 60k pseudo-random samples each, exact normals
    // (enough points that the data term outweighs the translation prior, cloud_matcher.cpp:153)
    auto full_cloud_with_normals = std::make_shared<PointCloud<PointNormal>>();
    PointCloud<PointXYZ> full_cloud;
    auto add = [&](float x, float y, float z, float nx, float ny, float nz) {
        PointNormal p(x, y, z);
        p.normal_x = nx, p.normal_y = ny, p.normal_z = nz;
        full_cloud_with_normals->points.push_back(p);
        full_cloud.points.emplace_back(x, y, z);
    };
    uint32_t lcg = 12345u;
    auto rnd = [&lcg]() {
        lcg = lcg * 1664525u + 1013904223u;
        return (float)((lcg >> 8) / 16777216.0);
    };
    for (int w = 0; w < 3; w++)
        for (int k = 0; k < 60000; k++) {
            const float u = 1.f + 25.f * rnd(), v = 1.f + 25.f * rnd();
            if (w == 0) add(u, v, 1.f, 0, 0, 1);
            if (w == 1) add(u, 1.f, v, 0, 1, 0);
            if (w == 2) add(1.f, u, v, 1, 0, 0);
        }
    VoxelGrid keyframe(0.25, 20);
    keyframe.addCloud(*full_cloud_with_normals);
    VoxelGrid voxel_filter(0.5, 1);
    voxel_filter.addCloudWithoutNormals(full_cloud);
    auto subsampled_cloud = voxel_filter.getCloudWithoutNormals();
    EXPECT(subsampled_cloud->size() > 3000);

    CloudMatcher matcher;
    const float d = 3.14159265358979f / 180.f;
    const std::vector<Pose3D> guess_poses{
        Pose3D({0.0f, 0.0f, 0.0f}, Quaternionf::Identity()),
        Pose3D({0.0f, 0.0f, 0.1f}, Quaternionf::Identity()),
        Pose3D({0.1f, 0.1f, 0.1f}, Quaternionf::Identity()),
        Pose3D({-0.1f, -0.1f, -0.1f}, Quaternionf::Identity()),
        Pose3D({0.1f, -0.1f, 0.f}, Quaternionf::Identity()),
        Pose3D({0.0f, 0.0f, 0.0f}, angleAxis(-1.0f * d, 0, 0, 1)),
        Pose3D({-0.2f, 0.0f, 0.0f}, angleAxis(2.0f * d, 0, 0, 1)),
    };
    for (const auto &guess_pose : guess_poses) {
        auto guess_cloud = CloudTransformer::transform(*subsampled_cloud, guess_pose.inverse());
        auto final_transform = matcher.align(keyframe, *guess_cloud, Pose3D());
        auto error = final_transform.relativeTo(guess_pose);
        const double rotation_error = 1.0 - std::fabs(final_transform.rotation.dot(guess_pose.rotation));
        EXPECT(error.translation.norm() < 0.05);  // test.cpp:261
        EXPECT(rotation_error < 0.01);            // test.cpp:262
        EXPECT(matcher.last_stats.outer_iterations >= 5 && matcher.last_stats.outer_iterations <= 35);
    }
    // `const VoxelGrid&` (cloud_matcher.h:15-16): the seven guesses again, one thread each, all at once against the one
    // keyframe -- every thread must get the pose the serial loop above got for its guess, bit for bit
    {
        std::vector<Pose3D> serial, parallel(guess_poses.size());
        for (const auto &guess_pose : guess_poses)
            serial.push_back(matcher.align(keyframe, *CloudTransformer::transform(*subsampled_cloud, guess_pose.inverse()), Pose3D()));
        const VoxelGrid &shared = keyframe;
        std::vector<std::thread> threads;
        for (size_t i = 0; i < guess_poses.size(); i++)
            threads.emplace_back([&, i]() {
                CloudMatcher mine;  // stateless, like the reference's
                auto cloud = CloudTransformer::transform(*subsampled_cloud, guess_poses[i].inverse());
                for (int rep = 0; rep < 5; rep++) parallel[i] = mine.align(shared, *cloud, Pose3D());
                auto some = shared.findMatchingPairs(*cloud, Pose3D(), 0.3f);
                if (some.empty()) parallel[i].translation.v[0] = 1e9f;
            });
        for (auto &t : threads) t.join();
        int equal = 0;
        for (size_t i = 0; i < guess_poses.size(); i++) {
            bool same = true;
            for (int k = 0; k < 3; k++) same = same && serial[i].translation.v[k] == parallel[i].translation.v[k];
            for (int k = 0; k < 4; k++) same = same && serial[i].rotation.q[k] == parallel[i].rotation.q[k];
            equal += same;
            const double terr = parallel[i].relativeTo(guess_poses[i]).translation.norm();
            EXPECT(terr < 0.05);
        }
        EXPECT(equal >= (int)guess_poses.size() - 1);  // (a solve that found the GPU too full redoes itself host-driven: 1e-6)
    }
    // findMatchingPairs / getCorrespondence shapes
    auto pairs = keyframe.findMatchingPairs(*subsampled_cloud, Pose3D(), 0.3f);
    EXPECT(!pairs.empty() && pairs.size() <= subsampled_cloud->size());
    auto c = keyframe.getCorrespondence(Vector3f(12.5f, 12.5f, 1.02f), 0.3f * 0.3f);
    EXPECT(c.valid && std::fabs(c.plane_normal[2] - 1.0) < 1e-6 && std::fabs(c.plane_origin[2] - 1.0) < 1e-6);
    // radiusCleanup keeps the voxels whose first point is within the radius (voxel_grid.h:236-246)
    const size_t before = keyframe.size();
    keyframe.radiusCleanup(Vector3f(2, 2, 1), 5.0f);
    EXPECT(keyframe.size() < before && keyframe.size() > 0);
}

// LidarOdometry (lidar_odometry.h:65-76) over a synthetic room: 16 beams x 900 azimuth steps ray-cast
// against an axis-aligned box, the sensor moving 0.1 m per frame along +x
static void LidarOdometry_RoomSequence()
{
    LidarOdometry::Params params;
    LidarOdometry odometry(params);
    const float half[3] = {24.f, 17.f, 0.f};  // walls at x = +-24, y = +-17; floor z = -2, ceiling z = 7
    const float zlo = -2.f, zhi = 7.f;
    const int n_frames = 8;
    for (int f = 0; f < n_frames; f++) {
        const float sx = 0.1f * (float)f;
        LidarOdometry::CloudType cloud;
        for (int ring = 0; ring < 16; ring++) {
            const float el = (-15.f + 2.f * (float)ring) * 3.14159265358979f / 180.f;
            for (int a = 0; a < 900; a++) {
                const float az = (float)a * (2.f * 3.14159265358979f / 900.f);
                const float d[3] = {std::cos(el) * std::cos(az), -std::cos(el) * std::sin(az), std::sin(el)};
                float t = 1e9f;
                if (d[0] > 1e-6f) t = std::fmin(t, (half[0] - sx) / d[0]);
                if (d[0] < -1e-6f) t = std::fmin(t, (-half[0] - sx) / d[0]);
                if (d[1] > 1e-6f) t = std::fmin(t, half[1] / d[1]);
                if (d[1] < -1e-6f) t = std::fmin(t, -half[1] / d[1]);
                if (d[2] > 1e-6f) t = std::fmin(t, zhi / d[2]);
                if (d[2] < -1e-6f) t = std::fmin(t, zlo / d[2]);
                lom_point_xyzirt p{};
                p.x = t * d[0];
                p.y = t * d[1];
                p.z = t * d[2];
                p.intensity = 1.f;
                p.ring = (uint16_t)ring;
                p.time = (float)a / 900.f * 0.1f;
                cloud.points.push_back(p);
            }
        }
        odometry.processCloud(cloud);
        const auto st = odometry.lastFrameStats();
        EXPECT(st.planar_points > 2000 && st.filtered_points > 1000);
        if (f == 0) EXPECT(st.initialised_keyframe == 1);
        if (f > 0) EXPECT(st.outer_iterations >= 5 && st.matching_points > 200);
    }
    const Pose3D pose = odometry.getCurrentPose();
    EXPECT(std::fabs(pose.translation.x() - 0.1f * (float)(n_frames - 1)) < 0.1f);
    EXPECT(std::fabs(pose.translation.y()) < 0.05f && std::fabs(pose.translation.z()) < 0.05f);
    EXPECT(std::fabs(pose.rotation.w()) > 0.9999f);
    auto sparse = odometry.getKeyFrameCloud();
    auto full = odometry.getFullKeyFrameCloud();
    EXPECT(sparse->size() > 1000 && full->size() >= sparse->size());
}

// The reference's own orchestration (src/lidar_odometry.cpp:22-77), written with the mirror's classes exactly as the
// reference writes it with PCL / Eigen ones -- what a maintainer who keeps lidar_odometry.cpp compiles.  Its poses must
// equal lom::LidarOdometry's (which runs the same stages on the device) frame by frame.
struct ReferenceStyleOdometry {
    LidarOdometry::Params config_;
    VoxelGrid keyframe_;
    Pose3D previous_transform_, current_transform_;
    explicit ReferenceStyleOdometry(const LidarOdometry::Params &c)
        : config_(c), keyframe_(c.keyframe_voxel_size, c.keyframe_max_points_cnt)
    {
    }
    void processCloud(const LidarOdometry::CloudType &input_cloud)
    {
        auto time_normalized_cloud = utils::pointTimeNormalize(input_cloud);                                       // :25
        const auto relative_transform = previous_transform_.relativeTo(current_transform_);                         // :27
        previous_transform_ = current_transform_;                                                                   // :28
        auto deskewed_input_cloud =
            CloudTransformer::transformNonRigid(*time_normalized_cloud, relative_transform.inverse(), Pose3D());  // :30
        auto [planar_cloud, unclassified_cloud] = CloudClassifier::classify(*deskewed_input_cloud);                // :33
        auto filtered_planar_cloud =
            utils::rangeFilter(*planar_cloud, config_.lidar_min_range, config_.lidar_max_range);                   // :35
        VoxelGrid keyframe_downsampler(config_.keyframe_update_voxel_size, 1);                                      // :37
        keyframe_downsampler.addCloud(*filtered_planar_cloud);                                                      // :38
        if (keyframe_.size() == 0) {                                                                                // :40
            keyframe_.addCloud(*keyframe_downsampler.getCloud());                                                   // :42
            return;
        }
        VoxelGrid matching_downsampler(config_.keyframe_matching_voxel_size, 1);                                    // :46
        PointCloud<PointXYZ> filtered_xyz;
        for (const auto &p : filtered_planar_cloud->points) filtered_xyz.points.emplace_back(p.x, p.y, p.z);
        matching_downsampler.addCloudWithoutNormals(filtered_xyz);                                                  // :47
        CloudMatcher matcher;                                                                                       // :49
        const auto guess = current_transform_.compose(relative_transform);                                          // :51
        auto result = matcher.align(keyframe_, *matching_downsampler.getCloudWithoutNormals(), guess);              // :50
        current_transform_ = result;  // (the divergence guard of :53-63 never fires in this sequence)             // :65
        keyframe_.radiusCleanup(current_transform_.translation, config_.keyframe_cleanup_range);                    // :67
        auto update_cloud = CloudTransformer::transformWithNormals(*keyframe_downsampler.getCloud(), current_transform_);  // :69
        keyframe_.addCloud(*update_cloud);                                                                          // :70
    }
};

static LidarOdometry::CloudType room_frame(float sx)
{
    const float half[2] = {24.f, 17.f};
    const float zlo = -2.f, zhi = 7.f;
    LidarOdometry::CloudType cloud;
    for (int ring = 0; ring < 16; ring++) {
        const float el = (-15.f + 2.f * (float)ring) * 3.14159265358979f / 180.f;
        for (int a = 0; a < 900; a++) {
            const float az = (float)a * (2.f * 3.14159265358979f / 900.f);
            const float d[3] = {std::cos(el) * std::cos(az), -std::cos(el) * std::sin(az), std::sin(el)};
            float t = 1e9f;
            if (d[0] > 1e-6f) t = std::fmin(t, (half[0] - sx) / d[0]);
            if (d[0] < -1e-6f) t = std::fmin(t, (-half[0] - sx) / d[0]);
            if (d[1] > 1e-6f) t = std::fmin(t, half[1] / d[1]);
            if (d[1] < -1e-6f) t = std::fmin(t, -half[1] / d[1]);
            if (d[2] > 1e-6f) t = std::fmin(t, zhi / d[2]);
            if (d[2] < -1e-6f) t = std::fmin(t, zlo / d[2]);
            lom_point_xyzirt p{};
            p.x = t * d[0];
            p.y = t * d[1];
            p.z = t * d[2];
            p.intensity = 1.f;
            p.ring = (uint16_t)ring;
            p.time = (float)a / 900.f * 0.1f;
            cloud.points.push_back(p);
        }
    }
    return cloud;
}

static void ReferenceOrchestration_EqualsLidarOdometry()
{
    LidarOdometry::Params params;
    LidarOdometry odometry(params);
    ReferenceStyleOdometry by_hand(params);
    for (int f = 0; f < 5; f++) {
        const auto cloud = room_frame(0.1f * (float)f);
        // the stages on their own: sizes and value ranges
        auto normalized = utils::pointTimeNormalize(cloud);
        EXPECT(normalized->size() == cloud.size());
        float tmin = 1e9f, tmax = -1e9f;
        for (const auto &p : normalized->points) tmin = std::fmin(tmin, p.time), tmax = std::fmax(tmax, p.time);
        EXPECT(tmin == 0.f && tmax == 1.f);
        auto [planar, rest] = CloudClassifier::classify(*normalized);
        EXPECT(planar->size() > 2000 && planar->size() + rest->size() <= 16 * 900);
        for (size_t i = 0; i < planar->size(); i += 97) {
            const auto &p = planar->points[i];
            const float nn = p.normal_x * p.normal_x + p.normal_y * p.normal_y + p.normal_z * p.normal_z;
            EXPECT(std::fabs(nn - 1.f) < 1e-4f);
        }
        auto near = utils::rangeFilter(*planar, 0.f, 10.f);
        EXPECT(near->size() > 0 && near->size() < planar->size());
        for (const auto &p : near->points) EXPECT(p.x * p.x + p.y * p.y + p.z * p.z <= 100.f);
        // the two orchestrations
        odometry.processCloud(cloud);
        by_hand.processCloud(cloud);
        const Pose3D a = odometry.getCurrentPose(), b = by_hand.current_transform_;
        for (int k = 0; k < 3; k++) EXPECT(a.translation.v[k] == b.translation.v[k]);
        for (int k = 0; k < 4; k++) EXPECT(a.rotation.q[k] == b.rotation.q[k]);
    }
    EXPECT((size_t)odometry.lastFrameStats().keyframe_voxels == by_hand.keyframe_.size());
}

int main()
{
    try {
        VoxelGrid_UniquePoints();
        VoxelGrid_DuplicatePoints();
        Pose3D_ComposeRelativeInverse();
        CloudTransformer_RigidTransform();
        CloudMatcher_MatchingTest();
        LidarOdometry_RoomSequence();
        ReferenceOrchestration_EqualsLidarOdometry();
    } catch (const lom::Error &e) {
        std::printf("lom::Error %d: %s\n", e.code, e.what());
        return 2;
    }
    std::printf(g_fail ? "FAILED (%d)\n" : "ALL PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
