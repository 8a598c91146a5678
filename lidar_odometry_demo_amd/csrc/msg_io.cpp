// On-wire side of the reference's node: sensor_msgs/PointCloud2 payloads in and out,
// without ROS or PCL.  The node converts its input with pcl::fromROSMsg into PointCloud<PointXYZIRT>
// (src/lidar_odometry_node.cpp:47-48) and its outputs with pcl::toROSMsg (:61, :71); both are restated here
// from PCL's published conversion rules (pcl/conversions.h, PCL 1.12 as shipped with ROS2 Humble; PCL is
// absent from this image, so the restatement is pinned by the message definition only):
//   in:  every field registered for the point type (src/lidar_point_type.h:24-31: x y z intensity f32,
//        ring u16, time f32) is looked up in the message by name AND datatype AND count; a field with no such
//        match stays zero (PCL warns and leaves the value-initialised member); points are read at
//        row * row_step + col * point_step.
//   out: records are the point structs themselves (PointXYZ: 16 bytes, x y z at 0 4 8, fourth float 1.0;
//        PointXYZIRT: the 32-byte lom_point_xyzirt), height 1, width n, little endian.
// Host code; a frame is 26.6k points, one pass over it is a few microseconds beside the 0.37 ms frame.
#include <cstring>
#include <string>

#include "../../include/lidar_odometry_amd.h"

namespace {

struct Wanted {
    const char *name;
    uint8_t datatype;
    size_t dst_offset;
    size_t bytes;
};

// POINT_CLOUD_REGISTER_POINT_STRUCT(lidar_point::PointXYZIRT, ...), src/lidar_point_type.h:24-31
const Wanted kXyzirt[6] = {
    {"x", LOM_PF_FLOAT32, offsetof(lom_point_xyzirt, x), 4},
    {"y", LOM_PF_FLOAT32, offsetof(lom_point_xyzirt, y), 4},
    {"z", LOM_PF_FLOAT32, offsetof(lom_point_xyzirt, z), 4},
    {"intensity", LOM_PF_FLOAT32, offsetof(lom_point_xyzirt, intensity), 4},
    {"ring", LOM_PF_UINT16, offsetof(lom_point_xyzirt, ring), 2},
    {"time", LOM_PF_FLOAT32, offsetof(lom_point_xyzirt, time), 4},
};

thread_local std::string g_msg_error;

int64_t fail(int code, const std::string &what)
{
    g_msg_error = what;
    return code;
}

} // namespace

extern "C" {

const char *lom_pointcloud2_last_error(void) { return g_msg_error.c_str(); }

int64_t lom_pointcloud2_unpack(const lom_pc2_view *msg, lom_point_xyzirt *out, size_t cap, uint32_t *missing_mask)
{
    if (!msg || (!msg->fields && msg->n_fields)) return fail(LOM_ERR_ARG, "null message");
    if (msg->is_bigendian) return fail(LOM_ERR_ARG, "big-endian PointCloud2 payloads are not supported");
    const uint64_t n = (uint64_t)msg->width * msg->height;
    // source offset of each wanted field, or -1
    int64_t src[6];
    uint32_t missing = 0;
    for (int k = 0; k < 6; k++) {
        src[k] = -1;
        for (uint32_t i = 0; i < msg->n_fields; i++) {
            const lom_pc2_field &f = msg->fields[i];
            // pcl::FieldMatches: same name, same datatype, count 1 (0 is read as 1)
            if (!f.name || std::strcmp(f.name, kXyzirt[k].name) != 0) continue;
            if (f.datatype != kXyzirt[k].datatype || f.count > 1) continue;
            if ((uint64_t)f.offset + kXyzirt[k].bytes > msg->point_step)
                return fail(LOM_ERR_ARG, std::string("field '") + f.name + "' reaches past point_step");
            src[k] = f.offset;
            break;
        }
        if (src[k] < 0) missing |= 1u << k;
    }
    if (missing_mask) *missing_mask = missing;
    if (n == 0) return 0;
    if (msg->point_step == 0 || msg->row_step < (uint64_t)msg->width * msg->point_step)
        return fail(LOM_ERR_ARG, "row_step smaller than width * point_step");
    const uint64_t need = (uint64_t)(msg->height - 1) * msg->row_step + (uint64_t)msg->width * msg->point_step;
    if (!msg->data || msg->data_bytes < need) return fail(LOM_ERR_ARG, "payload shorter than height x row_step");
    if (!out || cap == 0) return (int64_t)n;
    uint64_t written = 0;
    for (uint32_t r = 0; r < msg->height && written < cap; r++) {
        const uint8_t *row = msg->data + (uint64_t)r * msg->row_step;
        for (uint32_t c = 0; c < msg->width && written < cap; c++) {
            const uint8_t *p = row + (uint64_t)c * msg->point_step;
            lom_point_xyzirt q;
            std::memset(&q, 0, sizeof q);
            for (int k = 0; k < 6; k++)
                if (src[k] >= 0) std::memcpy((uint8_t *)&q + kXyzirt[k].dst_offset, p + src[k], kXyzirt[k].bytes);
            out[written++] = q;
        }
    }
    return (int64_t)n;
}

int lom_pointcloud2_layout(int kind, lom_pc2_field fields_out[6], uint32_t *point_step_out)
{
    if (!fields_out || !point_step_out) return LOM_ERR_ARG;
    if (kind == LOM_PC2_XYZ) {
        for (int k = 0; k < 3; k++) fields_out[k] = {kXyzirt[k].name, (uint32_t)(4 * k), LOM_PF_FLOAT32, 1};
        *point_step_out = 16;
        return 3;
    }
    if (kind == LOM_PC2_XYZIRT) {
        for (int k = 0; k < 6; k++)
            fields_out[k] = {kXyzirt[k].name, (uint32_t)kXyzirt[k].dst_offset, kXyzirt[k].datatype, 1};
        *point_step_out = (uint32_t)sizeof(lom_point_xyzirt);
        return 6;
    }
    return LOM_ERR_ARG;
}

int64_t lom_pointcloud2_pack_xyz(const float *xyz, size_t n, size_t stride_bytes, uint8_t *data_out, size_t cap_bytes)
{
    if (stride_bytes == 0) stride_bytes = 12;
    if (stride_bytes < 12 || (n && !xyz)) return fail(LOM_ERR_ARG, "bad xyz input");
    const uint64_t need = (uint64_t)n * 16;
    if (!data_out) return (int64_t)need;
    if (cap_bytes < need) return fail(LOM_ERR_ARG, "output buffer smaller than 16 bytes per point");
    for (size_t i = 0; i < n; i++) {
        float rec[4];
        std::memcpy(rec, (const uint8_t *)xyz + i * stride_bytes, 12);
        rec[3] = 1.0f;
        std::memcpy(data_out + i * 16, rec, 16);
    }
    return (int64_t)need;
}

} // extern "C"
