// PCD (Point Cloud Data v0.7) reader for the clouds the path is fed from files: the reference's only
// file input is pcl::io::loadPCDFile<pcl::PointXYZ> in its matcher test (test/test.cpp:194); its
// shipped data file (test/test_data/intersection00056.pcd) is `DATA binary` with the layout
// FIELDS rgb _ x y z _ / SIZE 4 1 4 4 4 1 / COUNT 1 12 1 1 1 4 (32-byte records, x at byte 16).
// Host code, no PCL: header parsed field by field (FIELDS / SIZE / TYPE / COUNT / WIDTH / HEIGHT / POINTS /
// DATA), x y z (and normal_x normal_y normal_z when present) picked by name, `ascii` and `binary`
// bodies; `binary_compressed` (LZF) is not produced by anything on this path and is refused.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lidar_odometry_amd.h"

namespace {

struct Field {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
    size_t offset = 0;
};

thread_local std::string g_pcd_error;

int fail(int code, const std::string &what)
{
    g_pcd_error = what;
    return code;
}

std::vector<std::string> split(const std::string &line)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) i++;
        size_t j = i;
        while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') j++;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

// value of one field element as f32 (PCD TYPE F / I / U with SIZE 1, 2, 4, 8)
float element_as_float(const unsigned char *p, const Field &f)
{
    switch (f.type) {
    case 'F':
        if (f.size == 4) {
            float v;
            std::memcpy(&v, p, 4);
            return v;
        }
        if (f.size == 8) {
            double v;
            std::memcpy(&v, p, 8);
            return (float)v;
        }
        break;
    case 'I':
        if (f.size == 1) return (float)*reinterpret_cast<const int8_t *>(p);
        if (f.size == 2) {
            int16_t v;
            std::memcpy(&v, p, 2);
            return (float)v;
        }
        if (f.size == 4) {
            int32_t v;
            std::memcpy(&v, p, 4);
            return (float)v;
        }
        if (f.size == 8) {
            int64_t v;
            std::memcpy(&v, p, 8);
            return (float)v;
        }
        break;
    case 'U':
        if (f.size == 1) return (float)*p;
        if (f.size == 2) {
            uint16_t v;
            std::memcpy(&v, p, 2);
            return (float)v;
        }
        if (f.size == 4) {
            uint32_t v;
            std::memcpy(&v, p, 4);
            return (float)v;
        }
        if (f.size == 8) {
            uint64_t v;
            std::memcpy(&v, p, 8);
            return (float)v;
        }
        break;
    }
    return 0.f;
}

}  // namespace

extern "C" {

const char *lom_pcd_last_error(void) { return g_pcd_error.c_str(); }

static int64_t pcd_read(const char *path, float *xyz_out, float *nrm_out, size_t cap, lom_pcd_info *info);

// Nothing may leave an extern "C" function as an exception (the callers are C, ctypes, JNI ...): allocation
// failures of the parser's own buffers come back as LOM_ERR_OOM.
int64_t lom_pcd_read(const char *path, float *xyz_out, float *nrm_out, size_t cap, lom_pcd_info *info)
{
    try {
        return pcd_read(path, xyz_out, nrm_out, cap, info);
    } catch (const std::bad_alloc &) {
        return fail(LOM_ERR_OOM, "out of memory while reading the PCD file");
    } catch (const std::exception &e) {
        return fail(LOM_ERR_ARG, std::string("PCD reader: ") + e.what());
    }
}

}  // extern "C"

static int64_t pcd_read(const char *path, float *xyz_out, float *nrm_out, size_t cap, lom_pcd_info *info)
{
    if (!path || (cap && !xyz_out)) return fail(LOM_ERR_ARG, "null argument");
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return fail(LOM_ERR_ARG, std::string("cannot open ") + path + ": " + std::strerror(errno));
    struct Closer {
        std::FILE *f;
        ~Closer() { std::fclose(f); }
    } closer{f};

    std::vector<Field> fields;
    uint64_t width = 0, height = 1, points = 0;
    bool have_points = false, have_size = false;
    int data_kind = -1;  // 0 ascii, 1 binary
    // header: text lines up to and including DATA
    for (;;) {
        std::string line;
        int c;
        while ((c = std::fgetc(f)) != EOF && c != '\n') {
            if (line.size() >= 65536) return fail(LOM_ERR_ARG, "PCD header line too long");  // before it is buffered
            line.push_back((char)c);
        }
        if (c == EOF && line.empty()) return fail(LOM_ERR_ARG, "PCD header ends before DATA");
        const std::vector<std::string> tok = split(line);
        if (tok.empty() || tok[0][0] == '#') continue;
        const std::string &key = tok[0];
        if (key == "FIELDS" || key == "COLUMNS") {
            fields.clear();
            for (size_t i = 1; i < tok.size(); i++) {
                Field fd;
                fd.name = tok[i];
                fields.push_back(fd);
            }
        } else if (key == "SIZE" || key == "TYPE" || key == "COUNT") {
            if (tok.size() - 1 != fields.size()) return fail(LOM_ERR_ARG, key + " does not match FIELDS");
            for (size_t i = 1; i < tok.size(); i++) {
                Field &fd = fields[i - 1];
                if (key == "SIZE") {
                    fd.size = std::atoi(tok[i].c_str());
                    if (fd.size != 1 && fd.size != 2 && fd.size != 4 && fd.size != 8)
                        return fail(LOM_ERR_ARG, "unsupported SIZE " + tok[i]);
                    have_size = true;
                } else if (key == "TYPE") {
                    fd.type = tok[i][0];
                    if (fd.type != 'F' && fd.type != 'I' && fd.type != 'U') return fail(LOM_ERR_ARG, "unsupported TYPE " + tok[i]);
                } else {
                    fd.count = std::atoi(tok[i].c_str());
                    if (fd.count < 0 || fd.count > (1 << 20)) return fail(LOM_ERR_ARG, "unsupported COUNT " + tok[i]);
                }
            }
        } else if (key == "WIDTH" && tok.size() > 1) {
            width = std::strtoull(tok[1].c_str(), nullptr, 10);
        } else if (key == "HEIGHT" && tok.size() > 1) {
            height = std::strtoull(tok[1].c_str(), nullptr, 10);
        } else if (key == "POINTS" && tok.size() > 1) {
            points = std::strtoull(tok[1].c_str(), nullptr, 10);
            have_points = true;
        } else if (key == "DATA" && tok.size() > 1) {
            if (tok[1] == "ascii")
                data_kind = 0;
            else if (tok[1] == "binary")
                data_kind = 1;
            else
                return fail(LOM_ERR_ARG, "DATA " + tok[1] + " is not supported (ascii and binary are)");
            break;
        }  // VERSION, VIEWPOINT and unknown keys are skipped
        if (c == EOF) return fail(LOM_ERR_ARG, "PCD header ends before DATA");
    }
    if (fields.empty() || !have_size) return fail(LOM_ERR_ARG, "PCD header has no FIELDS / SIZE");
    if (!have_points) points = width * height;
    if (points >= 0x7FFFFFFFull) return fail(LOM_ERR_ARG, "too many points");
    size_t step = 0;
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
    for (size_t i = 0; i < fields.size(); i++) {
        fields[i].offset = step;
        step += (size_t)fields[i].size * (size_t)fields[i].count;
        if (fields[i].count < 1) continue;
        const std::string &nm = fields[i].name;
        if (nm == "x") ix = (int)i;
        if (nm == "y") iy = (int)i;
        if (nm == "z") iz = (int)i;
        if (nm == "normal_x") inx = (int)i;
        if (nm == "normal_y") iny = (int)i;
        if (nm == "normal_z") inz = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return fail(LOM_ERR_ARG, "PCD file has no x / y / z fields");
    // SIZE and COUNT come from the file: a record of more than 64 KiB is not a point cloud this reader is for
    // (64 fields of COUNT 2^20 x 8 bytes would ask for half a terabyte of chunk buffer below)
    if (step == 0 || step > 65536) return fail(LOM_ERR_ARG, "PCD record size (sum of SIZE x COUNT) must be 1 .. 65536 bytes");
    const bool has_normals = inx >= 0 && iny >= 0 && inz >= 0;
    if (info) {
        info->points = points;
        info->width = (uint32_t)width;
        info->height = (uint32_t)height;
        info->point_step = (uint32_t)step;
        info->has_normals = has_normals ? 1 : 0;
        info->data_kind = data_kind;
    }
    const size_t take = (size_t)points < cap ? (size_t)points : cap;
    if (!take) return (int64_t)points;
    if (data_kind == 1) {
        // records in chunks: the file may be far larger than what the caller wants of it
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(65536, ((size_t)16 << 20) / step));  // <= 16 MiB of buffer
        std::vector<unsigned char> buf(chunk * step);
        size_t done = 0;
        while (done < take) {
            const size_t want = std::min(chunk, take - done);
            if (std::fread(buf.data(), step, want, f) != want) return fail(LOM_ERR_ARG, "PCD body is shorter than POINTS says");
            for (size_t i = 0; i < want; i++) {
                const unsigned char *r = buf.data() + i * step;
                float *o = xyz_out + (done + i) * 3;
                o[0] = element_as_float(r + fields[ix].offset, fields[ix]);
                o[1] = element_as_float(r + fields[iy].offset, fields[iy]);
                o[2] = element_as_float(r + fields[iz].offset, fields[iz]);
                if (nrm_out) {
                    float *no = nrm_out + (done + i) * 3;
                    no[0] = has_normals ? element_as_float(r + fields[inx].offset, fields[inx]) : 0.f;
                    no[1] = has_normals ? element_as_float(r + fields[iny].offset, fields[iny]) : 0.f;
                    no[2] = has_normals ? element_as_float(r + fields[inz].offset, fields[inz]) : 0.f;
                }
            }
            done += want;
        }
        return (int64_t)points;
    }
    // ascii: one point per line, every field element a token ("nan" included, as strtof reads it)
    size_t n_tok = 0;
    std::vector<size_t> first_tok(fields.size());
    for (size_t i = 0; i < fields.size(); i++) {
        first_tok[i] = n_tok;
        n_tok += (size_t)fields[i].count;
    }
    for (size_t p = 0; p < take; p++) {
        std::string line;
        int c;
        do {
            line.clear();
            while ((c = std::fgetc(f)) != EOF && c != '\n') {
                if (line.size() >= ((size_t)1 << 22)) return fail(LOM_ERR_ARG, "PCD data line too long");
                line.push_back((char)c);
            }
        } while (c != EOF && split(line).empty());
        const std::vector<std::string> tok = split(line);
        if (tok.size() < n_tok) return fail(LOM_ERR_ARG, "PCD body is shorter than POINTS says");
        auto val = [&](int fi) { return std::strtof(tok[first_tok[(size_t)fi]].c_str(), nullptr); };
        float *o = xyz_out + p * 3;
        o[0] = val(ix), o[1] = val(iy), o[2] = val(iz);
        if (nrm_out) {
            float *no = nrm_out + p * 3;
            no[0] = has_normals ? val(inx) : 0.f;
            no[1] = has_normals ? val(iny) : 0.f;
            no[2] = has_normals ? val(inz) : 0.f;
        }
    }
    return (int64_t)points;
}


