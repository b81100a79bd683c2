// scene_parse.cpp -- host-side scene construction: the reference's JSON schema
// (raytracer-lib/src/scene/mod.rs:27-376) and parse_scene (scene/mod.rs:378-715)
// restated in C++, producing the flat RtScene of include/rt_scene.h.
//
// Pure host code; no HIP.  Everything here runs once per scene.

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "json_min.h"
#include "rt_error.h"

using jsonmin::Value;

namespace {

struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct Invalid : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct IoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Owner of every array an RtScene points into.  `scene` must stay the first
// member: rt_scene_destroy() casts the RtScene* back to this struct.
struct SceneOwner {
    RtScene scene;
    std::vector<RtShape> shapes, protos;
    std::vector<RtMesh> meshes;
    std::vector<std::vector<double>> mesh_vertices;
    std::vector<RtShader> shaders;
    std::vector<RtLight> lights;
};

// ---- small f64 linear algebra with the accumulation order of the reference's
// ---- matrix library (column-by-column axpy: ((a_i0*b_0 + a_i1*b_1) + a_i2*b_2) ...)

struct M4 {
    double m[16];  // row-major
    double &at(int r, int c) { return m[r * 4 + c]; }
    double at(int r, int c) const { return m[r * 4 + c]; }
};

M4 m4_identity() {
    M4 r{};
    for (int i = 0; i < 4; ++i) r.at(i, i) = 1.0;
    return r;
}

M4 m4_mul(const M4 &a, const M4 &b) {
    M4 c{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            double acc = a.at(i, 0) * b.at(0, j);
            for (int k = 1; k < 4; ++k) acc = acc + a.at(i, k) * b.at(k, j);
            c.at(i, j) = acc;
        }
    return c;
}

M4 m4_transpose(const M4 &a) {
    M4 t{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t.at(i, j) = a.at(j, i);
    return t;
}

struct M3 {
    double m[9];
    double &at(int r, int c) { return m[r * 3 + c]; }
    double at(int r, int c) const { return m[r * 3 + c]; }
};

M3 m3_identity() {
    M3 r{};
    for (int i = 0; i < 3; ++i) r.at(i, i) = 1.0;
    return r;
}

M3 m3_mul(const M3 &a, const M3 &b) {
    M3 c{};
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            double acc = a.at(i, 0) * b.at(0, j);
            for (int k = 1; k < 3; ++k) acc = acc + a.at(i, k) * b.at(k, j);
            c.at(i, j) = acc;
        }
    return c;
}

// Rotation3::from_axis_angle for a coordinate axis (scene/mod.rs:641-647):
// the axis-angle formula with a unit basis vector as axis.
M3 rotation_about_axis(int axis, double angle) {
    const double u[3] = {axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0};
    const double s = std::sin(angle), c = std::cos(angle), omc = 1.0 - c;
    const double sqx = u[0] * u[0], sqy = u[1] * u[1], sqz = u[2] * u[2];
    M3 r;
    r.at(0, 0) = sqx + (1.0 - sqx) * c;
    r.at(0, 1) = u[0] * u[1] * omc - u[2] * s;
    r.at(0, 2) = u[0] * u[2] * omc + u[1] * s;
    r.at(1, 0) = u[0] * u[1] * omc + u[2] * s;
    r.at(1, 1) = sqy + (1.0 - sqy) * c;
    r.at(1, 2) = u[1] * u[2] * omc - u[0] * s;
    r.at(2, 0) = u[0] * u[2] * omc - u[1] * s;
    r.at(2, 1) = u[1] * u[2] * omc + u[0] * s;
    r.at(2, 2) = sqz + (1.0 - sqz) * c;
    return r;
}

M4 m4_from_rotation(const M3 &r) {
    M4 m = m4_identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.at(i, j) = r.at(i, j);
    return m;
}

M4 m4_translation(const RtVec3 &t) {
    M4 m = m4_identity();
    m.at(0, 3) = t.x;
    m.at(1, 3) = t.y;
    m.at(2, 3) = t.z;
    return m;
}

M4 m4_scale(const RtVec3 &s) {
    M4 m = m4_identity();
    m.at(0, 0) = s.x;
    m.at(1, 1) = s.y;
    m.at(2, 2) = s.z;
    return m;
}

// Matrix4::transform_point (geometry/bbox.rs:55): (M3*p + t) / (0*p + m33).
RtVec3 m4_transform_point(const M4 &m, const RtVec3 &p) {
    double r[3];
    for (int i = 0; i < 3; ++i) {
        double acc = m.at(i, 0) * p.x;
        acc = acc + m.at(i, 1) * p.y;
        acc = acc + m.at(i, 2) * p.z;
        r[i] = acc + m.at(i, 3);
    }
    double n = m.at(3, 0) * p.x + m.at(3, 1) * p.y + m.at(3, 2) * p.z + m.at(3, 3);
    if (n != 0.0) {
        r[0] /= n;
        r[1] /= n;
        r[2] /= n;
    }
    return RtVec3{r[0], r[1], r[2]};
}

RtVec3 v_sub(const RtVec3 &a, const RtVec3 &b) { return RtVec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RtVec3 v_add(const RtVec3 &a, const RtVec3 &b) { return RtVec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
double v_dot(const RtVec3 &a, const RtVec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RtVec3 v_cross(const RtVec3 &a, const RtVec3 &b) {
    return RtVec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RtVec3 v_normalize(const RtVec3 &a) {
    double n = std::sqrt(v_dot(a, a));
    return RtVec3{a.x / n, a.y / n, a.z / n};
}

// ---- vec3 fields: "x y z" string or [x, y, z] array (scene/parse_vec3.rs:31-76)

bool parse_full_double(const std::string &tok, double &out) {
    if (tok.empty()) return false;
    char *end = nullptr;
    out = std::strtod(tok.c_str(), &end);
    return end == tok.c_str() + tok.size();
}

bool parse_full_float(const std::string &tok, float &out) {
    if (tok.empty()) return false;
    char *end = nullptr;
    out = std::strtof(tok.c_str(), &end);
    return end == tok.c_str() + tok.size();
}

std::vector<std::string> split_whitespace(const std::string &s) {
    std::vector<std::string> out;
    std::istringstream iss(s);
    std::string tok;
    while (iss >> tok) out.push_back(tok);
    return out;
}

RtVec3 vec3_f64(const Value &v, const char *what) {
    if (v.is_string()) {
        auto toks = split_whitespace(v.str);
        double d[3];
        if (toks.size() != 3) throw Invalid(std::string(what) + ": expected exactly 3 space-separated numbers");
        for (int i = 0; i < 3; ++i)
            if (!parse_full_double(toks[i], d[i]))
                throw Invalid(std::string(what) + ": failed to parse number in vector string");
        return RtVec3{d[0], d[1], d[2]};
    }
    if (v.is_array()) {
        if (v.arr.size() != 3) throw Invalid(std::string(what) + ": array must have exactly 3 numbers");
        for (const auto &e : v.arr)
            if (!e.is_number()) throw Invalid(std::string(what) + ": array must hold numbers");
        return RtVec3{v.arr[0].num, v.arr[1].num, v.arr[2].num};
    }
    throw Invalid(std::string(what) + ": expected \"x y z\" string or [x, y, z] array");
}

void vec3_f32(const Value &v, const char *what, float out[3]) {
    if (v.is_string()) {
        auto toks = split_whitespace(v.str);
        if (toks.size() != 3) throw Invalid(std::string(what) + ": expected exactly 3 space-separated numbers");
        for (int i = 0; i < 3; ++i)
            if (!parse_full_float(toks[i], out[i]))
                throw Invalid(std::string(what) + ": failed to parse number in vector string");
        return;
    }
    if (v.is_array()) {
        if (v.arr.size() != 3) throw Invalid(std::string(what) + ": array must have exactly 3 numbers");
        for (int i = 0; i < 3; ++i) {
            if (!v.arr[i].is_number()) throw Invalid(std::string(what) + ": array must hold numbers");
            out[i] = (float)v.arr[i].num;  // JSON number -> f64 -> f32
        }
        return;
    }
    throw Invalid(std::string(what) + ": expected \"x y z\" string or [x, y, z] array");
}

const Value &require(const Value &obj, std::initializer_list<const char *> names, const char *ctx) {
    const Value *v = obj.find(names);
    if (!v) throw Invalid(std::string(ctx) + ": missing field `" + *names.begin() + "`");
    return *v;
}

double require_number(const Value &obj, std::initializer_list<const char *> names, const char *ctx) {
    const Value &v = require(obj, names, ctx);
    if (!v.is_number()) throw Invalid(std::string(ctx) + ": field `" + *names.begin() + "` must be a number");
    return v.num;
}

std::string require_string(const Value &obj, std::initializer_list<const char *> names, const char *ctx) {
    const Value &v = require(obj, names, ctx);
    if (!v.is_string()) throw Invalid(std::string(ctx) + ": field `" + *names.begin() + "` must be a string");
    return v.str;
}

// A material property is a colour, or a texture reference the reference
// cannot render (scene/mod.rs:210-220, 465-479).
void material_color(const Value &v, const char *what, float out[3]) {
    if (v.is_object()) throw Unsupported(std::string(what) + ": texture for material property not implemented");
    vec3_f32(v, what, out);
}

// ---- OBJ (geometry/mesh.rs:18-58): positions as f32, fan triangulation,
// ---- exactly one model.

void load_obj_triangles(const std::string &path, std::vector<double> &tri_vertices) {
    std::ifstream in(path);
    if (!in) throw IoError("Failed to load model for mesh: " + path);
    std::vector<float> pos;
    std::vector<long> face;
    int models = 0;
    bool open_model_has_faces = false;
    std::string line;
    while (std::getline(in, line)) {
        size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos || line[b] == '#') continue;
        std::istringstream ls(line.substr(b));
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            std::string sx, sy, sz;
            ls >> sx >> sy >> sz;
            float f[3];
            if (!parse_full_float(sx, f[0]) || !parse_full_float(sy, f[1]) || !parse_full_float(sz, f[2]))
                throw Invalid("OBJ: bad vertex line in " + path);
            pos.insert(pos.end(), f, f + 3);
        } else if (tag == "f") {
            face.clear();
            std::string tok;
            const long nv = (long)(pos.size() / 3);
            while (ls >> tok) {
                long idx = std::strtol(tok.c_str(), nullptr, 10);  // stops at '/'
                if (idx == 0) throw Invalid("OBJ: bad face index in " + path);
                idx = idx > 0 ? idx - 1 : nv + idx;
                if (idx < 0 || idx >= nv) throw Invalid("OBJ: face index out of range in " + path);
                face.push_back(idx);
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                const long tri[3] = {face[0], face[k], face[k + 1]};
                for (long vi : tri)
                    for (int c = 0; c < 3; ++c) tri_vertices.push_back((double)pos[vi * 3 + c]);
            }
            if (face.size() >= 3) open_model_has_faces = true;
        } else if (tag == "o" || tag == "g") {
            if (open_model_has_faces) {
                ++models;
                open_model_has_faces = false;
            }
        }
    }
    if (open_model_has_faces) ++models;
    if (models != 1) {
        throw Invalid("expected exactly one model, found " + std::to_string(models) + " for mesh " + path);
    }
}

// ---- constructors of derived scene data

// CoordinateSystem::new (math/coordinate_system.rs:12-39)
void camera_basis(const RtVec3 &view_direction, RtVec3 &u_out, RtVec3 &v_out, RtVec3 &w_out) {
    RtVec3 n = v_normalize(view_direction);
    RtVec3 w{-n.x, -n.y, -n.z};
    RtVec3 up{0.0, 1.0, 0.0};
    double tdotw = v_dot(up, w);
    if (std::fabs(tdotw) > 0.999) {
        up = w;
        double x = std::fabs(up.x), y = std::fabs(up.y), z = std::fabs(up.z);
        if (x <= y && x <= z) up.x = 1.0;
        else if (y <= x) up.y = 1.0;
        else up.z = 1.0;
    }
    RtVec3 u = v_cross(up, w);
    RtVec3 v = v_cross(w, u);
    u_out = v_normalize(u);
    v_out = v_normalize(v);
    w_out = w;
}

struct Box3 {
    RtVec3 min, max;
};

Box3 box_of_points(const RtVec3 *p, int n) {
    Box3 b{p[0], p[0]};
    for (int i = 1; i < n; ++i) {
        b.min.x = std::fmin(b.min.x, p[i].x);
        b.min.y = std::fmin(b.min.y, p[i].y);
        b.min.z = std::fmin(b.min.z, p[i].z);
        b.max.x = std::fmax(b.max.x, p[i].x);
        b.max.y = std::fmax(b.max.y, p[i].y);
        b.max.z = std::fmax(b.max.z, p[i].z);
    }
    return b;
}

// Bounding box of a prototype as its constructor computes it
// (sphere.rs:23-26, cuboid.rs:18, triangle.rs:23-32, mesh.rs:59-60).
Box3 proto_bbox(const RtShape &s, const SceneOwner &so) {
    switch (s.type) {
    case RT_SHAPE_SPHERE: {
        RtVec3 r{s.radius, s.radius, s.radius};
        return Box3{v_sub(s.p0, r), v_add(s.p0, r)};
    }
    case RT_SHAPE_BOX: return Box3{s.p0, s.p1};
    case RT_SHAPE_TRIANGLE: {
        RtVec3 p[3] = {s.p0, s.p1, s.p2};
        return box_of_points(p, 3);
    }
    case RT_SHAPE_MESH: {
        const std::vector<double> &v = so.mesh_vertices[(size_t)s.mesh];
        if (v.empty()) throw Invalid("mesh has no triangles");
        Box3 b{RtVec3{v[0], v[1], v[2]}, RtVec3{v[0], v[1], v[2]}};
        for (size_t i = 0; i + 2 < v.size(); i += 3) {
            b.min.x = std::fmin(b.min.x, v[i]);
            b.min.y = std::fmin(b.min.y, v[i + 1]);
            b.min.z = std::fmin(b.min.z, v[i + 2]);
            b.max.x = std::fmax(b.max.x, v[i]);
            b.max.y = std::fmax(b.max.y, v[i + 1]);
            b.max.z = std::fmax(b.max.z, v[i + 2]);
        }
        return b;
    }
    default: throw Invalid("An instanced shape can not be type instance");
    }
}

// BBox::transform (geometry/bbox.rs:38-74)
Box3 transform_box(const Box3 &b, const M4 &m) {
    const RtVec3 corners[8] = {
        b.min,
        {b.max.x, b.min.y, b.min.z},
        {b.min.x, b.max.y, b.min.z},
        {b.max.x, b.max.y, b.min.z},
        {b.min.x, b.min.y, b.max.z},
        {b.max.x, b.min.y, b.max.z},
        {b.min.x, b.max.y, b.max.z},
        b.max,
    };
    RtVec3 t[8];
    for (int i = 0; i < 8; ++i) t[i] = m4_transform_point(m, corners[i]);
    return box_of_points(t, 8);
}

// Instance::new (geometry/instance.rs:21-52)
void instance_matrices(const RtVec3 &translate, const M3 &rotation, const RtVec3 &scale,
                       const Box3 &proto_box, const std::string &name, RtShape &out) {
    M4 T = m4_translation(translate), R = m4_from_rotation(rotation), S = m4_scale(scale);
    M4 transform = m4_mul(m4_mul(T, R), S);
    M3 rt;  // rotation.inverse() == transpose
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rt.at(i, j) = rotation.at(j, i);
    M4 inv_rotate = m4_from_rotation(rt);
    if (scale.x == 0.0 || scale.y == 0.0 || scale.z == 0.0)
        throw Invalid("The scaling applied to " + name + " is not invertible");
    M4 inv_scale = m4_scale(RtVec3{1.0 / scale.x, 1.0 / scale.y, 1.0 / scale.z});
    M4 inv_translate = m4_translation(RtVec3{-translate.x, -translate.y, -translate.z});
    M4 inv_scale_rot = m4_mul(inv_scale, inv_rotate);
    M4 inv_transform = m4_mul(inv_scale_rot, inv_translate);
    M4 normal_matrix = m4_transpose(inv_scale_rot);
    std::memcpy(out.inv_transform, inv_transform.m, sizeof(out.inv_transform));
    std::memcpy(out.normal_matrix, normal_matrix.m, sizeof(out.normal_matrix));
    Box3 wb = transform_box(proto_box, transform);
    out.bbox_min = wb.min;
    out.bbox_max = wb.max;
}

std::string join_path(const std::string &dir, const std::string &file) {
    if (!file.empty() && file[0] == '/') return file;  // Path::join replaces on absolute
    if (dir.empty()) return file;
    if (dir.back() == '/') return dir + file;
    return dir + "/" + file;
}

std::string to_lower(std::string s) {
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

// A shape entry without its shader: sphere / box / triangle / mesh payloads
// (scene/mod.rs:286-332, 501-546, 579-620).
RtShape geometric_shape(const Value &sh, const std::string &type, const std::string &name,
                        const std::string &data_path, SceneOwner &so) {
    RtShape s{};
    s.mesh = -1;
    s.proto = -1;
    s.shader = -1;
    const char *ctx = name.c_str();
    if (type == "sphere") {
        s.type = RT_SHAPE_SPHERE;
        s.p0 = vec3_f64(require(sh, {"center"}, ctx), "center");
        s.radius = require_number(sh, {"radius"}, ctx);
    } else if (type == "box") {
        s.type = RT_SHAPE_BOX;
        const Value *mn = sh.find({"min", "minPt"});
        const Value *mx = sh.find({"max", "maxPt"});
        if (mn && mx) {
            s.p0 = vec3_f64(*mn, "minPt");
            s.p1 = vec3_f64(*mx, "maxPt");
        } else {
            RtVec3 center = vec3_f64(require(sh, {"center"}, ctx), "center");
            RtVec3 extent = vec3_f64(require(sh, {"extent"}, ctx), "extent");
            RtVec3 half{extent.x / 2.0, extent.y / 2.0, extent.z / 2.0};
            s.p0 = v_sub(center, half);
            s.p1 = v_add(center, half);
        }
    } else if (type == "triangle") {
        s.type = RT_SHAPE_TRIANGLE;
        s.p0 = vec3_f64(require(sh, {"a", "v0"}, ctx), "v0");
        s.p1 = vec3_f64(require(sh, {"b", "v1"}, ctx), "v1");
        s.p2 = vec3_f64(require(sh, {"c", "v2"}, ctx), "v2");
    } else if (type == "mesh") {
        s.type = RT_SHAPE_MESH;
        std::string file = require_string(sh, {"model_path", "file"}, ctx);
        so.mesh_vertices.emplace_back();
        load_obj_triangles(join_path(data_path, file), so.mesh_vertices.back());
        s.mesh = (int32_t)so.mesh_vertices.size() - 1;
    } else {
        throw Invalid("unknown shape _type `" + type + "` for " + name);
    }
    return s;
}

std::string shader_ref_name(const Value &shape, const char *ctx) {
    const Value &r = require(shape, {"_shader", "shader"}, ctx);
    if (r.is_string()) return r.str;
    if (r.is_object()) return require_string(r, {"_ref"}, ctx);
    throw Invalid(std::string(ctx) + ": shader reference must be a string or {\"_ref\": name}");
}

SceneOwner *build_scene(const char *json_text, const std::string &data_path, const RtParseArgs &args) {
    Value doc = jsonmin::parse(json_text);
    if (!doc.is_object()) throw Invalid("scene file must be a JSON object");
    const Value &scene = require(doc, {"scene"}, "root");
    if (!scene.is_object()) throw Invalid("`scene` must be an object");

    auto so = std::make_unique<SceneOwner>();

    // Image size or default; aspect ratio (scene/mod.rs:396-400)
    const uint32_t image_width = args.image_width ? args.image_width : 360u;
    const uint32_t image_height = args.image_height ? args.image_height : 360u;
    const double aspect_ratio =
        args.aspect_ratio > 0.0 ? args.aspect_ratio : (double)image_width / (double)image_height;

    // ---- camera (scene/mod.rs:402-457)
    const Value &cameras = require(scene, {"cameras", "camera"}, "scene");
    if (!cameras.is_array()) throw Invalid("`camera` must be an array");
    if (cameras.arr.empty()) throw Invalid("scene must have at least one camera");
    const Value *params = scene.find({"scene_parameters", "sceneParameters"});
    size_t camera_index = 0;
    for (const auto &c : cameras.arr) require_string(c, {"_name"}, "camera");
    if (cameras.arr.size() != 1) {
        std::string camera_name = "main";
        if (params) {
            const Value *cn = params->find({"camera"});
            if (cn && cn->is_string()) camera_name = cn->str;
        }
        bool found = false;
        for (size_t i = 0; i < cameras.arr.size(); ++i) {
            if (require_string(cameras.arr[i], {"_name"}, "camera") == camera_name) {
                camera_index = i;
                found = true;
                break;
            }
        }
        if (!found) throw Invalid("camera " + camera_name + " not found");
    }
    {
        const Value &cam = cameras.arr[camera_index];
        std::string ctype = require_string(cam, {"_type"}, "camera");
        RtCamera &rc = so->scene.camera;
        std::memset(&rc, 0, sizeof(rc));
        if (ctype == "perspective" || ctype == "Perspective") rc.type = RT_CAMERA_PERSPECTIVE;
        else if (ctype == "orthographic" || ctype == "Orthographic") rc.type = RT_CAMERA_ORTHOGRAPHIC;
        else throw Invalid("unknown camera _type `" + ctype + "`");
        rc.position = vec3_f64(require(cam, {"position"}, "camera"), "position");
        RtVec3 view;
        if (const Value *vd = cam.find({"view_dir", "viewDir"})) {
            view = vec3_f64(*vd, "viewDir");
        } else if (const Value *la = cam.find({"lookat_point", "lookatPoint"})) {
            view = v_sub(vec3_f64(*la, "lookatPoint"), rc.position);
        } else {
            throw Invalid("camera needs `viewDir` or `lookatPoint`");
        }
        if (rc.type == RT_CAMERA_PERSPECTIVE)
            rc.focal_length = require_number(cam, {"focal_length", "focalLength"}, "camera");
        // CameraBase::new (camera/mod.rs:37-50); imagePlaneWidth is parsed but never read.
        const double image_plane_width = 0.5;
        const double image_plane_height = image_plane_width / aspect_ratio;
        camera_basis(view, rc.u, rc.v, rc.w);
        rc.left = -image_plane_width / 2.0;
        rc.right = image_plane_width / 2.0;
        rc.top = image_plane_height / 2.0;
        rc.bottom = -image_plane_height / 2.0;
        rc.pixels_x = image_width;
        rc.pixels_y = image_height;
    }

    // ---- shaders (scene/mod.rs:459-494): later duplicates replace earlier ones
    std::map<std::string, int32_t> shader_index;
    const Value &shaders = require(scene, {"shaders", "shader"}, "scene");
    if (!shaders.is_array()) throw Invalid("`shader` must be an array");
    for (const auto &sd : shaders.arr) {
        std::string name = require_string(sd, {"_name"}, "shader");
        std::string type = require_string(sd, {"_type"}, name.c_str());
        RtShader rs{};
        if (type == "Lambertian") {
            rs.type = RT_SHADER_LAMBERTIAN;
            material_color(require(sd, {"diffuse"}, name.c_str()), "diffuse", rs.diffuse);
        } else if (type == "BlinnPhong") {
            rs.type = RT_SHADER_BLINN_PHONG;
            material_color(require(sd, {"diffuse"}, name.c_str()), "diffuse", rs.diffuse);
            material_color(require(sd, {"specular"}, name.c_str()), "specular", rs.specular);
            rs.shininess = (float)require_number(sd, {"shininess", "phongExp"}, name.c_str());
        } else if (type == "PerfectMirror" || type == "Mirror") {
            rs.type = RT_SHADER_PERFECT_MIRROR;
        } else if (type == "GGXMirror") {
            rs.type = RT_SHADER_GGX_MIRROR;
            // GGXMirrorShader::new clamps (ggx_mirror.rs:14-17); RtShader holds what the constructor holds
            rs.roughness = std::fmin(std::fmax(require_number(sd, {"roughness"}, name.c_str()), 0.0), 1.0);
            rs.samples = (uint32_t)require_number(sd, {"samples"}, name.c_str());
        } else if (type == "Diffuse" || type == "BlinnPhongMirror" || type == "BlinnPhongMirrored" ||
                   type == "Glaze" || type == "Dielectric") {
            rs.type = RT_SHADER_NULL;
        } else {
            throw Invalid("unknown shader _type `" + type + "`");
        }
        so->shaders.push_back(rs);
        shader_index[name] = (int32_t)so->shaders.size() - 1;
    }
    // Shaders every scene carries implicitly: NullShader for prototypes
    // (scene/mod.rs:500) and NormalShader for --render-normals (scene/mod.rs:549).
    RtShader null_shader{};
    null_shader.type = RT_SHADER_NULL;
    so->shaders.push_back(null_shader);
    const int32_t null_shader_index = (int32_t)so->shaders.size() - 1;
    RtShader normal_shader{};
    normal_shader.type = RT_SHADER_NORMAL;
    so->shaders.push_back(normal_shader);
    const int32_t normal_shader_index = (int32_t)so->shaders.size() - 1;

    // ---- instance prototypes (scene/mod.rs:496-546): later duplicates replace
    std::map<std::string, int32_t> proto_index;
    if (const Value *insts = scene.find({"instances", "instance"})) {
        if (!insts->is_array()) throw Invalid("`instance` must be an array");
        for (const auto &sh : insts->arr) {
            std::string name = require_string(sh, {"_name"}, "instance");
            (void)shader_ref_name(sh, name.c_str());  // required by the schema, ignored
            std::string type = require_string(sh, {"_type"}, name.c_str());
            if (type == "instance") throw Unsupported("An instanced shape can not be type instance");
            RtShape s = geometric_shape(sh, type, name, data_path, *so);
            s.shader = null_shader_index;
            so->protos.push_back(s);
            proto_index[name] = (int32_t)so->protos.size() - 1;
        }
    }

    // ---- shapes (scene/mod.rs:551-667)
    std::set<std::string> shape_names;
    const Value &shapes = require(scene, {"shapes", "shape"}, "scene");
    if (!shapes.is_array()) throw Invalid("`shape` must be an array");
    for (const auto &sh : shapes.arr) {
        std::string name = require_string(sh, {"_name"}, "shape");
        std::string shader_name = shader_ref_name(sh, name.c_str());
        std::string type = require_string(sh, {"_type"}, name.c_str());
        int32_t shader;
        if (!args.render_normals) {
            auto it = shader_index.find(shader_name);
            if (it == shader_index.end()) throw Invalid("shape references non-existent shader");
            shader = it->second;
        } else {
            shader = normal_shader_index;
        }
        if (!shape_names.insert(name).second) throw Invalid("shape names must be unique");

        RtShape s{};
        if (type == "instance") {
            s.type = RT_SHAPE_INSTANCE;
            s.mesh = -1;
            std::string id = require_string(sh, {"instance_of", "_id"}, name.c_str());
            auto it = proto_index.find(id);
            if (it == proto_index.end()) throw Invalid("instance ID is not a valid instance");
            s.proto = it->second;
            const Value &xf = require(sh, {"transform", "xform"}, name.c_str());
            if (!xf.is_array()) throw Invalid("`xform` must be an array");
            RtVec3 translate{0.0, 0.0, 0.0}, scale{1.0, 1.0, 1.0};
            M3 rot[3] = {m3_identity(), m3_identity(), m3_identity()};
            for (const auto &t : xf.arr) {
                std::string tt = require_string(t, {"type"}, "xform");
                if (tt == "translate") {
                    translate = v_add(translate, vec3_f64(require(t, {"amount"}, "translate"), "amount"));
                } else if (tt == "scale") {
                    RtVec3 a = vec3_f64(require(t, {"amount"}, "scale"), "amount");
                    scale = RtVec3{scale.x * a.x, scale.y * a.y, scale.z * a.z};
                } else if (tt == "rotate" || tt == "rotation") {
                    std::string axis = to_lower(require_string(t, {"axis"}, "rotate"));
                    double degrees = require_number(t, {"degrees", "amount"}, "rotate");
                    double angle = M_PI * degrees / 180.0;
                    int ax = axis == "x" ? 0 : axis == "y" ? 1 : axis == "z" ? 2 : -1;
                    if (ax < 0) throw Invalid("rotation axis must be X, Y or Z");
                    rot[ax] = rotation_about_axis(ax, angle);
                } else {
                    throw Invalid("unknown xform type `" + tt + "`");
                }
            }
            M3 rotation = m3_mul(m3_mul(rot[2], rot[1]), rot[0]);
            Box3 pb = proto_bbox(so->protos[(size_t)s.proto], *so);
            instance_matrices(translate, rotation, scale, pb, name, s);
        } else {
            s = geometric_shape(sh, type, name, data_path, *so);
        }
        s.shader = shader;
        so->shapes.push_back(s);
    }
    if (so->shapes.empty()) throw Invalid("scene has no shapes (the reference's BVH build panics on an empty list)");

    // ---- lights (scene/mod.rs:669-683)
    if (const Value *lights = scene.find({"lights", "light"})) {
        if (!lights->is_array()) throw Invalid("`light` must be an array");
        for (const auto &l : lights->arr) {
            std::string type = require_string(l, {"_type"}, "light");
            RtLight rl{};
            if (type == "point") {
                rl.type = RT_LIGHT_POINT;
                rl.position = vec3_f64(require(l, {"position"}, "light"), "position");
                vec3_f32(require(l, {"intensity"}, "light"), "intensity", rl.intensity);
            } else if (type == "ambient") {
                rl.type = RT_LIGHT_AMBIENT;
                vec3_f32(require(l, {"intensity"}, "light"), "intensity", rl.intensity);
            } else if (type == "area" || type == "shape") {
                throw Unsupported("light type not implemented yet");
            } else {
                throw Invalid("unknown light _type `" + type + "`");
            }
            so->lights.push_back(rl);
        }
    }

    // ---- background (scene/mod.rs:685-697)
    float bg[3] = {0.198f, 0.198f, 0.198f};
    if (args.render_normals) {
        bg[0] = bg[1] = bg[2] = 0.0f;
    } else if (params) {
        if (params->find({"env_map_prefix", "envMapPrefix", "env_map_vert_cross", "envMapVertCross"}))
            throw Unsupported("environment maps aren't implemented yet");
        if (const Value *c = params->find({"background_color", "bgColor", "_bgColor"}))
            vec3_f32(*c, "bgColor", bg);
    }
    std::memcpy(so->scene.background, bg, sizeof(bg));

    // ---- assemble the view
    for (size_t i = 0; i < so->mesh_vertices.size(); ++i) {
        RtMesh m;
        m.n_triangles = so->mesh_vertices[i].size() / 9;
        m.vertices = so->mesh_vertices[i].data();
        if (m.n_triangles == 0) throw Invalid("mesh has no triangles");
        so->meshes.push_back(m);
    }
    RtScene &rs = so->scene;
    rs.recursion_depth = (uint16_t)(args.recursion_depth >= 0 ? args.recursion_depth : 3);
    rs.disable_shadows = args.disable_shadows ? 1 : 0;
    rs.image_width = image_width;
    rs.image_height = image_height;
    rs.n_shapes = (uint32_t)so->shapes.size();
    rs.n_protos = (uint32_t)so->protos.size();
    rs.n_meshes = (uint32_t)so->meshes.size();
    rs.n_shaders = (uint32_t)so->shaders.size();
    rs.n_lights = (uint32_t)so->lights.size();
    rs.shapes = so->shapes.data();
    rs.protos = so->protos.data();
    rs.meshes = so->meshes.data();
    rs.shaders = so->shaders.data();
    rs.lights = so->lights.data();
    return so.release();
}

int parse_guarded(const char *json_text, const std::string &data_path, const RtParseArgs *args,
                  RtScene **scene_out) {
    if (!json_text || !scene_out) return rt::set_error(RT_ERR_INVALID, "null argument");
    *scene_out = nullptr;
    RtParseArgs a{};
    a.recursion_depth = -1;
    if (args) a = *args;
    try {
        SceneOwner *so = build_scene(json_text, data_path, a);
        *scene_out = &so->scene;
        return RT_OK;
    } catch (const Unsupported &e) {
        return rt::set_error(RT_ERR_UNSUPPORTED, e.what());
    } catch (const Invalid &e) {
        return rt::set_error(RT_ERR_INVALID, e.what());
    } catch (const IoError &e) {
        return rt::set_error(RT_ERR_IO, e.what());
    } catch (const std::exception &e) {
        return rt::set_error(RT_ERR_PARSE, e.what());
    }
}

}  // namespace

extern "C" int rt_parse_scene(const char *scene_json, const char *scene_data_path,
                              const RtParseArgs *args, RtScene **scene_out) {
    return parse_guarded(scene_json, scene_data_path ? scene_data_path : "", args, scene_out);
}

extern "C" int rt_parse_scene_file(const char *scene_path, const RtParseArgs *args, RtScene **scene_out) {
    if (!scene_path || !scene_out) return rt::set_error(RT_ERR_INVALID, "null argument");
    *scene_out = nullptr;
    std::ifstream in(scene_path, std::ios::binary);
    if (!in) return rt::set_error(RT_ERR_IO, std::string("cannot read scene file ") + scene_path);
    std::stringstream ss;
    ss << in.rdbuf();
    std::string path(scene_path);
    size_t slash = path.find_last_of('/');
    std::string dir = slash == std::string::npos ? std::string("") : path.substr(0, slash);
    return parse_guarded(ss.str().c_str(), dir, args, scene_out);
}

extern "C" void rt_scene_destroy(RtScene *scene) {
    if (scene) delete reinterpret_cast<SceneOwner *>(scene);
}

extern "C" int rt_framebuffer_to_rgb8(const float *fb_rgb, uint32_t width, uint32_t height,
                                      uint8_t *rgb8_out) {
    if (!fb_rgb || !rgb8_out) return rt::set_error(RT_ERR_INVALID, "null argument");
    for (uint32_t y = 0; y < height; ++y) {
        const float *row = fb_rgb + (size_t)(height - y - 1) * width * 3;
        uint8_t *out = rgb8_out + (size_t)y * width * 3;
        for (size_t k = 0; k < (size_t)width * 3; ++k) {
            float c = row[k];
            // f32::clamp(0, 1): NaN stays NaN, and `NaN as u8` is 0.
            float cl = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
            float r = std::round(cl * 255.0f);  // round half away from zero, like f32::round
            out[k] = (r == r) ? (uint8_t)r : (uint8_t)0;
        }
    }
    return RT_OK;
}
