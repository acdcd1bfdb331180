// gltf_scene.cpp -- host side of the scene rows (include/mipt_scene.h): glTF 2.0 / GLB loader into the reference's data
// model, animation sampler + player, and the per-frame host walk that feeds the path-tracing context.
//
//   class Scene            <- class Gltf (Source/Gltf.h:16-232, Source/Gltf.cpp)
//   convert / AccessorIter <- tinygltf::tools (Source/TinyGltfTools.h:45-389)
//   sample_channel         <- Animation::Channel::GetTransform (Source/Animation.cpp:73-122)
//   gs_frame               <- Renderer::PerformSkinning / GatherLights / GatherMaterials (Source/Renderer.cpp:399-500) and the
//                             instance walk of Pathtracer::BuildTlas (Source/Pathtracer.cpp:185-257)
// tinygltf itself (JSON, base64, GLB chunks, stb image decode) is an empty submodule upstream: restated from the glTF 2.0
// specification in json.h / image_decode.cpp / this file.  Plain C++17, no HIP.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include "../../../include/mipt_scene.h"
#include "glm_lite.h"
#include "image_decode.h"
#include "json.h"

using hostjson::Value;
using namespace glml;

namespace {

thread_local std::string g_error;
int fail(int code, const std::string& msg) { g_error = msg; return code; }

enum { CT_BYTE = 5120, CT_UBYTE = 5121, CT_SHORT = 5122, CT_USHORT = 5123, CT_INT = 5124, CT_UINT = 5125, CT_FLOAT = 5126 };

int component_size(int ct) { switch (ct) { case CT_BYTE: case CT_UBYTE: return 1; case CT_SHORT: case CT_USHORT: return 2; case CT_INT: case CT_UINT: case CT_FLOAT: return 4; default: return 0; } }
int type_components(const std::string& t) {
    if (t == "SCALAR") return 1; if (t == "VEC2") return 2; if (t == "VEC3") return 3; if (t == "VEC4") return 4;
    if (t == "MAT2") return 4; if (t == "MAT3") return 9; if (t == "MAT4") return 16;
    return 0;
}

struct BufferView { int buffer = -1; size_t offset = 0, length = 0, stride = 0; };
struct Accessor {
    int view = -1; size_t offset = 0; int component = 0; bool normalized = false; size_t count = 0; int ncomp = 0;
    bool sparse = false; size_t sparse_count = 0; int si_view = -1; size_t si_offset = 0; int si_component = 0; int sv_view = -1; size_t sv_offset = 0;
    std::vector<double> minv, maxv;
};

struct MorphTarget { int flags = 0; std::vector<float> position; std::vector<uint32_t> tangent_space; int h_position = -1, h_tangent_space = -1; };
struct Primitive {
    int flags = 0, topology = 4, num_vertices = 0, num_indices = 0, index_format = 0, material_id = 0;
    bool valid = true;                     // false: unsupported topology (Gltf.cpp:190-205 returns before creating the mesh)
    std::vector<uint8_t> index; std::vector<float> position; std::vector<uint32_t> tangent_space; std::vector<float> texcoord[2];
    std::vector<uint16_t> color; std::vector<uint16_t> joint_weight;       // 8 x u16 per vertex
    std::vector<MorphTarget> targets;
    int h_index = -1, h_position = -1, h_tangent_space = -1, h_texcoord[2] = {-1, -1}, h_color = -1, h_joint_weight = -1;
};
struct Mesh { std::string name; std::vector<Primitive> prims; std::vector<float> weights; };
struct Trs { vec3 t{0, 0, 0}; quat r{0, 0, 0, 1}; vec3 s{1, 1, 1}; };
struct Node {
    std::string name;
    int child = -1, sibling = -1, mesh = -1, skin = -1, dynamic_mesh = -1, camera = -1, light = -1;
    Trs rest, local;
    mat4 global = identity(), previous_global = identity();
    std::vector<float> weights, current_weights;
};
struct MatTex { int texture = -1, sampler = 0, tex_coord = 0; float offset[2] = {0, 0}, scale[2] = {1, 1}, rotation = 0; };
struct Material {                            // Gltf::Material (Gltf.h:84-176), defaults included
    uint32_t flags = 0;
    float base_color_factor[4] = {1, 1, 1, 1}, metalness_factor = 1, roughness_factor = 1, occlusion_factor = 1, emissive_factor[3] = {0, 0, 0}, normal_map_scale = 1;
    MatTex albedo, normal, metallic_roughness, occlusion, emissive;
    int alpha_mode = 0; float alpha_cutoff = 0.5f;
    float anisotropy_strength = 0, anisotropy_rotation = 0; MatTex anisotropy;
    float clearcoat_factor = 0; MatTex clearcoat; float clearcoat_roughness_factor = 0; MatTex clearcoat_roughness; float clearcoat_normal_scale = 1; MatTex clearcoat_normal;
    float dispersion = 0, emissive_strength = 1, ior = 1.5f;
    float iridescence_factor = 0, iridescence_ior = 1.3f, iridescence_thickness_minimum = 100, iridescence_thickness_maximum = 400; MatTex iridescence, iridescence_thickness;
    float sheen_color_factor[3] = {0, 0, 0}; MatTex sheen_color; float sheen_roughness_factor = 0; MatTex sheen_roughness;
    float specular_factor = 1; MatTex specular; float specular_color_factor[3] = {1, 1, 1}; MatTex specular_color;
    float transmission_factor = 0; MatTex transmission;
    float thickness_factor = 0, attenuation_distance = 0, attenuation_color[3] = {1, 1, 1}; MatTex thickness;
};
struct Texture { std::string name; bool loaded = false, srgb = false; hostimg::Image8 image; int handle = -1; };
struct Light { int type = 0; float color[3] = {1, 1, 1}, intensity = 1, cutoff = 0, inner = 0, outer = 0; };
struct Skin { std::vector<mat4> inverse_bind; std::vector<uint32_t> joints; };
struct Channel {
    int node = -1, path = 0, interpolation = 1, format = 0, width = 0;
    std::vector<float> times; std::vector<uint8_t> transforms;
};
struct Animation { std::string name; float length = 0; std::vector<Channel> channels; };
struct DynamicMesh { int flags = 0, num_vertices = 0, h_position = -1, h_tangent_space = -1; };
struct DynamicPrimitives { std::vector<DynamicMesh> meshes; };

}  // namespace

// Gltf::LoadCameras (Gltf.cpp:642-655) + Camera::GetViewToClip (Camera.h:80-92) for one entry of the file's `cameras` array
static gs_camera_info parse_camera(const Value& c) {
    gs_camera_info info;
    gs_camera_info* o = &info;
    memset(o, 0, sizeof(*o));
    const std::string type = c.get("type").string_or("");
    o->upstream_type_matches = (type == "Perspective" || type == "Orthographic") ? 1 : 0;
    o->type = (type == "perspective" || type == "Perspective") ? 0 : ((type == "orthographic" || type == "Orthographic") ? 1 : -1);
    double m[16] = {0};
    if (o->type == 0) {
        const Value& p = c.get("perspective");
        o->aspect_ratio = (float)p.get("aspectRatio").number_or(0.0); o->y_fov = (float)p.get("yfov").number_or(0.0);
        o->z_near = (float)p.get("znear").number_or(0.0); o->z_far = (float)p.get("zfar").number_or(0.0);
        // Camera::GetViewToClip (Camera.h:80-88): perspectiveRH_ZO(y_fov, aspect, zNear <- far (100000 if 0), zFar <- near); SURVEY section 11
        const double aspect = o->aspect_ratio > 0 ? o->aspect_ratio : 1.0, zn = o->z_far != 0.0f ? o->z_far : 100000.0, zf = o->z_near;
        const double t = tan((double)o->y_fov / 2.0);
        if (t != 0.0 && zn != zf) {
            m[0] = 1.0 / (aspect * t); m[5] = 1.0 / t; m[10] = zf / (zn - zf); m[11] = -1.0; m[14] = -(zf * zn) / (zf - zn);
        }
    } else if (o->type == 1) {
        const Value& p = c.get("orthographic");
        o->x_mag = (float)p.get("xmag").number_or(0.0); o->y_mag = (float)p.get("ymag").number_or(0.0);
        o->z_near = (float)p.get("znear").number_or(0.0); o->z_far = (float)p.get("zfar").number_or(0.0);
        o->aspect_ratio = o->y_mag != 0.0f ? o->x_mag / o->y_mag : 0.0f;                 // Camera::Orthographic, Camera.h:31-40
        // orthoRH_ZO(l = -1/xmag, r = 1/xmag, b = -1/ymag, t = 1/ymag, zNear <- far, zFar <- near) (Camera.h:91)
        if (o->x_mag != 0.0f && o->y_mag != 0.0f && o->z_near != o->z_far) {
            const double zn = o->z_far, zf = o->z_near;
            m[0] = (double)o->x_mag; m[5] = (double)o->y_mag; m[10] = -1.0 / (zf - zn); m[14] = -zn / (zf - zn); m[15] = 1.0;
        }
    }
    for (int k = 0; k < 16; k++) o->view_to_clip[k] = (float)m[k];
    return info;
}

struct gs_scene {
    std::string filename, base_dir;
    Value json;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<BufferView> views;
    std::vector<Accessor> accessors;

    std::vector<std::vector<int>> scenes = std::vector<std::vector<int>>(1);
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<Node> nodes;
    std::vector<Skin> skins;
    std::vector<DynamicPrimitives> dynamic;
    std::vector<Animation> animations;
    std::vector<Light> lights;
    std::vector<Texture> textures;
    std::vector<pt_sampler_desc> samplers;
    std::vector<int> sampler_handles;
    int num_cameras = 0;
    std::vector<gs_camera_info> cameras;   // Gltf::LoadCameras
    bool uploaded = false;

    // ---------------------------------------------------------------- accessors (TinyGltfTools.h)
    int stride_of(const Accessor& a) const {                                  // GetStride :50-58
        if (a.view < 0) return 0;
        const BufferView& v = views[a.view];
        return v.stride == 0 ? component_size(a.component) * a.ncomp : (int)v.stride;
    }
    const uint8_t* view_ptr(int view, size_t extra, size_t need) const {
        if (view < 0 || view >= (int)views.size()) return nullptr;
        const BufferView& v = views[view];
        if (v.buffer < 0 || v.buffer >= (int)buffers.size()) return nullptr;
        const std::vector<uint8_t>& b = buffers[v.buffer];
        if (v.offset + extra + need > b.size()) return nullptr;
        return b.data() + v.offset + extra;
    }
    // Element i of the accessor as a raw pointer (RawIterator::Get :253-264), or nullptr where the reference would read a
    // null base (accessor without bufferView).  Deviation, documented: the reference indexes the sparse VALUE array with
    // the vertex index (`sparse_index * stride`, :258) instead of the running sparse counter, which reads the wrong
    // (possibly out-of-bounds) element; the specified behaviour (values[sparse_i]) is implemented instead.
    struct RawIter {
        const gs_scene* s; const Accessor* a;
        const uint8_t* data = nullptr; int data_stride = 0;
        const uint8_t* sidx = nullptr; int sidx_stride = 0; const uint8_t* sval = nullptr; int sval_stride = 0;
        size_t sparse_i = 0, data_i = 0; uint32_t sparse_index = 0xffffffffu;
        bool ok = true;
        uint32_t sparse_at(size_t i) const {
            const uint8_t* p = sidx + (size_t)sidx_stride * i;
            switch (a->si_component) { case CT_UBYTE: return *p; case CT_USHORT: { uint16_t v; memcpy(&v, p, 2); return v; } case CT_UINT: { uint32_t v; memcpy(&v, p, 4); return v; } default: return 0; }
        }
        RawIter(const gs_scene* sc, const Accessor* ac) : s(sc), a(ac) {
            const int elem = component_size(a->component) * a->ncomp;
            data_stride = s->stride_of(*a);
            if (a->view >= 0) {
                size_t need = a->count ? (a->count - 1) * (size_t)data_stride + elem : 0;
                data = s->view_ptr(a->view, a->offset, need);
                if (!data && a->count) ok = false;
            }
            if (a->sparse && a->sparse_count) {
                int isz = component_size(a->si_component);
                const BufferView* iv = a->si_view >= 0 && a->si_view < (int)s->views.size() ? &s->views[a->si_view] : nullptr;
                const BufferView* vv = a->sv_view >= 0 && a->sv_view < (int)s->views.size() ? &s->views[a->sv_view] : nullptr;
                if (!iv || !vv || !isz) { ok = false; return; }
                sidx_stride = iv->stride ? (int)iv->stride : isz;
                sval_stride = vv->stride ? (int)vv->stride : elem;
                sidx = s->view_ptr(a->si_view, a->si_offset, (a->sparse_count - 1) * (size_t)sidx_stride + isz);
                sval = s->view_ptr(a->sv_view, a->sv_offset, (a->sparse_count - 1) * (size_t)sval_stride + elem);
                if (!sidx || !sval) { ok = false; return; }
                sparse_index = sparse_at(0);
            }
        }
        bool at_end() const { return data_i >= a->count; }
        const uint8_t* get() const {
            if (sidx && sparse_i < a->sparse_count && sparse_index == data_i) return sval + sparse_i * (size_t)sval_stride;
            return data ? data + data_i * (size_t)data_stride : nullptr;
        }
        void next() {
            if (sidx && sparse_i < a->sparse_count && sparse_index == data_i) { sparse_i++; if (sparse_i < a->sparse_count) sparse_index = sparse_at(sparse_i); }
            data_i++;
        }
    };
};

namespace {

// ---- component conversion (TinyGltfTools.h:137-222)
float unpack_normalized(const uint8_t* d, int ct) {                               // UnpackNormalizedValue :137-158
    switch (ct) {
        case CT_UBYTE: return (float)*d / 255.0f;                                // unpackUnorm1x8
        case CT_BYTE: { float v = (float)*(const int8_t*)d / 127.0f; return v < -1.f ? -1.f : (v > 1.f ? 1.f : v); }
        case CT_USHORT: { uint16_t v; memcpy(&v, d, 2); return (float)v / 65535.0f; }
        case CT_SHORT: { int16_t v; memcpy(&v, d, 2); float f = (float)v / 32767.0f; return f < -1.f ? -1.f : (f > 1.f ? 1.f : f); }
        case CT_UINT: { uint32_t v; memcpy(&v, d, 4); return (float)v / 4294967295.0f; }
        case CT_INT: { int32_t v; memcpy(&v, d, 4); float f = (float)v / 2147483647.0f; return f < -1.f ? -1.f : (f > 1.f ? 1.f : f); }
        case CT_FLOAT: { float v; memcpy(&v, d, 4); return v; }
        default: return 0.f;
    }
}
template <typename T> T pack_normalized(float x) {                                // PackNormalizedValue :160-171 (glm::packUnorm / packSnorm)
    if constexpr (std::is_same_v<T, float>) return x;
    else if constexpr (std::is_unsigned_v<T>) { float c = x < 0.f ? 0.f : (x > 1.f ? 1.f : x); return (T)roundf(c * (float)std::numeric_limits<T>::max()); }
    else { float c = x < -1.f ? -1.f : (x > 1.f ? 1.f : x); return (T)roundf(c * (float)std::numeric_limits<T>::max()); }
}
template <typename T> bool same_type(int ct) {
    return (std::is_same_v<T, float> && ct == CT_FLOAT) || (std::is_same_v<T, int32_t> && ct == CT_INT) || (std::is_same_v<T, uint32_t> && ct == CT_UINT) ||
           (std::is_same_v<T, int16_t> && ct == CT_SHORT) || (std::is_same_v<T, uint16_t> && ct == CT_USHORT) || (std::is_same_v<T, int8_t> && ct == CT_BYTE) ||
           (std::is_same_v<T, uint8_t> && ct == CT_UBYTE);
}
template <typename T> T convert_plain(const uint8_t* d, int ct) {                 // Convert<T>(data, type) :173-193: C casts
    switch (ct) {
        case CT_UBYTE: return (T)*d;
        case CT_BYTE: return (T) * (const int8_t*)d;
        case CT_USHORT: { uint16_t v; memcpy(&v, d, 2); return (T)v; }
        case CT_SHORT: { int16_t v; memcpy(&v, d, 2); return (T)v; }
        case CT_UINT: { uint32_t v; memcpy(&v, d, 4); return (T)v; }
        case CT_INT: { int32_t v; memcpy(&v, d, 4); return (T)v; }
        case CT_FLOAT: { float v; memcpy(&v, d, 4); return (T)v; }
        default: return (T)0;
    }
}
template <typename T, bool NORMALIZE> T convert_component(const uint8_t* d, bool normalized, int ct) {   // :195-209
    if (!d) return (T)0;
    if (same_type<T>(ct)) { T v; memcpy(&v, d, sizeof(T)); return v; }
    if (normalized || NORMALIZE) return pack_normalized<T>(unpack_normalized(d, ct));
    return convert_plain<T>(d, ct);
}
// Convert<L, T, NORMALIZE> :211-222: missing components are filled with 1
template <int L, typename T, bool NORMALIZE> void convert_element(const uint8_t* d, const Accessor& a, T* out) {
    int n = a.ncomp < L ? a.ncomp : L;
    for (int i = 0; i < n; i++) out[i] = convert_component<T, NORMALIZE>(d ? d + component_size(a.component) * i : nullptr, a.normalized, a.component);
    for (int i = a.ncomp; i < L; i++) out[i] = (T)1;
}
// Copy<L, T, NORMALIZE> :340-356
template <int L, typename T, bool NORMALIZE = false> bool copy_typed(const gs_scene& s, const Accessor& a, T* out) {
    gs_scene::RawIter it(&s, &a);
    if (!it.ok) return false;
    for (size_t i = 0; !it.at_end(); i++, it.next()) convert_element<L, T, NORMALIZE>(it.get(), a, out + i * L);
    return true;
}
// Copy(std::byte*, ...) :359-375: raw element copy without conversion
bool copy_raw(const gs_scene& s, const Accessor& a, uint8_t* out) {
    gs_scene::RawIter it(&s, &a);
    if (!it.ok) return false;
    const int elem = component_size(a.component) * a.ncomp;
    for (size_t i = 0; !it.at_end(); i++, it.next()) {
        const uint8_t* d = it.get();
        if (d) memcpy(out + i * elem, d, elem); else memset(out + i * elem, 0, elem);
    }
    return true;
}

// ---- tangent-space encoding (Gltf.cpp:23-104)
void encode_octahedral(const float n[3], float out[2]) {
    float l1 = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
    float o[3] = {n[0] / l1, n[1] / l1, n[2] / l1};
    if (o[2] >= 0.f) { out[0] = o[0]; out[1] = o[1]; }
    else { out[0] = (o[0] >= 0.f ? 1.f : -1.f) * (1.f - fabsf(o[1])); out[1] = (o[1] >= 0.f ? 1.f : -1.f) * (1.f - fabsf(o[0])); }
}
void decode_octahedral(const float e[2], float out[3]) {
    float z = (float)(1. - fabs((double)e[0]) - fabs((double)e[1]));              // `1. -` makes this a double expression upstream (:44)
    float x, y;
    if (z >= 0.) { x = e[0]; y = e[1]; }
    else { x = (e[0] >= 0.f ? 1.f : -1.f) * (1.f - fabsf(e[1])); y = (e[1] >= 0.f ? 1.f : -1.f) * (1.f - fabsf(e[0])); }
    float il = 1.0f / sqrtf(x * x + y * y + z * z);                               // glm::normalize = v * inversesqrt(dot(v, v))
    out[0] = x * il; out[1] = y * il; out[2] = z * il;
}
uint32_t quantize10(float v) {                                                    // clamp(v, 0, 1) * 1023 + 0.5 -> uint (:69, :87, :101)
    float c = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return (uint32_t)(c * 1023.0f + 0.5f);
}
uint32_t encode_normal(const float n[3]) {                                        // EncodeNormal :65-77
    float e[2];
    encode_octahedral(n, e);
    return quantize10(0.5f * e[0] + 0.5f) | (quantize10(0.5f * e[1] + 0.5f) << 10) | (3u << 30);
}
uint32_t encode_tangent_space(const float n_in[3], const float t[4]) {            // EncodeTangentSpace :79-104
    float e[2];
    encode_octahedral(n_in, e);
    uint32_t qx = quantize10(0.5f * e[0] + 0.5f), qy = quantize10(0.5f * e[1] + 0.5f);
    float u[2] = {2.0f * ((float)qx / 1023.0f) - 1.0f, 2.0f * ((float)qy / 1023.0f) - 1.0f};
    float n[3];
    decode_octahedral(u, n);
    const float sign = n[2] >= 0.0f ? 1.0f : -1.0f;                                // CreateBasis :55-63
    const float a = -1.0f / (sign + n[2]);
    const float b = n[0] * n[1] * a;
    float ct[3] = {1.0f + sign * n[0] * n[0] * a, sign * b, -sign * n[0]};
    float cb[3] = {b, sign + n[1] * n[1] * a, -n[1]};
    float angle = atan2f(t[0] * cb[0] + t[1] * cb[1] + t[2] * cb[2], t[0] * ct[0] + t[1] * ct[1] + t[2] * ct[2]);
    float enc = (angle / 6.28318530717958647692f) + 0.5f;
    uint32_t qt = quantize10(enc);
    uint32_t qw = t[3] == 1.0f ? 3u : 0u;
    return qx | (qy << 10) | (qt << 20) | (qw << 30);
}

// ---- base64 (data URIs)
bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    static int8_t table[256];
    static bool ready = false;
    if (!ready) {
        memset(table, -1, sizeof(table));
        const char* al = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; i++) table[(uint8_t)al[i]] = (int8_t)i;
        ready = true;
    }
    out.clear();
    out.reserve(n * 3 / 4);
    uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = (uint8_t)s[i];
        if (c == '=' ) break;
        if (c == '\n' || c == '\r' || c == ' ') continue;
        int v = table[c];
        if (v < 0) return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); acc &= (1u << bits) - 1; }
    }
    return true;
}
std::string percent_decode(const std::string& s) {
    std::string o;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((uint8_t)s[i + 1]) && isxdigit((uint8_t)s[i + 2])) { o += (char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; }
        else o += s[i];
    }
    return o;
}
bool load_uri(const gs_scene& sc, const std::string& uri, std::vector<uint8_t>& out, std::string& err) {
    if (uri.compare(0, 5, "data:") == 0) {
        size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma) { err = "unsupported data URI"; return false; }
        if (!base64_decode(uri.c_str() + comma + 1, uri.size() - comma - 1, out)) { err = "bad base64 payload"; return false; }
        return true;
    }
    return hostimg::read_file(sc.base_dir + percent_decode(uri), out, err);
}

// ---- loader
struct Loader {
    gs_scene& s;
    std::string err;
    explicit Loader(gs_scene& sc) : s(sc) {}

    bool parse_container(const std::vector<uint8_t>& file, bool glb) {
        std::vector<uint8_t> bin;
        bool have_bin = false;
        const char* text = (const char*)file.data();
        size_t text_n = file.size();
        if (glb) {
            auto le32 = [&](size_t p) { return (uint32_t)file[p] | ((uint32_t)file[p + 1] << 8) | ((uint32_t)file[p + 2] << 16) | ((uint32_t)file[p + 3] << 24); };
            if (file.size() < 20 || le32(0) != 0x46546C67u) { err = "GLB: bad magic"; return false; }
            if (le32(4) != 2) { err = "GLB: unsupported container version"; return false; }
            size_t total = le32(8);
            if (total > file.size()) { err = "GLB: length exceeds file size"; return false; }
            size_t pos = 12;
            text = nullptr;
            while (pos + 8 <= total) {
                uint32_t clen = le32(pos), ctype = le32(pos + 4);
                if (pos + 8 + clen > total) { err = "GLB: truncated chunk"; return false; }
                if (ctype == 0x4E4F534Au && !text) { text = (const char*)file.data() + pos + 8; text_n = clen; }
                else if (ctype == 0x004E4942u && !have_bin) { bin.assign(file.begin() + pos + 8, file.begin() + pos + 8 + clen); have_bin = true; }
                pos += 8 + (size_t)clen + ((4 - (clen & 3)) & 3);
            }
            if (!text) { err = "GLB: no JSON chunk"; return false; }
        }
        hostjson::Parser p;
        if (!p.parse(text, text_n, s.json, err)) return false;
        if (!s.json.is_object()) { err = "glTF: top level is not an object"; return false; }
        const Value& asset = s.json.get("asset");
        if (!asset.is_object() || asset.get("version").string_or("").compare(0, 2, "2.") != 0) { err = "glTF: asset.version 2.x required"; return false; }
        // buffers
        const Value& bufs = s.json.get("buffers");
        for (size_t i = 0; i < bufs.size(); i++) {
            const Value& b = bufs.at(i);
            std::vector<uint8_t> data;
            if (b.has("uri")) { if (!load_uri(s, b.get("uri").string_or(""), data, err)) return false; }
            else if (glb && i == 0 && have_bin) data = bin;
            else { err = "glTF: buffer without uri"; return false; }
            size_t want = (size_t)b.get("byteLength").number_or(0);
            if (data.size() < want) { err = "glTF: buffer shorter than byteLength"; return false; }
            s.buffers.push_back(std::move(data));
        }
        const Value& vs = s.json.get("bufferViews");
        for (size_t i = 0; i < vs.size(); i++) {
            const Value& v = vs.at(i);
            BufferView bv;
            bv.buffer = v.get("buffer").int_or(-1);
            bv.offset = (size_t)v.get("byteOffset").number_or(0);
            bv.length = (size_t)v.get("byteLength").number_or(0);
            bv.stride = (size_t)v.get("byteStride").number_or(0);
            if (bv.buffer < 0 || bv.buffer >= (int)s.buffers.size() || bv.offset + bv.length > s.buffers[bv.buffer].size()) { err = "glTF: bufferView out of range"; return false; }
            s.views.push_back(bv);
        }
        const Value& as = s.json.get("accessors");
        for (size_t i = 0; i < as.size(); i++) {
            const Value& a = as.at(i);
            Accessor ac;
            ac.view = a.get("bufferView").int_or(-1);
            ac.offset = (size_t)a.get("byteOffset").number_or(0);
            ac.component = a.get("componentType").int_or(0);
            ac.normalized = a.get("normalized").type == Value::Bool && a.get("normalized").b;
            ac.count = (size_t)a.get("count").number_or(0);
            ac.ncomp = type_components(a.get("type").string_or(""));
            if (!component_size(ac.component) || !ac.ncomp) { err = "glTF: accessor with bad componentType or type"; return false; }
            if (ac.view >= (int)s.views.size()) { err = "glTF: accessor bufferView out of range"; return false; }
            for (size_t k = 0; k < a.get("min").size(); k++) ac.minv.push_back(a.get("min").at(k).number_or(0));
            for (size_t k = 0; k < a.get("max").size(); k++) ac.maxv.push_back(a.get("max").at(k).number_or(0));
            const Value& sp = a.get("sparse");
            if (sp.is_object()) {
                ac.sparse = true;
                ac.sparse_count = (size_t)sp.get("count").number_or(0);
                ac.si_view = sp.get("indices").get("bufferView").int_or(-1);
                ac.si_offset = (size_t)sp.get("indices").get("byteOffset").number_or(0);
                ac.si_component = sp.get("indices").get("componentType").int_or(0);
                ac.sv_view = sp.get("values").get("bufferView").int_or(-1);
                ac.sv_offset = (size_t)sp.get("values").get("byteOffset").number_or(0);
            }
            s.accessors.push_back(ac);
        }
        return true;
    }

    const Accessor* accessor(int i) { if (i < 0 || i >= (int)s.accessors.size()) { err = "glTF: accessor index out of range"; return nullptr; } return &s.accessors[i]; }
    static int attr(const Value& attrs, const char* name) { const Value* v = attrs.find(name); return v && v->is_number() ? (int)v->num : -1; }

    bool load_tangent_space(int normal_acc, int tangent_acc, std::vector<uint32_t>& out) {        // Gltf.cpp:258-285, 341-368
        const Accessor* na = accessor(normal_acc);
        if (!na) return false;
        gs_scene::RawIter nit(&s, na);
        if (!nit.ok) { err = "glTF: NORMAL accessor out of range"; return false; }
        if (tangent_acc >= 0) {
            const Accessor* ta = accessor(tangent_acc);
            if (!ta) return false;
            gs_scene::RawIter tit(&s, ta);
            if (!tit.ok) { err = "glTF: TANGENT accessor out of range"; return false; }
            size_t i = 0;
            for (; !nit.at_end() && !tit.at_end() && i < out.size(); nit.next(), tit.next(), i++) {
                float n[3], t[4];
                convert_element<3, float, false>(nit.get(), *na, n);
                convert_element<4, float, false>(tit.get(), *ta, t);
                out[i] = encode_tangent_space(n, t);
            }
        } else {
            size_t i = 0;
            for (; !nit.at_end() && i < out.size(); nit.next(), i++) {
                float n[3];
                convert_element<3, float, false>(nit.get(), *na, n);
                out[i] = encode_normal(n);
            }
        }
        return true;
    }

    bool load_primitive(const Value& gp, Primitive& p) {                                            // Gltf::LoadPrimitive :178-319
        int mode = gp.get("mode").int_or(4);
        p.topology = mode;
        if (mode == 2 || mode == 6) { p.valid = false; return true; }                               // line loop / triangle fan: skipped with a warning
        const Value& attrs = gp.get("attributes");
        int pos_acc = attr(attrs, "POSITION");
        if (pos_acc < 0) { err = "glTF: primitive without POSITION"; return false; }
        int idx_acc = gp.get("indices").int_or(-1);
        int nrm = attr(attrs, "NORMAL"), tan = attr(attrs, "TANGENT"), tc0 = attr(attrs, "TEXCOORD_0"), tc1 = attr(attrs, "TEXCOORD_1"), col = attr(attrs, "COLOR_0"),
            jnt = attr(attrs, "JOINTS_0"), wgt = attr(attrs, "WEIGHTS_0");
        p.flags = (idx_acc != -1 ? PT_MESH_FLAG_INDEX : 0) | (nrm >= 0 ? PT_MESH_FLAG_TANGENT_SPACE : 0) | (tc0 >= 0 ? PT_MESH_FLAG_TEXCOORD_0 : 0) |
                  (tc1 >= 0 ? PT_MESH_FLAG_TEXCOORD_1 : 0) | (col >= 0 ? PT_MESH_FLAG_COLOR : 0) | ((jnt >= 0 && wgt >= 0) ? PT_MESH_FLAG_JOINT_WEIGHT : 0);
        const Accessor* pa = accessor(pos_acc);
        if (!pa) return false;
        p.num_vertices = (int)pa->count;
        if (p.flags & PT_MESH_FLAG_INDEX) {
            const Accessor* ia = accessor(idx_acc);
            if (!ia) return false;
            p.num_indices = (int)ia->count;
            if (ia->ncomp != 1) { err = "glTF: index accessor must be SCALAR"; return false; }       // (the raw copy below writes ncomp components per element)
            if (ia->component == CT_UBYTE || ia->component == CT_USHORT) p.index_format = PT_FORMAT_R16_UINT;
            else if (ia->component == CT_UINT) p.index_format = PT_FORMAT_R32_UINT;
            else { err = "glTF: index accessor must be u8 / u16 / u32"; return false; }
            bool ok;
            if (ia->component == CT_UBYTE) { p.index.resize(ia->count * 2); ok = copy_typed<1, uint16_t>(s, *ia, (uint16_t*)p.index.data()); }     // 8-bit indices widened (:242-244)
            else { p.index.resize(ia->count * (size_t)component_size(ia->component)); ok = copy_raw(s, *ia, p.index.data()); }
            if (!ok) { err = "glTF: index accessor out of range"; return false; }
            // host-side guard the reference does not have: an index beyond the vertex count would fault the GPU
            for (size_t i = 0; i < ia->count; i++) {
                uint32_t v = p.index_format == PT_FORMAT_R16_UINT ? ((const uint16_t*)p.index.data())[i] : ((const uint32_t*)p.index.data())[i];
                if (v >= (uint32_t)p.num_vertices) { err = "glTF: index out of the POSITION accessor's range"; return false; }
            }
        }
        p.position.resize((size_t)p.num_vertices * 3);
        if (!copy_typed<3, float>(s, *pa, p.position.data())) { err = "glTF: POSITION accessor out of range"; return false; }
        if (p.flags & PT_MESH_FLAG_TANGENT_SPACE) { p.tangent_space.assign(p.num_vertices, 0); if (!load_tangent_space(nrm, tan, p.tangent_space)) return false; }
        for (int k = 0; k < 2; k++) {
            int acc = k ? tc1 : tc0;
            if (acc < 0) continue;
            const Accessor* a = accessor(acc);
            if (!a) return false;
            p.texcoord[k].assign((size_t)p.num_vertices * 2, 0.f);
            if (a->count > (size_t)p.num_vertices) { err = "glTF: TEXCOORD accessor longer than POSITION"; return false; }
            if (!copy_typed<2, float>(s, *a, p.texcoord[k].data())) { err = "glTF: TEXCOORD accessor out of range"; return false; }
        }
        if (p.flags & PT_MESH_FLAG_COLOR) {
            const Accessor* a = accessor(col);
            if (!a) return false;
            if (a->count > (size_t)p.num_vertices) { err = "glTF: COLOR_0 accessor longer than POSITION"; return false; }
            p.color.assign((size_t)p.num_vertices * 4, 0);
            if (!copy_typed<4, uint16_t, true>(s, *a, p.color.data())) { err = "glTF: COLOR_0 accessor out of range"; return false; }
        }
        if (p.flags & PT_MESH_FLAG_JOINT_WEIGHT) {                                                  // :296-308
            const Accessor *ja = accessor(jnt), *wa = accessor(wgt);
            if (!ja || !wa) return false;
            if (ja->count > (size_t)p.num_vertices || wa->count > (size_t)p.num_vertices) { err = "glTF: JOINTS_0 / WEIGHTS_0 longer than POSITION"; return false; }
            p.joint_weight.assign((size_t)p.num_vertices * 8, 0);
            std::vector<uint16_t> tmp((size_t)p.num_vertices * 4, 0);
            if (!copy_typed<4, uint16_t, false>(s, *ja, tmp.data())) { err = "glTF: JOINTS_0 accessor out of range"; return false; }
            for (size_t i = 0; i < ja->count; i++) memcpy(&p.joint_weight[i * 8], &tmp[i * 4], 8);
            if (!copy_typed<4, uint16_t, true>(s, *wa, tmp.data())) { err = "glTF: WEIGHTS_0 accessor out of range"; return false; }
            for (size_t i = 0; i < wa->count; i++) memcpy(&p.joint_weight[i * 8 + 4], &tmp[i * 4], 8);
        }
        p.material_id = gp.get("material").int_or(-1) + 1;                                          // :311-312
        const Value& targets = gp.get("targets");
        p.targets.resize(targets.size());
        for (size_t t = 0; t < targets.size(); t++) {                                               // CreateMorphTarget :321-369
            const Value& tg = targets.at(t);
            MorphTarget& mt = p.targets[t];
            int tp = attr(tg, "POSITION"), tn = attr(tg, "NORMAL"), tt = attr(tg, "TANGENT");
            mt.flags = (tp >= 0 ? 1 : 0) | (tn >= 0 ? 2 : 0);
            if (tp >= 0) {
                const Accessor* a = accessor(tp);
                if (!a) return false;
                if (a->count > (size_t)p.num_vertices) { err = "glTF: morph POSITION longer than the mesh"; return false; }
                mt.position.assign((size_t)p.num_vertices * 3, 0.f);
                if (!copy_typed<3, float>(s, *a, mt.position.data())) { err = "glTF: morph POSITION accessor out of range"; return false; }
            }
            if (tn >= 0) { mt.tangent_space.assign(p.num_vertices, 0); if (!load_tangent_space(tn, tt, mt.tangent_space)) return false; }
        }
        return true;
    }

    // ---- materials
    static void texture_transform(const Value& v, MatTex& t) {                                      // GetTextureTransform :371-400
        t.offset[0] = t.offset[1] = 0; t.rotation = 0; t.scale[0] = t.scale[1] = 1;
        if (!v.is_object()) return;
        const Value& o = v.get("offset");
        if (o.size() == 2) { t.offset[0] = (float)o.at(0).number_or(0); t.offset[1] = (float)o.at(1).number_or(0); }
        const Value& r = v.get("rotation");
        if (r.is_number()) t.rotation = (float)r.num;
        const Value& sc = v.get("scale");
        if (sc.size() == 2) { t.scale[0] = (float)sc.at(0).number_or(0); t.scale[1] = (float)sc.at(1).number_or(0); }
        const Value& tc = v.get("texCoord");
        if (tc.is_number() && tc.is_int) { int val = (int)tc.num; if (val >= 0 && val < 2) t.tex_coord = val; }
    }
    bool ensure_texture(int source, bool srgb) {                                                    // LoadTexture :1047-1077 (decode part)
        Texture& t = s.textures[source];
        if (t.loaded) return true;
        const Value& img = s.json.get("images").at(source);
        std::vector<uint8_t> bytes;
        if (img.has("uri")) { if (!load_uri(s, img.get("uri").string_or(""), bytes, err)) return false; }
        else {
            int bv = img.get("bufferView").int_or(-1);
            if (bv < 0 || bv >= (int)s.views.size()) { err = "glTF: image without uri or bufferView"; return false; }
            const BufferView& v = s.views[bv];
            bytes.assign(s.buffers[v.buffer].begin() + v.offset, s.buffers[v.buffer].begin() + v.offset + v.length);
        }
        if (!hostimg::decode_image8(bytes.data(), bytes.size(), t.image, err)) { err = "image " + std::to_string(source) + ": " + err; return false; }
        t.name = img.get("name").string_or("");
        t.loaded = true; t.srgb = srgb;
        return true;
    }
    bool get_texture(int texture_index, int tex_coord, const Value& transform, bool srgb, MatTex& out) {   // GetTexture :402-427
        out = MatTex();
        if (texture_index == -1) return true;
        const Value& tex = s.json.get("textures").at(texture_index);
        if (!tex.is_object()) { err = "glTF: texture index out of range"; return false; }
        int source = tex.get("source").int_or(-1);
        if (source == -1) return true;
        if (source < 0 || source >= (int)s.textures.size()) { err = "glTF: texture source out of range"; return false; }
        if (!ensure_texture(source, srgb)) return false;
        int smp = tex.get("sampler").int_or(-1);
        if (smp >= (int)s.samplers.size()) { err = "glTF: sampler index out of range"; return false; }
        out.texture = source;
        out.sampler = smp == -1 ? 0 : smp + 1;
        out.tex_coord = tex_coord < 2 ? tex_coord : 0;
        if (out.tex_coord < 0) out.tex_coord = 0;
        texture_transform(transform, out);
        return true;
    }
    // the tinygltf::TextureInfo flavours (:429-447): a missing object has index -1, texCoord 0
    bool core_texture(const Value& info, bool srgb, MatTex& out, float* scale = nullptr, const char* scale_key = "scale") {
        if (!info.is_object()) { out = MatTex(); return true; }
        if (scale && info.get(scale_key).is_number()) *scale = (float)info.get(scale_key).num;
        const Value& ext = info.get("extensions");
        return get_texture(info.get("index").int_or(-1), info.get("texCoord").int_or(0), ext.is_object() ? ext.get("KHR_texture_transform") : Value(), srgb, out);
    }
    // GetTexture(const tinygltf::Value*, scale, srgb) :449-474: for extension objects; GetNumberAsInt of a missing index is 0
    bool ext_texture(const Value& info, float* scale, bool srgb, MatTex& out) {
        out = MatTex();
        if (!info.is_object()) return true;
        int index = info.get("index").is_number() ? (int)info.get("index").num : 0;
        int tc = info.get("texCoord").is_number() ? (int)info.get("texCoord").num : 0;
        if (scale && info.get("scale").is_number()) *scale = (float)info.get("scale").num;
        const Value& ext = info.get("extensions");
        return get_texture(index, tc, ext.is_object() ? ext.get("KHR_texture_transform") : Value(), srgb, out);
    }
    static void get_f(const Value& o, const char* k, float* out) { if (o.has(k)) *out = (float)o.get(k).number_or(0); }      // tools::GetValue :377-382
    template <int L> static void get_v(const Value& o, const char* k, float* out) {                                            // :384-396
        const Value& v = o.get(k);
        if (v.is_array() && v.size() == (size_t)L) for (int i = 0; i < L; i++) out[i] = (float)v.at(i).number_or(0);
    }
    bool load_materials() {                                                                          // LoadMaterials :476-633
        const Value& ms = s.json.get("materials");
        s.materials.assign(ms.size() + 1, Material());
        for (size_t i = 0; i < ms.size(); i++) {
            const Value& gm = ms.at(i);
            Material& m = s.materials[i + 1];
            const Value& pbr = gm.get("pbrMetallicRoughness");
            if (!core_texture(gm.get("normalTexture"), false, m.normal, &m.normal_map_scale)) return false;
            if (!core_texture(pbr.get("baseColorTexture"), true, m.albedo)) return false;
            get_v<4>(pbr, "baseColorFactor", m.base_color_factor);
            if (!core_texture(pbr.get("metallicRoughnessTexture"), false, m.metallic_roughness)) return false;
            m.metalness_factor = (float)pbr.get("metallicFactor").number_or(1.0);
            m.roughness_factor = (float)pbr.get("roughnessFactor").number_or(1.0);
            if (!core_texture(gm.get("occlusionTexture"), false, m.occlusion)) return false;
            if (!core_texture(gm.get("emissiveTexture"), true, m.emissive)) return false;
            get_v<3>(gm, "emissiveFactor", m.emissive_factor);
            std::string am = gm.get("alphaMode").string_or("OPAQUE");
            if (am == "OPAQUE") m.alpha_mode = 0; else if (am == "MASK") m.alpha_mode = 1; else if (am == "BLEND") m.alpha_mode = 2;
            m.alpha_cutoff = (float)gm.get("alphaCutoff").number_or(0.5);
            if (gm.get("doubleSided").type == Value::Bool && gm.get("doubleSided").b) m.flags |= 1u;
            const Value& ex = gm.get("extensions");
            const Value* e;
            if ((e = ex.find("KHR_materials_anisotropy"))) {
                get_f(*e, "anisotropyStrength", &m.anisotropy_strength); get_f(*e, "anisotropyRotation", &m.anisotropy_rotation);
                if (!ext_texture(e->get("anisotropyTexture"), nullptr, false, m.anisotropy)) return false;
            }
            if ((e = ex.find("KHR_materials_clearcoat"))) {
                get_f(*e, "clearcoatFactor", &m.clearcoat_factor); get_f(*e, "clearcoatRoughnessFactor", &m.clearcoat_roughness_factor);
                if (!ext_texture(e->get("clearcoatTexture"), nullptr, false, m.clearcoat)) return false;
                if (!ext_texture(e->get("clearcoatRoughnessTexture"), nullptr, false, m.clearcoat_roughness)) return false;
                if (!ext_texture(e->get("clearcoatNormalTexture"), &m.clearcoat_normal_scale, false, m.clearcoat_normal)) return false;
            }
            if ((e = ex.find("KHR_dispersion"))) get_f(*e, "dispersion", &m.dispersion);
            if ((e = ex.find("KHR_materials_emissive_strength"))) get_f(*e, "emissiveStrength", &m.emissive_strength);
            if ((e = ex.find("KHR_materials_ior"))) get_f(*e, "ior", &m.ior);
            if ((e = ex.find("KHR_materials_iridescence"))) {
                get_f(*e, "iridescenceFactor", &m.iridescence_factor); get_f(*e, "iridescenceIor", &m.iridescence_ior);
                get_f(*e, "iridescenceThicknessMinimum", &m.iridescence_thickness_minimum); get_f(*e, "iridescenceThicknessMaximum", &m.iridescence_thickness_maximum);
                if (!ext_texture(e->get("iridescenceTexture"), nullptr, false, m.iridescence)) return false;
                if (!ext_texture(e->get("iridescenceThicknessTexture"), nullptr, false, m.iridescence_thickness)) return false;
            }
            if ((e = ex.find("KHR_materials_sheen"))) {
                get_v<3>(*e, "sheenColorFactor", m.sheen_color_factor); get_f(*e, "sheenRoughnessFactor", &m.sheen_roughness_factor);
                if (!ext_texture(e->get("sheenColorTexture"), nullptr, true, m.sheen_color)) return false;
                if (!ext_texture(e->get("sheenRoughnessTexture"), nullptr, false, m.sheen_roughness)) return false;
            }
            if ((e = ex.find("KHR_materials_specular"))) {
                get_f(*e, "specularFactor", &m.specular_factor); get_v<3>(*e, "specularColorFactor", m.specular_color_factor);
                if (!ext_texture(e->get("specularTexture"), nullptr, false, m.specular)) return false;
                if (!ext_texture(e->get("specularColorTexture"), nullptr, true, m.specular_color)) return false;
            }
            if ((e = ex.find("KHR_materials_transmission"))) {
                get_f(*e, "transmissionFactor", &m.transmission_factor);
                if (!ext_texture(e->get("transmissionTexture"), nullptr, false, m.transmission)) return false;
            }
            if ((e = ex.find("KHR_materials_volume"))) {
                get_f(*e, "thicknessFactor", &m.thickness_factor);
                if (!ext_texture(e->get("thicknessTexture"), nullptr, false, m.thickness)) return false;
                get_f(*e, "attenuationDistance", &m.attenuation_distance); get_v<3>(*e, "attenuationColor", m.attenuation_color);
            }
            if (ex.find("KHR_materials_unlit")) m.flags |= 2u;
        }
        return true;
    }

    bool load_samplers() {                                                                           // LoadSamplers :836-852, TinyGltfTools.h:16-43
        const Value& ss = s.json.get("samplers");
        for (size_t i = 0; i < ss.size(); i++) {
            const Value& g = ss.at(i);
            auto addr = [](int w) { return w == 33071 ? PT_ADDRESS_CLAMP : (w == 33648 ? PT_ADDRESS_MIRROR : PT_ADDRESS_WRAP); };
            int minf = g.get("minFilter").int_or(-1), magf = g.get("magFilter").int_or(-1);
            pt_sampler_desc d;
            d.address_u = addr(g.get("wrapS").int_or(10497)); d.address_v = addr(g.get("wrapT").int_or(10497));
            d.min_filter = (minf == 9728 || minf == 9986 || minf == 9984) ? PT_FILTER_POINT : PT_FILTER_LINEAR;
            d.mag_filter = magf == 9728 ? PT_FILTER_POINT : PT_FILTER_LINEAR;
            s.samplers.push_back(d);
        }
        return true;
    }

    bool load_nodes() {                                                                              // LoadNodes :654-706
        const Value& ns = s.json.get("nodes");
        s.nodes.assign(ns.size(), Node());
        for (size_t i = 0; i < ns.size(); i++) {
            const Value& g = ns.at(i);
            Node& n = s.nodes[i];
            n.name = g.get("name").string_or("");
            const Value& mx = g.get("matrix");
            if (mx.size() != 0) {
                if (mx.size() != 16) { err = "glTF: node.matrix must have 16 entries"; return false; }
                mat4 m;
                for (int k = 0; k < 16; k++) m.m[k] = (float)mx.at(k).number_or(0);
                if (!decompose(m, n.rest.s, n.rest.r, n.rest.t)) { n.rest = Trs(); }
            } else {
                const Value &t = g.get("translation"), &r = g.get("rotation"), &sc = g.get("scale");
                n.rest.t = t.size() ? vec3{(float)t.at(0).number_or(0), (float)t.at(1).number_or(0), (float)t.at(2).number_or(0)} : vec3{0, 0, 0};
                // a default-constructed glm::quat is the identity under GLM_FORCE_CTOR_INIT only; upstream leaves it
                // uninitialised.  Identity is what every glTF viewer means.
                n.rest.r = r.size() ? quat{(float)r.at(0).number_or(0), (float)r.at(1).number_or(0), (float)r.at(2).number_or(0), (float)r.at(3).number_or(1)} : quat{0, 0, 0, 1};
                n.rest.s = sc.size() ? vec3{(float)sc.at(0).number_or(1), (float)sc.at(1).number_or(1), (float)sc.at(2).number_or(1)} : vec3{1, 1, 1};
            }
            n.local = n.rest;
            n.mesh = g.get("mesh").int_or(-1);
            n.skin = g.get("skin").int_or(-1);
            if (n.mesh >= (int)s.meshes.size() || n.skin >= (int)s.json.get("skins").size()) { err = "glTF: node mesh / skin index out of range"; return false; }
            for (size_t k = 0; k < g.get("weights").size(); k++) n.weights.push_back((float)g.get("weights").at(k).number_or(0));
            if (n.mesh != -1 && !s.meshes[n.mesh].prims.empty()) n.current_weights.assign(s.meshes[n.mesh].prims[0].targets.size(), 0.0f);
            n.camera = g.get("camera").int_or(-1);
            const Value& le = g.get("extensions").get("KHR_lights_punctual");
            n.light = le.is_object() ? le.get("light").int_or(-1) : -1;
            const Value& ch = g.get("children");
            if (ch.size() > 0) {                                                                     // child / sibling binary tree :697-703
                for (size_t k = 0; k < ch.size(); k++) { int c = ch.at(k).int_or(-1); if (c < 0 || c >= (int)ns.size()) { err = "glTF: child index out of range"; return false; } }
                n.child = ch.at(0).int_or(-1);
                for (size_t k = 1; k < ch.size(); k++) s.nodes[ch.at(k - 1).int_or(0)].sibling = ch.at(k).int_or(-1);
            }
        }
        // guard against cycles, which would hang the traversal (the reference would recurse forever)
        std::vector<int> seen(s.nodes.size(), 0);
        std::function<bool(int, int)> walk = [&](int n, int depth) -> bool {
            if (depth > (int)s.nodes.size()) return false;
            for (int c = s.nodes[n].child; c != -1; c = s.nodes[c].sibling) { if (++seen[c] > 1) return false; if (!walk(c, depth + 1)) return false; }
            return true;
        };
        for (size_t i = 0; i < s.nodes.size(); i++) { std::fill(seen.begin(), seen.end(), 0); if (!walk((int)i, 0)) { err = "glTF: node hierarchy is not a tree"; return false; } }
        return true;
    }

    bool load_skins() {                                                                              // LoadSkins :807-834
        const Value& sk = s.json.get("skins");
        for (size_t i = 0; i < sk.size(); i++) {
            Skin skin;
            const Value& js = sk.at(i).get("joints");
            for (size_t k = 0; k < js.size(); k++) {
                int jn = js.at(k).int_or(-1);
                if (jn < 0 || jn >= (int)s.nodes.size()) { err = "glTF: skin joint out of range"; return false; }
                skin.joints.push_back((uint32_t)jn);
            }
            skin.inverse_bind.assign(js.size(), identity());
            int ibm = sk.at(i).get("inverseBindMatrices").int_or(-1);
            if (ibm != -1) {
                const Accessor* a = accessor(ibm);
                if (!a) return false;
                if (a->ncomp != 16 || a->component != CT_FLOAT || a->count > js.size()) { err = "glTF: inverseBindMatrices must be float MAT4, one per joint"; return false; }
                if (!copy_raw(s, *a, (uint8_t*)skin.inverse_bind.data())) { err = "glTF: inverseBindMatrices accessor out of range"; return false; }
            }
            // (without the accessor upstream push_backs identities AFTER resizing, doubling the vector; only the first joints.size()
            // entries are ever read, and those are identity either way)
            s.skins.push_back(std::move(skin));
        }
        return true;
    }

    bool load_animations() {                                                                         // LoadAnimations :708-805
        const Value& an = s.json.get("animations");
        for (size_t i = 0; i < an.size(); i++) {
            const Value& ga = an.at(i);
            Animation anim;
            anim.name = ga.get("name").string_or("");
            const Value &chs = ga.get("channels"), &sms = ga.get("samplers");
            for (size_t j = 0; j < chs.size(); j++) {
                const Value& gc = chs.at(j);
                const Value& smp = sms.at(gc.get("sampler").int_or(-1) < 0 ? sms.size() : (size_t)gc.get("sampler").int_or(0));
                if (!smp.is_object()) { err = "glTF: animation sampler out of range"; return false; }
                anim.channels.emplace_back();                                                       // emplaced before validation, like upstream
                Channel& c = anim.channels.back();
                c.node = gc.get("target").get("node").int_or(-1);
                std::string path = gc.get("target").get("path").string_or("");
                if (path == "rotation") c.path = 1; else if (path == "translation") c.path = 0; else if (path == "scale") c.path = 2; else if (path == "weights") c.path = 3;
                else continue;
                std::string ip = smp.get("interpolation").string_or("LINEAR");
                if (ip == "STEP") c.interpolation = 0; else if (ip == "LINEAR") c.interpolation = 1; else if (ip == "CUBICSPLINE") c.interpolation = 2;
                else continue;
                if (c.node < 0 || c.node >= (int)s.nodes.size()) { err = "glTF: animation target node out of range"; return false; }
                const Accessor* in = accessor(smp.get("input").int_or(-1));
                if (!in) return false;
                c.times.assign(in->count, 0.f);
                if (!copy_typed<1, float>(s, *in, c.times.data())) { err = "glTF: animation input accessor out of range"; return false; }
                float end_time = in->maxv.empty() ? (c.times.empty() ? 0.f : c.times.back()) : (float)in->maxv[0];
                const Accessor* outa = accessor(smp.get("output").int_or(-1));
                if (!outa) return false;
                switch (outa->component) {                                                           // :760-779 (SHORT maps to SNORM_8 upstream; kept)
                    case CT_FLOAT: c.format = 0; break;
                    case CT_USHORT: c.format = 2; break;
                    case CT_SHORT: c.format = 3; break;
                    case CT_UBYTE: c.format = 1; break;
                    case CT_BYTE: c.format = 3; break;
                    default: continue;
                }
                if (c.path == 3) {
                    int mesh = s.nodes[c.node].mesh;
                    c.width = (mesh >= 0 && !s.meshes[mesh].prims.empty()) ? (int)s.meshes[mesh].prims[0].targets.size() : 0;
                } else c.width = c.path == 1 ? 4 : 3;
                size_t num_values = (size_t)c.width * outa->count;
                if (c.path == 3) num_values = outa->count;
                if (c.interpolation == 2) num_values *= 3;
                size_t bytes = num_values * (size_t)component_size(outa->component);
                size_t have = outa->count * (size_t)outa->ncomp * (size_t)component_size(outa->component);
                c.transforms.assign(std::max(bytes, have), 0);                                       // upstream sizes by `bytes` and copies `have` (can overflow); take the larger
                if (!copy_raw(s, *outa, c.transforms.data())) { err = "glTF: animation output accessor out of range"; return false; }
                anim.length = std::max(anim.length, end_time);
            }
            s.animations.push_back(std::move(anim));
        }
        return true;
    }

    bool load_lights() {                                                                             // LoadLights :854-881
        const Value& ls = s.json.get("extensions").get("KHR_lights_punctual").get("lights");
        for (size_t i = 0; i < ls.size(); i++) {
            const Value& g = ls.at(i);
            Light l;
            std::string t = g.get("type").string_or("");
            if (t == "directional") l.type = PT_LIGHT_DIRECTIONAL; else if (t == "point") l.type = PT_LIGHT_POINT; else if (t == "spot") l.type = PT_LIGHT_SPOT;
            l.intensity = (float)g.get("intensity").number_or(1.0);
            l.cutoff = (float)g.get("range").number_or(0.0);
            if (g.get("color").size() == 3) for (int k = 0; k < 3; k++) l.color[k] = (float)g.get("color").at(k).number_or(1.0);
            l.inner = (float)g.get("spot").get("innerConeAngle").number_or(0.0);
            l.outer = (float)g.get("spot").get("outerConeAngle").number_or(0.7853981634);
            s.lights.push_back(l);
        }
        for (auto& n : s.nodes) if (n.light >= (int)s.lights.size()) { err = "glTF: node light index out of range"; return false; }
        return true;
    }

    bool load(const std::string& path) {
        size_t dot = path.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : path.substr(dot);
        bool glb;
        if (ext == ".glb") glb = true; else if (ext == ".gltf") glb = false; else { err = "unsupported file extension (need .gltf or .glb)"; return false; }   // Gltf.cpp:893-908
        size_t slash = path.find_last_of("/\\");
        s.base_dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
        s.filename = slash == std::string::npos ? path : path.substr(slash + 1);
        std::vector<uint8_t> file;
        if (!hostimg::read_file(path, file, err)) return false;
        if (!parse_container(file, glb)) return false;
        static const char* allowed[] = {"KHR_lights_punctual", "KHR_texture_transform", "KHR_materials_ior", "KHR_materials_specular", "KHR_materials_anisotropy", "KHR_materials_sheen"};
        const Value& req = s.json.get("extensionsRequired");                                         // :919-933
        for (size_t i = 0; i < req.size(); i++) {
            bool ok = false;
            for (auto a : allowed) ok = ok || req.at(i).string_or("") == a;
            if (!ok) { err = "required extension not supported: " + req.at(i).string_or("?"); return false; }
        }
        if (!load_samplers()) return false;
        s.textures.assign(s.json.get("images").size(), Texture());                                   // ReserveTextures
        const Value& ms = s.json.get("meshes");                                                      // LoadMeshes / LoadMesh :157-176
        s.meshes.assign(ms.size(), Mesh());
        for (size_t i = 0; i < ms.size(); i++) {
            const Value& gm = ms.at(i);
            Mesh& m = s.meshes[i];
            m.name = gm.get("name").string_or("");
            const Value& ps = gm.get("primitives");
            m.prims.assign(ps.size(), Primitive());
            for (size_t k = 0; k < ps.size(); k++) if (!load_primitive(ps.at(k), m.prims[k])) { err = "mesh " + std::to_string(i) + " primitive " + std::to_string(k) + ": " + err; return false; }
            for (size_t k = 0; k < gm.get("weights").size(); k++) m.weights.push_back((float)gm.get("weights").at(k).number_or(0));
        }
        if (!load_materials()) return false;
        for (auto& m : s.meshes) for (auto& p : m.prims) if (p.material_id < 0 || p.material_id >= (int)s.materials.size()) { err = "glTF: primitive material out of range"; return false; }
        const Value& sc = s.json.get("scenes");                                                      // LoadScenes :635-643
        s.scenes.assign(sc.size(), {});
        for (size_t i = 0; i < sc.size(); i++)
            for (size_t k = 0; k < sc.at(i).get("nodes").size(); k++) s.scenes[i].push_back(sc.at(i).get("nodes").at(k).int_or(-1));
        s.num_cameras = (int)s.json.get("cameras").size();
        for (int k = 0; k < s.num_cameras; k++) s.cameras.push_back(parse_camera(s.json.get("cameras").at((size_t)k)));
        if (!load_nodes()) return false;
        for (auto& roots : s.scenes) for (int n : roots) if (n < 0 || n >= (int)s.nodes.size()) { err = "glTF: scene root out of range"; return false; }
        if (!load_skins()) return false;
        if (!load_animations()) return false;
        if (!load_lights()) return false;
        for (size_t i = 0; i < s.nodes.size(); i++) {                                                // CreateDynamicMesh :949-974
            Node& n = s.nodes[i];
            if (n.skin == -1 && n.current_weights.empty()) { n.dynamic_mesh = -1; continue; }
            if (n.mesh < 0) { n.dynamic_mesh = -1; continue; }                                       // upstream indexes meshes[-1] here; a skin without a mesh has nothing to skin
            DynamicPrimitives d;
            for (auto& p : s.meshes[n.mesh].prims) {
                DynamicMesh dm;
                dm.num_vertices = p.num_vertices;
                dm.flags = PT_DYNAMIC_MESH_FLAG_POSITION | ((p.flags & PT_MESH_FLAG_TANGENT_SPACE) ? PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE : 0);
                d.meshes.push_back(dm);
            }
            s.dynamic.push_back(std::move(d));
            n.dynamic_mesh = (int)s.dynamic.size() - 1;
        }
        return true;
    }
};

// ---- animation (Animation.cpp)
int format_size(int f) { return f == 0 ? 4 : ((f == 2 || f == 4) ? 2 : 1); }                       // FormatSize :38-50
float unpack_data(const Channel& c, size_t keyframe, int component) {                               // UnpackData :52-71
    int fs = format_size(c.format);
    size_t off = keyframe * (size_t)c.width * fs + (size_t)component * fs;
    if (off + fs > c.transforms.size()) return 0.f;                                                  // (upstream reads out of bounds)
    const uint8_t* d = &c.transforms[off];
    switch (c.format) {
        case 0: { float v; memcpy(&v, d, 4); return v; }
        case 2: { uint16_t v; memcpy(&v, d, 2); return (float)v / 65535.0f; }
        case 4: { int16_t v; memcpy(&v, d, 2); float f = (float)v / 32767.0f; return f < -1.f ? -1.f : (f > 1.f ? 1.f : f); }
        case 1: return (float)*d / 255.0f;
        default: { float f = (float)*(const int8_t*)d / 127.0f; return f < -1.f ? -1.f : (f > 1.f ? 1.f : f); }
    }
}
float cubic_spline(float p0, float m0, float p1, float m1, float dt, float t) {                     // CubicSpline :21-28
    float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * p0 + dt * (t3 - 2 * t2 + t) * m0 + (-2 * t3 + 3 * t2) * p1 + dt * (t3 - t2) * m1;
}
float lerp_std(float a, float b, float t) {                                                         // std::lerp (C++20) as libstdc++ defines it
    if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if (t == 1) return b;
    const float x = a + t * (b - a);
    return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}
void sample_channel(const Channel& c, float time, bool fix_cubic, float* out) {                      // GetTransform :73-122
    if (c.times.empty() || c.width <= 0) return;
    time = time < c.times.front() ? c.times.front() : (time > c.times.back() ? c.times.back() : time);
    size_t ks = 0;
    for (size_t i = 1; i < c.times.size() && c.times[i] <= time; i++) ks = i;
    size_t ke = ks;
    if (ke + 1 < c.times.size() && c.times[ke] < time) ke++;
    auto factor = [&]() { float diff = c.times[ke] - c.times[ks]; return diff == 0.0f ? 0.0f : (time - c.times[ks]) / diff; };   // GetInterpolationFactor :9-19
    switch (c.interpolation) {
        case 0: for (int i = 0; i < c.width; i++) out[i] = unpack_data(c, ks, i); break;
        case 1: {
            float f = factor();
            if (c.path == 1) {
                quat a{unpack_data(c, ks, 0), unpack_data(c, ks, 1), unpack_data(c, ks, 2), unpack_data(c, ks, 3)};
                quat b{unpack_data(c, ke, 0), unpack_data(c, ke, 1), unpack_data(c, ke, 2), unpack_data(c, ke, 3)};
                quat r = slerp(a, b, f);
                out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
            } else
                for (int i = 0; i < c.width; i++) { float a = unpack_data(c, ks, i), b = unpack_data(c, ke, i); out[i] = lerp_std(a, b, f); }
        } break;
        default: {
            float f = factor();
            for (int i = 0; i < c.width; i++) {
                float duration = c.times[ke] - c.times[ks];
                float sv, st, ev, et;
                if (!fix_cubic) {            // upstream reads value AND tangent from element keyframe*3 (the in-tangent slot; "TODO: I think this is wrong")
                    sv = st = unpack_data(c, ks * 3, i);
                    ev = et = unpack_data(c, ke * 3, i);
                } else {                     // glTF: per keyframe [in-tangent, value, out-tangent]
                    sv = unpack_data(c, ks * 3 + 1, i); st = unpack_data(c, ks * 3 + 2, i);
                    ev = unpack_data(c, ke * 3 + 1, i); et = unpack_data(c, ke * 3, i);
                }
                out[i] = cubic_spline(sv, st, ev, et, duration, f);
            }
            if (c.path == 1) { quat q = normalize(quat{out[0], out[1], out[2], out[3]}); out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w; }
        } break;
    }
}

void traverse(const gs_scene& s, int node, const std::function<void(int)>& f) {                     // TraverseNode :113-120
    f(node);
    for (int c = s.nodes[node].child; c != -1; c = s.nodes[c].sibling) traverse(s, c, f);
}
void traverse_scene(const gs_scene& s, int scene, const std::function<void(int)>& f) { for (int n : s.scenes[scene]) traverse(s, n, f); }

void global_transforms(gs_scene& s, Node& n, const mat4& parent) {                                   // CalculateGlobalTransforms :1027-1041
    n.previous_global = n.global;
    n.global = mul(mul(mul(parent, translate(n.local.t)), mat4_cast(n.local.r)), scale(n.local.s));
    for (int c = n.child; c != -1; c = s.nodes[c].sibling) global_transforms(s, s.nodes[c], n.global);
}

void fill_sample(pt_texture_sample& o, const MatTex& t) {                                            // TextureSample(const Gltf::Material::Texture&) Renderer.h:77-85
    o.descriptor = t.texture; o.sampler = t.sampler; o.tex_coord = t.tex_coord; o.rotation = t.rotation;
    o.offset[0] = t.offset[0]; o.offset[1] = t.offset[1]; o.scale[0] = t.scale[0]; o.scale[1] = t.scale[1];
}
void gpu_material(const Material& m, pt_material& g) {                                               // GpuMaterial(const Gltf::Material&) Renderer.h:125-170
    memset(&g, 0, sizeof(g));
    g.flags = m.flags; g.alpha_mode = m.alpha_mode; g.metalness_factor = m.metalness_factor; g.roughness_factor = m.roughness_factor;
    g.occlusion_factor = m.occlusion_factor;
    for (int i = 0; i < 3; i++) g.emissive_factor[i] = m.emissive_strength * m.emissive_factor[i];
    memcpy(g.base_color_factor, m.base_color_factor, 16);
    g.normal_scale = m.normal_map_scale;
    fill_sample(g.normal, m.normal); fill_sample(g.albedo, m.albedo); fill_sample(g.metallic_roughness, m.metallic_roughness);
    fill_sample(g.occlusion, m.occlusion); fill_sample(g.emissive, m.emissive);
    g.alpha_cutoff = m.alpha_mode == 1 ? m.alpha_cutoff : 0.0f;
    g.ior = m.ior;
    memcpy(g.specular_color_factor, m.specular_color_factor, 12); g.specular_factor = m.specular_factor;
    fill_sample(g.specular, m.specular); fill_sample(g.specular_color, m.specular_color);
    g.clearcoat_factor = m.clearcoat_factor; g.clearcoat_roughness_factor = m.clearcoat_roughness_factor; g.clearcoat_normal_scale = m.clearcoat_normal_scale;
    fill_sample(g.clearcoat, m.clearcoat); fill_sample(g.clearcoat_roughness, m.clearcoat_roughness); fill_sample(g.clearcoat_normal, m.clearcoat_normal);
    g.anisotropy_strength = m.anisotropy_strength; g.anisotropy_rotation = m.anisotropy_rotation; fill_sample(g.anisotropy, m.anisotropy);
    memcpy(g.sheen_color_factor, m.sheen_color_factor, 12); g.sheen_roughness_factor = m.sheen_roughness_factor;
    fill_sample(g.sheen_color, m.sheen_color); fill_sample(g.sheen_roughness, m.sheen_roughness);
    g.transmission_factor = m.transmission_factor; fill_sample(g.transmission, m.transmission);
    g.thickness_factor = m.thickness_factor; g.attenuation_distance = m.attenuation_distance; memcpy(g.attenuation_color, m.attenuation_color, 12);
    fill_sample(g.thickness, m.thickness);
}

const Primitive* flat_primitive(const gs_scene* s, int flat, int* mesh_out, int* idx_out) {
    if (flat < 0) return nullptr;
    for (size_t m = 0; m < s->meshes.size(); m++) {
        if (flat < (int)s->meshes[m].prims.size()) { if (mesh_out) *mesh_out = (int)m; if (idx_out) *idx_out = flat; return &s->meshes[m].prims[flat]; }
        flat -= (int)s->meshes[m].prims.size();
    }
    return nullptr;
}

int gather_bones(const gs_scene* s, int node, std::vector<pt_bone>& out) {                           // Renderer.cpp:408-417
    const Node& n = s->nodes[node];
    if (n.skin < 0) return 0;
    const Skin& sk = s->skins[n.skin];
    out.resize(sk.joints.size());
    mat4 ninv = affine_inverse(n.global);
    for (size_t i = 0; i < sk.joints.size(); i++) {
        mat4 t = mul(mul(ninv, s->nodes[sk.joints[i]].global), sk.inverse_bind[i]);
        mat4 it = inverse_transpose3(t);
        memcpy(out[i].transform, t.m, 64);
        memcpy(out[i].inverse_transpose, it.m, 64);
    }
    return (int)out.size();
}

}  // namespace

// ===================================================================================================== C-ABI
extern "C" {

const char* gs_last_error(void) { return g_error.c_str(); }

int gs_load_file(const char* path, gs_scene** out) {
    if (!path || !out) return fail(PT_ERR_INVALID_ARGUMENT, "gs_load_file: null argument");
    *out = nullptr;
    gs_scene* s = new gs_scene();
    Loader l(*s);
    if (!l.load(path)) { std::string e = l.err; delete s; return fail(PT_ERR_INVALID_ARGUMENT, std::string(path) + ": " + e); }
    s->json = Value();                 // the DOM is not needed after loading
    s->buffers.clear(); s->buffers.shrink_to_fit();
    *out = s;
    return PT_OK;
}
void gs_free(gs_scene* s) { delete s; }

int gs_get_counts(const gs_scene* s, gs_counts* o) {
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_counts: null argument");
    int prims = 0;
    for (auto& m : s->meshes) prims += (int)m.prims.size();
    *o = {(int)s->meshes.size(), prims, (int)s->materials.size(), (int)s->nodes.size(), (int)s->scenes.size(), (int)s->skins.size(), (int)s->animations.size(),
          (int)s->lights.size(), (int)s->textures.size(), (int)s->samplers.size(), s->num_cameras, (int)s->dynamic.size()};
    return PT_OK;
}

int gs_get_primitive(const gs_scene* s, int flat, gs_primitive_info* o) {
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_primitive: null argument");
    int mesh = 0, idx = 0;
    const Primitive* p = flat_primitive(s, flat, &mesh, &idx);
    if (!p) return fail(PT_ERR_BAD_HANDLE, "gs_get_primitive: index out of range");
    memset(o, 0, sizeof(*o));
    o->mesh = mesh; o->index_in_mesh = idx; o->flags = p->valid ? p->flags : 0; o->topology = p->topology; o->num_vertices = p->num_vertices; o->num_indices = p->num_indices;
    o->index_format = p->index_format; o->material_id = p->material_id; o->num_targets = (int)p->targets.size();
    o->index = p->index.empty() ? nullptr : p->index.data();
    o->position = p->position.empty() ? nullptr : p->position.data();
    o->tangent_space = p->tangent_space.empty() ? nullptr : p->tangent_space.data();
    for (int k = 0; k < 2; k++) o->texcoord[k] = p->texcoord[k].empty() ? nullptr : p->texcoord[k].data();
    o->color = p->color.empty() ? nullptr : p->color.data();
    o->joint_weight = p->joint_weight.empty() ? nullptr : p->joint_weight.data();
    return PT_OK;
}
int gs_get_morph_target(const gs_scene* s, int flat, int target, int* flags, const float** pos, const uint32_t** ts) {
    if (!s) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_morph_target: null scene");
    const Primitive* p = flat_primitive(s, flat, nullptr, nullptr);
    if (!p || target < 0 || target >= (int)p->targets.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_morph_target: index out of range");
    const MorphTarget& t = p->targets[target];
    if (flags) *flags = t.flags;
    if (pos) *pos = t.position.empty() ? nullptr : t.position.data();
    if (ts) *ts = t.tangent_space.empty() ? nullptr : t.tangent_space.data();
    return PT_OK;
}

int gs_get_material(const gs_scene* s, int i, pt_material* o) {
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_material: null argument");
    if (i < 0 || i >= (int)s->materials.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_material: index out of range");
    gpu_material(s->materials[i], *o);
    if (s->uploaded) {
        pt_texture_sample* slots[15] = {&o->normal, &o->albedo, &o->metallic_roughness, &o->occlusion, &o->emissive, &o->specular, &o->specular_color, &o->clearcoat,
                                        &o->clearcoat_roughness, &o->clearcoat_normal, &o->anisotropy, &o->sheen_color, &o->sheen_roughness, &o->transmission, &o->thickness};
        for (auto* t : slots) {
            if (t->descriptor >= 0) t->descriptor = s->textures[t->descriptor].handle;
            t->sampler = t->sampler > 0 ? s->sampler_handles[t->sampler - 1] : 0;
        }
    }
    return PT_OK;
}
int gs_get_texture(const gs_scene* s, int i, int* w, int* h, int* srgb, int* loaded, const uint8_t** rgba) {
    if (!s) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_texture: null scene");
    if (i < 0 || i >= (int)s->textures.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_texture: index out of range");
    const Texture& t = s->textures[i];
    if (w) *w = t.image.width; if (h) *h = t.image.height; if (srgb) *srgb = t.srgb; if (loaded) *loaded = t.loaded;
    if (rgba) *rgba = t.loaded ? t.image.rgba.data() : nullptr;
    return PT_OK;
}
int gs_get_sampler(const gs_scene* s, int i, pt_sampler_desc* o) {
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_sampler: null argument");
    if (i < 0 || i >= (int)s->samplers.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_sampler: index out of range");
    *o = s->samplers[i];
    return PT_OK;
}

int gs_get_camera(const gs_scene* s, int i, gs_camera_info* o) {                       // Gltf::LoadCameras, Gltf.cpp:642-655
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_camera: null argument");
    if (i < 0 || i >= (int)s->cameras.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_camera: index out of range");
    *o = s->cameras[(size_t)i];
    return PT_OK;
}

int gs_get_node(const gs_scene* s, int i, gs_node_info* o) {
    if (!s || !o) return fail(PT_ERR_INVALID_ARGUMENT, "gs_get_node: null argument");
    if (i < 0 || i >= (int)s->nodes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_node: index out of range");
    const Node& n = s->nodes[i];
    o->child = n.child; o->sibling = n.sibling; o->mesh = n.mesh; o->skin = n.skin; o->dynamic_mesh = n.dynamic_mesh; o->camera = n.camera; o->light = n.light;
    auto put = [](const Trs& t, float* tt, float* rr, float* ss) { tt[0] = t.t.x; tt[1] = t.t.y; tt[2] = t.t.z; rr[0] = t.r.x; rr[1] = t.r.y; rr[2] = t.r.z; rr[3] = t.r.w; ss[0] = t.s.x; ss[1] = t.s.y; ss[2] = t.s.z; };
    put(n.rest, o->rest_translation, o->rest_rotation, o->rest_scale);
    put(n.local, o->local_translation, o->local_rotation, o->local_scale);
    memcpy(o->global_transform, n.global.m, 64);
    o->num_current_weights = (int)n.current_weights.size();
    return PT_OK;
}
int gs_get_node_weights(const gs_scene* s, int i, float* out, int cap) {
    if (!s || i < 0 || i >= (int)s->nodes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_node_weights: index out of range");
    const auto& w = s->nodes[i].current_weights;
    for (int k = 0; k < (int)w.size() && k < cap && out; k++) out[k] = w[k];
    return (int)w.size();
}
int gs_get_scene_nodes(const gs_scene* s, int scene, int* out, int cap) {
    if (!s || scene < 0 || scene >= (int)s->scenes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_scene_nodes: index out of range");
    const auto& r = s->scenes[scene];
    for (int k = 0; k < (int)r.size() && k < cap && out; k++) out[k] = r[k];
    return (int)r.size();
}
int gs_get_skin(const gs_scene* s, int i, int* nj, const uint32_t** joints, const float** ibp) {
    if (!s || i < 0 || i >= (int)s->skins.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_skin: index out of range");
    const Skin& k = s->skins[i];
    if (nj) *nj = (int)k.joints.size();
    if (joints) *joints = k.joints.data();
    if (ibp) *ibp = k.inverse_bind.empty() ? nullptr : k.inverse_bind[0].m;
    return PT_OK;
}
int gs_get_animation(const gs_scene* s, int i, float* length, int* nch) {
    if (!s || i < 0 || i >= (int)s->animations.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_animation: index out of range");
    if (length) *length = s->animations[i].length;
    if (nch) *nch = (int)s->animations[i].channels.size();
    return PT_OK;
}
int gs_get_channel(const gs_scene* s, int a, int c, gs_channel_info* o) {
    if (!s || !o || a < 0 || a >= (int)s->animations.size() || c < 0 || c >= (int)s->animations[a].channels.size()) return fail(PT_ERR_BAD_HANDLE, "gs_get_channel: index out of range");
    const Channel& ch = s->animations[a].channels[c];
    *o = {ch.node, ch.path, ch.interpolation, ch.format, ch.width, (int)ch.times.size(), (int)ch.transforms.size(), ch.times.data(), ch.transforms.data()};
    return PT_OK;
}
int gs_sample_channel(const gs_scene* s, int a, int c, float time, int fix, float* out) {
    if (!s || !out || a < 0 || a >= (int)s->animations.size() || c < 0 || c >= (int)s->animations[a].channels.size()) return fail(PT_ERR_BAD_HANDLE, "gs_sample_channel: index out of range");
    sample_channel(s->animations[a].channels[c], time, fix != 0, out);
    return PT_OK;
}

int gs_apply_rest_transforms(gs_scene* s) {                                                          // ApplyRestTransforms :976-990
    if (!s) return fail(PT_ERR_INVALID_ARGUMENT, "gs_apply_rest_transforms: null scene");
    for (Node& n : s->nodes) {
        n.local = n.rest;
        if (!n.weights.empty()) n.current_weights = n.weights;
        else if (n.mesh != -1 && !s->meshes[n.mesh].weights.empty()) n.current_weights = s->meshes[n.mesh].weights;
        else n.current_weights.assign(n.current_weights.size(), 0.0f);
    }
    return PT_OK;
}
int gs_animate(gs_scene* s, int a, float time) {                                                     // Animate :992-1014
    if (!s || a < 0 || a >= (int)s->animations.size()) return fail(PT_ERR_BAD_HANDLE, "gs_animate: animation out of range");
    gs_apply_rest_transforms(s);
    for (const Channel& c : s->animations[a].channels) {
        if (c.node < 0 || c.node >= (int)s->nodes.size() || c.times.empty()) continue;
        Node& n = s->nodes[c.node];
        float v[4] = {0, 0, 0, 0};
        switch (c.path) {
            case 0: v[0] = n.local.t.x; v[1] = n.local.t.y; v[2] = n.local.t.z; sample_channel(c, time, false, v); n.local.t = {v[0], v[1], v[2]}; break;
            case 1: v[0] = n.local.r.x; v[1] = n.local.r.y; v[2] = n.local.r.z; v[3] = n.local.r.w; sample_channel(c, time, false, v); n.local.r = {v[0], v[1], v[2], v[3]}; break;
            case 2: v[0] = n.local.s.x; v[1] = n.local.s.y; v[2] = n.local.s.z; sample_channel(c, time, false, v); n.local.s = {v[0], v[1], v[2]}; break;
            default: if ((int)n.current_weights.size() >= c.width && c.width > 0) sample_channel(c, time, false, n.current_weights.data()); break;
        }
    }
    return PT_OK;
}
int gs_calculate_global_transforms(gs_scene* s, int scene) {                                         // :1016-1025: glTF Y-up -> the renderer's Z-up
    if (!s || scene < 0 || scene >= (int)s->scenes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_calculate_global_transforms: scene out of range");
    mat4 cs = {{1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1}};
    for (int n : s->scenes[scene]) global_transforms(*s, s->nodes[n], cs);
    return PT_OK;
}
int gs_player_tick(gs_scene* s, gs_player* p, float dt) {                                            // AnimationPlayer::Tick
    if (!s || !p) return fail(PT_ERR_INVALID_ARGUMENT, "gs_player_tick: null argument");
    if (p->animation < (int)s->animations.size() && p->animation >= 0) {
        float len = s->animations[p->animation].length;
        if (p->playing) p->playhead += dt;
        if (len < p->playhead) {
            if (p->loop) p->playhead = len != 0 ? fmodf(p->playhead, len) : 0;
            else { p->playhead = len; p->playing = 0; }
        }
        return gs_animate(s, p->animation, p->playhead);
    }
    return PT_OK;
}

int gs_gather_lights(const gs_scene* s, int scene, pt_light* out, int cap) {                         // GatherLights :459-492
    if (!s || scene < 0 || scene >= (int)s->scenes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_gather_lights: scene out of range");
    int count = 0;
    traverse_scene(*s, scene, [&](int id) {
        const Node& n = s->nodes[id];
        if (n.light == -1) return;
        if (count < cap && out) {
            const Light& sl = s->lights[n.light];
            pt_light l;
            memset(&l, 0, sizeof(l));
            l.type = sl.type;
            memcpy(l.color, sl.color, 12); l.intensity = sl.intensity; l.cutoff = sl.cutoff;
            l.position[0] = n.global.m[12]; l.position[1] = n.global.m[13]; l.position[2] = n.global.m[14];
            mat4 it = inverse_transpose(n.global);
            float d[4] = {-it.m[8], -it.m[9], -it.m[10], -it.m[11]};                                  // it * (0, 0, -1, 0)
            float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);                 // normalize(vec4): w included
            l.direction[0] = d[0] / len; l.direction[1] = d[1] / len; l.direction[2] = d[2] / len;
            l.inner_angle = sl.inner; l.outer_angle = sl.outer;
            out[count] = l;
        }
        count++;
    });
    return count;
}
int gs_gather_bones(const gs_scene* s, int node, pt_bone* out, int cap) {
    if (!s || node < 0 || node >= (int)s->nodes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_gather_bones: node out of range");
    std::vector<pt_bone> b;
    int n = gather_bones(s, node, b);
    for (int i = 0; i < n && i < cap && out; i++) out[i] = b[i];
    return n;
}

int gs_upload(gs_scene* s, pt_ctx* ctx) {
    if (!s || !ctx) return fail(PT_ERR_INVALID_ARGUMENT, "gs_upload: null argument");
    if (s->uploaded) return fail(PT_ERR_INVALID_ARGUMENT, "gs_upload: scene already uploaded");
#define GS_TRY(call) do { int _r = (call); if (_r != PT_OK) return fail(_r, std::string(#call) + ": " + pt_last_error(ctx)); } while (0)
    for (auto& m : s->meshes)
        for (auto& p : m.prims) {
            if (!p.valid || p.topology != 4) continue;                                                // only triangle lists reach the BLAS builder (RayTracingAccelerationStructure.cpp:228-290)
            if (!p.index.empty()) GS_TRY(pt_buffer_create(ctx, p.index.data(), p.index.size(), p.index_format, &p.h_index));
            GS_TRY(pt_buffer_create(ctx, p.position.data(), p.position.size() * 4, PT_FORMAT_R32G32B32_FLOAT, &p.h_position));
            if (!p.tangent_space.empty()) GS_TRY(pt_buffer_create(ctx, p.tangent_space.data(), p.tangent_space.size() * 4, PT_FORMAT_R10G10B10A2_UNORM, &p.h_tangent_space));
            for (int k = 0; k < 2; k++) if (!p.texcoord[k].empty()) GS_TRY(pt_buffer_create(ctx, p.texcoord[k].data(), p.texcoord[k].size() * 4, PT_FORMAT_R32G32_FLOAT, &p.h_texcoord[k]));
            if (!p.color.empty()) GS_TRY(pt_buffer_create(ctx, p.color.data(), p.color.size() * 2, PT_FORMAT_R16G16B16A16_UNORM, &p.h_color));
            if (!p.joint_weight.empty()) GS_TRY(pt_buffer_create(ctx, p.joint_weight.data(), p.joint_weight.size() * 2, PT_FORMAT_JOINT_WEIGHT, &p.h_joint_weight));
            for (auto& t : p.targets) {
                if (!t.position.empty()) GS_TRY(pt_buffer_create(ctx, t.position.data(), t.position.size() * 4, PT_FORMAT_R32G32B32_FLOAT, &t.h_position));
                if (!t.tangent_space.empty()) GS_TRY(pt_buffer_create(ctx, t.tangent_space.data(), t.tangent_space.size() * 4, PT_FORMAT_R10G10B10A2_UNORM, &t.h_tangent_space));
            }
        }
    for (auto& d : s->dynamic)
        for (auto& dm : d.meshes) {
            if (dm.num_vertices <= 0) continue;
            GS_TRY(pt_buffer_create(ctx, nullptr, (size_t)dm.num_vertices * 12, PT_FORMAT_R32G32B32_FLOAT, &dm.h_position));
            if (dm.flags & PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE) GS_TRY(pt_buffer_create(ctx, nullptr, (size_t)dm.num_vertices * 4, PT_FORMAT_R10G10B10A2_UNORM, &dm.h_tangent_space));
        }
    for (auto& t : s->textures) if (t.loaded) GS_TRY(pt_texture_create(ctx, t.image.rgba.data(), t.image.width, t.image.height, t.srgb ? 1 : 0, &t.handle));
    s->sampler_handles.assign(s->samplers.size(), 0);
    for (size_t i = 0; i < s->samplers.size(); i++) GS_TRY(pt_sampler_create(ctx, &s->samplers[i], &s->sampler_handles[i]));
    s->uploaded = true;
    return PT_OK;
}

// Gltf::Unload (Gltf.cpp:123-157), as LoadGltf calls it before loading the next file (Main.cpp:43-54): the per-frame tables that
// point into this scene's resources are emptied first, then every stream, dynamic-mesh output and texture is released.
int gs_unload(gs_scene* s, pt_ctx* ctx) {
    if (!s || !ctx) return fail(PT_ERR_INVALID_ARGUMENT, "gs_unload: null argument");
    if (!s->uploaded) return PT_OK;
    GS_TRY(pt_scene_set_instances(ctx, nullptr, 0));
    GS_TRY(pt_scene_set_materials(ctx, nullptr, 0));
    auto drop = [&](int& h) -> int { if (h < 0) return PT_OK; int r = pt_buffer_destroy(ctx, h); h = -1; return r; };
    for (auto& m : s->meshes)
        for (auto& p : m.prims) {
            GS_TRY(drop(p.h_index)); GS_TRY(drop(p.h_position)); GS_TRY(drop(p.h_tangent_space)); GS_TRY(drop(p.h_texcoord[0])); GS_TRY(drop(p.h_texcoord[1]));
            GS_TRY(drop(p.h_color)); GS_TRY(drop(p.h_joint_weight));
            for (auto& t : p.targets) { GS_TRY(drop(t.h_position)); GS_TRY(drop(t.h_tangent_space)); }
        }
    for (auto& d : s->dynamic)
        for (auto& dm : d.meshes) { GS_TRY(drop(dm.h_position)); GS_TRY(drop(dm.h_tangent_space)); }
    for (auto& t : s->textures)
        if (t.loaded && t.handle >= 0) { GS_TRY(pt_texture_destroy(ctx, t.handle)); t.handle = -1; }
    s->uploaded = false;                                       // samplers are plain records of the context and stay
    return PT_OK;
}

int gs_frame(gs_scene* s, pt_ctx* ctx, int scene, int* light_count_out) {
    if (!s || !ctx) return fail(PT_ERR_INVALID_ARGUMENT, "gs_frame: null argument");
    if (!s->uploaded) return fail(PT_ERR_NOT_READY, "gs_frame: call gs_upload first");
    if (scene < 0 || scene >= (int)s->scenes.size()) return fail(PT_ERR_BAD_HANDLE, "gs_frame: scene out of range");
    int rc = PT_OK;
    std::string msg;
    // ---- PerformSkinning (Renderer.cpp:399-457)
    traverse_scene(*s, scene, [&](int id) {
        if (rc != PT_OK) return;
        const Node& n = s->nodes[id];
        bool skinned = n.skin != -1, morphed = !n.current_weights.empty();
        if (!(skinned || morphed) || n.dynamic_mesh < 0 || n.mesh < 0) return;
        std::vector<pt_bone> bones;
        if (skinned) gather_bones(s, id, bones);
        auto& prims = s->meshes[n.mesh].prims;
        auto& dyn = s->dynamic[n.dynamic_mesh].meshes;
        for (size_t i = 0; i < prims.size() && rc == PT_OK; i++) {
            Primitive& p = prims[i];
            if (p.h_position < 0 || dyn[i].h_position < 0) continue;
            int nt = 0;                                                                              // the four largest positive weights :425-443
            float w[PT_MAX_SIMULTANEOUS_MORPH_TARGETS] = {0, 0, 0, 0};
            int which[PT_MAX_SIMULTANEOUS_MORPH_TARGETS] = {-1, -1, -1, -1};
            for (size_t j = 0; j < n.current_weights.size() && j < p.targets.size(); j++) {
                float cw = n.current_weights[j];
                if (!(cw > 0.0f)) continue;
                if (nt < PT_MAX_SIMULTANEOUS_MORPH_TARGETS) { w[nt] = cw; which[nt] = (int)j; nt++; }
                else { int mi = (int)(std::min_element(w, w + PT_MAX_SIMULTANEOUS_MORPH_TARGETS) - w); if (w[mi] < cw) { w[mi] = cw; which[mi] = (int)j; } }
            }
            pt_skin_params sp;
            memset(&sp, 0, sizeof(sp));
            sp.num_of_vertices = (uint32_t)p.num_vertices;
            sp.input_mesh_flags = (uint32_t)p.flags & ~(skinned ? 0u : (uint32_t)PT_MESH_FLAG_JOINT_WEIGHT);
            sp.output_mesh_flags = (uint32_t)dyn[i].flags;
            sp.input_position = p.h_position; sp.input_tangent_space = p.h_tangent_space; sp.input_joint_weight = skinned ? p.h_joint_weight : -1;
            sp.output_position = dyn[i].h_position; sp.output_tangent_space = dyn[i].h_tangent_space;
            sp.num_of_morph_targets = nt;
            for (int k = 0; k < PT_MAX_SIMULTANEOUS_MORPH_TARGETS; k++) {
                sp.morph_weights[k] = w[k];
                sp.morph_position[k] = which[k] >= 0 ? p.targets[which[k]].h_position : -1;
                sp.morph_tangent_space[k] = which[k] >= 0 ? p.targets[which[k]].h_tangent_space : -1;
            }
            sp.use_mfma = 1;
            rc = pt_skin_run(ctx, &sp, bones.empty() ? nullptr : bones.data(), (int)bones.size());
            if (rc != PT_OK) msg = std::string("pt_skin_run: ") + pt_last_error(ctx);
        }
    });
    if (rc != PT_OK) return fail(rc, msg);
    // ---- GatherLights / GatherMaterials (Renderer.cpp:459-500)
    int nl = gs_gather_lights(s, scene, nullptr, 0);
    std::vector<pt_light> lights((size_t)(nl > 0 ? nl : 0));
    if (nl > 0) gs_gather_lights(s, scene, lights.data(), nl);
    GS_TRY(pt_scene_set_lights(ctx, lights.data(), nl));
    std::vector<pt_material> mats(s->materials.size());
    for (size_t i = 0; i < mats.size(); i++) gs_get_material(s, (int)i, &mats[i]);
    GS_TRY(pt_scene_set_materials(ctx, mats.data(), (int)mats.size()));
    // ---- BuildTlas' instance walk (Pathtracer.cpp:185-257)
    std::vector<pt_instance_desc> inst;
    traverse_scene(*s, scene, [&](int id) {
        const Node& n = s->nodes[id];
        if (n.mesh == -1) return;
        auto& prims = s->meshes[n.mesh].prims;
        for (size_t i = 0; i < prims.size(); i++) {
            const Primitive& p = prims[i];
            if (p.h_position < 0) continue;                                                           // no BLAS: AddTlasInstance fails and the primitive is skipped (:248-250)
            if ((p.num_indices ? p.num_indices : p.num_vertices) < 3) continue;
            const Material& m = s->materials[p.material_id];
            pt_instance_desc d;
            memset(&d, 0, sizeof(d));
            memcpy(d.gpu.transform, n.global.m, 64);
            mat4 nt = inverse_transpose(n.global);
            memcpy(d.gpu.normal_transform, nt.m, 64);
            d.gpu.index_descriptor = p.h_index; d.gpu.position_descriptor = p.h_position; d.gpu.tangent_space_descriptor = p.h_tangent_space;
            d.gpu.texcoord_descriptors[0] = p.h_texcoord[0]; d.gpu.texcoord_descriptors[1] = p.h_texcoord[1];
            d.gpu.color_descriptor = p.h_color; d.gpu.material_id = p.material_id;
            d.instance_flags = ((m.flags & 1u) ? PT_INSTANCE_FLAG_TRIANGLE_CULL_DISABLE : 0u) | (m.alpha_mode == 1 ? PT_INSTANCE_FLAG_FORCE_NON_OPAQUE : 0u);
            d.instance_mask = m.alpha_mode == 2 ? 2u : 1u;                                           // MASK_ALPHA_BLEND / MASK_NONE :192-195
            d.num_of_vertices = (uint32_t)p.num_vertices;
            d.num_of_indices = (uint32_t)(p.num_indices ? p.num_indices : p.num_vertices);
            d.num_of_indices -= d.num_of_indices % 3;
            if (n.dynamic_mesh != -1 && i < s->dynamic[n.dynamic_mesh].meshes.size()) {
                const DynamicMesh& dm = s->dynamic[n.dynamic_mesh].meshes[i];
                if (dm.h_position >= 0) d.gpu.position_descriptor = dm.h_position;
                if ((dm.flags & PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE) && dm.h_tangent_space >= 0) d.gpu.tangent_space_descriptor = dm.h_tangent_space;
                d.dynamic = 1;
            }
            inst.push_back(d);
        }
    });
    GS_TRY(pt_scene_set_instances(ctx, inst.data(), (int)inst.size()));
    if (light_count_out) *light_count_out = nl;
    return PT_OK;
#undef GS_TRY
}

// ---- image files
static int give(const std::vector<uint8_t>& v, uint8_t** out) { *out = (uint8_t*)malloc(v.size() ? v.size() : 1); if (!*out) return PT_ERR_OUT_OF_MEMORY; memcpy(*out, v.data(), v.size()); return PT_OK; }
int img_decode_rgba8(const void* data, size_t bytes, int* w, int* h, uint8_t** out) {
    if (!data || !w || !h || !out) return fail(PT_ERR_INVALID_ARGUMENT, "img_decode_rgba8: null argument");
    hostimg::Image8 im;
    std::string err;
    if (!hostimg::decode_image8((const uint8_t*)data, bytes, im, err)) return fail(PT_ERR_INVALID_ARGUMENT, err);
    *w = im.width; *h = im.height;
    return give(im.rgba, out);
}
int img_load_rgba8(const char* path, int* w, int* h, uint8_t** out) {
    if (!path) return fail(PT_ERR_INVALID_ARGUMENT, "img_load_rgba8: null path");
    std::vector<uint8_t> f;
    std::string err;
    if (!hostimg::read_file(path, f, err)) return fail(PT_ERR_INVALID_ARGUMENT, err);
    return img_decode_rgba8(f.data(), f.size(), w, h, out);
}
int img_decode_rgb32f(const void* data, size_t bytes, int is_exr, int* w, int* h, int* half, float** out) {
    if (!data || !w || !h || !out) return fail(PT_ERR_INVALID_ARGUMENT, "img_decode_rgb32f: null argument");
    hostimg::ImageF im;
    std::string err;
    bool ok = is_exr ? hostimg::decode_exr((const uint8_t*)data, bytes, im, err, is_exr == 2) : hostimg::decode_hdr((const uint8_t*)data, bytes, im, err);
    if (!ok) return fail(PT_ERR_INVALID_ARGUMENT, err);
    *w = im.width; *h = im.height;
    if (half) *half = im.half_source ? 1 : 0;
    *out = (float*)malloc(im.rgb.size() * 4 + 4);
    if (!*out) return fail(PT_ERR_OUT_OF_MEMORY, "img_decode_rgb32f: out of memory");
    memcpy(*out, im.rgb.data(), im.rgb.size() * 4);
    return PT_OK;
}
int img_load_rgb32f(const char* path, int* w, int* h, int* half, float** out) {
    if (!path) return fail(PT_ERR_INVALID_ARGUMENT, "img_load_rgb32f: null path");
    std::string p(path);
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot);
    int is_exr;
    if (ext == ".exr") is_exr = 1; else if (ext == ".hdr") is_exr = 0; else return fail(PT_ERR_INVALID_ARGUMENT, "img_load_rgb32f: need .hdr or .exr");
    std::vector<uint8_t> f;
    std::string err;
    if (!hostimg::read_file(p, f, err)) return fail(PT_ERR_INVALID_ARGUMENT, err);
    return img_decode_rgb32f(f.data(), f.size(), is_exr, w, h, half, out);
}
void img_free(void* p) { free(p); }

int img_write_png(const char* path, const uint8_t* rgba8, int width, int height, int channels) {
    if (!path || !rgba8) return fail(PT_ERR_INVALID_ARGUMENT, "img_write_png: null argument");
    std::string err;
    return hostimg::write_png(path, rgba8, width, height, channels, err) ? PT_OK : fail(PT_ERR_INVALID_ARGUMENT, err);
}
int img_write_pfm(const char* path, const float* rgb, int width, int height) {
    if (!path || !rgb) return fail(PT_ERR_INVALID_ARGUMENT, "img_write_pfm: null argument");
    std::string err;
    return hostimg::write_pfm(path, rgb, width, height, err) ? PT_OK : fail(PT_ERR_INVALID_ARGUMENT, err);
}
int img_write_exr(const char* path, const float* rgb, int width, int height, int half) {
    if (!path || !rgb) return fail(PT_ERR_INVALID_ARGUMENT, "img_write_exr: null argument");
    std::string err;
    return hostimg::write_exr(path, rgb, width, height, half != 0, err) ? PT_OK : fail(PT_ERR_INVALID_ARGUMENT, err);
}

}  // extern "C"
