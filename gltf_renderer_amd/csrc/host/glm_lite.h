// glm_lite.h -- the glm closed forms the scene side needs, restated: glm is an empty submodule in the
// reference tree.  Column-major float[16] matrices (M[col*4+row]), quaternions stored x,y,z,w (GLM_FORCE_QUAT_DATA_XYZW).
#pragma once
#include <cmath>
#include <cstring>

namespace glml {

struct vec3 { float x, y, z; };
struct quat { float x, y, z, w; };
struct mat4 { float m[16]; };

inline mat4 identity() { mat4 r; memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
inline mat4 translate(vec3 v) { mat4 r = identity(); r.m[12] = v.x; r.m[13] = v.y; r.m[14] = v.z; return r; }
inline mat4 scale(vec3 v) { mat4 r = identity(); r.m[0] = v.x; r.m[5] = v.y; r.m[10] = v.z; return r; }
inline mat4 mat4_cast(quat q) {                       // glm::mat3_cast
    mat4 r = identity();
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r.m[0] = 1.f - 2.f * (qyy + qzz); r.m[1] = 2.f * (qxy + qwz); r.m[2] = 2.f * (qxz - qwy);
    r.m[4] = 2.f * (qxy - qwz); r.m[5] = 1.f - 2.f * (qxx + qzz); r.m[6] = 2.f * (qyz + qwx);
    r.m[8] = 2.f * (qxz + qwy); r.m[9] = 2.f * (qyz - qwx); r.m[10] = 1.f - 2.f * (qxx + qyy);
    return r;
}
// general inverse by cofactors, evaluated in double and rounded once (as the hot path's camera inverse does)
inline bool inverse(const mat4& a, mat4& out) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = a.m[i];
    double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[9] - m[8] * m[1], s2 = m[0] * m[13] - m[12] * m[1];
    double s3 = m[4] * m[9] - m[8] * m[5], s4 = m[4] * m[13] - m[12] * m[5], s5 = m[8] * m[13] - m[12] * m[9];
    double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[6] * m[15] - m[14] * m[7], c3 = m[6] * m[11] - m[10] * m[7];
    double c2 = m[2] * m[15] - m[14] * m[3], c1 = m[2] * m[11] - m[10] * m[3], c0 = m[2] * m[7] - m[6] * m[3];
    double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    if (det == 0) return false;
    double id = 1.0 / det;
    inv[0] = (m[5] * c5 - m[9] * c4 + m[13] * c3) * id;   inv[4] = (-m[4] * c5 + m[8] * c4 - m[12] * c3) * id;
    inv[8] = (m[7] * s5 - m[11] * s4 + m[15] * s3) * id;  inv[12] = (-m[6] * s5 + m[10] * s4 - m[14] * s3) * id;
    inv[1] = (-m[1] * c5 + m[9] * c2 - m[13] * c1) * id;  inv[5] = (m[0] * c5 - m[8] * c2 + m[12] * c1) * id;
    inv[9] = (-m[3] * s5 + m[11] * s2 - m[15] * s1) * id; inv[13] = (m[2] * s5 - m[10] * s2 + m[14] * s1) * id;
    inv[2] = (m[1] * c4 - m[5] * c2 + m[13] * c0) * id;   inv[6] = (-m[0] * c4 + m[4] * c2 - m[12] * c0) * id;
    inv[10] = (m[3] * s4 - m[7] * s2 + m[15] * s0) * id;  inv[14] = (-m[2] * s4 + m[6] * s2 - m[14] * s0) * id;
    inv[3] = (-m[1] * c3 + m[5] * c1 - m[9] * c0) * id;   inv[7] = (m[0] * c3 - m[4] * c1 + m[8] * c0) * id;
    inv[11] = (-m[3] * s3 + m[7] * s1 - m[11] * s0) * id; inv[15] = (m[2] * s3 - m[6] * s1 + m[10] * s0) * id;
    for (int i = 0; i < 16; i++) out.m[i] = (float)inv[i];      // same routine as mat4_inverse in mipt_api.hip (layout-agnostic)
    return true;
}
inline mat4 transpose(const mat4& a) { mat4 r; for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) r.m[c * 4 + row] = a.m[row * 4 + c]; return r; }
inline mat4 inverse_transpose(const mat4& a) { mat4 i; if (!inverse(a, i)) { for (auto& v : i.m) v = NAN; } return transpose(i); }
// glm::affineInverse: R = inverse(mat3(m)); [R, -R * m[3].xyz; 0 0 0 1]
inline mat4 affine_inverse(const mat4& a) {
    double m00 = a.m[0], m01 = a.m[1], m02 = a.m[2], m10 = a.m[4], m11 = a.m[5], m12 = a.m[6], m20 = a.m[8], m21 = a.m[9], m22 = a.m[10];
    double det = m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02);
    double id = 1.0 / det;
    double r[9];                                    // column-major 3x3 inverse
    r[0] = (m11 * m22 - m21 * m12) * id; r[1] = -(m01 * m22 - m21 * m02) * id; r[2] = (m01 * m12 - m11 * m02) * id;
    r[3] = -(m10 * m22 - m20 * m12) * id; r[4] = (m00 * m22 - m20 * m02) * id; r[5] = -(m00 * m12 - m10 * m02) * id;
    r[6] = (m10 * m21 - m20 * m11) * id; r[7] = -(m00 * m21 - m20 * m01) * id; r[8] = (m00 * m11 - m10 * m01) * id;
    mat4 o = identity();
    for (int c = 0; c < 3; c++) for (int row = 0; row < 3; row++) o.m[c * 4 + row] = (float)r[c * 3 + row];
    double tx = a.m[12], ty = a.m[13], tz = a.m[14];
    o.m[12] = (float)-(r[0] * tx + r[3] * ty + r[6] * tz);
    o.m[13] = (float)-(r[1] * tx + r[4] * ty + r[7] * tz);
    o.m[14] = (float)-(r[2] * tx + r[5] * ty + r[8] * tz);
    return o;
}
// inverseTranspose(mat3(m)) widened to a mat4 with identity elsewhere (GpuSkin::Bone::inverse_transpose)
inline mat4 inverse_transpose3(const mat4& a) {
    mat4 lin = identity();
    for (int c = 0; c < 3; c++) for (int row = 0; row < 3; row++) lin.m[c * 4 + row] = a.m[c * 4 + row];
    return inverse_transpose(lin);
}
inline float dot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline quat normalize(quat q) {
    float len = sqrtf(dot(q, q));
    if (len <= 0) return {0, 0, 0, 1};
    float il = 1.f / len;
    return {q.x * il, q.y * il, q.z * il, q.w * il};
}
inline quat slerp(quat x, quat y, float a) {         // glm::slerp: shortest path, lerp when nearly parallel
    quat z = y;
    float cos_theta = dot(x, y);
    if (cos_theta < 0.f) { z = {-y.x, -y.y, -y.z, -y.w}; cos_theta = -cos_theta; }
    if (cos_theta > 1.f - 1.1920929e-07f) return {x.x + a * (z.x - x.x), x.y + a * (z.y - x.y), x.z + a * (z.z - x.z), x.w + a * (z.w - x.w)};   // glm::mix
    float angle = acosf(cos_theta);
    float s0 = sinf((1.f - a) * angle), s1 = sinf(a * angle), sd = sinf(angle);
    return {(s0 * x.x + s1 * z.x) / sd, (s0 * x.y + s1 * z.y) / sd, (s0 * x.z + s1 * z.z) / sd, (s0 * x.w + s1 * z.w) / sd};
}
// glm::decompose (gtx/matrix_decompose, after Graphics Gems "unmatrix"): scale, rotation, translation; skew and perspective
// are computed by glm but discarded by the caller (Gltf.cpp:673-675).  Returns false for a singular matrix.
inline bool decompose(const mat4& model, vec3& scl, quat& rot, vec3& trans) {
    mat4 l = model;
    if (fabsf(l.m[15]) <= 1.1920929e-07f) return false;
    for (int i = 0; i < 16; i++) l.m[i] /= model.m[15];
    l.m[3] = l.m[7] = l.m[11] = 0; l.m[15] = 1;      // perspective partition cleared
    trans = {l.m[12], l.m[13], l.m[14]};
    float row[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) row[i][j] = l.m[i * 4 + j];
    auto len = [](const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    auto scl_to = [&](float* v, float desired) { float n = len(v); if (n != 0) { float s = desired / n; v[0] *= s; v[1] *= s; v[2] *= s; } };
    auto dot3 = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto combine = [](float* a, const float* b, float as, float bs) { for (int i = 0; i < 3; i++) a[i] = as * a[i] + bs * b[i]; };
    float sx = len(row[0]); scl_to(row[0], 1.f);
    float skew_z = dot3(row[0], row[1]); combine(row[1], row[0], 1.f, -skew_z);
    float sy = len(row[1]); scl_to(row[1], 1.f);
    float skew_y = dot3(row[0], row[2]); combine(row[2], row[0], 1.f, -skew_y);
    float skew_x = dot3(row[1], row[2]); combine(row[2], row[1], 1.f, -skew_x);
    float sz = len(row[2]); scl_to(row[2], 1.f);
    float cr[3] = {row[1][1] * row[2][2] - row[1][2] * row[2][1], row[1][2] * row[2][0] - row[1][0] * row[2][2], row[1][0] * row[2][1] - row[1][1] * row[2][0]};
    if (dot3(row[0], cr) < 0) { sx = -sx; sy = -sy; sz = -sz; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) row[i][j] = -row[i][j]; }
    scl = {sx, sy, sz};
    float q[4];
    float trace = row[0][0] + row[1][1] + row[2][2], root;
    if (trace > 0.f) {
        root = sqrtf(trace + 1.f);
        q[3] = 0.5f * root;
        root = 0.5f / root;
        q[0] = root * (row[1][2] - row[2][1]); q[1] = root * (row[2][0] - row[0][2]); q[2] = root * (row[0][1] - row[1][0]);
    } else {
        static const int next[3] = {1, 2, 0};
        int i = 0;
        if (row[1][1] > row[0][0]) i = 1;
        if (row[2][2] > row[i][i]) i = 2;
        int j = next[i], k = next[j];
        root = sqrtf(row[i][i] - row[j][j] - row[k][k] + 1.f);
        q[i] = 0.5f * root;
        root = 0.5f / root;
        q[j] = root * (row[i][j] + row[j][i]); q[k] = root * (row[i][k] + row[k][i]); q[3] = root * (row[j][k] - row[k][j]);
    }
    rot = {q[0], q[1], q[2], q[3]};
    return true;
}

}  // namespace glml
