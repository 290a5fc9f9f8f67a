// host_math.h -- 4x4 matrices and the Transform pair (m, mInv) of the scene front end.
//
// The camera matrices handed to the GPU must be the same float32 numbers the reference computes, so the
// arithmetic (and its order) follows core/transform.cpp:31-194, core/transform.h:33-57 and the float
// Gauss-Jordan inverse with full pivoting of core/util.cpp:127-184:
// LookAt inverts numerically, Transform*Transform multiplies the stored inverses rather than re-inverting.
#pragma once
#include <cmath>
#include <cstring>
#include <utility>

namespace pbrthip {

struct Mat4 {
    float m[4][4];
    Mat4() { std::memset(m, 0, sizeof m); m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1.f; }
    Mat4(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13,
         float a20, float a21, float a22, float a23, float a30, float a31, float a32, float a33) {
        float t[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
        std::memcpy(m, t, sizeof m);
    }
    Mat4 transposed() const {
        Mat4 r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
        return r;
    }
    static Mat4 mul(const Mat4 &a, const Mat4 &b) {                      // core/pbrt.h:525-536
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
        return r;
    }
    // Gauss-Jordan elimination, full pivoting, float throughout (util.cpp:127-184)
    Mat4 inverse(bool *singular = nullptr) const {
        int colIdx[4], rowIdx[4], pivoted[4] = {0, 0, 0, 0};
        float a[4][4]; std::memcpy(a, m, sizeof a);
        for (int step = 0; step < 4; ++step) {
            int pr = -1, pc = -1; float big = 0.f;
            for (int r = 0; r < 4; ++r) {
                if (pivoted[r] == 1) continue;
                for (int c = 0; c < 4; ++c) {
                    if (pivoted[c] == 0) { if (std::fabs(a[r][c]) >= big) { big = float(std::fabs(a[r][c])); pr = r; pc = c; } }
                    else if (pivoted[c] > 1 && singular) *singular = true;
                }
            }
            ++pivoted[pc];
            if (pr != pc) for (int k = 0; k < 4; ++k) std::swap(a[pr][k], a[pc][k]);
            rowIdx[step] = pr; colIdx[step] = pc;
            if (a[pc][pc] == 0.f && singular) *singular = true;
            float pivinv = 1.f / a[pc][pc];
            a[pc][pc] = 1.f;
            for (int k = 0; k < 4; ++k) a[pc][k] *= pivinv;
            for (int r = 0; r < 4; ++r) {
                if (r == pc) continue;
                float save = a[r][pc];
                a[r][pc] = 0;
                for (int k = 0; k < 4; ++k) a[r][k] -= a[pc][k] * save;
            }
        }
        for (int j = 3; j >= 0; --j)
            if (rowIdx[j] != colIdx[j]) for (int k = 0; k < 4; ++k) std::swap(a[k][rowIdx[j]], a[k][colIdx[j]]);
        Mat4 r; std::memcpy(r.m, a, sizeof a);
        return r;
    }
};

struct Xform {
    Mat4 m, inv;
    Xform() {}
    explicit Xform(const Mat4 &mm) : m(mm), inv(mm.inverse()) {}
    Xform(const Mat4 &mm, const Mat4 &ii) : m(mm), inv(ii) {}
    Xform inverse() const { return Xform(inv, m); }
    Xform operator*(const Xform &o) const { return Xform(Mat4::mul(m, o.m), Mat4::mul(o.inv, inv)); }
    bool swaps_handedness() const {                                       // transform.cpp:166-176
        float det = ((m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1])) -
                     (m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0])) +
                     (m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0])));
        return det < 0.f;
    }
    void point(const float p[3], float out[3]) const {                   // transform.h:73-92
        float x = p[0], y = p[1], z = p[2];
        float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
        float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
        float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
        float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
        if (wp == 1.f) { out[0] = xp; out[1] = yp; out[2] = zp; }
        else { float inv_w = 1.f / wp; out[0] = inv_w * xp; out[1] = inv_w * yp; out[2] = inv_w * zp; }
    }
};

inline float radians(float deg) { return (3.14159265358979323846f / 180.f) * deg; }

inline Xform Translate(float x, float y, float z) {
    return Xform(Mat4(1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1), Mat4(1, 0, 0, -x, 0, 1, 0, -y, 0, 0, 1, -z, 0, 0, 0, 1));
}
inline Xform Scale(float x, float y, float z) {
    return Xform(Mat4(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1),
                 Mat4(1.f / x, 0, 0, 0, 0, 1.f / y, 0, 0, 0, 0, 1.f / z, 0, 0, 0, 0, 1));
}
inline Xform Rotate(float angle, float ax, float ay, float az) {         // transform.cpp:82-112
    float len = std::sqrt(ax * ax + ay * ay + az * az);
    float il = 1.f / len; float x = ax * il, y = ay * il, z = az * il;
    float s = sinf(radians(angle)), c = cosf(radians(angle));
    Mat4 r;
    r.m[0][0] = x * x + (1.f - x * x) * c; r.m[0][1] = x * y * (1.f - c) - z * s; r.m[0][2] = x * z * (1.f - c) + y * s; r.m[0][3] = 0;
    r.m[1][0] = x * y * (1.f - c) + z * s; r.m[1][1] = y * y + (1.f - y * y) * c; r.m[1][2] = y * z * (1.f - c) - x * s; r.m[1][3] = 0;
    r.m[2][0] = x * z * (1.f - c) - y * s; r.m[2][1] = y * z * (1.f - c) + x * s; r.m[2][2] = z * z + (1.f - z * z) * c; r.m[2][3] = 0;
    r.m[3][0] = 0; r.m[3][1] = 0; r.m[3][2] = 0; r.m[3][3] = 1;
    return Xform(r, r.transposed());
}
inline Xform LookAt(const float pos[3], const float look[3], const float up[3]) {   // transform.cpp:113-138
    auto norm = [](float v[3]) { float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); float il = 1.f / l; v[0] *= il; v[1] *= il; v[2] *= il; };
    auto cross = [](const float a[3], const float b[3], float r[3]) {
        r[0] = (a[1] * b[2]) - (a[2] * b[1]); r[1] = (a[2] * b[0]) - (a[0] * b[2]); r[2] = (a[0] * b[1]) - (a[1] * b[0]); };
    float dir[3] = {look[0] - pos[0], look[1] - pos[1], look[2] - pos[2]}; norm(dir);
    float right[3]; cross(dir, up, right); norm(right);
    float newUp[3]; cross(right, dir, newUp);
    Mat4 c2w;
    c2w.m[0][3] = pos[0]; c2w.m[1][3] = pos[1]; c2w.m[2][3] = pos[2]; c2w.m[3][3] = 1;
    c2w.m[0][0] = right[0]; c2w.m[1][0] = right[1]; c2w.m[2][0] = right[2]; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp[0]; c2w.m[1][1] = newUp[1]; c2w.m[2][1] = newUp[2]; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir[0]; c2w.m[1][2] = dir[1]; c2w.m[2][2] = dir[2]; c2w.m[3][2] = 0.;
    return Xform(c2w.inverse(), c2w);
}
inline Xform Perspective(float fov, float n, float f) {                   // transform.cpp:182-194
    float inv_denom = 1.f / (f - n);
    Mat4 persp(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f * inv_denom, -f * n * inv_denom, 0, 0, 1, 0);
    float invTanAng = 1.f / tanf(radians(fov) / 2.f);
    return Scale(invTanAng, invTanAng, 1) * Xform(persp);
}

}  // namespace pbrthip
