// rt_math.h -- device-side small-vector arithmetic.
//
// Every operator spells out the association order of the reference's inline operators
// (core/geometry.h:30-120,296-345 for Vector/Point/Normal, core/color.h:54-116 for the
// 3-sample Spectrum) because the translation unit is compiled with -ffp-contract=off and
// parity with the reference's non-FMA SSE build is judged at the 1e-5 level: e.g. a
// division by a scalar is "multiply by 1.f/f" there, never a true per-component divide.
#pragma once
#include <hip/hip_runtime.h>

#define RT_DEV __device__ __forceinline__

#define RT_PI 3.14159265358979323846f        /* core/pbrt.h:204 (float literal) */
#define RT_INV_PI 0.31830988618379067154f    /* core/pbrt.h:205 */
#define RT_INV_TWOPI 0.15915494309189533577f
#define RT_RAY_EPSILON 1e-3f                 /* core/pbrt.h:211 */
#define RT_INF __builtin_inff()

struct V3 { float x, y, z; };

RT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 mk3(float v) { return mk3(v, v, v); }
RT_DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV V3 operator*(V3 a, float f) { return mk3(f * a.x, f * a.y, f * a.z); }
RT_DEV V3 operator*(float f, V3 a) { return mk3(f * a.x, f * a.y, f * a.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }   // Spectrum*Spectrum
RT_DEV V3 div_s(V3 a, float f) { float inv = 1.f / f; return mk3(a.x * inv, a.y * inv, a.z * inv); }
RT_DEV V3 div_c(V3 a, V3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }      // Spectrum/Spectrum
RT_DEV float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV float absdot3(V3 a, V3 b) { return fabsf(dot3(a, b)); }
RT_DEV V3 cross3(V3 a, V3 b) {
    return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
RT_DEV float len3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_DEV V3 normalize3(V3 a) { return div_s(a, len3(a)); }
RT_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RT_DEV bool is_black(V3 s) { return !(s.x != 0.f) && !(s.y != 0.f) && !(s.z != 0.f); }   // color.h:104-108
RT_DEV float lum_y(V3 s) {                                                               // color.h:185-190
    float v = 0.f;
    v += 0.212671f * s.x; v += 0.715160f * s.y; v += 0.072169f * s.z;
    return v;
}
RT_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// core/transform.h:73-92 : point transform with the w != 1 divide; :94-105 vector transform
RT_DEV V3 xform_point(const float *m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp != 1.f) { float inv = 1.f / wp; return mk3(xp * inv, yp * inv, zp * inv); }   // Point::operator/=
    return mk3(xp, yp, zp);
}
RT_DEV V3 xform_vector(const float *m, V3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// ---- counter-based RNG: identical definition in oracle/ref/keyed_rng.cpp and oracle/pbrt_oracle.cpp.
RT_DEV uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
RT_DEV uint32_t rng_base(uint32_t key, uint32_t seed) { return pcg_hash(key + seed * 0x9E3779B9u); }
RT_DEV uint32_t rng_u32(uint32_t base, uint32_t ctr) { return pcg_hash(ctr + base); }
// RandomFloat(): core/util.cpp:377-380
RT_DEV float rng_f32(uint32_t base, uint32_t ctr) { return float(rng_u32(base, ctr) & 0xffffffu) / float(1 << 24); }
