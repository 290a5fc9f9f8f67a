// rt_shade.h -- device-side surface interaction: differential geometry of a triangle hit,
// the BSDF set the three materials can produce, and the two light types.
// Each function names the reference code whose arithmetic (and association order) it follows.
#pragma once
#include "rt_traverse.h"

namespace rt {

// BxDFType bits (core/reflection.h:52-68)
enum { BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16,
       BX_ALL = 31 };

struct Vertex {
    V3 p, nn, sn, tn;   // hit point, shading normal, BSDF frame (BSDF ctor reflection.cpp:471-479)
    V3 ng;              // geometric normal, carried by the EXT kernels only (per-vertex N / S make it differ from nn): every read goes
                        // through vertex_ng<EXT>(), so the other kernels never keep it alive (3 VGPRs decide 3 vs 2 waves per SIMD on C2)
    V3 wo;
    int mat, light;     // material index, area-light index of the primitive or -1
};

// DifferentialGeometry for a triangle without per-vertex uv/N/S:
//   Triangle::Intersect trianglemesh.cpp:248-274 with GetUVs' defaults (0,0),(1,0),(1,1) (:321-326):
//   du1=-1 du2=0 dv1=-1 dv2=-1, determinant 1  =>  dpdu = (dv2*dp1 - dv1*dp2)*invdet,  dpdv = (-du2*dp1 + du1*dp2)*invdet
//   DifferentialGeometry ctor shape.cpp:37-51: nn = Normalize(Cross(dpdu,dpdv)), flipped iff
//   reverseOrientation ^ transformSwapsHandedness.
//   GetShadingGeometry is the identity (trianglemesh.cpp:71-75, no N/S) and Material::Bump with the always
//   present constant-0 bump texture (paramset.cpp:452-465, material.cpp:29-71) re-derives the same nn:
//   dpdu + 0/du*nn + 0*dndu leaves dpdu/dpdv unchanged, and the final "face the geometric normal" flip is
//   a no-op because both normals are the same vector.  BSDF frame: reflection.cpp:471-479.
RT_DEV void tri_frame(V3 p1, V3 p2, V3 p3, bool flip, V3 &nn, V3 &dpdu) {
    const float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    V3 dp1 = p1 - p3, dp2 = p2 - p3;
    const float determinant = du1 * dv2 - dv1 * du2;
    const float invdet = 1.f / determinant;
    dpdu = (dv2 * dp1 - dv1 * dp2) * invdet;
    V3 dpdv = (-du2 * dp1 + du1 * dp2) * invdet;
    nn = normalize3(cross3(dpdu, dpdv));
    if (flip) nn = nn * -1.f;
}

// DifferentialGeometry + BSDF frame of a quadric hit (sphere.cpp:141-209, disk.cpp:85-103, cylinder.cpp:108-142, shape.cpp:37-51,
// reflection.cpp:471-479): the hit
// point is re-derived exactly as Sphere::Intersect did (same object-space ray, t = the accepted hit parameter)
RT_DEV void quadric_frame(const DevScene &sc, unsigned qi, bool flip, V3 ow, V3 dw, float thit, V3 &vp, V3 &vnn, V3 &vsn) {
    const DevQuadric RT_G &q = RT_GPTR(const DevQuadric, sc.quadrics)[qi];
    const V3 o = xform_point(q.w2o, ow), d = xform_vector(q.w2o, dw);
    const V3 phit = o + d * thit;
    const float radius = q.radius, phiMax = q.phi_max, thetaMin = q.theta_min, thetaMax = q.theta_max;
    float cosphi, sinphi; V3 dpdu, dpdv;
    if (q.type == RT_QUADRIC_DISK) {                                            // disk.cpp:85-93 (zmax = innerRadius)
        const float dist2 = phit.x * phit.x + phit.y * phit.y;
        const float vv = 1.f - ((sqrtf(dist2) - q.zmax) / (radius - q.zmax));
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f) * (phiMax * RT_INV_TWOPI);
        dpdv = mk3(-phit.x / (1 - vv), -phit.y / (1 - vv), 0.f) * ((radius - q.zmax) / radius);
    } else if (q.type == RT_QUADRIC_CYLINDER) {                                 // cylinder.cpp:112-115
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f);
        dpdv = mk3(0.f, 0.f, q.zmax - q.zmin);
    } else if (q.type == RT_QUADRIC_CONE) {                                     // cone.cpp:95-100 (zmax = height)
        const float vv = phit.z / q.zmax;
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f);
        dpdv = mk3(-phit.x / (1.f - vv), -phit.y / (1.f - vv), q.zmax);
    } else if (q.type == RT_QUADRIC_PARABOLOID) {                               // paraboloid.cpp:97-102
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f);
        dpdv = mk3(phit.x / (2.f * phit.z), phit.y / (2.f * phit.z), 1.f) * (q.zmax - q.zmin);
    } else if (q.type == RT_QUADRIC_HYPERBOLOID) {                              // hyperboloid.cpp:122-130
        const float phi = quadric_phi(q, phit);
        cosphi = cosf(phi); sinphi = sinf(phi);
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f);
        dpdv = mk3((q.p2[0] - q.p1[0]) * cosphi - (q.p2[1] - q.p1[1]) * sinphi, (q.p2[0] - q.p1[0]) * sinphi + (q.p2[1] - q.p1[1]) * cosphi, q.p2[2] - q.p1[2]);
    } else {
    const float theta = acosf(clampf(phit.z / radius, -1.f, 1.f));
    const float zradius = sqrtf(phit.x * phit.x + phit.y * phit.y);
    if (zradius == 0) {
        cosphi = 0; sinphi = 1;
        dpdv = mk3(phit.z * cosphi, phit.z * sinphi, -radius * sinf(theta)) * (thetaMax - thetaMin);
        dpdu = cross3(dpdv, phit);
    } else {
        const float invzradius = 1.f / zradius;
        cosphi = phit.x * invzradius; sinphi = phit.y * invzradius;
        dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.f);
        dpdv = mk3(phit.z * cosphi, phit.z * sinphi, -radius * sinf(theta)) * (thetaMax - thetaMin);
    }
    }
    vp = xform_point(q.o2w, phit);
    const V3 dpduW = xform_vector(q.o2w, dpdu), dpdvW = xform_vector(q.o2w, dpdv);
    vnn = normalize3(cross3(dpduW, dpdvW));
    if (flip) vnn = vnn * -1.f;
    vsn = normalize3(dpduW);
}
// Triangle::GetShadingGeometry (trianglemesh.cpp:71-133) -> Material::Bump with the always-present constant-0 displacement
// (material.cpp:29-71) -> BSDF ctor (reflection.cpp:471-479) for a triangle with per-vertex N and / or S: the shading normal and
// the tangent the BSDF frame is built from.  ngeom: geometric normal (flipped), gdpdu: Triangle::Intersect's dpdu (both per-triangle
// constants evaluated on the host); b1, b2: the hit's barycentrics.
RT_DEV V3 xform_normal_rm(const float RT_G *mi, V3 n) {                            // Transform::operator()(Normal) transform.h:106-111
    return mk3(mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z);
}
RT_DEV V3 xform_vec3_rm(const float RT_G *m, V3 v) {                               // Transform::operator()(Vector) transform.h:93-98
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
RT_DEV void shading_frame(const DevScene &sc, const DevTriShading RT_G &r, float hb1, float hb2, V3 ngeom, bool flip, V3 &nn_out, V3 &sn_dir) {
    const float RT_G *o2w = RT_GPTR(const float, sc.xforms) + size_t(32) * r.xform;
    const float RT_G *o2wInv = o2w + 16;
    float b0, b1, b2;
    const float hb0 = 1 - hb1 - hb2;                                               // trianglemesh.cpp:270-272
    const float tu = hb0 * r.uv[0] + hb1 * r.uv[2] + hb2 * r.uv[4];
    const float tv_ = hb0 * r.uv[1] + hb1 * r.uv[3] + hb2 * r.uv[5];
    const float A00 = r.uv[2] - r.uv[0], A01 = r.uv[4] - r.uv[0], A10 = r.uv[3] - r.uv[1], A11 = r.uv[5] - r.uv[1];
    const float C0 = tu - r.uv[0], C1 = tv_ - r.uv[1];
    const float det = A00 * A11 - A01 * A10;                                       // SolveLinearSystem2x2 util.cpp:99-108
    if (fabsf(det) < 1e-5) b0 = b1 = b2 = 1.f / 3.f;
    else {
        const float invDet = 1.0f / det;
        b1 = (A11 * C0 - A01 * C1) * invDet;
        b2 = (A00 * C1 - A10 * C0) * invDet;
        b0 = 1.f - b1 - b2;
    }
    const V3 n0 = mk3(r.n[0], r.n[1], r.n[2]), n1 = mk3(r.n[3], r.n[4], r.n[5]), n2 = mk3(r.n[6], r.n[7], r.n[8]);
    V3 ns, ss, ts;
    if (r.flags & RT_SHADING_N) ns = normalize3(xform_normal_rm(o2wInv, (n0 * b0 + n1 * b1) + n2 * b2)); else ns = ngeom;
    if (r.flags & RT_SHADING_S) {
        const V3 s0 = mk3(r.s[0], r.s[1], r.s[2]), s1 = mk3(r.s[3], r.s[4], r.s[5]), s2 = mk3(r.s[6], r.s[7], r.s[8]);
        ss = normalize3(xform_vec3_rm(o2w, (s0 * b0 + s1 * b1) + s2 * b2));
    } else ss = normalize3(mk3(r.dpdu[0], r.dpdu[1], r.dpdu[2]));
    ts = normalize3(cross3(ss, ns));
    ss = cross3(ts, ns);
    V3 dndu = mk3(0.f), dndv = mk3(0.f);
    if (r.flags & RT_SHADING_N) {
        const float du1 = r.uv[0] - r.uv[4], du2 = r.uv[2] - r.uv[4], dv1 = r.uv[1] - r.uv[5], dv2 = r.uv[3] - r.uv[5];
        const V3 dn1 = n0 - n2, dn2 = n1 - n2;
        const float determinant = du1 * dv2 - dv1 * du2;
        if (determinant != 0) {
            const float invdet = 1.f / determinant;
            dndu = (dv2 * dn1 - dv1 * dn2) * invdet;
            dndv = (-du2 * dn1 + du1 * dn2) * invdet;
        }
    }
    dndu = xform_vec3_rm(o2w, dndu); dndv = xform_vec3_rm(o2w, dndv);
    V3 snn = normalize3(cross3(ss, ts));                                           // DifferentialGeometry ctor shape.cpp:37-51
    if (flip) snn = snn * -1.f;
    const float du = .01f, zero = 0.f;                                             // Bump: displacement 0 everywhere; du only divides 0
    const V3 bdpdu = (ss + snn * ((zero - zero) / du)) + dndu * zero;
    const V3 bdpdv = (ts + snn * ((zero - zero) / du)) + dndv * zero;
    V3 bnn = normalize3(cross3(bdpdu, bdpdv));
    if (flip) bnn = bnn * -1.f;
    if (dot3(ngeom, bnn) < 0.f) bnn = bnn * -1.f;
    nn_out = bnn; sn_dir = bdpdu;
}
template <bool EXT> RT_DEV V3 vertex_ng(const Vertex &v) { return EXT ? v.ng : v.nn; }
template <bool EXT>
RT_DEV void make_vertex(const DevScene &sc, const Trav &tv, Vertex &v) {
    const float4 RT_G *q = RT_GPTR(const float4, sc.tri_shade) + size_t(2) * unsigned(tv.hit_prim);
    const float4 a = q[0], b = q[1];
    const unsigned bits = __float_as_uint(a.w);
    if (EXT && (bits & RT_PRIM_QUADRIC)) {
        const unsigned qi = __float_as_uint(RT_GPTR(const DevTri, sc.tris)[unsigned(tv.hit_prim)].q0.x);
        quadric_frame(sc, qi, (bits & 0x10000u) != 0, tv.o, tv.d, tv.maxt, v.p, v.nn, v.sn);
    } else {
        v.p = tv.o + tv.d * tv.maxt;                     // ray(t), geometry.h:210
        v.nn = mk3(a.x, a.y, a.z);                       // tri_frame(), precomputed per triangle on the host
        v.sn = mk3(b.x, b.y, b.z);
    }
    if (EXT) v.ng = v.nn;
    if (EXT && (bits & RT_PRIM_SHADING)) {               // per-vertex N / S: the shading frame depends on where the triangle was hit
        const DevTriShading RT_G &r = RT_GPTR(const DevTriShading, sc.tri_shading)[RT_GPTR(const int, sc.tri_shading_idx)[unsigned(tv.hit_prim)]];
        V3 sdir;
        shading_frame(sc, r, tv.b1, tv.b2, v.ng, (bits & 0x10000u) != 0, v.nn, sdir);
        v.sn = normalize3(sdir);
    }
    v.tn = cross3(v.nn, v.sn);
    v.wo = -tv.d;
    v.mat = int(bits & 0xffffu);
    v.light = __float_as_int(b.w);
}
// geometric normal (orientation flip applied) and area-light index of a primitive
template <bool EXT>
RT_DEV void prim_normal_light(const DevScene &sc, const Trav &tv, V3 &nn, int &light) {
    const unsigned prim = unsigned(tv.hit_prim);
    const float4 RT_G *q = RT_GPTR(const float4, sc.tri_shade) + size_t(2) * prim;
    const float4 a = q[0];
    nn = mk3(a.x, a.y, a.z);
    light = __float_as_int(q[1].w);
    if (EXT && (__float_as_uint(a.w) & RT_PRIM_QUADRIC) && light >= 0) {       // an emitting quadric: its normal at this hit
        V3 hp, sn;
        quadric_frame(sc, __float_as_uint(RT_GPTR(const DevTri, sc.tris)[prim].q0.x), (__float_as_uint(a.w) & 0x10000u) != 0, tv.o, tv.d, tv.maxt, hp, nn, sn);
    }
}

RT_DEV V3 to_local(const Vertex &v, V3 w) { return mk3(dot3(w, v.sn), dot3(w, v.tn), dot3(w, v.nn)); }
RT_DEV V3 to_world(const Vertex &v, V3 w) {
    return mk3(v.sn.x * w.x + v.tn.x * w.y + v.nn.x * w.z, v.sn.y * w.x + v.tn.y * w.y + v.nn.y * w.z,
               v.sn.z * w.x + v.tn.z * w.y + v.nn.z * w.z);
}

// ---- BxDFs -------------------------------------------------------------------------------
RT_DEV float sin_theta(V3 w) { return sqrtf(fmaxf(0.f, 1.f - w.z * w.z)); }
RT_DEV float sin_theta2(V3 w) { return fmaxf(0.f, 1.f - w.z * w.z); }
RT_DEV float cos_phi(V3 w) { float s = sin_theta(w); if (s == 0.f) return 1.f; return clampf(w.x / s, -1.f, 1.f); }
RT_DEV float sin_phi(V3 w) { float s = sin_theta(w); if (s == 0.f) return 0.f; return clampf(w.y / s, -1.f, 1.f); }

template <class P> RT_DEV V3 mat_color(P c) { return mk3(c[0], c[1], c[2]); }

// Lambertian::f reflection.cpp:128-131 / OrenNayar::f :132-156
RT_DEV V3 diffuse_f(MatRef m, V3 wo, V3 wi) {
    V3 R = mat_color(m.r);
    if (m.on_b < 0.f) return R * RT_INV_PI;
    float sinthetai = sin_theta(wi), sinthetao = sin_theta(wo);
    float maxcos = 0.f;
    if (sinthetai > 1e-4 && sinthetao > 1e-4) {
        float sinphii = sin_phi(wi), cosphii = cos_phi(wi);
        float sinphio = sin_phi(wo), cosphio = cos_phi(wo);
        float dcos = cosphii * cosphio + sinphii * sinphio;
        maxcos = fmaxf(0.f, dcos);
    }
    float sinalpha, tanbeta;
    if (fabsf(wi.z) > fabsf(wo.z)) { sinalpha = sinthetao; tanbeta = sinthetai / fabsf(wi.z); }
    else { sinalpha = sinthetai; tanbeta = sinthetao / fabsf(wo.z); }
    return R * RT_INV_PI * (m.on_a + m.on_b * maxcos * sinalpha * tanbeta);
}

// FresnelDielectric::Evaluate reflection.cpp:77-95 + FrDiel :31-39 (all three channels equal)
RT_DEV float fresnel_dielectric(float cosi, float eta_i, float eta_t) {
    cosi = clampf(cosi, -1.f, 1.f);
    bool entering = cosi > 0.f;
    float ei = eta_i, et = eta_t;
    if (!entering) { float tmp = ei; ei = et; et = tmp; }
    float sint = ei / et * sqrtf(fmaxf(0.f, 1.f - cosi * cosi));
    if (sint >= 1.f) return 1.f;
    float cost = sqrtf(fmaxf(0.f, 1.f - sint * sint));
    float ac = fabsf(cosi);
    float Rparl = ((et * ac) - (ei * cost)) / ((et * ac) + (ei * cost));
    float Rperp = ((ei * ac) - (et * cost)) / ((ei * ac) + (et * cost));
    return (Rparl * Rparl + Rperp * Rperp) * (1.f / 2.f);
}

// ConcentricSampleDisk core/mc.cpp:92-135
RT_DEV void concentric_disk(float u1, float u2, float &dx, float &dy) {
    float r, theta;
    float sx = 2 * u1 - 1;
    float sy = 2 * u2 - 1;
    if (sx == 0.0f && sy == 0.0f) { dx = 0.f; dy = 0.f; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.0f) theta = sy / r; else theta = 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= RT_PI / 4.f;
    dx = r * cosf(theta);
    dy = r * sinf(theta);
}

// Microfacet::f reflection.cpp:163-175 with Blinn::D (reflection.h:315-320), Microfacet::G (:293-301) and
// FresnelDielectric(1.5, 1) (plastic.cpp:59-60)
RT_DEV float min_std(float a, float b) { return (b < a) ? b : a; }            // std::min
RT_DEV V3 microfacet_f(MatRef m, V3 wo, V3 wi) {
    const float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
    if (cosThetaI == 0.f || cosThetaO == 0.f) return mk3(0.f);
    V3 wh = wi + wo;
    if (wh.x == 0.f && wh.y == 0.f && wh.z == 0.f) return mk3(0.f);
    wh = normalize3(wh);
    const float cosThetaH = dot3(wi, wh);
    const float F = fresnel_dielectric(cosThetaH, 1.5f, 1.f);
    const float D = (m.exponent + 2) * RT_INV_TWOPI * powf(fabsf(wh.z), m.exponent);
    const float NdotWh = fabsf(wh.z), NdotWo = fabsf(wo.z), NdotWi = fabsf(wi.z), WOdotWh = absdot3(wo, wh);
    const float G = min_std(1.f, min_std((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
    return div_s(((mat_color(m.ks) * D) * G) * mk3(F), 4.f * cosThetaI * cosThetaO);
}
// Blinn::Pdf reflection.cpp:263-272
RT_DEV float blinn_pdf(float exponent, V3 wo, V3 wi) {
    const V3 H = normalize3(wo + wi);
    const float costheta = fabsf(H.z);
    float p = ((exponent + 1.f) * powf(costheta, exponent)) / (2.f * RT_PI * 4.f * dot3(wo, H));
    if (dot3(wo, H) <= 0.f) p = 0.f;
    return p;
}
// BxDF::Pdf of the two non-specular lobes: reflection.cpp:227-230 (cosine) and Microfacet::Pdf :241-245
RT_DEV float diffuse_pdf(V3 wo, V3 wi) { return (wo.z * wi.z > 0.f) ? fabsf(wi.z) * RT_INV_PI : 0.f; }
RT_DEV float glossy_pdf(MatRef m, V3 wo, V3 wi) { return (wo.z * wi.z > 0.f) ? blinn_pdf(m.exponent, wo, wi) : 0.f; }

// lobes in the order the material adds them: matte {diffuse}; plastic {diffuse, glossy} (plastic.cpp:66-67);
// mirror {specular R}; glass {specular R, specular T} (glass.cpp:56-61, only the non-black ones)
// non-specular lobes of a material: matte {D}; plastic {D, G}; uber {D if op*Kd != 0, G if op*Ks != 0} (uber.cpp:67-79)
template <bool EXT> RT_DEV bool mat_has_diffuse(MatRef m) { return m.type == RT_MAT_MATTE || (EXT && (m.type == RT_MAT_PLASTIC || (m.type == RT_MAT_UBER && m.has_r))); }
template <bool EXT> RT_DEV bool mat_has_glossy(MatRef m) { return EXT && (m.type == RT_MAT_PLASTIC || (m.type == RT_MAT_UBER && m.has_g)); }
RT_DEV bool flags_match(int type, int flags) { return (type & flags) == type; }

template <bool EXT>
RT_DEV int bsdf_num_components(MatRef m, int flags) {
    int n = 0;
    if (EXT && m.type == RT_MAT_UBER) {               // lobe order T, D, G, R (uber.cpp:62-87)
        if (m.has_t && flags_match(BX_TRANSMISSION | BX_SPECULAR, flags)) ++n;
        if (m.has_r && flags_match(BX_REFLECTION | BX_DIFFUSE, flags)) ++n;
        if (m.has_g && flags_match(BX_REFLECTION | BX_GLOSSY, flags)) ++n;
        if (m.has_kr && flags_match(BX_REFLECTION | BX_SPECULAR, flags)) ++n;
        return n;
    }
    if (m.type == RT_MAT_MATTE || (EXT && m.type == RT_MAT_PLASTIC)) {
        if (((BX_REFLECTION | BX_DIFFUSE) & flags) == (BX_REFLECTION | BX_DIFFUSE)) ++n;
        if (EXT && m.type == RT_MAT_PLASTIC && ((BX_REFLECTION | BX_GLOSSY) & flags) == (BX_REFLECTION | BX_GLOSSY)) ++n;
    } else {
        if (m.has_r && ((BX_REFLECTION | BX_SPECULAR) & flags) == (BX_REFLECTION | BX_SPECULAR)) ++n;
        if (m.type == RT_MAT_GLASS && m.has_t && ((BX_TRANSMISSION | BX_SPECULAR) & flags) == (BX_TRANSMISSION | BX_SPECULAR)) ++n;
    }
    return n;
}
template <bool EXT> RT_DEV int bsdf_total_components(MatRef m) { return bsdf_num_components<EXT>(m, BX_ALL); }

// sum of f over the non-specular reflection lobes matching `flags`, in lobe order (BSDF::f's loop, reflection.cpp:489-492)
template <bool EXT>
RT_DEV V3 bsdf_f_lobes(MatRef m, V3 wo, V3 wi, int flags) {
    V3 f = mk3(0.f);
    if (mat_has_diffuse<EXT>(m) && flags_match(BX_REFLECTION | BX_DIFFUSE, flags)) f = f + diffuse_f(m, wo, wi);
    if (mat_has_glossy<EXT>(m) && flags_match(BX_REFLECTION | BX_GLOSSY, flags)) f = f + microfacet_f(m, wo, wi);
    return f;
}

// BSDF::f reflection.cpp:480-494 (flags = BSDF_ALL): only the non-specular lobes have a non-zero f
template <bool EXT>
RT_DEV V3 bsdf_f(MatRef m, const Vertex &v, V3 woW, V3 wiW) {
    if (!mat_has_diffuse<EXT>(m) && !mat_has_glossy<EXT>(m)) return mk3(0.f);
    V3 wi = to_local(v, wiW), wo = to_local(v, woW);
    if (dot3(wiW, vertex_ng<EXT>(v)) * dot3(woW, vertex_ng<EXT>(v)) > 0) return bsdf_f_lobes<EXT>(m, wo, wi, BX_ALL & ~BX_TRANSMISSION);   // BRDFs only
    return mk3(0.f);                                                                                         // BTDFs only: none
}

// BSDF::Pdf reflection.cpp:458-470 (flags = BSDF_ALL)
template <bool EXT>
RT_DEV float bsdf_pdf(MatRef m, const Vertex &v, V3 woW, V3 wiW) {
    int nc = bsdf_total_components<EXT>(m);
    if (nc == 0) return 0.f;
    if (!mat_has_diffuse<EXT>(m) && !mat_has_glossy<EXT>(m)) return 0.f / float(nc);
    V3 wo = to_local(v, woW), wi = to_local(v, wiW);
    float pdf = 0.f;
    if (mat_has_diffuse<EXT>(m)) pdf += diffuse_pdf(wo, wi);
    if (mat_has_glossy<EXT>(m)) pdf += glossy_pdf(m, wo, wi);
    return pdf / nc;
}

// BSDF::Sample_f reflection.cpp:402-457.  Returns f; pdf == 0 means "no sample".
template <bool EXT>
RT_DEV V3 bsdf_sample_f(MatRef m, const Vertex &v, V3 woW, V3 &wiW, float u1, float u2, float u3,
                        float &pdf, int flags, int &sampled) {
    sampled = 0; pdf = 0.f;
    int matching = bsdf_num_components<EXT>(m, flags);
    if (matching == 0) return mk3(0.f);
    int which = min(int(floorf(u3 * matching)), matching - 1);     // Floor2Int(double(u3*matching))
    V3 wo = to_local(v, woW);
    V3 wi, f;
    const bool diffuse_has = mat_has_diffuse<EXT>(m) && flags_match(BX_REFLECTION | BX_DIFFUSE, flags);
    const bool glossy_has = mat_has_glossy<EXT>(m) && flags_match(BX_REFLECTION | BX_GLOSSY, flags);
    // which lobe `which` designates: matte / plastic {D, G}; uber {T, D, G, R}; mirror / glass {R, T}
    int pick = 0;                                                   // 1 = diffuse, 2 = glossy, 3 = specular R, 4 = specular T
    if (EXT && m.type == RT_MAT_UBER) {
        int idx = which;
        const bool mT = m.has_t && flags_match(BX_TRANSMISSION | BX_SPECULAR, flags), mR = m.has_kr && flags_match(BX_REFLECTION | BX_SPECULAR, flags);
        if (mT && idx-- == 0) pick = 4;
        else if (diffuse_has && idx-- == 0) pick = 1;
        else if (glossy_has && idx-- == 0) pick = 2;
        else if (mR) pick = 3;
    } else if (diffuse_has || glossy_has) pick = (!glossy_has || (diffuse_has && which == 0)) ? 1 : 2;   // glossy_has is constant false without EXT
    if (pick == 1 || pick == 2) {
        if (pick == 1) {
            // BxDF::Sample_f reflection.cpp:219-226 with CosineSampleHemisphere mc.h:38-44
            float dx, dy; concentric_disk(u1, u2, dx, dy);
            wi = mk3(dx, dy, sqrtf(fmaxf(0.f, 1.f - dx * dx - dy * dy)));
            if (wo.z < 0.f) wi.z *= -1.f;
            pdf = diffuse_pdf(wo, wi);
            if (pdf == 0.f) return mk3(0.f);
            sampled = BX_REFLECTION | BX_DIFFUSE;
            if (glossy_has) pdf += glossy_pdf(m, wo, wi);                        // the other matching non-specular lobe, :436-443
        } else {
            // Microfacet::Sample_f reflection.cpp:235-240 with Blinn::Sample_f :246-262
            const float costheta = powf(u1, 1.f / (m.exponent + 1));
            const float sintheta = sqrtf(fmaxf(0.f, 1.f - costheta * costheta));
            const float phi = u2 * 2.f * RT_PI;
            V3 H = mk3(sintheta * cosf(phi), sintheta * sinf(phi), costheta);
            if (!(wo.z * H.z > 0.f)) H = -H;
            wi = -wo + H * (2.f * dot3(wo, H));
            pdf = ((m.exponent + 1.f) * powf(costheta, m.exponent)) / (2.f * RT_PI * 4.f * dot3(wo, H));
            if (dot3(wo, H) <= 0.f) pdf = 0.f;
            if (pdf == 0.f) return mk3(0.f);
            sampled = BX_REFLECTION | BX_GLOSSY;
            if (diffuse_has) pdf += diffuse_pdf(wo, wi);
        }
        wiW = to_world(v, wi);
        if (matching > 1) pdf /= matching;
        f = mk3(0.f);
        if (dot3(wiW, vertex_ng<EXT>(v)) * dot3(woW, vertex_ng<EXT>(v)) > 0) f = bsdf_f_lobes<EXT>(m, wo, wi, flags & ~BX_TRANSMISSION);
        return f;
    }
    // specular lobes, in the order the material added them (glass.cpp:56-61, mirror.cpp:51-53)
    bool reflect_has = m.has_r && ((BX_REFLECTION | BX_SPECULAR) & flags) == (BX_REFLECTION | BX_SPECULAR);
    bool pick_reflect = (EXT && m.type == RT_MAT_UBER) ? pick == 3 : (reflect_has && which == 0);
    if (pick_reflect) {
        // SpecularReflection::Sample_f reflection.cpp:96-103
        wi = mk3(-wo.x, -wo.y, wo.z);
        pdf = 1.f;
        const bool uber = EXT && m.type == RT_MAT_UBER;
        float F = (m.type == RT_MAT_GLASS) ? fresnel_dielectric(wo.z, 1.f, m.ior) : (uber ? fresnel_dielectric(wo.z, 1.5f, 1.f) : 1.f);
        f = div_s(mk3(F) * (uber ? mat_color(m.kr) : mat_color(m.r)), fabsf(wi.z));
        sampled = BX_REFLECTION | BX_SPECULAR;
    } else {
        // SpecularTransmission::Sample_f reflection.cpp:104-127
        bool entering = wo.z > 0.f;
        float ei = 1.f, et = m.ior;
        if (!entering) { float tmp = ei; ei = et; et = tmp; }
        float sini2 = sin_theta2(wo);
        float eta = ei / et;
        float sint2 = eta * eta * sini2;
        if (sint2 >= 1.f) return mk3(0.f);                 // pdf stays 0
        float cost = sqrtf(fmaxf(0.f, 1.f - sint2));
        if (entering) cost = -cost;
        wi = mk3(eta * -wo.x, eta * -wo.y, cost);
        pdf = 1.f;
        float F = fresnel_dielectric(wo.z, 1.f, m.ior);
        f = div_s(((mk3(1.f) - mk3(F)) * ((et * et) / (ei * ei))) * mat_color(m.t), fabsf(wi.z));
        sampled = BX_TRANSMISSION | BX_SPECULAR;
    }
    wiW = to_world(v, wi);
    if (matching > 1) pdf /= matching;
    return f;
}

// ---- lights ------------------------------------------------------------------------------
RT_DEV void light_tri(const DevScene &sc, unsigned k, V3 &p1, V3 &p2, V3 &p3) {
    const float RT_G *t = RT_GPTR(const float, sc.light_tris) + size_t(k) * 16;
    p1 = mk3(t[0], t[1], t[2]); p2 = mk3(t[3], t[4], t[5]); p3 = mk3(t[6], t[7], t[8]);
}

// ---- delta lights: {Point,Spot,Distant}Light::Sample_L(p, &wi, &visibility) (point.cpp:55-60, spot.cpp:61-79,
// distant.cpp:57-62).  Returns the incident radiance; `sd`, `smax` = the visibility ray's direction and maxt
// (SetSegment: p -> light position, maxt 1 - eps; SetRay: p along wi, unbounded; light.h:78-83).
RT_DEV bool light_is_delta(LightRef Lt) { return Lt.type != RT_LIGHT_AREA; }
RT_DEV V3 delta_light_sample(LightRef Lt, V3 p, V3 &wi, V3 &sd, float &smax) {
    if (Lt.type == RT_LIGHT_DISTANT) {
        wi = mk3(Lt.dir[0], Lt.dir[1], Lt.dir[2]);
        sd = wi; smax = RT_INF;
        return mat_color(Lt.color);
    }
    const V3 lp = mat_color(Lt.pos);
    const V3 dd = lp - p;
    wi = normalize3(dd);
    sd = dd; smax = 1.f - RT_RAY_EPSILON;
    const float d2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;                  // DistanceSquared geometry.h
    if (Lt.type == RT_LIGHT_POINT) return div_s(mat_color(Lt.color), d2);
    // SpotLight::Falloff(-wi)
    const V3 w = -wi;
    const V3 wl = normalize3(mk3(Lt.w2l[0] * w.x + Lt.w2l[1] * w.y + Lt.w2l[2] * w.z, Lt.w2l[3] * w.x + Lt.w2l[4] * w.y + Lt.w2l[5] * w.z,
                                 Lt.w2l[6] * w.x + Lt.w2l[7] * w.y + Lt.w2l[8] * w.z));
    const float costheta = wl.z;
    float fall;
    if (costheta < Lt.cos_total) fall = 0.f;
    else if (costheta > Lt.cos_falloff) fall = 1.f;
    else { const float delta = (costheta - Lt.cos_total) / (Lt.cos_falloff - Lt.cos_total); fall = delta * delta * delta * delta; }
    return div_s(mat_color(Lt.color) * fall, d2);
}

// ---- area lights on quadrics: {Sphere,Disk,Cylinder}::Sample / ::Pdf / ::Area (sphere.cpp:36-86,:251-253, disk.cpp:37-46,:124-127,
// cylinder.cpp:38-45,:180-182) and the Shape defaults (shape.h:89-107).  EXT kernels only.
RT_DEV V3 xform_normal(const float *minv, V3 n) {                               // Transform::operator()(Normal) transform.h:106-112
    return mk3(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z, minv[2] * n.x + minv[6] * n.y + minv[10] * n.z);
}
RT_DEV float dist_sq(V3 a, V3 b) { const V3 d = a - b; return d.x * d.x + d.y * d.y + d.z * d.z; }
RT_DEV V3 quadric_sample_uniform(const DevQuadric RT_G &q, bool ro, float u1, float u2, V3 &ns) {   // Shape::Sample(u1, u2, Ns)
    V3 p, n;
    if (q.type == RT_QUADRIC_DISK) {
        float dx, dy; concentric_disk(u1, u2, dx, dy);
        p = mk3(dx * q.radius, dy * q.radius, q.zmin);
        n = mk3(0.f, 0.f, 1.f);
    } else if (q.type == RT_QUADRIC_CYLINDER) {
        const float z = (1.f - u1) * q.zmin + u1 * q.zmax;                         // Lerp
        const float t = u2 * q.phi_max;
        p = mk3(q.radius * cosf(t), q.radius * sinf(t), z);
        n = mk3(p.x, p.y, 0.f);
    } else {
        const float z = 1.f - 2.f * u1;                                             // UniformSampleSphere mc.cpp:74-81
        const float r = sqrtf(fmaxf(0.f, 1.f - z * z));
        const float phi = 2.f * RT_PI * u2;
        p = mk3(0.f) + mk3(r * cosf(phi), r * sinf(phi), z) * q.radius;
        n = p;
    }
    ns = normalize3(xform_normal(q.w2o, n));
    if (ro) ns = ns * -1.f;
    return xform_point(q.o2w, p);
}
RT_DEV V3 quadric_sample(const DevScene &sc, unsigned qi, bool ro, V3 p, float u1, float u2, V3 &ns) {   // Shape::Sample(p, u1, u2, Ns)
    const DevQuadric RT_G &q = RT_GPTR(const DevQuadric, sc.quadrics)[qi];
    if (q.type != RT_QUADRIC_SPHERE) return quadric_sample_uniform(q, ro, u1, u2, ns);
    const V3 Pcenter = xform_point(q.o2w, mk3(0.f));                                 // sphere.cpp:46-72
    const V3 wc = normalize3(Pcenter - p);
    V3 wcX, wcY;
    if (fabsf(wc.x) > fabsf(wc.y)) { const float il = 1.f / sqrtf(wc.x * wc.x + wc.z * wc.z); wcX = mk3(-wc.z * il, 0.f, wc.x * il); }   // CoordinateSystem
    else { const float il = 1.f / sqrtf(wc.y * wc.y + wc.z * wc.z); wcX = mk3(0.f, wc.z * il, -wc.y * il); }
    wcY = cross3(wc, wcX);
    if (dist_sq(p, Pcenter) - q.radius * q.radius < 1e-4f) return quadric_sample_uniform(q, ro, u1, u2, ns);
    const float cosThetaMax = sqrtf(fmaxf(0.f, 1.f - q.radius * q.radius / dist_sq(p, Pcenter)));
    const float costheta = (1.f - u1) * cosThetaMax + u1 * 1.f;                     // UniformSampleCone mc.cpp:154-161
    const float sintheta = sqrtf(1.f - costheta * costheta);
    const float phi = u2 * 2.f * RT_PI;
    const V3 rd = wcX * (cosf(phi) * sintheta) + wcY * (sinf(phi) * sintheta) + wc * costheta;
    float thit;
    if (!quadric_test(sc, qi, p, rd, RT_RAY_EPSILON, RT_INF, thit)) thit = dot3(Pcenter - p, normalize3(rd));
    const V3 ps = p + rd * thit;
    ns = normalize3(ps - Pcenter);
    if (ro) ns = ns * -1.f;
    return ps;
}
RT_DEV float quadric_light_pdf(const DevScene &sc, unsigned qi, V3 p, V3 wi) {   // Shape::Pdf(p, wi) shape.h:96-107; Sphere::Pdf sphere.cpp:73-86
    const DevQuadric RT_G &q = RT_GPTR(const DevQuadric, sc.quadrics)[qi];
    if (q.type == RT_QUADRIC_SPHERE) {
        const V3 Pcenter = xform_point(q.o2w, mk3(0.f));
        if (!(dist_sq(p, Pcenter) - q.radius * q.radius < 1e-4f)) {
            const float cosThetaMax = sqrtf(fmaxf(0.f, 1.f - q.radius * q.radius / dist_sq(p, Pcenter)));
            return 1.f / (2.f * RT_PI * (1.f - cosThetaMax));                      // UniformConePdf mc.cpp:142-144
        }
    }
    float thit;
    if (!quadric_test(sc, qi, p, wi, RT_RAY_EPSILON, RT_INF, thit)) return 0.f;
    V3 hp, nn, sn; quadric_frame(sc, qi, false, p, wi, thit, hp, nn, sn);
    const float area = q.type == RT_QUADRIC_DISK ? q.phi_max * 0.5f * (q.radius * q.radius - q.zmax * q.zmax)
                     : (q.type == RT_QUADRIC_CYLINDER ? (q.zmax - q.zmin) * q.phi_max * q.radius : q.phi_max * q.radius * (q.zmax - q.zmin));
    float pdf = dist_sq(p, p + wi * thit) / (absdot3(nn, -wi) * area);
    if (absdot3(nn, -wi) == 0.f) pdf = 0.f;
    return pdf;
}

// Shape::Pdf(p, wi) shape.h:96-107 evaluated on the emitter's own triangles:
// ShapeSet::Intersect (shape.h:150-156) keeps the LAST triangle hit, the ray's maxt is never shortened.
template <bool EXT>
RT_DEV float area_light_pdf(const DevScene &sc, LightRef L, V3 p, V3 wi) {
    if (EXT && L.quadric >= 0) return quadric_light_pdf(sc, unsigned(L.quadric), p, wi);
    bool any = false; float thit = 0.f; V3 nl = mk3(0.f);
    for (unsigned k = 0; k < L.n_tris; ++k) {
        V3 p1, p2, p3; light_tri(sc, L.first_tri + k, p1, p2, p3);
        float t, b1, b2;
        if (tri_test(p1, p2 - p1, p3 - p1, p, wi, RT_RAY_EPSILON, RT_INF, t, b1, b2)) {
            any = true; thit = t;
            const float RT_G *ltr = RT_GPTR(const float, sc.light_tris) + size_t(L.first_tri + k) * 16;
            nl = mk3(ltr[12], ltr[13], ltr[14]);             // tri_frame() of the emitter triangle, precomputed
        }
    }
    if (!any) return 0.f;
    V3 ph = p + wi * thit;
    V3 dd = p - ph;
    float ad = absdot3(nl, -wi);
    float pdf = (dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) / (ad * L.area);
    if (ad == 0.f) pdf = 0.f;
    return pdf;
}

// AreaLight::L light.h:93-96
RT_DEV V3 area_L(LightRef L, V3 n, V3 w) { return dot3(n, w) > 0 ? mat_color(L.color) : mk3(0.f); }

// shape->Sample(p,u1,u2,&ns): ShapeSet::Sample shape.h:115-121 (one extra RandomFloat when the emitter has
// more than one triangle) + Triangle::Sample trianglemesh.cpp:336-349 + UniformSampleTriangle mc.cpp:136-141
template <bool EXT, class RNG>
RT_DEV V3 area_sample_point(const DevScene &sc, LightRef L, V3 pref, float u1, float u2, RNG &rng, V3 &ns) {
    if (EXT && L.quadric >= 0) return quadric_sample(sc, unsigned(L.quadric), L.reverse_orientation != 0, pref, u1, u2, ns);
    unsigned k = 0;
    if (L.n_tris > 1) {
        float ls = rng.next_float();
        for (k = 0; k < L.n_tris - 1; ++k)
            if (ls < RT_GPTR(const float, sc.light_tris)[size_t(L.first_tri + k) * 16 + 10]) break;
    }
    V3 p1, p2, p3; light_tri(sc, L.first_tri + k, p1, p2, p3);
    float su1 = sqrtf(u1);
    float b1 = 1.f - su1, b2 = u2 * su1;
    V3 ps = b1 * p1 + b2 * p2 + (1.f - b1 - b2) * p3;
    ns = normalize3(cross3(p2 - p1, p3 - p1));
    if (L.reverse_orientation) ns = ns * -1.f;
    return ps;
}

}  // namespace rt
