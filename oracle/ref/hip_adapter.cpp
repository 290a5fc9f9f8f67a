// oracle/ref/hip_adapter.cpp -- the reference-side binding of the drop-in boundary, as a real pbrt-v1 plugin.
//
// TEST INFRASTRUCTURE built only in the authoring container (oracle/ref/Makefile -> oracle/_ref/bin/hip.so); the reference's
// headers and plugin sources are #included where they lie under /root/reference, nothing is copied.
//
// pbrt-v1 loads plugins by name and resolves unmangled factory symbols (core/dynload.cpp:185-205, :485-509).  hip.so exports two:
//     Primitive         *CreateAccelerator(const vector<Reference<Primitive> >&, const ParamSet&)   (accelerators/kdtree.cpp:489)
//     SurfaceIntegrator *CreateSurfaceIntegrator(const ParamSet&)                                      (integrators/whitted.cpp:141)
// used from an ordinary scene file as
//     SurfaceIntegrator "hip" "string inner" ["path"] "integer maxdepth" [5]
//     Accelerator "hip" "string inner" ["kdtree"]
// CreateAccelerator sees the primitives api.cpp hands to the accelerator (api.cpp:497-502) and keeps them; the integrator's
// Preprocess(scene) (called by Scene::Render before the sample loop, core/scene.cpp:38) walks the reference's OWN objects --
// Scene, Camera, Film, Sampler, Light, GeometricPrimitive, TriangleMesh, Material, VolumeRegion -- and flattens them into the
// RtSceneDesc / RtRenderDesc of include/pbrt_hip.h.  Then, depending on the environment:
//     PBRT_HIP_DESC_DUMP=<file>   writes rt_desc_serialize()'s byte image (tests compare it with the product's front end);
//     PBRT_HIP_LIB=<libpbrt_hip.so> renders the frame through the C ABI (rt_scene_create, rt_film_bind, rt_render,
//                                 rt_samples_read) and Li() then returns the device's radiance for each camera sample, so the
//                                 reference's own Scene::Render loop, ImageFilm and image writer produce the picture;
//     neither                     Li() forwards to the reference's inner integrator (the plugin is then a pass-through).
// Class internals are read through `#define private public`: this file is an inspector of the reference, not part of it.
#include <vector>
#include <string>
#include <map>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <iostream>
#include <sstream>
#include <dlfcn.h>
#define private public
#define protected public
#include "pbrt.h"
#include "scene.h"
#include "primitive.h"
#include "shape.h"
#include "light.h"
#include "camera.h"
#include "sampling.h"
#include "film.h"
#include "material.h"
#include "texture.h"
#include "transport.h"
#include "volume.h"
#include "paramset.h"
#include "dynload.h"
// plugin classes live in .cpp files: include those where they lie, with their factory symbols renamed
#define CreateShape CreateShape_trianglemesh_
#include "shapes/trianglemesh.cpp"
#undef CreateShape
#define CreateMaterial CreateMaterial_matte_
#include "materials/matte.cpp"
#undef CreateMaterial
#define CreateMaterial CreateMaterial_glass_
#include "materials/glass.cpp"
#undef CreateMaterial
#define CreateMaterial CreateMaterial_mirror_
#include "materials/mirror.cpp"
#undef CreateMaterial
#define CreateMaterial CreateMaterial_plastic_
#include "materials/plastic.cpp"
#undef CreateMaterial
#define CreateMaterial CreateMaterial_uber_
#include "materials/uber.cpp"
#undef CreateMaterial
#define CreateShape CreateShape_sphere_
#include "shapes/sphere.cpp"
#undef CreateShape
#define CreateShape CreateShape_disk_
#include "shapes/disk.cpp"
#undef CreateShape
#define CreateShape CreateShape_cylinder_
#include "shapes/cylinder.cpp"
#undef CreateShape
#define CreateShape CreateShape_cone_
#include "shapes/cone.cpp"
#undef CreateShape
#define CreateShape CreateShape_paraboloid_
#include "shapes/paraboloid.cpp"
#undef CreateShape
#define CreateShape CreateShape_hyperboloid_
#include "shapes/hyperboloid.cpp"
#undef CreateShape
#define CreateLight CreateLight_point_
#include "lights/point.cpp"
#undef CreateLight
#define CreateLight CreateLight_spot_
#include "lights/spot.cpp"
#undef CreateLight
#define CreateLight CreateLight_distant_
#include "lights/distant.cpp"
#undef CreateLight
#define CreateCamera CreateCamera_orthographic_
#include "cameras/orthographic.cpp"
#undef CreateCamera
#define CreateCamera CreateCamera_environment_
#include "cameras/environment.cpp"
#undef CreateCamera
#define CreateCamera CreateCamera_perspective_
#include "cameras/perspective.cpp"
#undef CreateCamera
#define CreateSampler CreateSampler_stratified_
#include "samplers/stratified.cpp"
#undef CreateSampler
#define CreateSampler CreateSampler_ld_
#include "samplers/lowdiscrepancy.cpp"
#undef CreateSampler
#define CreateSampler CreateSampler_random_
#include "samplers/random.cpp"
#undef CreateSampler
#define CreateFilm CreateFilm_image_
#include "film/image.cpp"
#undef CreateFilm
#define CreateVolumeRegion CreateVolumeRegion_homogeneous_
#include "volumes/homogeneous.cpp"
#undef CreateVolumeRegion
#define CreateVolumeIntegrator CreateVolumeIntegrator_emission_
#include "integrators/emission.cpp"
#undef CreateVolumeIntegrator
#define CreateVolumeIntegrator CreateVolumeIntegrator_single_
#include "integrators/single.cpp"
#undef CreateVolumeIntegrator
#undef private
#undef protected
#include "../../include/pbrt_hip_desc.h"

namespace {

[[noreturn]] void die(const char *what) { fprintf(stderr, "hip_adapter: %s is not handled by the adapter\n", what); abort(); }
// is the dynamic type the (global-namespace) class `name`?  Compared by mangled name: the object's class was compiled into another
// plugin .so, so its type_info object is not the one this file would get from typeid(Class).
bool type_is(const std::type_info &ti, const char *name) {
    const char *n = ti.name(); if (*n == '*') ++n;
    char want[96]; snprintf(want, sizeof want, "%zu%s", strlen(name), name);
    return !strcmp(n, want);
}
extern "C" const void *g_keyed_inner_sampler;
extern "C" unsigned g_keyed_seed;

// ---- what CreateAccelerator saw (api.cpp:497-502)
std::vector<Reference<Primitive> > g_prims;
RtAccelParams g_accel;

struct Flat {
    std::vector<float> tri_verts; std::vector<uint16_t> tri_material; std::vector<int32_t> tri_light; std::vector<uint8_t> tri_flags;
    std::vector<RtMaterial> materials; std::vector<RtLight> lights; std::vector<float> light_tris;
    std::vector<int32_t> tri_shading; std::vector<RtTriShading> shading; std::vector<float> xforms; std::vector<RtQuadric> quadrics;
    RtSceneDesc scene; RtRenderDesc render;
};

void spec3(const Spectrum &s, float out[3]) { for (int k = 0; k < 3; ++k) out[k] = s.c[k]; }
DifferentialGeometry dummy_dg() { static ShapeSet *none = NULL; (void)none; DifferentialGeometry dg; memset((void *)&dg, 0, sizeof dg); return dg; }

RtMaterial flatten_material(const Material *m) {
    RtMaterial o; memset(&o, 0, sizeof o);
    const DifferentialGeometry dg = dummy_dg();                               // constant textures ignore it (texture.h:113-123)
    const std::type_info &ti = typeid(*m);
    if (type_is(ti, "Matte")) {                                              // matte.cpp:46-64: Kd.Clamp(), Clamp(sigma, 0, 90)
        const Matte *x = static_cast<const Matte *>(m);
        spec3(x->Kd->Evaluate(dg).Clamp(), o.kd); o.sigma = Clamp(x->sigma->Evaluate(dg), 0.f, 90.f); o.type = RT_MAT_MATTE; o.ior = 1.f;
    } else if (type_is(ti, "Plastic")) {                                     // plastic.cpp:47-69
        const Plastic *x = static_cast<const Plastic *>(m);
        spec3(x->Kd->Evaluate(dg).Clamp(), o.kd); spec3(x->Ks->Evaluate(dg).Clamp(), o.ks); o.roughness = x->roughness->Evaluate(dg); o.type = RT_MAT_PLASTIC; o.ior = 1.f;
    } else if (type_is(ti, "Mirror")) {                                      // mirror.cpp:42-55
        const Mirror *x = static_cast<const Mirror *>(m);
        spec3(x->Kr->Evaluate(dg).Clamp(), o.kd); o.type = RT_MAT_MIRROR; o.ior = 1.f;
    } else if (type_is(ti, "Glass")) {                                       // glass.cpp:46-63
        const Glass *x = static_cast<const Glass *>(m);
        spec3(x->Kr->Evaluate(dg).Clamp(), o.kd); spec3(x->Kt->Evaluate(dg).Clamp(), o.kt); o.ior = x->index->Evaluate(dg); o.type = RT_MAT_GLASS;
    } else if (type_is(ti, "UberMaterial")) {                                // uber.cpp:52-89: op = opacity.Clamp(); T = -op + 1, D = op * Kd.Clamp(), G = op * Ks.Clamp(), R = op * Kr.Clamp()
        const UberMaterial *x = static_cast<const UberMaterial *>(m);
        const Spectrum op = x->opacity->Evaluate(dg).Clamp();
        spec3(-op + Spectrum(1.), o.kt); spec3(op * x->Kd->Evaluate(dg).Clamp(), o.kd); spec3(op * x->Ks->Evaluate(dg).Clamp(), o.ks); spec3(op * x->Kr->Evaluate(dg).Clamp(), o.kr);
        o.roughness = x->roughness->Evaluate(dg); o.type = RT_MAT_UBER; o.ior = 1.f;
    } else die(ti.name());
    return o;
}

void put_m(const Reference<Matrix4x4> &m, float *out);
// a quadric (shapes/{sphere,disk,cylinder,cone,paraboloid,hyperboloid}.cpp) as the constructors left it; false for any other shape
bool flatten_quadric(const Shape *sh, RtQuadric &q) {
    memset(&q, 0, sizeof q);
    const std::type_info &ti = typeid(*sh);
    if (type_is(ti, "Sphere")) {
        const Sphere *x = static_cast<const Sphere *>(sh);
        q.type = RT_QUADRIC_SPHERE; q.radius = x->radius; q.zmin = x->zmin; q.zmax = x->zmax; q.theta_min = x->thetaMin; q.theta_max = x->thetaMax; q.phi_max = x->phiMax;
    } else if (type_is(ti, "Disk")) {
        const Disk *x = static_cast<const Disk *>(sh);
        q.type = RT_QUADRIC_DISK; q.radius = x->radius; q.zmin = x->height; q.zmax = x->innerRadius; q.phi_max = x->phiMax;
    } else if (type_is(ti, "Cylinder")) {
        const Cylinder *x = static_cast<const Cylinder *>(sh);
        q.type = RT_QUADRIC_CYLINDER; q.radius = x->radius; q.zmin = x->zmin; q.zmax = x->zmax; q.phi_max = x->phiMax;
    } else if (type_is(ti, "Cone")) {
        const Cone *x = static_cast<const Cone *>(sh);
        q.type = RT_QUADRIC_CONE; q.radius = x->radius; q.zmin = 0.f; q.zmax = x->height; q.phi_max = x->phiMax;
    } else if (type_is(ti, "Paraboloid")) {
        const Paraboloid *x = static_cast<const Paraboloid *>(sh);
        q.type = RT_QUADRIC_PARABOLOID; q.radius = x->radius; q.zmin = x->zmin; q.zmax = x->zmax; q.phi_max = x->phiMax;
    } else if (type_is(ti, "Hyperboloid")) {
        const Hyperboloid *x = static_cast<const Hyperboloid *>(sh);
        q.type = RT_QUADRIC_HYPERBOLOID; q.radius = x->rmax; q.zmin = x->zmin; q.zmax = x->zmax; q.phi_max = x->phiMax;
        q.p1[0] = x->p1.x; q.p1[1] = x->p1.y; q.p1[2] = x->p1.z; q.p2[0] = x->p2.x; q.p2[1] = x->p2.y; q.p2[2] = x->p2.z; q.a = x->a; q.c = x->c;
    } else return false;
    put_m(sh->ObjectToWorld.m, q.object_to_world); put_m(sh->ObjectToWorld.mInv, q.world_to_object);
    return true;
}

void put_m(const Reference<Matrix4x4> &m, float *out) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = m->m[r][c]; }

void flatten(const Scene *scene, const ParamSet &ips, Flat &F) {
    memset(&F.scene, 0, sizeof F.scene); memset(&F.render, 0, sizeof F.render);
    // ---- primitives in the order KdTreeAccel refines them (kdtree.cpp:146-148: prims[i]->FullyRefine appends); the k-th quadric among
    // them is quadrics[k] (an area light on a quadric refers to it by that index)
    std::vector<Reference<Primitive> > refined;
    for (size_t i = 0; i < g_prims.size(); ++i) g_prims[i]->FullyRefine(refined);
    std::map<const Shape *, int> quadric_index;
    for (size_t i = 0; i < refined.size(); ++i) {
        if (!type_is(typeid(*refined[i].ptr), "GeometricPrimitive")) die("a primitive that is not a GeometricPrimitive");
        const GeometricPrimitive *gp = static_cast<const GeometricPrimitive *>(refined[i].ptr);
        RtQuadric q;
        if (flatten_quadric(gp->shape.ptr, q)) { quadric_index[gp->shape.ptr] = int(F.quadrics.size()); F.quadrics.push_back(q); }
    }
    // ---- lights (their indices are what tri_light refers to): Scene::lights in creation order (api.cpp:339-352, :362-366)
    std::map<const Light *, int> light_index;
    for (size_t i = 0; i < scene->lights.size(); ++i) {
        const Light *l = scene->lights[i];
        RtLight L; memset(&L, 0, sizeof L);
        L.n_samples = l->nSamples;
        const std::type_info &ti = typeid(*l);
        if (type_is(ti, "PointLight")) {                                    // point.cpp:49-54
            const PointLight *p = static_cast<const PointLight *>(l);
            L.type = RT_LIGHT_POINT; spec3(p->Intensity, L.color); L.pos[0] = p->lightPos.x; L.pos[1] = p->lightPos.y; L.pos[2] = p->lightPos.z;
        } else if (type_is(ti, "SpotLight")) {                               // spot.cpp:54-60; Falloff uses WorldToLight (spot.cpp:68-79)
            const SpotLight *p = static_cast<const SpotLight *>(l);
            L.type = RT_LIGHT_SPOT; spec3(p->Intensity, L.color); L.pos[0] = p->lightPos.x; L.pos[1] = p->lightPos.y; L.pos[2] = p->lightPos.z;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L.world_to_light[3 * r + c] = p->WorldToLight.m->m[r][c];
            L.cos_total_width = p->cosTotalWidth; L.cos_falloff_start = p->cosFalloffStart;
        } else if (type_is(ti, "DistantLight")) {                            // distant.cpp:51-56
            const DistantLight *p = static_cast<const DistantLight *>(l);
            L.type = RT_LIGHT_DISTANT; spec3(p->L, L.color); L.dir[0] = p->lightDir.x; L.dir[1] = p->lightDir.y; L.dir[2] = p->lightDir.z;
        } else if (type_is(ti, "AreaLight") && quadric_index.count(static_cast<const AreaLight *>(l)->shape.ptr)) {
            // an emitter on a quadric: AreaLight keeps a shape that CanIntersect as it is (area.cpp:38-39)
            const AreaLight *a = static_cast<const AreaLight *>(l);
            L.type = RT_LIGHT_AREA; spec3(a->Lemit, L.color);
            L.first_tri = 0; L.n_tris = 0; L.quadric_plus1 = quadric_index[a->shape.ptr] + 1;
            if (F.quadrics[L.quadric_plus1 - 1].type > RT_QUADRIC_CYLINDER) die("an area light on a cone / paraboloid / hyperboloid (Shape::Sample unimplemented, shape.h:84-88)");
            L.reverse_orientation = a->shape->reverseOrientation ? 1 : 0;
            L.flip_normal = (a->shape->reverseOrientation ^ a->shape->transformSwapsHandedness) ? 1 : 0;
        } else if (type_is(ti, "AreaLight")) {                               // area.cpp:28-54
            const AreaLight *a = static_cast<const AreaLight *>(l);
            L.type = RT_LIGHT_AREA; spec3(a->Lemit, L.color);
            std::vector<Reference<Shape> > parts;
            if (type_is(typeid(*a->shape.ptr), "ShapeSet")) parts = static_cast<const ShapeSet *>(a->shape.ptr)->shapes;
            else parts.push_back(a->shape);
            L.first_tri = uint32_t(F.light_tris.size() / 9); L.n_tris = uint32_t(parts.size());
            for (size_t k = 0; k < parts.size(); ++k) {
                if (!type_is(typeid(*parts[k].ptr), "Triangle")) die("an area light on a non-triangle shape");
                const Triangle *t = static_cast<const Triangle *>(parts[k].ptr);
                for (int c = 0; c < 3; ++c) { const Point &p = t->mesh->p[t->v[c]]; F.light_tris.push_back(p.x); F.light_tris.push_back(p.y); F.light_tris.push_back(p.z); }
                L.reverse_orientation = t->reverseOrientation ? 1 : 0;
                L.flip_normal = (t->reverseOrientation ^ t->transformSwapsHandedness) ? 1 : 0;
            }
        } else die(ti.name());
        light_index[l] = int(F.lights.size());
        F.lights.push_back(L);
    }
    std::map<const Material *, int> mat_index;
    std::map<const TriangleMesh *, uint32_t> mesh_xform;
    bool any_shading = false;
    for (size_t i = 0; i < refined.size(); ++i) {
        if (!type_is(typeid(*refined[i].ptr), "GeometricPrimitive")) die("a primitive that is not a GeometricPrimitive");
        const GeometricPrimitive *gp = static_cast<const GeometricPrimitive *>(refined[i].ptr);
        if (quadric_index.count(gp->shape.ptr)) continue;
        if (!type_is(typeid(*gp->shape.ptr), "Triangle")) die("a shape that is neither a triangle nor a quadric");
        const TriangleMesh *mesh = static_cast<const Triangle *>(gp->shape.ptr)->mesh.ptr;
        any_shading = any_shading || mesh->n || mesh->s || mesh->uvs;
    }
    for (size_t i = 0; i < refined.size(); ++i) {
        const GeometricPrimitive *gp = static_cast<const GeometricPrimitive *>(refined[i].ptr);
        if (quadric_index.count(gp->shape.ptr)) {
            // one primitive slot: the shape's world bound as a degenerate triangle {pMin, pMax, pMin} (pbrt_hip.h), flags bit 1
            const Shape *sh = gp->shape.ptr;
            const BBox wb = sh->WorldBound();
            const float v[9] = {wb.pMin.x, wb.pMin.y, wb.pMin.z, wb.pMax.x, wb.pMax.y, wb.pMax.z, wb.pMin.x, wb.pMin.y, wb.pMin.z};
            F.tri_verts.insert(F.tri_verts.end(), v, v + 9);
            const Material *m = gp->material.ptr;
            if (!mat_index.count(m)) { mat_index[m] = int(F.materials.size()); F.materials.push_back(flatten_material(m)); }
            F.tri_material.push_back(uint16_t(mat_index[m]));
            F.tri_light.push_back(gp->areaLight ? light_index[gp->areaLight] : -1);
            F.tri_flags.push_back(uint8_t(((sh->reverseOrientation ^ sh->transformSwapsHandedness) ? 1 : 0) | 2));
            if (any_shading) F.tri_shading.push_back(-1);
            continue;
        }
        const Triangle *t = static_cast<const Triangle *>(gp->shape.ptr);
        const TriangleMesh *mesh = t->mesh.ptr;
        for (int c = 0; c < 3; ++c) { const Point &p = mesh->p[t->v[c]]; F.tri_verts.push_back(p.x); F.tri_verts.push_back(p.y); F.tri_verts.push_back(p.z); }
        const Material *m = gp->material.ptr;
        if (!mat_index.count(m)) { mat_index[m] = int(F.materials.size()); F.materials.push_back(flatten_material(m)); }
        F.tri_material.push_back(uint16_t(mat_index[m]));
        F.tri_light.push_back(gp->areaLight ? light_index[gp->areaLight] : -1);
        F.tri_flags.push_back(uint8_t((t->reverseOrientation ^ t->transformSwapsHandedness) ? 1 : 0));
        if (any_shading) {
            if (!(mesh->n || mesh->s || mesh->uvs)) F.tri_shading.push_back(-1);
            else {
                if (!mesh_xform.count(mesh)) {
                    mesh_xform[mesh] = uint32_t(F.xforms.size() / 32);
                    float mm[32]; put_m(t->ObjectToWorld.m, mm); put_m(t->ObjectToWorld.mInv, mm + 16);
                    F.xforms.insert(F.xforms.end(), mm, mm + 32);
                }
                RtTriShading r; memset(&r, 0, sizeof r);
                r.flags = (mesh->uvs ? RT_SHADING_UV : 0) | (mesh->n ? RT_SHADING_N : 0) | (mesh->s ? RT_SHADING_S : 0);
                r.xform = mesh_xform[mesh];
                float uv[3][2]; t->GetUVs(uv);
                for (int c = 0; c < 3; ++c) {
                    r.uv[2 * c] = uv[c][0]; r.uv[2 * c + 1] = uv[c][1];
                    if (mesh->n) { const Normal &n = mesh->n[t->v[c]]; r.n[3 * c] = n.x; r.n[3 * c + 1] = n.y; r.n[3 * c + 2] = n.z; }
                    if (mesh->s) { const Vector &s = mesh->s[t->v[c]]; r.s[3 * c] = s.x; r.s[3 * c + 1] = s.y; r.s[3 * c + 2] = s.z; }
                }
                F.tri_shading.push_back(int32_t(F.shading.size())); F.shading.push_back(r);
            }
        }
    }
    // ---- camera (camera.cpp:50-70, perspective.cpp:37-50) and film (image.cpp:69-101)
    const Camera *anycam = scene->camera;
    if (!type_is(typeid(*anycam->film), "ImageFilm")) die(typeid(*anycam->film).name());
    const ImageFilm *film = static_cast<const ImageFilm *>(anycam->film);
    RtCamera &C = F.scene.camera;
    C.x_res = film->xResolution; C.y_res = film->yResolution;
    put_m(anycam->CameraToWorld.m, C.camera_to_world);
    C.hither = anycam->ClipHither; C.yon = anycam->ClipYon; C.shutter_open = anycam->ShutterOpen; C.shutter_close = anycam->ShutterClose;
    if (type_is(typeid(*anycam), "EnvironmentCamera")) C.type = RT_CAMERA_ENVIRONMENT;     // environment.cpp:37-61: no projection, no lens
    else {
        if (type_is(typeid(*anycam), "PerspectiveCamera")) C.type = RT_CAMERA_PERSPECTIVE;
        else if (type_is(typeid(*anycam), "OrthoCamera")) C.type = RT_CAMERA_ORTHOGRAPHIC;   // orthographic.cpp:40-47
        else die(typeid(*anycam).name());
        const ProjectiveCamera *cam = static_cast<const ProjectiveCamera *>(anycam);
        put_m(cam->RasterToCamera.m, C.raster_to_camera);
        C.lens_radius = cam->LensRadius; C.focal_distance = cam->FocalDistance;
    }
    RtRenderDesc &R = F.render;
    R.x_res = film->xResolution; R.y_res = film->yResolution;
    R.x_pixel_start = film->xPixelStart; R.y_pixel_start = film->yPixelStart; R.x_pixel_count = film->xPixelCount; R.y_pixel_count = film->yPixelCount;
    film->GetSampleExtent(&R.x_start, &R.x_end, &R.y_start, &R.y_end);
    R.filter_x_width = film->filter->xWidth; R.filter_y_width = film->filter->yWidth;
    memcpy(R.filter_table, film->filterTable, sizeof R.filter_table);         // FILTER_TABLE_SIZE^2 = 256 (image.cpp:53-64)
    // ---- sampler (stratified.cpp:51-86, lowdiscrepancy.cpp:57-75, random.cpp:45-75); the keyed wrapper exposes its inner sampler
    const Sampler *smp = scene->sampler;
    R.seed = 0; R.pixel_samples = 4; R.x_samples = R.y_samples = 2; R.jitter = 1;
    if (type_is(typeid(*smp), "KeyedSampler")) {                               // oracle/ref/keyed_sampler.cpp
        if (!g_keyed_inner_sampler) die("a keyed sampler that did not publish its inner sampler");
        smp = static_cast<const Sampler *>(g_keyed_inner_sampler); R.seed = g_keyed_seed;
    }
    if (type_is(typeid(*smp), "StratifiedSampler")) {
        const StratifiedSampler *s = static_cast<const StratifiedSampler *>(smp);
        R.sampler = RT_SAMPLER_STRATIFIED; R.x_samples = s->xPixelSamples; R.y_samples = s->yPixelSamples; R.jitter = s->jitterSamples ? 1 : 0;
    } else if (type_is(typeid(*smp), "LDSampler")) {
        // rounded up to a power of two by the constructor (lowdiscrepancy.cpp:62-66)
        R.sampler = RT_SAMPLER_LOWDISCREPANCY; R.pixel_samples = static_cast<const LDSampler *>(smp)->pixelSamples;
    } else if (type_is(typeid(*smp), "RandomSampler")) {
        const RandomSampler *s = static_cast<const RandomSampler *>(smp);
        R.sampler = RT_SAMPLER_RANDOM; R.x_samples = s->xPixelSamples; R.y_samples = s->yPixelSamples;
    } else die(typeid(*smp).name());
    // ---- integrators: this plugin's own parameters name the surface integrator it stands for
    const string inner = ips.FindOneString("inner", "path");
    R.integrator = inner == "whitted" ? RT_INTEGRATOR_WHITTED : inner == "directlighting" ? RT_INTEGRATOR_DIRECT : RT_INTEGRATOR_PATH;
    R.max_depth = ips.FindOneInt("maxdepth", 5);
    R.strategy = ips.FindOneString("strategy", "all") == "one" ? RT_STRATEGY_ONE : RT_STRATEGY_ALL;
    R.volume_integrator = RT_VOLUME_EMISSION; R.step_size = 1.f;
    if (type_is(typeid(*scene->volumeIntegrator), "EmissionIntegrator")) { R.volume_integrator = RT_VOLUME_EMISSION; R.step_size = static_cast<const EmissionIntegrator *>(scene->volumeIntegrator)->stepSize; }
    else if (type_is(typeid(*scene->volumeIntegrator), "SingleScattering")) { R.volume_integrator = RT_VOLUME_SINGLE; R.step_size = static_cast<const SingleScattering *>(scene->volumeIntegrator)->stepSize; }
    else die(typeid(*scene->volumeIntegrator).name());
    R.shard_index = 0; R.shard_count = 1; R.tile_pixels = 64;
    // ---- medium (homogeneous.cpp:27-42)
    if (scene->volumeRegion) {
        if (!type_is(typeid(*scene->volumeRegion), "HomogeneousVolume")) die(typeid(*scene->volumeRegion).name());
        const HomogeneousVolume *v = static_cast<const HomogeneousVolume *>(scene->volumeRegion);
        RtVolume &V = F.scene.volume; V.present = 1;
        put_m(v->WorldToVolume.m, V.world_to_volume);
        V.p0[0] = v->extent.pMin.x; V.p0[1] = v->extent.pMin.y; V.p0[2] = v->extent.pMin.z;
        V.p1[0] = v->extent.pMax.x; V.p1[1] = v->extent.pMax.y; V.p1[2] = v->extent.pMax.z;
        spec3(v->sig_a, V.sigma_a); spec3(v->sig_s, V.sigma_s); spec3(v->le, V.le); V.g = v->g;
    }
    F.scene.accel = g_accel;
    F.scene.n_tris = uint32_t(F.tri_material.size());
    F.scene.tri_verts = F.tri_verts.data(); F.scene.tri_material = F.tri_material.data(); F.scene.tri_light = F.tri_light.data(); F.scene.tri_flags = F.tri_flags.data();
    F.scene.n_materials = uint32_t(F.materials.size()); F.scene.materials = F.materials.data();
    F.scene.n_lights = uint32_t(F.lights.size()); F.scene.lights = F.lights.data();
    F.scene.n_light_tris = uint32_t(F.light_tris.size() / 9); F.scene.light_tris = F.light_tris.data();
    F.scene.tri_shading = F.tri_shading.empty() ? NULL : F.tri_shading.data();
    F.scene.n_shading = uint32_t(F.shading.size()); F.scene.shading = F.shading.empty() ? NULL : F.shading.data();
    F.scene.n_xforms = uint32_t(F.xforms.size() / 32); F.scene.xforms = F.xforms.empty() ? NULL : F.xforms.data();
    F.scene.n_quadrics = uint32_t(F.quadrics.size()); F.scene.quadrics = F.quadrics.empty() ? NULL : F.quadrics.data();
}

// ---- the C ABI of libpbrt_hip.so, bound at run time (include/pbrt_hip.h)
struct HipLib {
    void *h;
    int (*scene_create)(const RtSceneDesc *, int, RtScene **); int (*scene_destroy)(RtScene *);
    int (*film_bind)(RtScene *, void *, int32_t, int32_t); int (*render)(RtScene *, const RtRenderDesc *); int (*sync)(RtScene *);
    int (*samples_read)(RtScene *, uint64_t, uint64_t, float *); const char *(*last_error)(void);
    bool open(const char *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "hip_adapter: %s\n", dlerror()); return false; }
#define BIND(f, name) *(void **)(&f) = dlsym(h, name); if (!f) { fprintf(stderr, "hip_adapter: %s lacks %s\n", path, name); return false; }
        BIND(scene_create, "rt_scene_create") BIND(scene_destroy, "rt_scene_destroy") BIND(film_bind, "rt_film_bind") BIND(render, "rt_render")
        BIND(sync, "rt_sync") BIND(samples_read, "rt_samples_read") BIND(last_error, "rt_last_error")
#undef BIND
        return true;
    }
};

class HipIntegrator : public SurfaceIntegrator {
public:
    HipIntegrator(SurfaceIntegrator *in, const ParamSet &ps) : inner(in), params(ps), next(0) {}
    ~HipIntegrator() { delete inner; }
    void RequestSamples(Sample *sample, const Scene *scene) { inner->RequestSamples(sample, scene); }
    void Preprocess(const Scene *scene) {
        inner->Preprocess(scene);
        Flat F;
        flatten(scene, params, F);
        if (const char *dump = getenv("PBRT_HIP_DESC_DUMP")) {
            std::vector<unsigned char> buf(rt_desc_serialize(&F.scene, &F.render, NULL));
            rt_desc_serialize(&F.scene, &F.render, buf.data());
            FILE *f = fopen(dump, "wb");
            if (!f || fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "hip_adapter: cannot write %s\n", dump); abort(); }
            fclose(f);
        }
        if (const char *lib = getenv("PBRT_HIP_LIB")) {
            HipLib L;
            if (!L.open(lib)) abort();
            RtScene *s = NULL;
            if (L.scene_create(&F.scene, -1, &s) || L.film_bind(s, NULL, F.render.x_pixel_count, F.render.y_pixel_count) || L.render(s, &F.render) || L.sync(s)) {
                fprintf(stderr, "hip_adapter: %s\n", L.last_error()); abort();
            }
            const uint64_t n = uint64_t(F.render.x_end - F.render.x_start) * uint64_t(F.render.y_end - F.render.y_start) * uint64_t(scene->sampler->samplesPerPixel);
            radiance.resize(size_t(n) * 8);
            if (L.samples_read(s, 0, n, radiance.data())) { fprintf(stderr, "hip_adapter: %s\n", L.last_error()); abort(); }
            L.scene_destroy(s);
        }
    }
    // SurfaceIntegrator::Li (transport.h:35-50).  With device results: sample k of the frame, in the sampler's own order.
    Spectrum Li(const Scene *scene, const RayDifferential &ray, const Sample *sample, float *alpha) const {
        if (radiance.empty()) return inner->Li(scene, ray, sample, alpha);
        const float *r = &radiance[8 * next++];
        if (r[4] != sample->imageX || r[5] != sample->imageY) { fprintf(stderr, "hip_adapter: sample %zu is at (%a, %a) here and (%a, %a) on the device (use the keyed sampler)\n", next - 1, sample->imageX, sample->imageY, r[4], r[5]); abort(); }
        if (alpha) *alpha = r[3];
        float c[3] = {r[0], r[1], r[2]};
        return Spectrum(c);
    }
private:
    SurfaceIntegrator *inner; ParamSet params;
    std::vector<float> radiance; mutable size_t next;
};

}  // namespace

extern "C" DLLEXPORT Primitive *CreateAccelerator(const vector<Reference<Primitive> > &prims, const ParamSet &ps) {
    g_prims = prims;
    memset(&g_accel, 0, sizeof g_accel);
    const string inner = ps.FindOneString("inner", "kdtree");
    g_accel.kind = inner == "grid" ? RT_ACCEL_GRID : RT_ACCEL_KDTREE;
    g_accel.isect_cost = ps.FindOneInt("intersectcost", 80); g_accel.trav_cost = ps.FindOneInt("traversalcost", 1);     // kdtree.cpp:489-498
    g_accel.empty_bonus = ps.FindOneFloat("emptybonus", 0.5f); g_accel.max_prims = ps.FindOneInt("maxprims", 1); g_accel.max_depth = ps.FindOneInt("maxdepth", -1);
    return MakeAccelerator(inner, prims, ps);
}

extern "C" DLLEXPORT SurfaceIntegrator *CreateSurfaceIntegrator(const ParamSet &ps) {
    SurfaceIntegrator *inner = MakeSurfaceIntegrator(ps.FindOneString("inner", "path"), ps);
    if (!inner) return NULL;
    return new HipIntegrator(inner, ps);
}
