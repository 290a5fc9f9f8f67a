/* pbrt_hip_desc.h -- canonical byte image of an (RtSceneDesc, RtRenderDesc) pair.
 *
 * The boundary of this library is "flat descriptors in, film out" (pbrt_hip.h).  Two producers of descriptors exist:
 *   - the product's own host front end (pbrt-v1_amd/csrc/host: scene parser + pbrt* API state machine + plugin factories), and
 *   - the reference-side adapter oracle/ref/hip_adapter.cpp, a pbrt-v1 plugin (extern "C" CreateSurfaceIntegrator /
 *     CreateAccelerator, core/dynload.cpp:185-205) that builds the descriptors from the reference's OWN objects
 *     (Scene, Camera, Light, GeometricPrimitive, TriangleMesh ...) inside the unmodified reference binary.
 * Serialising both with this one function makes "the same description" a byte comparison (tests/test_boundary.py).
 * Pointers are replaced by the arrays they point to, each behind an 8-byte tag and a 64-bit element count.
 * Fields that do not describe the scene (accel.build_threads, the shard_* work partition) are written as zero. */
#ifndef PBRT_HIP_DESC_H
#define PBRT_HIP_DESC_H
#include "pbrt_hip.h"
#include <string.h>

static inline size_t rt_desc_put(unsigned char *out, size_t at, const char *tag, const void *data, size_t elem, uint64_t count) {
    char t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t bytes = elem * (size_t)count;
    strncpy(t, tag, 8);
    if (out) {
        memcpy(out + at, t, 8);
        memcpy(out + at + 8, &count, 8);
        if (bytes) memcpy(out + at + 16, data, bytes);
    }
    return at + 16 + ((bytes + 7) & ~(size_t)7);
}

/* returns the number of bytes; call with out == NULL to size the buffer */
static inline size_t rt_desc_serialize(const RtSceneDesc *s, const RtRenderDesc *r, unsigned char *out) {
    size_t at = 0;
    RtAccelParams acc = s->accel;
    RtRenderDesc rd = *r;
    acc.build_threads = 0;
    rd.shard_index = 0; rd.shard_count = 0; rd.tile_pixels = 0;
    at = rt_desc_put(out, at, "TRIVERT", s->tri_verts, 9 * sizeof(float), s->n_tris);
    at = rt_desc_put(out, at, "TRIMAT", s->tri_material, sizeof(uint16_t), s->n_tris);
    at = rt_desc_put(out, at, "TRILIGHT", s->tri_light, sizeof(int32_t), s->n_tris);
    at = rt_desc_put(out, at, "TRIFLAGS", s->tri_flags, sizeof(uint8_t), s->n_tris);
    at = rt_desc_put(out, at, "MATERIAL", s->materials, sizeof(RtMaterial), s->n_materials);
    at = rt_desc_put(out, at, "LIGHTS", s->lights, sizeof(RtLight), s->n_lights);
    at = rt_desc_put(out, at, "LIGHTTRI", s->light_tris, 9 * sizeof(float), s->n_light_tris);
    at = rt_desc_put(out, at, "CAMERA", &s->camera, sizeof(RtCamera), 1);
    at = rt_desc_put(out, at, "VOLUME", &s->volume, sizeof(RtVolume), 1);
    at = rt_desc_put(out, at, "ACCEL", &acc, sizeof(RtAccelParams), 1);
    at = rt_desc_put(out, at, "QUADRICS", s->quadrics, sizeof(RtQuadric), s->n_quadrics);
    at = rt_desc_put(out, at, "TRISHIDX", s->tri_shading, sizeof(int32_t), s->tri_shading ? s->n_tris : 0);
    at = rt_desc_put(out, at, "SHADING", s->shading, sizeof(RtTriShading), s->n_shading);
    at = rt_desc_put(out, at, "XFORMS", s->xforms, 32 * sizeof(float), s->n_xforms);
    at = rt_desc_put(out, at, "RENDER", &rd, sizeof(RtRenderDesc), 1);
    return at;
}
#endif /* PBRT_HIP_DESC_H */
