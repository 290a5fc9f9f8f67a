// oracle/ref/probe_integrator.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A SurfaceIntegrator plugin (interface core/transport.h:35-50; factory symbol
// CreateSurfaceIntegrator as in integrators/whitted.cpp:141) that records, for
// every camera sample, the generated camera ray and what Scene::Intersect /
// Scene::IntersectP return for it -- golden vectors for the camera
// (cameras/perspective.cpp:51-82), kd-tree/grid traversal (kdtree.cpp:313-488)
// and Triangle::Intersect (trianglemesh.cpp:213-314) in isolation.
//   SurfaceIntegrator "probe" "string dump" ["rays.bin"] "point target" [x y z]
// record = 20 floats: o[3] d[3] mint maxt | hit t p[3] nn[3] u v | occluded pad
// `occluded` = IntersectP of the shadow segment (light.h:78-83) from the hit
// point to `target`.  Li returns (t, u, v) so the film is a depth/uv image.
#include "pbrt.h"
#include "transport.h"
#include "scene.h"
#include "light.h"
#include "paramset.h"
class ProbeIntegrator : public SurfaceIntegrator {
public:
    ProbeIntegrator(const string &fn, const Point &t) : target(t) { f = fopen(fn.c_str(), "wb"); }
    ~ProbeIntegrator() { if (f) fclose(f); }
    Spectrum Li(const Scene *scene, const RayDifferential &ray, const Sample *, float *alpha) const {
        float rec[20]; memset(rec, 0, sizeof rec);
        rec[0] = ray.o.x; rec[1] = ray.o.y; rec[2] = ray.o.z;
        rec[3] = ray.d.x; rec[4] = ray.d.y; rec[5] = ray.d.z;
        rec[6] = ray.mint; rec[7] = ray.maxt;
        Intersection isect;
        float c[3] = {0, 0, 0};
        if (alpha) *alpha = 0.f;
        if (scene->Intersect(ray, &isect)) {
            if (alpha) *alpha = 1.f;
            rec[8] = 1.f; rec[9] = ray.maxt;
            rec[10] = isect.dg.p.x; rec[11] = isect.dg.p.y; rec[12] = isect.dg.p.z;
            rec[13] = isect.dg.nn.x; rec[14] = isect.dg.nn.y; rec[15] = isect.dg.nn.z;
            rec[16] = isect.dg.u; rec[17] = isect.dg.v;
            VisibilityTester vis; vis.SetSegment(isect.dg.p, target);
            rec[18] = vis.Unoccluded(scene) ? 0.f : 1.f;
            c[0] = ray.maxt; c[1] = isect.dg.u; c[2] = isect.dg.v;
        }
        if (f) fwrite(rec, sizeof(float), 20, f);
        return Spectrum(c);
    }
private:
    FILE *f;
    Point target;
};
extern "C" DLLEXPORT SurfaceIntegrator *CreateSurfaceIntegrator(const ParamSet &params) {
    return new ProbeIntegrator(params.FindOneString("dump", "probe_rays.bin"),
                               params.FindOnePoint("target", Point(278, 540, 280)));
}
