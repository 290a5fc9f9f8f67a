// oracle/ref/count_accel.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// An Aggregate plugin (interface core/primitive.h:31-47,102-108; factory symbol
// CreateAccelerator as in accelerators/kdtree.cpp:489) that wraps one of the
// reference's accelerators, counts every Scene::Intersect / Scene::IntersectP
// (core/scene.h:40-45) -- the metric's definition of a "ray" -- and times the
// inner accelerator's construction.
//   Accelerator "countaccel" "string inner" ["kdtree"] <inner params...>
#include "primitive.h"
#include "paramset.h"
#include "dynload.h"
extern "C" unsigned long long g_ref_counters[8];
extern "C" double g_ref_times[4];
extern "C" double ref_now();

class CountAccel : public Aggregate {
public:
    CountAccel(Primitive *in) : inner(in) {}
    BBox WorldBound() const { return inner->WorldBound(); }
    bool CanIntersect() const { return true; }
    bool Intersect(const Ray &r, Intersection *in) const {
        ++g_ref_counters[0];
        bool h = inner->Intersect(r, in);
        if (h) ++g_ref_counters[2];
        return h;
    }
    bool IntersectP(const Ray &r) const {
        ++g_ref_counters[1];
        bool h = inner->IntersectP(r);
        if (h) ++g_ref_counters[3];
        return h;
    }
private:
    Reference<Primitive> inner;
};

extern "C" DLLEXPORT Primitive *CreateAccelerator(const vector<Reference<Primitive> > &prims,
                                                  const ParamSet &ps) {
    string innerName = ps.FindOneString("inner", "kdtree");
    double t0 = ref_now();
    Primitive *in = MakeAccelerator(innerName, prims, ps);
    double t1 = ref_now();
    g_ref_times[0] = t1 - t0;
    g_ref_times[1] = t1;
    if (!in) return NULL;
    return new CountAccel(in);
}
