// oracle/ref/keyed_sampler.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A Sampler plugin (interface core/sampling.h:30-47, factory symbol
// CreateSampler as in samplers/stratified.cpp:132) that wraps one of the
// reference's own samplers and re-keys the counter RNG (keyed_rng.cpp) with the
// running camera-sample index before forwarding every GetNextSample().
//   Sampler "keyed" "string inner" ["stratified"] <inner sampler params...>
#include "sampling.h"
#include "paramset.h"
#include "film.h"
#include "dynload.h"
#include <stdint.h>
extern "C" void keyed_rng_set_key(uint32_t key);
extern "C" void keyed_rng_set_seed(uint32_t seed);
extern "C" const void *g_keyed_inner_sampler;     // ref_driver.cpp: read by hip_adapter.cpp (the description names the inner sampler)
extern "C" unsigned g_keyed_seed;

class KeyedSampler : public Sampler {
public:
    KeyedSampler(Sampler *in)
        : Sampler(in->xPixelStart, in->xPixelEnd, in->yPixelStart, in->yPixelEnd, in->samplesPerPixel),
          inner(in), index(0) {}
    ~KeyedSampler() { delete inner; }
    int RoundSize(int size) const { return inner->RoundSize(size); }
    bool GetNextSample(Sample *sample) {
        keyed_rng_set_key(index++);
        return inner->GetNextSample(sample);
    }
private:
    Sampler *inner;
    uint32_t index;
};

extern "C" DLLEXPORT Sampler *CreateSampler(const ParamSet &params, const Film *film) {
    string innerName = params.FindOneString("inner", "stratified");
    keyed_rng_set_seed((uint32_t)params.FindOneInt("seed", 0));
    keyed_rng_set_key(0xFFFFFFFFu);           // constructor draws of the inner sampler
    Sampler *in = MakeSampler(innerName, params, film);
    if (!in) return NULL;
    g_keyed_inner_sampler = in; g_keyed_seed = (unsigned)params.FindOneInt("seed", 0);
    return new KeyedSampler(in);
}
