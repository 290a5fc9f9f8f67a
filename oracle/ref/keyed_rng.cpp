// oracle/ref/keyed_rng.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Link-time replacement for the reference's global MT19937 entry points
// genrand_int32 / genrand_real2 (core/util.cpp:340-380; declared pbrt.h:228-230).
// Linked BEFORE util.o with -Wl,--allow-multiple-definition, so every
// RandomFloat()/RandomUInt() (pbrt.h:636-644) in the core and in every plugin
// resolves here.  No reference source is modified.
//
// Why: one global sequential stream consumed in scanline order cannot be
// replayed by a parallel renderer (SURVEY.md section 7 "RNG semantics"); and a single
// last-bit difference de-synchronises every later sample.  Here
// the stream is counter based:  draw #c of camera sample #n is
//        u32 = pcg(c + pcg(n + seed*0x9E3779B9))
// with pcg() the PCG-RXS-M-XS 32-bit output hash.  The key n is set by the
// wrapper Sampler plugin (keyed_sampler.cpp) on every GetNextSample().  Draws
// made before the first sample (StratifiedSampler's constructor,
// samplers/stratified.cpp:66-84) use key 0xFFFFFFFF.
// The same definition is implemented by oracle/pbrt_oracle.cpp and by the HIP
// kernels (pbrt-v1_amd/csrc/hip/rt_rng.h) -- that is the "fixed sample seed".
#include <stdint.h>
extern "C" {
static inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
static uint32_t g_base = 0, g_ctr = 0, g_seed = 0;
unsigned long long g_keyed_draws = 0;
void keyed_rng_set_seed(uint32_t seed) { g_seed = seed; }
void keyed_rng_set_key(uint32_t key) { g_base = pcg(key + g_seed * 0x9E3779B9u); g_ctr = 0; }
uint32_t keyed_rng_counter() { return g_ctr; }
}
static struct KeyedInit { KeyedInit() { keyed_rng_set_key(0xFFFFFFFFu); } } g_keyedInit;

unsigned long genrand_int32(void) { ++g_keyed_draws; return pcg(g_ctr++ + g_base); }
// util.cpp:377-380: (RandomUInt() & 0xffffff) / float(1 << 24)
float genrand_real2(void) { return (genrand_int32() & 0xffffff) / float(1 << 24); }
