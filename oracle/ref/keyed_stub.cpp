// oracle/ref/keyed_stub.cpp -- TEST INFRASTRUCTURE.  In the plain (un-keyed)
// reference executable the helper plugins still resolve these two symbols;
// they do nothing, so the reference's own MT19937 stream runs untouched.
#include <stdint.h>
extern "C" void keyed_rng_set_key(uint32_t) {}
extern "C" void keyed_rng_set_seed(uint32_t) {}
