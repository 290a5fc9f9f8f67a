// rt_mega_d.hip -- rt::render_kernel for integrator 1 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 1
#define RT_TU_NAT_WAVES 3
#define RT_TU_TABLE g_render_kernels_direct
#include "rt_mega_tu.inc"
