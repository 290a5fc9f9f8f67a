// rt_mega_w.hip -- rt::render_kernel for integrator 0 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 0
#define RT_TU_TABLE g_render_kernels_whitted
#include "rt_mega_tu.inc"
