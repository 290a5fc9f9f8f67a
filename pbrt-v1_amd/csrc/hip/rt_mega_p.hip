// rt_mega_p.hip -- rt::render_kernel for integrator 2 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 2
#define RT_TU_TABLE g_render_kernels_path
#include "rt_mega_tu.inc"
