// rt_pipe_d.hip -- rt::pipe_shade_kernel for integrator 1 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 1
#define RT_TU_TABLE g_pipe_shade_direct
#include "rt_pipe_tu.inc"
