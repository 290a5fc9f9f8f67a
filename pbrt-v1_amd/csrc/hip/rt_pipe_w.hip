// rt_pipe_w.hip -- rt::pipe_shade_kernel for integrator 0 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 0
#define RT_TU_TABLE g_pipe_shade_whitted
#include "rt_pipe_tu.inc"
