// rt_pipe_p.hip -- rt::pipe_shade_kernel for integrator 2 (0 whitted, 1 directlighting, 2 path)
#define RT_TU_INTEG 2
#define RT_TU_TABLE g_pipe_shade_path
#include "rt_pipe_tu.inc"
