// rt_march.hip -- rt::pipe_march_kernel (rt_pipe_march.h): k = ACCEL*3 + f, f = 0: timed, 1: counting twin (carries the glossy / quadric code), 2: timed with that code (EXT)
#include "rt_pipe_march.h"
namespace rt {
#define RT_K(C, A, G) pipe_march_kernel<C, A, G>
extern const PipeMarchFn g_pipe_march[6];
const PipeMarchFn g_pipe_march[6] = {RT_K(false, 0, false), RT_K(true, 0, true), RT_K(false, 0, true),
                                     RT_K(false, 1, false), RT_K(true, 1, true), RT_K(false, 1, true)};
#undef RT_K
}  // namespace rt
