// rt_pipe_v.hip -- rt::pipe_vertex_kernel (rt_pipe_vertex.h): f = 0: timed, 1: counting twin (carries the glossy / quadric code), 2: timed with that code (EXT)
#include "rt_pipe_vertex.h"
namespace rt {
extern const PipeShadeFn g_pipe_vertex[3];
const PipeShadeFn g_pipe_vertex[3] = {pipe_vertex_kernel<false, false>, pipe_vertex_kernel<true, true>, pipe_vertex_kernel<false, true>};
}  // namespace rt
