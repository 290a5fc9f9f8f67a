// rt_trace.hip -- rt::pipe_trace_kernel (the dominant kernel of the queue pipeline): k = ACCEL*4 + f, f = 0: timed, 1: counting twin with the
// glossy / quadric code, 2: timed with that code (EXT), 3: counting twin without it (per-primitive records + leaf entries, like 0)
#include "rt_pipeline.h"
namespace rt {
#define RT_K(C, A, G) pipe_trace_kernel<C, A, G>
extern const PipeTraceFn g_pipe_trace[8];
const PipeTraceFn g_pipe_trace[8] = {RT_K(false, 0, false), RT_K(true, 0, true), RT_K(false, 0, true), RT_K(true, 0, false),
                                     RT_K(false, 1, false), RT_K(true, 1, true), RT_K(false, 1, true), RT_K(true, 1, false)};
#undef RT_K
}  // namespace rt
