// rt_trace.hip -- rt::pipe_trace_kernel (the dominant kernel of the queue pipeline): k = ACCEL*3 + f, f as in rt_pipe_tu.inc
#include "rt_pipeline.h"
namespace rt {
#define RT_K(C, A, G) pipe_trace_kernel<C, A, G>
extern const PipeTraceFn g_pipe_trace[6];
const PipeTraceFn g_pipe_trace[6] = {RT_K(false, 0, false), RT_K(true, 0, true), RT_K(false, 0, true),
                                     RT_K(false, 1, false), RT_K(true, 1, true), RT_K(false, 1, true)};
#undef RT_K
}  // namespace rt
