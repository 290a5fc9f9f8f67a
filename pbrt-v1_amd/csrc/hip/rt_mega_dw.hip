// rt_mega_dw.hip -- rt::render_kernel for DirectLighting "weighted" (RT_INTEG_DIRECT_WEIGHTED; three passes per frame, rt_weighted.h): 8 instantiations,
// k = (VOL*2 + ACCEL)*2 + COUNT, all with the glossy / quadric code (EXT) at natural register allocation -- this strategy's frame time is set by its
// sequential recurrence, not by these kernels
#include "rt_render_kernel.h"
namespace rt {
#define RT_K(C, A, V) render_kernel<C, RT_INTEG_DIRECT_WEIGHTED, A, V, RT_MIN_WAVES, true>
extern const RenderKernelFn g_render_kernels_weighted[8];
const RenderKernelFn g_render_kernels_weighted[8] = {RT_K(false, 0, false), RT_K(true, 0, false), RT_K(false, 1, false), RT_K(true, 1, false),
                                                     RT_K(false, 0, true),  RT_K(true, 0, true),  RT_K(false, 1, true),  RT_K(true, 1, true)};
#undef RT_K
}  // namespace rt
