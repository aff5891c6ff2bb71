// float32 instantiations of the small-scenario kernels for simple_spread 3/3, simple_tag, simple_world_comm (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f32_core(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_group<float, GROUP_CORE>(s, B, mode, a, st);
}
}  // namespace mpe

#ifdef MPE_STAMPS
// Diagnostic build only: copy the per-workgroup timestamps to host memory.
extern "C" int mpe_debug_stamps(long long* host, long long count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mpe::g_stamps), (size_t)count * sizeof(long long));
}
#endif
