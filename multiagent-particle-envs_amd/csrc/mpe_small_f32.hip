// float32 instantiations of the small-scenario kernels (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f32(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_t<float>(s, B, mode, a, st);
}
}  // namespace mpe


#ifdef MPE_STAMPS
// Diagnostic build only: copy the per-workgroup timestamps to host memory.
extern "C" int mpe_debug_stamps(long long* host, long long count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mpe::g_stamps), (size_t)count * sizeof(long long));
}
#endif
