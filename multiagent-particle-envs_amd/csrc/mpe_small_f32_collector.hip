// float32 instantiations of the small-scenario kernels for collector (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f32_collector(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_group<float, GROUP_COLLECTOR>(s, B, mode, a, st);
}
}  // namespace mpe
