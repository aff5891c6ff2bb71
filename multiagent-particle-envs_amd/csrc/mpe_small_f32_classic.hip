// float32 instantiations of the small-scenario kernels for the six other classic scenarios (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f32_classic(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_group<float, GROUP_CLASSIC>(s, B, mode, a, st);
}
}  // namespace mpe
