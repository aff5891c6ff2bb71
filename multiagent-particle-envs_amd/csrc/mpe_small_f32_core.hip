// float32 instantiations of the small-scenario kernels for simple_spread 3/3, simple_tag, simple_world_comm (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f32_core(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_group<float, GROUP_CORE>(s, B, mode, a, st);
}
}  // namespace mpe
