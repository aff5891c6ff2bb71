// float64 instantiations of the small-scenario kernels for simple_predator_prey (see mpe_small_impl.hpp).
#include "mpe_small_impl.hpp"

namespace mpe {
hipError_t launch_small_f64_pp(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    return launch_small_group<double, GROUP_PREDATOR_PREY>(s, B, mode, a, st);
}
}  // namespace mpe
