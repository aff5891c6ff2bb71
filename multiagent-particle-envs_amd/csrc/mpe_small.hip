// Dispatch of the small-scenario kernels by dtype; the kernels are compiled per dtype in
// mpe_small_f32.hip / mpe_small_f64.hip from the templates in mpe_small_impl.hpp.
#include "mpe_launch.hpp"

namespace mpe {

hipError_t launch_small_f32(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st);
hipError_t launch_small_f64(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st);

bool small_supported(const HostSpec& s) {
    return (s.scenario == MPE_SCN_SPREAD && s.A == 3 && s.L == 3) || s.scenario == MPE_SCN_TAG || s.scenario == MPE_SCN_WORLD_COMM ||
           s.scenario == MPE_SCN_SIMPLE || s.scenario == MPE_SCN_ADVERSARY || s.scenario == MPE_SCN_PUSH ||
           s.scenario == MPE_SCN_SPEAKER_LISTENER || s.scenario == MPE_SCN_REFERENCE || s.scenario == MPE_SCN_CRYPTO ||
           s.scenario == MPE_SCN_PREDATOR_PREY || s.scenario == MPE_SCN_COLLECTOR;
}

hipError_t launch_small(const HostSpec& s, long long B, int dtype, int mode, const StepArgs& a, hipStream_t st) {
    return dtype == MPE_F64 ? launch_small_f64(s, B, mode, a, st) : launch_small_f32(s, B, mode, a, st);
}

}  // namespace mpe
