// Dispatch of the small-scenario kernels by scenario group and float type; the kernels are compiled per
// (group, type) in mpe_small_<type>_<group>.hip from the templates in mpe_small_impl.hpp.
#include "mpe_launch.hpp"

namespace mpe {

#define MPE_DECLARE_GROUP(name)                                                                                              \
    hipError_t launch_small_f32_##name(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st);       \
    hipError_t launch_small_f64_##name(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st);
MPE_DECLARE_GROUP(core)
MPE_DECLARE_GROUP(classic)
MPE_DECLARE_GROUP(pp)
MPE_DECLARE_GROUP(collector)
#undef MPE_DECLARE_GROUP

bool small_supported(const HostSpec& s) {
    return (s.scenario == MPE_SCN_SPREAD && s.A == 3 && s.L == 3) || s.scenario == MPE_SCN_TAG || s.scenario == MPE_SCN_WORLD_COMM ||
           s.scenario == MPE_SCN_SIMPLE || s.scenario == MPE_SCN_ADVERSARY || s.scenario == MPE_SCN_PUSH ||
           s.scenario == MPE_SCN_SPEAKER_LISTENER || s.scenario == MPE_SCN_REFERENCE || s.scenario == MPE_SCN_CRYPTO ||
           s.scenario == MPE_SCN_PREDATOR_PREY || s.scenario == MPE_SCN_COLLECTOR;
}

hipError_t launch_small(const HostSpec& s, long long B, int dtype, int mode, const StepArgs& a, hipStream_t st) {
    const bool f64 = dtype == MPE_F64;
    switch (s.scenario) {
        case MPE_SCN_SPREAD: case MPE_SCN_TAG: case MPE_SCN_WORLD_COMM:
            return f64 ? launch_small_f64_core(s, B, mode, a, st) : launch_small_f32_core(s, B, mode, a, st);
        case MPE_SCN_PREDATOR_PREY:
            return f64 ? launch_small_f64_pp(s, B, mode, a, st) : launch_small_f32_pp(s, B, mode, a, st);
        case MPE_SCN_COLLECTOR:
            return f64 ? launch_small_f64_collector(s, B, mode, a, st) : launch_small_f32_collector(s, B, mode, a, st);
        default:
            return f64 ? launch_small_f64_classic(s, B, mode, a, st) : launch_small_f32_classic(s, B, mode, a, st);
    }
}

}  // namespace mpe
