// On-device Scenario.reset_world: fresh uniform positions, zero velocity and comm state, goal landmark.
//
// One lane per world.  Entity e of world b in episode n takes its two coordinates from the
// Philox-4x32-10 block with key = seed and counter = (env_base + b, n * 256 + e), so a reset is reproducible
// and independent of batch size, launch geometry, of which other worlds are reset with it and -- with env_base =
// the global index of the batch's first env -- of how a job's worlds are sharded over handles and GPUs.
#include "mpe_launch.hpp"
#include "mpe_scenarios.hpp"

namespace mpe {

struct ResetSizes { double size_d[16]; };      // float64 Entity.size of scenarios whose reset tests placements

template <typename T>
__global__ __launch_bounds__(256) void reset_kernel(long long B, int E, int A, int comm_rows, int n_aux, bool env_major, int scenario,
                                                    const ResetSizes sizes, const ResetArgs rs,
                                                    const uint8_t* __restrict__ mask, T* __restrict__ pos,
                                                    T* __restrict__ vel, T* __restrict__ comm, int32_t* __restrict__ aux) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mask != nullptr && mask[b] == 0) return;
    // element (row r of n_rows, env b): env-minor r*B + b, env-major b*n_rows + r
    auto at = [=](int r, int n_rows) -> long long { return env_major ? b * n_rows + r : (long long)r * B + b; };
    if (scenario == MPE_SCN_PREDATOR_PREY) {
        // foods are redrawn until they are clear of the barrier and the forests; flags and respawn stream start fresh
        typedef PredatorPreyCfg C;
        V2<double> d[C::E];
        for (int e = 0; e < C::E; ++e) d[e] = reset_draw(rs, (unsigned long long)b, e, e < A);
        C::fix_reset(rs, (unsigned long long)b, d, sizes);
        for (int e = 0; e < C::E; ++e) {
            pos[at(2 * e, 2 * E)] = (T)d[e].x;
            pos[at(2 * e + 1, 2 * E)] = (T)d[e].y;
        }
        for (int k = 0; k < n_aux; ++k) aux[at(k, n_aux)] = C::initial_aux(rs, (unsigned long long)b, k);
    } else {
        for (int e = 0; e < E; ++e) {
            const V2<double> d = reset_draw(rs, (unsigned long long)b, e, e < A);
            pos[at(2 * e, 2 * E)] = (T)d.x;
            pos[at(2 * e + 1, 2 * E)] = (T)d.y;
        }
        for (int k = 0; k < n_aux; ++k)
            aux[at(k, n_aux)] = scenario == MPE_SCN_COLLECTOR ? CollectorCfg::initial_aux(rs, (unsigned long long)b, k)
                                                              : reset_aux(rs, (unsigned long long)b, k, E - A);
    }
    for (int r = 0; r < 2 * A; ++r) vel[at(r, 2 * A)] = T(0);
    for (int r = 0; r < comm_rows; ++r) comm[at(r, comm_rows)] = T(0);
}

hipError_t launch_reset(const HostSpec& s, long long B, int dtype, int comm_rows, bool env_major, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_base, const uint8_t* mask, void* pos, void* vel, void* comm, int32_t* aux,
                        hipStream_t st) {
    const ResetArgs rs{1, seed, episode, landmark_limit(s), 0, env_base};
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    ResetSizes sizes = {};
    for (int e = 0; e < s.E() && e < 16; ++e) sizes.size_d[e] = s.size[e];
    if (dtype == MPE_F64)
        hipLaunchKernelGGL(reset_kernel<double>, grid, block, 0, st, B, s.E(), s.A, comm_rows, s.n_aux(), env_major, s.scenario, sizes, rs, mask, (double*)pos, (double*)vel, (double*)comm, aux);
    else
        hipLaunchKernelGGL(reset_kernel<float>, grid, block, 0, st, B, s.E(), s.A, comm_rows, s.n_aux(), env_major, s.scenario, sizes, rs, mask, (float*)pos, (float*)vel, (float*)comm, aux);
    return hipGetLastError();
}

// The phase of an uneven deal over the XCDs (XcdDeal, mpe_launch.hpp): one workgroup stores the XCD it runs on, i.e. the
// XCD that workgroup 0 of the next launch on this stream will be dealt to.
__global__ __launch_bounds__(64) void xcd_phase_kernel(int* __restrict__ phase) {
    const unsigned id = hw_xcc_id();
    if (threadIdx.x == 0) *phase = (int)id;
}

hipError_t launch_xcd_phase(int* phase, hipStream_t st) {
    hipLaunchKernelGGL(xcd_phase_kernel, dim3(1), dim3(64), 0, st, phase);
    return hipGetLastError();
}

}  // namespace mpe
