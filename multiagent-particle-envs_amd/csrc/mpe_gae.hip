// Generalised advantage estimation over a batch of rollouts, the step that follows the rollout in the
// reference's training loop (multiagent/backup/runner_old.py:58-80): a backward scan over the steps of
// every env.  A lane walks t = T-1 .. 0 for one env (the reference's scan, whose accumulator runs through the
// env's agents) or for one (env, agent) element (independent scans: a wave's accesses of one step are then 64
// consecutive values).  Inputs and outputs are float32 (the runner casts rewards and values to
// float32, :61-62), the scan itself runs in float64 like the runner's (its `lastgaelam` stays a float64
// array, the advantages are rounded to float32 when stored).
#include "mpe_launch.hpp"

namespace mpe {

namespace {

// MODE_REFERENCE: runner_old.py:69-80 as written -- one `lastgaelam` carried through the agent loop
//   (agent i at step t continues from agent i-1 at step t, agent 0 from the last agent at step t+1) and
//   the non-terminal mask of step t taken from dones[t+1] (dones[T-1] for the last step, :72-76).
// MODE_STANDARD: an independent scan per agent with the mask of step t taken from dones[t], the flag the
//   env returned with that step's reward (Schulman et al. 2016, eq. 16 with episode boundaries).
template <int MODE>
__global__ __launch_bounds__(256) void gae_kernel(long long B, int A, int T, const float* __restrict__ rew, const float* __restrict__ val,
                                                  const uint8_t* __restrict__ dones, const float* __restrict__ last_val,
                                                  const uint8_t* __restrict__ last_dones, double gamma, double lam,
                                                  float* __restrict__ adv, float* __restrict__ ret) {
    // MODE_REFERENCE: one lane per env (the chain runs through the env's agents).  MODE_STANDARD: one lane per
    // (env, agent) element, so that a wave's accesses of one step are 64 consecutive values whatever A is.
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = (MODE == MPE_GAE_REFERENCE) ? tid : tid / A;
    if (b >= B) return;
    const float gamma_f = (float)gamma;         // gamma * float32 array is a float32 product in numpy
    const double gl = gamma * lam;
    // one (t, i) term of runner_old.py:77-79 given the running lastgaelam
    auto term = [&](int t, int i, double nnt, double prev) -> double {
        const long long at = ((long long)t * B + b) * A + i;
        const float nv = (t == T - 1) ? last_val[b * A + i] : val[at + B * A];
        const float v = val[at];
        const double delta = (double)rew[at] + (double)(gamma_f * nv) * nnt - (double)v;
        const double g = delta + gl * nnt * prev;
        const float a32 = (float)g;
        adv[at] = a32;
        ret[at] = a32 + v;
        return g;
    };
    if (MODE == MPE_GAE_REFERENCE) {
        double chain = 0.0;
        for (int t = T - 1; t >= 0; --t) {
            const double nnt = 1.0 - (double)((t == T - 1) ? last_dones[b] : dones[(long long)(t + 1) * B + b]);
            for (int i = 0; i < A; ++i) chain = term(t, i, nnt, chain);
        }
    } else {
        const int i = (int)(tid - b * A);
        double chain = 0.0;
        for (int t = T - 1; t >= 0; --t) chain = term(t, i, 1.0 - (double)dones[(long long)t * B + b], chain);
    }
}

template <int MODE>
hipError_t launch_gae_mode(long long B, int A, int T, const float* rew, const float* val, const uint8_t* dones, const float* last_val,
                           const uint8_t* last_dones, double gamma, double lam, float* adv, float* ret, hipStream_t st) {
    const long long lanes = (MODE == MPE_GAE_REFERENCE) ? B : B * A;
    const dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    hipLaunchKernelGGL((gae_kernel<MODE>), grid, block, 0, st, B, A, T, rew, val, dones, last_val, last_dones, gamma, lam, adv, ret);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gae(int mode, long long B, int A, int T, const float* rew, const float* val, const uint8_t* dones, const float* last_val,
                      const uint8_t* last_dones, double gamma, double lam, float* adv, float* ret, hipStream_t st) {
    return mode == MPE_GAE_REFERENCE ? launch_gae_mode<MPE_GAE_REFERENCE>(B, A, T, rew, val, dones, last_val, last_dones, gamma, lam, adv, ret, st)
                                     : launch_gae_mode<MPE_GAE_STANDARD>(B, A, T, rew, val, dones, last_val, last_dones, gamma, lam, adv, ret, st);
}

}  // namespace mpe
