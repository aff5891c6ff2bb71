// Fused step of the consensus scenario (multiagent/scenarios/consensus.py on multiagent/new_core.py):
// six non-colliding agents in the walled arena whose observations are the last ten positions and
// velocities of their two ring neighbours relative to their own.
//
// One lane owns one world, one wave a tile of 64 worlds.  Positions and velocities (24 values) live
// in registers; the history (240 values per world) stays in device memory as a ring of ten slots,
// hist T[(slot * A + agent) * 4 + {px, py, vx, vy}][B], whose oldest slot is aux[0][b]: a step
// overwrites the oldest slot with the new state and advances the head, which is what
// World.integrate_state's vstack([his[1:], new]) does to the ordered history (new_core.py:252-254).
// Rows are env-minor, so as long as the worlds of a wave share their head (they do unless they were
// reset at different times) every history access is a coalesced 256-byte row.
//
// The observation of a world-step is 6 x 80 values; it is produced one agent at a time (a window of
// three agents' histories in registers slides round the ring) into an
// LDS tile of 64 x 80 (row stride 81: conflict-free row writes) and streamed to the env-major
// output 4 bytes per lane, i.e. contiguous 256-byte runs.  Rewards leave through a small tile too.
#include "mpe_launch.hpp"
#include "mpe_math.hpp"

namespace mpe {

namespace {

constexpr int CA = 6;            // agents
constexpr int CH = 10;           // history length (consensus.py:56-58)
constexpr int COBS = 2 * 2 * CH * 2;     // per agent: 2 neighbours x (velocity, position) x CH x 2 coordinates
constexpr int CW = 4;            // walls
constexpr int CSTRIDE = COBS + 1;

template <typename T>
struct ConsensusParams {
    T size, mass, sens, force_scale, u_noise;       // uniform over the agents (mpe_create checks)
    T dt, one_minus_damping, contact_force, contact_margin;
    T wall_axis_pos[CW], wall_end0[CW], wall_end1[CW], wall_width[CW];
    int wall_vertical[CW], wall_hard[CW];          // dwords: scalar loads cannot fetch single bytes of the kernel arguments
    int ghost;
    long long n_envs;
};

// neighbors(), consensus.py:79-83: the ring neighbours of agent i in ascending order
__host__ __device__ constexpr int nb_lo(int i) { return i == 0 ? 1 : (i == CA - 1 ? 0 : i - 1); }
__host__ __device__ constexpr int nb_hi(int i) { return i == 0 ? CA - 1 : (i == CA - 1 ? CA - 2 : i + 1); }

template <typename T> __device__ __forceinline__ T f32r(T v) { return (T)(float)v; }     // .astype(np.float32), consensus.py:140

// MODE 0: mpe_step, 1: mpe_rollout / mpe_episode, 2: mpe_observe
// NOISE: agents with u_noise (its own instantiation: the generator is an out-of-line call)
template <typename T, int MODE, bool NOISE = false>
__global__ __launch_bounds__(64) void consensus_kernel(const ConsensusParams<T> p, const ResetArgs rs, const AutoArgs au, const NoiseArgs nz, const int* __restrict__ actions,
                                                       T* __restrict__ pos_g, T* __restrict__ vel_g, T* __restrict__ hist_g,
                                                       int* __restrict__ aux_g, T* __restrict__ obs_g, T* __restrict__ rew_g,
                                                       uint8_t* __restrict__ done_g, T* __restrict__ obs0_g, int n_steps) {
    __shared__ T tile_obs[64 * CSTRIDE];
    __shared__ T tile_rew[64 * CA];
    const int lane = threadIdx.x;
    const long long B = p.n_envs;
    const long long env0 = (long long)blockIdx.x * 64;
    const long long left = B - env0;
    const int nvalid = left < 64 ? (int)left : 64;
    const bool valid = lane < nvalid;
    const long long b = valid ? env0 + lane : env0;     // idle tail lanes shadow the wave's first env

    auto hrow = [&](int slot, int i, int c) -> long long { return ((long long)((slot * CA + i) * 4 + c)) * B + b; };

    V2<T> pos[CA], vel[CA];
    int head;
    // reset_world, consensus.py:46-59: agents U(-1,1)^2 at rest, every history row equal to the start
    auto reset_world = [&](const ResetArgs& r) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const V2<double> d = reset_draw(r, (unsigned long long)b, i, true);
            pos[i].x = (T)d.x; pos[i].y = (T)d.y;
            vel[i].x = T(0); vel[i].y = T(0);
        }
        head = 0;
        if (valid) {
            for (int s = 0; s < CH; ++s)
#pragma unroll
                for (int i = 0; i < CA; ++i) {
                    hist_g[hrow(s, i, 0)] = pos[i].x; hist_g[hrow(s, i, 1)] = pos[i].y;
                    hist_g[hrow(s, i, 2)] = T(0); hist_g[hrow(s, i, 3)] = T(0);
                }
        }
    };
    if (MODE == 1 && rs.enabled) {
        reset_world(rs);
    } else {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            pos[i].x = pos_g[(long long)(2 * i) * B + b]; pos[i].y = pos_g[(long long)(2 * i + 1) * B + b];
            vel[i].x = vel_g[(long long)(2 * i) * B + b]; vel[i].y = vel_g[(long long)(2 * i + 1) * B + b];
        }
        head = aux_g[b];
        head = (head < 0 || head >= CH) ? 0 : head;
    }

    // Episode launches with obs0: the observation of every freshly reset state (what MultiAgentEnv.reset returns,
    // environment.py:122-127) is produced by a pass through the loop body that skips the step and the rewards
    // ("pre") before the episode's first step, so the observation code exists once.
    const bool want_obs0 = (MODE == 1) && rs.enabled && obs0_g != nullptr;
    bool pre = want_obs0;
    int t = 0, last_reset = 0;
    while (t < n_steps) {
        const long long row = (long long)t * B + env0;
        if (MODE == 1 && rs.enabled && rs.period > 0 && t % rs.period == 0 && t != last_reset) {
            // next episode of a launch of several (mpe_episodes)
            ResetArgs nr = rs;
            nr.episode = rs.episode + (unsigned long long)(t / rs.period);
            reset_world(nr);
            last_reset = t;
            pre = want_obs0;
        }
        bool finished = false;
        if (MODE != 2 && !pre) {
            // ---- _set_action (environment.py:176-195) and World.step (new_core.py:188-205) ----
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int a = actions[((long long)t * B + b) * CA + i];
                V2<T> f = decode_move<T>(a, p.sens);
                f.x = p.force_scale * f.x; f.y = p.force_scale * f.y;           // new_core.py:214
                if constexpr (NOISE) {                                          // + randn(2) * u_noise, new_core.py:213
                    const V2<T> z = noise_pair<T>(nz, t, B, b, i);
                    f.x = f.x + z.x * p.u_noise; f.y = f.y + z.y * p.u_noise;
                }
                // nothing collides (consensus.py:30); the walls act on every agent (:35-41, new_core.py:230-236)
#pragma unroll
                for (int w = 0; w < CW; ++w) {
                    const T perp = p.wall_vertical[w] ? pos[i].x : pos[i].y;
                    const T cut = (sizeof(T) == 4 ? T(18) : T(60)) * p.contact_margin;     // beyond it the softplus is 0 (float32) or < e^-60
                    if (t_abs<T>(perp - p.wall_axis_pos[w]) - (p.size + T(0.5) * p.wall_width[w]) > cut) continue;
                    V2<T> wf;
                    if (wall_force_fast<T>(pos[i], p.size, p.ghost != 0, p.wall_vertical[w] != 0, p.wall_hard[w] != 0, p.wall_axis_pos[w],
                                           p.wall_end0[w], p.wall_end1[w], p.wall_width[w], p.contact_force, p.contact_margin, wf)) {
                        f.x = f.x + wf.x; f.y = f.y + wf.y;
                    }
                }
                integrate<T, false>(pos[i], vel[i], f, p.mass, T(0), p.one_minus_damping, p.dt);
            }
            // the new state replaces the oldest history slot
            if (valid) {
#pragma unroll
                for (int i = 0; i < CA; ++i) {
                    hist_g[hrow(head, i, 0)] = pos[i].x; hist_g[hrow(head, i, 1)] = pos[i].y;
                    hist_g[hrow(head, i, 2)] = vel[i].x; hist_g[hrow(head, i, 3)] = vel[i].y;
                }
            }
            head = (head + 1 == CH) ? 0 : head + 1;
        }

        // ---- reward(), consensus.py:91-113; shared by np.sum over the six agents (environment.py:103-105) ----
        const bool want_score = !pre && (MODE != 2 || rew_g != nullptr);
        if (want_score) {
            T own[CA];
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int n0 = nb_lo(i), n1 = nb_hi(i);
                const T d0 = norm2<T>(pos[n0].x - pos[i].x, pos[n0].y - pos[i].y);
                const T d1 = norm2<T>(pos[n1].x - pos[i].x, pos[n1].y - pos[i].y);
                T r = T(0) - (t_abs<T>(d0 - T(0.8)) + t_abs<T>(d1 - T(0.8))) * T(0.1);
                if (i == 0) r = r - t_abs<T>(norm2<T>(pos[0].x, pos[0].y) - T(0.8)) * T(0.1);
                r = r - T(2) * arena_bound<T>(t_abs<T>(pos[i].x));
                r = r - T(2) * arena_bound<T>(t_abs<T>(pos[i].y));
                own[i] = r;
            }
            T total = own[0];
#pragma unroll
            for (int i = 1; i < CA; ++i) total = total + own[i];
#pragma unroll
            for (int i = 0; i < CA; ++i) tile_rew[lane * CA + i] = (MODE == 2) ? own[i] : total;   // observe: what the callback returns
            if (MODE == 0 && au.step_count != nullptr) {
                // ---- episode bookkeeping (mpe_step_auto): returns, step counter, time-limit reset ----
                T* ep = (T*)au.ep_return + b * CA;
                const int cnt = au.step_count[b] + 1;
                finished = cnt >= au.max_steps;
                if (valid) {
#pragma unroll
                    for (int i = 0; i < CA; ++i) {
                        const T sum = ep[i] + total;
                        ep[i] = finished ? T(0) : sum;
                        if (finished) ((T*)au.last_return)[b * CA + i] = sum;
                    }
                    au.step_count[b] = finished ? 0 : cnt;
                }
                if (finished) {       // Scenario.reset_world with the next episode's draws; obs below is the fresh one
                    const int epi = au.episode_count[b] + 1;
                    if (valid) au.episode_count[b] = epi;
                    reset_world(ResetArgs{1, au.seed, (unsigned long long)epi, au.landmark_lim, 0});
                }
            }
        }

        // ---- observation(), consensus.py:122-140, one agent at a time ----
        // Agent i needs the histories of i - 1, i, i + 1: walking the ring in order, each step of the walk
        // loads one more agent's 40 values and drops one (agent 5's stay for the end, agent 0's are
        // fetched again for agent 5), 7 agent histories per world-step instead of 18.
        {
            T* const block = pre ? obs0_g + ((long long)(rs.period > 0 ? t / rs.period : 0) * B + env0) * (CA * COBS)
                                 : obs_g + row * (CA * COBS);      // the wave's 64 rows of the env-major array
            T H[CA][4 * CH];
            auto load_agent = [&](int j) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    int slot = head + k;
                    slot = slot >= CH ? slot - CH : slot;
#pragma unroll
                    for (int c = 0; c < 4; ++c) H[j][4 * k + c] = hist_g[hrow(slot, j, c)];
                }
            };
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                if (i == 0) { load_agent(CA - 1); load_agent(0); load_agent(1); }
                else if (i + 1 < CA - 1) load_agent(i + 1);
                else if (i == CA - 1) load_agent(0);
                const int n0 = nb_lo(i), n1 = nb_hi(i);
                T* orow = tile_obs + lane * CSTRIDE;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const T* me = &H[i][4 * k];
                    const T* lo = &H[n0][4 * k];
                    const T* hi = &H[n1][4 * k];
                    // [lo vel history (20), hi vel history (20), lo pos history (20), hi pos history (20)]
                    orow[2 * k] = f32r(lo[2] - me[2]); orow[2 * k + 1] = f32r(lo[3] - me[3]);
                    orow[2 * CH + 2 * k] = f32r(hi[2] - me[2]); orow[2 * CH + 2 * k + 1] = f32r(hi[3] - me[3]);
                    orow[4 * CH + 2 * k] = f32r(lo[0] - me[0]); orow[4 * CH + 2 * k + 1] = f32r(lo[1] - me[1]);
                    orow[6 * CH + 2 * k] = f32r(hi[0] - me[0]); orow[6 * CH + 2 * k + 1] = f32r(hi[1] - me[1]);
                }
                __syncthreads();
                // stream the 64 x 80 tile out as 16-byte vectors (an agent's 80 values start on a 16-byte boundary
                // of the row for both float types): vector g covers values [VEC * g, VEC * g + VEC) of row g / NVR
                constexpr int VEC = 16 / (int)sizeof(T), NVR = COBS / VEC;
                typedef T VT __attribute__((ext_vector_type(VEC)));
                T* gbase = block + i * COBS;
#pragma unroll 4
                for (int it = 0; it < NVR; ++it) {          // 64 * NVR vectors, 64 per iteration
                    const int g = it * 64 + lane;
                    const int r = g / NVR, j = (g - r * NVR) * VEC;
                    VT v;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[q] = tile_obs[r * CSTRIDE + j + q];
                    if (r < nvalid) *reinterpret_cast<VT*>(gbase + (long long)r * (CA * COBS) + j) = v;
                }
                __syncthreads();
            }
        }
        if (want_score) {
            const int n = nvalid * CA;
            for (int it = 0; it < CA; ++it) {
                const int g = it * 64 + lane;
                if (g < n) rew_g[row * CA + g] = tile_rew[g];
            }
            // no done callback (environment.py:143-146); mpe_step_auto flags the step that reaches its time limit
            if (MODE != 2 && valid) done_g[row + lane] = finished ? 1 : 0;
            __syncthreads();
        }
        if (pre) pre = false;
        else ++t;
    }

    if (MODE != 2 && valid) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            pos_g[(long long)(2 * i) * B + b] = pos[i].x; pos_g[(long long)(2 * i + 1) * B + b] = pos[i].y;
            vel_g[(long long)(2 * i) * B + b] = vel[i].x; vel_g[(long long)(2 * i + 1) * B + b] = vel[i].y;
        }
        aux_g[b] = head;
    }
}

// reset_world for the envs whose mask byte is set (all when mask is NULL)
template <typename T>
__global__ __launch_bounds__(256) void consensus_reset_kernel(long long B, const ResetArgs rs, const uint8_t* __restrict__ mask,
                                                              T* __restrict__ pos, T* __restrict__ vel, T* __restrict__ hist, int* __restrict__ aux) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mask != nullptr && mask[b] == 0) return;
    for (int i = 0; i < CA; ++i) {
        const V2<double> d = reset_draw(rs, (unsigned long long)b, i, true);
        pos[(long long)(2 * i) * B + b] = (T)d.x; pos[(long long)(2 * i + 1) * B + b] = (T)d.y;
        vel[(long long)(2 * i) * B + b] = T(0); vel[(long long)(2 * i + 1) * B + b] = T(0);
        for (int s = 0; s < CH; ++s) {
            const long long r = (long long)((s * CA + i) * 4) * B + b;
            hist[r] = (T)d.x; hist[r + B] = (T)d.y; hist[r + 2 * B] = T(0); hist[r + 3 * B] = T(0);
        }
    }
    aux[b] = 0;
}

template <typename T>
ConsensusParams<T> make_consensus_params(const HostSpec& s, long long B) {
    ConsensusParams<T> p;
    p.size = (T)s.size[0]; p.mass = (T)s.mass[0]; p.sens = (T)s.sens[0]; p.force_scale = (T)s.force_scale[0]; p.u_noise = (T)s.u_noise[0];
    p.dt = (T)s.dt; p.one_minus_damping = (T)(1 - s.damping); p.contact_force = (T)s.contact_force; p.contact_margin = (T)s.contact_margin;
    for (int w = 0; w < CW; ++w) {
        p.wall_axis_pos[w] = (T)s.wall_axis_pos[w]; p.wall_end0[w] = (T)s.wall_end0[w]; p.wall_end1[w] = (T)s.wall_end1[w];
        p.wall_width[w] = (T)s.wall_width[w]; p.wall_vertical[w] = s.wall_vertical[w]; p.wall_hard[w] = s.wall_hard[w];
    }
    p.ghost = s.ghost[0];
    p.n_envs = B;
    return p;
}

template <typename T>
hipError_t launch_consensus_t(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    const auto p = make_consensus_params<T>(s, B);
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    T *pos = (T*)a.pos, *vel = (T*)a.vel, *hist = (T*)a.comm, *obs = (T*)a.obs, *rew = (T*)a.rew;
    if (mode == 0 && a.noise.enabled)
        hipLaunchKernelGGL((consensus_kernel<T, 0, true>), grid, block, 0, st, p, a.reset, a.au, a.noise, a.actions, pos, vel, hist, a.aux, obs, rew, a.done, (T*)nullptr, 1);
    else if (mode == 1 && a.noise.enabled)
        hipLaunchKernelGGL((consensus_kernel<T, 1, true>), grid, block, 0, st, p, a.reset, a.au, a.noise, a.actions, pos, vel, hist, a.aux, obs, rew, a.done, (T*)a.obs0, a.n_steps);
    else if (mode == 0)
        hipLaunchKernelGGL((consensus_kernel<T, 0>), grid, block, 0, st, p, a.reset, a.au, a.noise, a.actions, pos, vel, hist, a.aux, obs, rew, a.done, (T*)nullptr, 1);
    else if (mode == 1)
        hipLaunchKernelGGL((consensus_kernel<T, 1>), grid, block, 0, st, p, a.reset, a.au, a.noise, a.actions, pos, vel, hist, a.aux, obs, rew, a.done, (T*)a.obs0, a.n_steps);
    else
        hipLaunchKernelGGL((consensus_kernel<T, 2>), grid, block, 0, st, p, a.reset, a.au, a.noise, a.actions, pos, vel, hist, a.aux, obs, rew, a.done, (T*)nullptr, 1);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_consensus(const HostSpec& s, long long B, int dtype, int mode, const StepArgs& a, hipStream_t st) {
    return dtype == MPE_F64 ? launch_consensus_t<double>(s, B, mode, a, st) : launch_consensus_t<float>(s, B, mode, a, st);
}

hipError_t launch_consensus_reset(const HostSpec& s, long long B, int dtype, unsigned long long seed, unsigned long long episode,
                                  const uint8_t* mask, void* pos, void* vel, void* hist, int32_t* aux, hipStream_t st) {
    const ResetArgs rs{1, seed, episode, 1.0};
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    if (dtype == MPE_F64)
        hipLaunchKernelGGL(consensus_reset_kernel<double>, grid, block, 0, st, B, rs, mask, (double*)pos, (double*)vel, (double*)hist, aux);
    else
        hipLaunchKernelGGL(consensus_reset_kernel<float>, grid, block, 0, st, B, rs, mask, (float*)pos, (float*)vel, (float*)hist, aux);
    return hipGetLastError();
}

}  // namespace mpe
