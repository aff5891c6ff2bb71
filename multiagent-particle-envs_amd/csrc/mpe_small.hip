// Fused step kernels for the small scenarios (simple_spread 3/3, simple_tag, simple_world_comm).
//
// One lane owns one world and keeps its whole state in registers: it decodes the int actions,
// runs World.step (action force, pairwise contact force, integration, comm state) and evaluates
// the scenario's observation / reward / info callbacks.  State is stored env-minor (SoA), so
// state loads and stores are coalesced 256-byte rows per wave.  The per-env output rows
// (observations, rewards, info) are env-major; each wave transposes them through LDS and streams
// them out as contiguous 16-byte vectors.
//
// A workgroup is a single 64-lane wave: the LDS tiles are private to the wave, so the
// __syncthreads() below are plain LDS waits, not hardware barriers.
#include "mpe_scenarios.hpp"
#include "mpe_launch.hpp"

namespace mpe {

enum { MODE_STEP = 0, MODE_ROLLOUT = 1, MODE_OBSERVE = 2 };

// ---- wave-private LDS tile holding W values for each of the wave's 64 consecutive envs ----------
// Lane l writes its env's row, then the wave streams the tile to the env-major global array.
//
// LINEAR: rows are stored back to back (row stride W), i.e. the tile is a byte image of the wave's
//   contiguous 64*W-element block of the global array, which goes out as 16-byte vectors with no
//   index arithmetic.  Row writes then land with lane stride W: fine when W*sizeof(V)/4 is not a
//   multiple of a large power of two (simple_spread 42, simple_tag 62: 2-way, free for ds_write).
// PADDED: row stride W|1 (odd) keeps both directions conflict-free for any W (simple_world_comm:
//   192 values per row would otherwise be a 32-way conflict) at the price of a divide per element.
template <typename V, int W, bool LINEAR>
struct Tile {
    static constexpr int STRIDE = LINEAR ? W : (W | 1);
    static constexpr int SIZE = 64 * STRIDE;
    static constexpr int VEC = 16 / (int)sizeof(V);

    __device__ static __forceinline__ V* row(V* tile, int lane) { return tile + lane * STRIDE; }

    // full: all 64 envs valid and gdst 16-byte aligned (wave-uniform)
    __device__ static __forceinline__ void flush(const V* tile, V* gdst, int lane, int nvalid, bool full) {
        if (LINEAR && full) {
            typedef V VT __attribute__((ext_vector_type(VEC)));
            constexpr int NV = 64 * W / VEC, FULL_IT = NV / 64, REM = NV % 64;
            const VT* src = reinterpret_cast<const VT*>(tile);
            VT* dst = reinterpret_cast<VT*>(gdst);
#pragma unroll
            for (int it = 0; it < FULL_IT; ++it) dst[it * 64 + lane] = src[it * 64 + lane];
            if (REM > 0 && lane < REM) dst[FULL_IT * 64 + lane] = src[FULL_IT * 64 + lane];
        } else {
            const int n = nvalid * W;
#pragma unroll
            for (int it = 0; it < W; ++it) {
                const int g = it * 64 + lane;
                if (LINEAR) {
                    if (g < n) gdst[g] = tile[g];
                } else {
                    const int r = g / W;
                    if (g < n) gdst[g] = tile[r * STRIDE + (g - r * W)];
                }
            }
        }
    }
};

template <typename T, class Cfg, bool CLAMP, int MODE>
__global__ __launch_bounds__(64) void small_kernel(const SmallParams<T, Cfg::E, Cfg::A> p,
                                                   const int* __restrict__ actions, T* __restrict__ pos_g,
                                                   T* __restrict__ vel_g, T* __restrict__ comm_g,
                                                   T* __restrict__ obs_g, T* __restrict__ rew_g,
                                                   uint8_t* __restrict__ done_g, int* __restrict__ info_i_g,
                                                   T* __restrict__ info_f_g, int n_steps) {
    constexpr int A = Cfg::A, E = Cfg::E, OT = Cfg::OBS_TOTAL, NA = Cfg::N_ACT;
    constexpr int NI = Cfg::NI, NF = Cfg::NF > 0 ? Cfg::NF : 1, DC = Cfg::DIM_C;
    constexpr bool LIN = ((OT * (int)sizeof(T) / 4) % 16) != 0;
    using ObsTile = Tile<T, OT, LIN>;
    using RewTile = Tile<T, A, true>;
    using IiTile = Tile<int, NI, true>;
    using IfTile = Tile<T, NF, true>;
    __shared__ __attribute__((aligned(16))) T tile_obs[ObsTile::SIZE];
    __shared__ __attribute__((aligned(16))) T tile_rew[RewTile::SIZE];
    __shared__ __attribute__((aligned(16))) int tile_ii[IiTile::SIZE];
    __shared__ __attribute__((aligned(16))) T tile_if[IfTile::SIZE];

    const int lane = threadIdx.x;
    const long long B = p.n_envs;
    const long long env0 = (long long)blockIdx.x * 64;
    const long long left = B - env0;
    const int nvalid = left < 64 ? (int)left : 64;
    const bool valid = lane < nvalid;
    const long long b = valid ? env0 + lane : env0;   // idle tail lanes shadow the wave's first env
    // every per-step output block of this wave starts at element (t*B + env0) * W: 16-byte aligned
    // for all t and W when B is a multiple of 4 (env0 is a multiple of 64)
    const bool full = (nvalid == 64) && ((B & 3) == 0);

    V2<T> pos[E];
    V2<T> vel[A];
    T comm[DC > 0 ? DC : 1];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        pos[e].x = pos_g[(long long)(2 * e) * B + b];
        pos[e].y = pos_g[(long long)(2 * e + 1) * B + b];
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        vel[i].x = vel_g[(long long)(2 * i) * B + b];
        vel[i].y = vel_g[(long long)(2 * i + 1) * B + b];
    }
    if (DC > 0) {
#pragma unroll
        for (int c = 0; c < DC; ++c) comm[c] = (MODE == MODE_OBSERVE) ? comm_g[(long long)c * B + b] : T(0);
    }
    const bool want_info = (info_i_g != nullptr);

    // actions of the step about to run; the next step's are fetched while this one computes
    int act[NA];
    if (MODE != MODE_OBSERVE) {
#pragma unroll
        for (int j = 0; j < NA; ++j) act[j] = actions[b * NA + j];
    }

    for (int t = 0; t < n_steps; ++t) {
        if (MODE != MODE_OBSERVE) {
            int act_next[NA];
            if (MODE == MODE_ROLLOUT) {
                const long long tn = (t + 1 < n_steps) ? t + 1 : t;
#pragma unroll
                for (int j = 0; j < NA; ++j) act_next[j] = actions[(tn * B + b) * NA + j];
            }
            // ---- MultiAgentEnv._set_action: int actions -> forces and comm one-hot ----
            V2<T> force[A];
            int col = 0;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                force[i] = decode_move<T>(act[col++], p.sens[i]);     // apply_action_force: u + 0
                if (Cfg::speaks(i)) {
                    const int word = act[col++];
#pragma unroll
                    for (int c = 0; c < DC; ++c) comm[c] = (word == c) ? T(1) : T(0);   // update_agent_state
                }
            }
            // ---- World.apply_environment_force: pairs in the reference's (a, b) order ----
#pragma unroll
            for (int a = 0; a < E; ++a) {
#pragma unroll
                for (int c = a + 1; c < E; ++c) {
                    if (!Cfg::collides(a) || !Cfg::collides(c)) continue;
                    if (a >= A && c >= A) continue;   // neither entity moves
                    const V2<T> f = contact_force<T>(pos[a], pos[c], p.size[a] + p.size[c], p.contact_force, p.contact_margin);
                    if (a < A) { force[a].x = f.x + force[a].x; force[a].y = f.y + force[a].y; }
                    if (c < A) { force[c].x = -f.x + force[c].x; force[c].y = -f.y + force[c].y; }
                }
            }
            // ---- World.integrate_state ----
#pragma unroll
            for (int i = 0; i < A; ++i)
                integrate<T, CLAMP>(pos[i], vel[i], force[i], p.mass[i], p.max_speed[i], p.one_minus_damping, p.dt);
            if (MODE == MODE_ROLLOUT) {
#pragma unroll
                for (int j = 0; j < NA; ++j) act[j] = act_next[j];
            }
        }

        // ---- scenario callbacks into the wave's LDS tiles ----
        const bool want_score = (MODE != MODE_OBSERVE) || (rew_g != nullptr);   // observe: rewards on request
        Cfg::template observe<T>(pos, vel, comm, p, ObsTile::row(tile_obs, lane));
        if (want_score)
            Cfg::template score<T>(pos, p, RewTile::row(tile_rew, lane), want_info, IiTile::row(tile_ii, lane), IfTile::row(tile_if, lane));
        __syncthreads();
        const long long row0 = (long long)t * B + env0;
        ObsTile::flush(tile_obs, obs_g + row0 * OT, lane, nvalid, full);
        if (want_score) {
            RewTile::flush(tile_rew, rew_g + row0 * A, lane, nvalid, full);
            if (MODE != MODE_OBSERVE && valid) done_g[row0 + lane] = 0;
            if (want_info) {
                IiTile::flush(tile_ii, info_i_g + row0 * NI, lane, nvalid, full);
                if (Cfg::NF > 0 && info_f_g != nullptr) IfTile::flush(tile_if, info_f_g + row0 * NF, lane, nvalid, full);
            }
        }
        if (MODE == MODE_ROLLOUT) __syncthreads();   // tiles are rewritten by the next step
    }

    if (MODE != MODE_OBSERVE && valid) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            pos_g[(long long)(2 * i) * B + b] = pos[i].x;
            pos_g[(long long)(2 * i + 1) * B + b] = pos[i].y;
            vel_g[(long long)(2 * i) * B + b] = vel[i].x;
            vel_g[(long long)(2 * i + 1) * B + b] = vel[i].y;
        }
        if (DC > 0) {
#pragma unroll
            for (int c = 0; c < DC; ++c) comm_g[(long long)c * B + b] = comm[c];
        }
    }
}

// ------------------------------------------------------------------------------------------
template <typename T, class Cfg>
static SmallParams<T, Cfg::E, Cfg::A> make_params(const HostSpec& s, long long n_envs) {
    SmallParams<T, Cfg::E, Cfg::A> p;
    for (int e = 0; e < Cfg::E; ++e) p.size[e] = (T)s.size[e];
    for (int i = 0; i < Cfg::A; ++i) {
        p.mass[i] = (T)s.mass[i];
        p.sens[i] = (T)s.sens[i];
        p.max_speed[i] = (T)s.max_speed[i];
    }
    p.dt = (T)s.dt;
    p.one_minus_damping = (T)(1 - s.damping);   // reference evaluates (1 - damping) in float64 first
    p.contact_force = (T)s.contact_force;
    p.contact_margin = (T)s.contact_margin;
    p.n_envs = n_envs;
    return p;
}

template <typename T, class Cfg, bool CLAMP>
static hipError_t launch_cfg(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    const auto p = make_params<T, Cfg>(s, B);
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    T* pos = (T*)a.pos; T* vel = (T*)a.vel; T* comm = (T*)a.comm; T* obs = (T*)a.obs; T* rew = (T*)a.rew; T* inf = (T*)a.info_f;
    switch (mode) {
        case MODE_STEP:
            hipLaunchKernelGGL((small_kernel<T, Cfg, CLAMP, MODE_STEP>), grid, block, 0, st, p, a.actions, pos, vel, comm, obs, rew, a.done, a.info_i, inf, 1);
            break;
        case MODE_ROLLOUT:
            hipLaunchKernelGGL((small_kernel<T, Cfg, CLAMP, MODE_ROLLOUT>), grid, block, 0, st, p, a.actions, pos, vel, comm, obs, rew, a.done, a.info_i, inf, a.n_steps);
            break;
        default:
            hipLaunchKernelGGL((small_kernel<T, Cfg, CLAMP, MODE_OBSERVE>), grid, block, 0, st, p, a.actions, pos, vel, comm, obs, rew, a.done, a.info_i, inf, 1);
            break;
    }
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_small_t(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    const bool clamp = s.any_speed_limit();
    switch (s.scenario) {
        case MPE_SCN_SPREAD:
            return clamp ? launch_cfg<T, SpreadCfg<3, 3>, true>(s, B, mode, a, st) : launch_cfg<T, SpreadCfg<3, 3>, false>(s, B, mode, a, st);
        case MPE_SCN_TAG:
            return launch_cfg<T, TagCfg, true>(s, B, mode, a, st);
        case MPE_SCN_WORLD_COMM:
            return launch_cfg<T, WorldCommCfg, true>(s, B, mode, a, st);
    }
    return hipErrorInvalidValue;
}

bool small_supported(const HostSpec& s) {
    return (s.scenario == MPE_SCN_SPREAD && s.A == 3 && s.L == 3) || s.scenario == MPE_SCN_TAG || s.scenario == MPE_SCN_WORLD_COMM;
}

hipError_t launch_small(const HostSpec& s, long long B, int dtype, int mode, const StepArgs& a, hipStream_t st) {
    return dtype == MPE_F64 ? launch_small_t<double>(s, B, mode, a, st) : launch_small_t<float>(s, B, mode, a, st);
}

}  // namespace mpe
