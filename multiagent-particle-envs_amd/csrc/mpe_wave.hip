// Workgroup-per-world kernels for worlds too large for one lane's registers (up to 64 agents and 64
// landmarks), driven by a run-time per-entity table.
//
// One 128-thread workgroup (2 waves) owns one world.  Entity positions, agent velocities and the
// per-entity properties are staged in LDS; thread i < A integrates agent i, walking its contact
// partners in ascending entity order (the accumulation order of the reference's pair loop, so
// float64 sums are bit-identical), and reads each partner's position as an LDS broadcast.  The
// observation block of a world (A x (4 + 2L + 2(A-1)) values, 66 KB at 64/64) is produced
// element-parallel by both waves and streamed out contiguously as 2-vectors.  For these worlds the
// state is stored env-major (one world's 2E coordinates are contiguous), so loads and stores of a
// world are coalesced rows.  Measured at 64/64, 2048 worlds, float32, 25-step launch: 64 threads
// per world 1.12 ms, 128 threads 0.99 ms, 256 threads 1.21 ms; splitting the reward loops over
// the waves as well made every variant slower, so they stay on the agent threads.
//
// Epilogues: EPI_NONE  -- World.step only; forces and comm actions come in as arrays
//            EPI_SPREAD -- int actions in, simple_spread observation / reward / benchmark_data out
#include <string.h>

#include <mutex>

#include "mpe_launch.hpp"
#include "mpe_math.hpp"

namespace mpe {

enum { EPI_NONE = 0, EPI_SPREAD = 1 };
enum { WMODE_STEP = 0, WMODE_OBSERVE = 2 };
constexpr int NT = 128;     // threads per world

template <typename T>
struct WaveParams {
    const DevTable* table;
    T dt, one_minus_damping, contact_force, contact_margin;
    long long n_envs;
    int A, L, dim_c, n_speak_rows;
    T done_bound;       // > 0: a world is done when an agent leaves [-done_bound, done_bound]^2 (mpe_set_done_bound)
};

template <typename T> __device__ __forceinline__ T spread_round(T v) { return (T)(float)v; }

// NOISE: agents with u_noise (its own instantiation: the generator is an out-of-line call).
template <typename T, int EPI, int MODE, bool CLAMP, bool NOISE = false>
__global__ __launch_bounds__(NT) void wave_kernel(const WaveParams<T> p, const ResetArgs rs, const AutoArgs au, const int* __restrict__ actions,
                                                  const T* __restrict__ u_g, const T* __restrict__ c_g,
                                                  T* __restrict__ pos_g, T* __restrict__ vel_g, T* __restrict__ comm_g,
                                                  T* __restrict__ obs_g, T* __restrict__ rew_g, uint8_t* __restrict__ done_g,
                                                  int* __restrict__ info_i_g, T* __restrict__ info_f_g, T* __restrict__ obs0_g, const NoiseArgs nz,
                                                  int n_steps) {
    // entity positions (agents first) and agent velocities in one array, so that the observation
    // loop can fetch either with one computed index
    __shared__ V2<T> PV[MAX_E + MPE_MAX_AGENTS];
    V2<T>* const P = PV;
    V2<T>* const Vel = PV + MAX_E;
    __shared__ T size_s[MAX_E];
    __shared__ int coll_idx[MAX_E];     // indices of the colliding entities, ascending
    __shared__ T mind_s[MPE_MAX_LANDMARKS];
    __shared__ int n_coll_s[2];        // colliding entities among the first / the second wave's 64
    // the colliding entities packed for the contact loops: (x, y, size) of coll_idx[k]
    __shared__ __attribute__((aligned(16))) T CP[MAX_E][4];

    const int lane = threadIdx.x;      // thread index within the world's workgroup
    const int A = p.A, L = p.L, E = A + L;
    const long long b = blockIdx.x;
    const long long B = p.n_envs;
    const DevTable* tb = p.table;

    // Scenario.reset_world into LDS (same draws as reset_kernel): positions, zero velocities; the landmarks,
    // which the step loop never writes back, go to pos_g at once.  Ends with the workgroup in step.
    auto draw_world = [&](const ResetArgs& r) {
        for (int e = lane; e < E; e += NT) {
            const V2<double> d = reset_draw(r, (unsigned long long)b, e, e < A);
            P[e].x = (T)d.x;
            P[e].y = (T)d.y;
            if (e >= A) {
                pos_g[b * (2 * E) + 2 * e] = P[e].x;
                pos_g[b * (2 * E) + 2 * e + 1] = P[e].y;
            }
        }
        if (lane < A) { Vel[lane].x = T(0); Vel[lane].y = T(0); }
        __syncthreads();
    };
    // ---- stage the world ----
    for (int e = lane; e < E; e += NT) size_s[e] = (T)tb->size[e];
    if (rs.enabled) {       // fused Scenario.reset_world
        draw_world(rs);
    } else {
        for (int e = lane; e < E; e += NT) {
            P[e].x = pos_g[b * (2 * E) + 2 * e];
            P[e].y = pos_g[b * (2 * E) + 2 * e + 1];
        }
        if (lane < A) {
            Vel[lane].x = vel_g[b * (2 * A) + 2 * lane];
            Vel[lane].y = vel_g[b * (2 * A) + 2 * lane + 1];
        }
    }
    // The list of colliding entities, ascending.  Thread e speaks for entity e (NT == MAX_E): a wave's ballot ranks
    // its entities, and the second wave's start where the first wave's count ends.  (One thread walking the table
    // entity by entity took a fifth of a single step's launch.)
    static_assert(NT == MAX_E && NT == 128, "one thread per entity, two waves");
    const bool listed = lane < E && tb->collide[lane] != 0;
    const unsigned long long wave_listed = __ballot(listed);
    if ((lane & 63) == 0) n_coll_s[lane >> 6] = __popcll(wave_listed);
    __syncthreads();
    const int first_wave_count = n_coll_s[0];
    const int n_coll = first_wave_count + n_coll_s[1];
    const int list_pos = (lane < 64 ? 0 : first_wave_count) + __popcll(wave_listed & ((1ull << (lane & 63)) - 1ull));
    if (listed) coll_idx[list_pos] = lane;
    __syncthreads();
    const bool is_agent = lane < A;
    const int me = is_agent ? lane : 0;
    const int my_rank = (is_agent && listed) ? list_pos : -1;       // position of this agent in the list of colliding entities
    for (int k = lane; k < n_coll; k += NT) {
        const int j = coll_idx[k];
        CP[k][0] = P[j].x; CP[k][1] = P[j].y; CP[k][2] = size_s[j];
        CP[k][3] = (tb->movable[j] != 0) ? (T)tb->mass[j] : T(0);      // mass of a movable partner, else 0
    }
    __syncthreads();
    // A partner further away than this contributes nothing: in float32 the softplus of
    // -(dist - dist_min) / k is exactly 0 beyond 16.7 k (1 + exp2(..) rounds to 1), in float64 it is
    // below exp(-60) = 9e-27 beyond 60 k, i.e. a force under 1e-27.
    const T cut = (sizeof(T) == 4 ? T(18) : T(60)) * p.contact_margin;
    const bool i_collide = tb->collide[me] != 0;
    const bool i_move = tb->movable[me] != 0;       // an immovable agent keeps its state (core.py:195)
    const T my_size = size_s[me];
    const T my_mass = (T)tb->mass[me];
    const T my_sens = (T)tb->sens[me];
    const T my_max_speed = (T)tb->max_speed[me];
    const T my_force_scale = (T)tb->force_scale[me];
    const bool new_core = (EPI == EPI_NONE) && tb->new_core != 0;
    const bool want_info = (info_i_g != nullptr);
    const int OD = 4 + 2 * L + 2 * (A - 1);      // simple_spread observation length

    // the packed copy of the colliding entities follows P (after a reset)
    auto repack = [&]() {
        for (int k = lane; k < n_coll; k += NT) {
            const int j = coll_idx[k];
            CP[k][0] = P[j].x; CP[k][1] = P[j].y;
        }
        __syncthreads();
    };
    // ---- observation(), simple_spread.py:110-130: rows of RP = OD/2 coordinate pairs ----
    // Pair r of agent i's row is Vel[i] (r = 0), P[i] (r = 1) or P[src] - P[i]; the loop is branch-
    // free (one computed index into PV, the subtrahend masked out for r < 2) so that the LDS
    // reads of several iterations are in flight together, and (i, r) advance incrementally.
    // (Plain stores also in a launch of one step: as non-temporal 8-byte stores, see put<>, the block took 37.4 us
    // instead of 35.4 at 64/64 x 2 048 worlds.)
    auto emit_obs = [&](T* world_block) {
        const int RP = OD / 2, total = A * RP;
        typedef T T2 __attribute__((ext_vector_type(2)));
        // the world's block starts at a multiple of A*OD values and OD is even: pair q is 2*sizeof(T)-aligned
        T2* dst = reinterpret_cast<T2*>(world_block) + lane;
        int i = (int)(((float)lane + 0.5f) / (float)RP), r = lane - i * RP;      // lane < 128: exact
        #pragma unroll 4
        for (int q = lane; q < total; q += NT, dst += NT) {
            const int m = r - 2 - L;
            const int src = (m < 0) ? A + r - 2 : m + (m >= i ? 1 : 0);
            const int first = (r == 0) ? MAX_E + i : (r == 1 ? i : src);
            const V2<T> sa = PV[first], sb = PV[i];
            const bool rel = r >= 2;
            T2 w;
            w.x = spread_round<T>(sa.x - (rel ? sb.x : T(0)));
            w.y = spread_round<T>(sa.y - (rel ? sb.y : T(0)));
            *dst = w;
            r += NT;
            while (r >= RP) { r -= RP; ++i; }       // once when RP >= 128; worlds with a short row have few pairs
        }
    };

    for (int t = 0; t < n_steps; ++t) {
        const long long row = (long long)t * B + b;
        if (EPI == EPI_SPREAD && MODE == WMODE_STEP && rs.enabled) {
            // episode launches: a launch of several episodes resets again every rs.period steps; the
            // observation of every freshly reset state goes to obs0 (what MultiAgentEnv.reset returns)
            const bool again = rs.period > 0 && t > 0 && t % rs.period == 0;
            if (again) {
                __syncthreads();
                ResetArgs nr = rs;
                nr.episode = rs.episode + (unsigned long long)(t / rs.period);
                draw_world(nr);
                repack();
            }
            if ((t == 0 || again) && obs0_g != nullptr) {
                const long long ep = rs.period > 0 ? t / rs.period : 0;
                emit_obs(obs0_g + (ep * B + b) * (long long)(A * OD));
            }
        }
        if (MODE != WMODE_OBSERVE) {
            V2<T> pi = P[me], vi = Vel[me];
            if (is_agent && i_move) {   // a wave without agents goes straight to the barrier
                // ---- action force ----
                V2<T> f;
                if (EPI == EPI_SPREAD) {
                    const int a = is_agent ? actions[row * A + lane] : 0;
                    f = decode_move<T>(a, my_sens);                       // environment.py:176-195
                    if constexpr (NOISE) {      // p_force = action.u + randn(2) * u_noise (core.py:165)
                        const T un = (T)tb->u_noise[me];
                        if (un != T(0)) {
                            const V2<T> z = noise_pair<T>(nz, t, B, b, lane);
                            f.x = f.x + z.x * un;
                            f.y = f.y + z.y * un;
                        }
                    }
                } else {
                    // agent.action.u as given; the scale is 1 (core.py:166) or mass (* accel) (new_core.py:214)
                    f.x = my_force_scale * u_g[b * (2 * A) + 2 * lane];
                    f.y = my_force_scale * u_g[b * (2 * A) + 2 * lane + 1];
                }
                // ---- contact forces: partners in ascending entity order (core.py:170-181) ----
                // Pass 1 marks the partners within reach (squared distances only, one LDS read each);
                // pass 2 evaluates the softplus force for those, every lane walking its own set bits
                // in ascending order.  A NaN distance counts as within reach, so NaNs spread exactly
                // as in the reference.
                if (i_collide) {
                    const T reach0 = my_size + cut;
                    unsigned near[MAX_E / 32];
#pragma unroll
                    for (int w = 0; w < MAX_E / 32; ++w) {
                        unsigned word = 0;
                        const int k0 = 32 * w, k1 = (n_coll - k0 < 32) ? n_coll - k0 : 32;
#pragma unroll 8
                        for (int kk = 0; kk < k1; ++kk) {
                            const T* c = CP[k0 + kk];
                            const T reach = reach0 + c[2];
                            const T s2 = sqdist<T>(pi.x - c[0], pi.y - c[1]);
                            word |= !(s2 >= reach * reach) ? (1u << kk) : 0u;
                        }
                        if ((my_rank >> 5) == w) word &= ~(1u << (my_rank & 31));     // not with itself
                        near[w] = word;
                    }
#pragma unroll
                    for (int w = 0; w < MAX_E / 32; ++w) {
                        unsigned word = near[w];
                        while (word != 0) {
                            const int kk = __ffs(word) - 1;
                            word &= word - 1;
                            const T* c = CP[32 * w + kk];
                            V2<T> pj;
                            pj.x = c[0]; pj.y = c[1];
                            V2<T> g = contact_force<T>(pi, pj, my_size + c[2], p.contact_force, p.contact_margin);
                            if (new_core && c[3] != T(0)) {
                                // two movable entities share the force by their masses (new_core.py:290-294): the
                                // lower-indexed one takes m_b / m_a of it, the other 1 / (m_b / m_a)
                                const T ratio = (32 * w + kk > my_rank) ? c[3] / my_mass : T(1) / (my_mass / c[3]);
                                g.x = ratio * g.x;
                                g.y = ratio * g.y;
                            }
                            f.x = g.x + f.x;
                            f.y = g.y + f.y;
                        }
                    }
                }
                // ---- wall contact, after the entity contacts (core.py:182-188) ----
                if (EPI == EPI_NONE) {
                    const int n_walls = tb->n_walls;
                    const bool my_ghost = tb->ghost[me] != 0;
                    for (int w = 0; w < n_walls; ++w) {
                        V2<T> wf;
                        if (wall_force<T>(pi, my_size, my_ghost, tb->wall_vertical[w] != 0, tb->wall_hard[w] != 0, (T)tb->wall_axis_pos[w],
                                          (T)tb->wall_end0[w], (T)tb->wall_end1[w], (T)tb->wall_width[w], p.contact_force,
                                          p.contact_margin, wf)) {
                            f.x = f.x + wf.x;
                            f.y = f.y + wf.y;
                        }
                    }
                }
                integrate<T, CLAMP>(pi, vi, f, my_mass, my_max_speed, p.one_minus_damping, p.dt);   // core.py:193-204
            }
            __syncthreads();            // every lane has read the old positions
            if (is_agent) {
                P[lane] = pi; Vel[lane] = vi;
                if (my_rank >= 0) { CP[my_rank][0] = pi.x; CP[my_rank][1] = pi.y; }
            }
            __syncthreads();
            if (EPI == EPI_NONE) {
                // update_agent_state (core.py:206-212): speakers copy their comm action
                for (int r = lane; r < p.n_speak_rows; r += NT) comm_g[b * p.n_speak_rows + r] = c_g[b * p.n_speak_rows + r];
            }
        }

        if (EPI == EPI_SPREAD) {
            bool finished = false;
            // done_callback (environment.py:98,110-113,143-146), any(done_n) over the agents: an agent beyond the bound
            bool outside = false;
            if (MODE != WMODE_OBSERVE && p.done_bound > T(0))
                outside = __syncthreads_or(is_agent && (t_abs<T>(P[me].x) > p.done_bound || t_abs<T>(P[me].y) > p.done_bound)) != 0;
            // The observation needs the new positions, not the rewards: in a launch of one step every other world
            // emits it before its rewards, so that one world's rows leave while its neighbour on the CU runs the reward
            // loops (worlds launched together are in step, and one step gives them no time to drift apart: 38.3 ->
            // 34.8 us at 64/64 x 2 048 worlds; a 25-step launch gains nothing from it).  Not under mpe_step_auto,
            // whose time-limit reset must come between the rewards and the observation.
            const bool obs_first = n_steps == 1 && au.step_count == nullptr && (blockIdx.x & 1u) != 0;
            if (obs_first) emit_obs(obs_g + row * (long long)(A * OD));
            if (MODE != WMODE_OBSERVE || rew_g != nullptr) {
                // ---- reward() :83-108 and benchmark_data() :58-74 ----
                // the second wave takes the landmark minima while the agents' wave counts collisions
                const int lm = lane - 64;
                if (lm >= 0 && lm < L) {
                    const V2<T> pl = P[A + lm];
                    // min over agents of the distance: minimum of the squared distances, rooted once
                    T m = sqdist<T>(P[0].x - pl.x, P[0].y - pl.y);
#pragma unroll 8
                    for (int a = 1; a < A; ++a) m = py_min<T>(m, sqdist<T>(P[a].x - pl.x, P[a].y - pl.y));
                    mind_s[lm] = root<T>(m);
                }
                int c = 0;              // collisions of this agent, counted while the other wave takes the minima
                if (is_agent) {
                    const V2<T> pi = P[lane];
#pragma unroll 8
                    for (int a = 0; a < A; ++a) {
                        if (a == lane) continue;
                        c += touching_many<T>(P[a], pi, size_s[a] + my_size) ? 1 : 0;
                    }
                }
                __syncthreads();
                if (is_agent) {
                    T prox = T(0), min_dists = T(0);
                    int occupied = 0;
#pragma unroll 8
                    for (int l = 0; l < L; ++l) {       // every agent thread repeats the short ordered sum
                        const T m = mind_s[l];
                        prox = prox - m;
                        min_dists = min_dists + m;
                        occupied += (m < T(0.1)) ? 1 : 0;
                    }
                    T bench = prox;
                    for (int n = 0; n < c; ++n) bench = bench - T(1);      // rew -= 1 per collision (:106)
                    const T r_me = (lane == 0 ? prox : T(0)) + (T)(-c);
                    rew_g[row * A + lane] = r_me;
                    if (MODE == WMODE_STEP && au.step_count != nullptr) {
                        T* ep = (T*)au.ep_return + b * A;
                        const T total = ep[lane] + r_me;
                        const bool fin = au.step_count[b] + 1 >= au.max_steps || outside;
                        ep[lane] = fin ? T(0) : total;
                        if (fin) ((T*)au.last_return)[b * A + lane] = total;
                    }
                    if (want_info) {
                        info_i_g[row * (A + 1) + lane] = c;
                        if (info_f_g) info_f_g[row * (A + 1) + 1 + lane] = bench;
                    }
                    if (lane == 0) {
                        if (want_info) {
                            info_i_g[row * (A + 1) + A] = occupied;
                            if (info_f_g) info_f_g[row * (A + 1)] = min_dists;
                        }
                    }
                }
                // No scenario ends an episode (environment.py:143-146): done is 0 unless mpe_step_auto's time limit
                // sets it (below).  A world's flag is one byte of a 128-byte line, so the world that holds the first
                // flag of a 128-flag block writes the whole block.
                if (MODE != WMODE_OBSERVE && au.step_count == nullptr && p.done_bound > T(0)) {
                    if (lane == 0) done_g[row] = outside ? 1 : 0;       // a batch with a done predicate: every world its own flag
                } else if (MODE != WMODE_OBSERVE && au.step_count == nullptr && (b & 127) == 0) {
                    uint8_t* d = done_g + row;
                    const long long n = B - b < 128 ? B - b : 128;
                    if ((((unsigned long long)(uintptr_t)d) & 3u) == 0 && n == 128) {
                        if (lane < 32) reinterpret_cast<unsigned*>(d)[lane] = 0u;
                    } else if (lane < n) {
                        d[lane] = 0;
                    }
                }
                __syncthreads();        // mind_s is rewritten by the next step; P may be reset below
            }
            if (MODE == WMODE_STEP && au.step_count != nullptr) {
                // ---- episode bookkeeping (mpe_step_auto): step counter and time-limit reset ----
                const int cnt = au.step_count[b] + 1;       // read by every thread before thread 0 updates it
                finished = cnt >= au.max_steps || outside;       // the time limit, or the world's own done
                const int epi = au.episode_count[b] + (finished ? 1 : 0);
                __syncthreads();
                if (lane == 0) {
                    au.step_count[b] = finished ? 0 : cnt;
                    au.episode_count[b] = epi;
                    done_g[row] = finished ? 1 : 0;
                }
                if (finished) {     // Scenario.reset_world with the next episode's draws
                    const ResetArgs nr{1, au.seed, (unsigned long long)epi, au.landmark_lim, 0, au.env_base};
                    draw_world(nr);
                    repack();
                }
            }
            if (!obs_first) emit_obs(obs_g + row * (long long)(A * OD));
        }
    }

    if (MODE != WMODE_OBSERVE && is_agent) {
        pos_g[b * (2 * E) + 2 * lane] = P[lane].x;
        pos_g[b * (2 * E) + 2 * lane + 1] = P[lane].y;
        vel_g[b * (2 * A) + 2 * lane] = Vel[lane].x;
        vel_g[b * (2 * A) + 2 * lane + 1] = Vel[lane].y;
    }
}

// ------------------------------------------------------------------------------------------
template <typename T, int EPI, int MODE>
static hipError_t launch_wave_t(const HostSpec& s, const DevTable* table, long long B, const StepArgs& a, const void* u, const void* c,
                                int n_speak_rows, hipStream_t st) {
    WaveParams<T> p;
    p.table = table;
    p.dt = (T)s.dt;
    p.one_minus_damping = (T)(1 - s.damping);
    p.contact_force = (T)s.contact_force;
    p.contact_margin = (T)s.contact_margin;
    p.n_envs = B;
    p.A = s.A; p.L = s.L; p.dim_c = s.dim_c; p.n_speak_rows = n_speak_rows;
    p.done_bound = (T)a.done_bound;
    const dim3 grid((unsigned)B), block(NT);
    if constexpr (EPI == EPI_SPREAD && MODE == WMODE_STEP) {
        if (a.noise.enabled) {
            if (s.any_speed_limit())
                hipLaunchKernelGGL((wave_kernel<T, EPI, MODE, true, true>), grid, block, 0, st, p, a.reset, a.au, a.actions, (const T*)u, (const T*)c, (T*)a.pos,
                                   (T*)a.vel, (T*)a.comm, (T*)a.obs, (T*)a.rew, a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.noise, a.n_steps);
            else
                hipLaunchKernelGGL((wave_kernel<T, EPI, MODE, false, true>), grid, block, 0, st, p, a.reset, a.au, a.actions, (const T*)u, (const T*)c, (T*)a.pos,
                                   (T*)a.vel, (T*)a.comm, (T*)a.obs, (T*)a.rew, a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.noise, a.n_steps);
            return hipGetLastError();
        }
    }
    if (s.any_speed_limit())
        hipLaunchKernelGGL((wave_kernel<T, EPI, MODE, true>), grid, block, 0, st, p, a.reset, a.au, a.actions, (const T*)u, (const T*)c, (T*)a.pos, (T*)a.vel,
                           (T*)a.comm, (T*)a.obs, (T*)a.rew, a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.noise, a.n_steps);
    else
        hipLaunchKernelGGL((wave_kernel<T, EPI, MODE, false>), grid, block, 0, st, p, a.reset, a.au, a.actions, (const T*)u, (const T*)c, (T*)a.pos, (T*)a.vel,
                           (T*)a.comm, (T*)a.obs, (T*)a.rew, a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.noise, a.n_steps);
    return hipGetLastError();
}

bool wave_supported(const HostSpec& s) { return s.scenario == MPE_SCN_SPREAD || s.scenario == MPE_SCN_NONE; }

// mode: 0 step, 1 rollout, 2 observe (scenario handles); physics-only when u is given
hipError_t launch_wave(const HostSpec& s, const DevTable* table, long long B, int dtype, int mode, const StepArgs& a, hipStream_t st) {
    if (B > 0x7fffffffLL) return hipErrorInvalidValue;
    if (mode == 2)
        return dtype == MPE_F64 ? launch_wave_t<double, EPI_SPREAD, WMODE_OBSERVE>(s, table, B, a, nullptr, nullptr, 0, st)
                                : launch_wave_t<float, EPI_SPREAD, WMODE_OBSERVE>(s, table, B, a, nullptr, nullptr, 0, st);
    return dtype == MPE_F64 ? launch_wave_t<double, EPI_SPREAD, WMODE_STEP>(s, table, B, a, nullptr, nullptr, 0, st)
                            : launch_wave_t<float, EPI_SPREAD, WMODE_STEP>(s, table, B, a, nullptr, nullptr, 0, st);
}

hipError_t launch_world_step(const HostSpec& s, const DevTable* table, long long B, int dtype, const void* u, const void* c, void* pos,
                             void* vel, void* comm, int n_speak_rows, hipStream_t st) {
    if (B > 0x7fffffffLL) return hipErrorInvalidValue;
    StepArgs a{};
    a.pos = pos; a.vel = vel; a.comm = comm; a.n_steps = 1;
    return dtype == MPE_F64 ? launch_wave_t<double, EPI_NONE, WMODE_STEP>(s, table, B, a, u, c, n_speak_rows, st)
                            : launch_wave_t<float, EPI_NONE, WMODE_STEP>(s, table, B, a, u, c, n_speak_rows, st);
}

// ---- World.step without a handle and without allocating: the table lives in a static device array ----
//
// mpe_create keeps the per-entity table of a wave-per-world handle in memory it gets from hipMalloc.
// Callers that must not allocate (code under HIP graph capture, serving loops, harnesses that forbid
// hipMalloc) get the same step with the table in `scratch`, a file-scope device array: every GPU the
// library runs on holds its own copy of the array, at its own address, until the process ends.  Each
// call rewrites the whole table (kernel arguments -> scratch, on the caller's stream) before the step
// kernel reads it, so nothing of an earlier call survives; two calls that overlap on one GPU (different
// streams) race for the array.
constexpr size_t SCRATCH_BYTES = MPE_STATIC_SCRATCH_BYTES;
static_assert(sizeof(DevTable) <= SCRATCH_BYTES && sizeof(DevTable) % 4 == 0, "the per-entity table must fit the static scratch array");
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int SCRATCH_MAX_DEVICES = 64;
static void* g_scratch_ptr[SCRATCH_MAX_DEVICES];    // device address of `scratch` on each GPU, resolved on first use
static std::mutex g_scratch_mutex;

// Address of the CURRENT device's copy of `scratch` (the caller has made `device` current).  In host code
// the array's name is only a handle for this lookup, never a device address.  The lookup itself
// (hipGetSymbolAddress, which may load the code object on that device) is not something to do while `capturing`
// stream is recording a graph: an address that is not known yet is then refused with
// hipErrorStreamCaptureUnsupported, before anything touches the stream, so the caller's capture stays valid.
static hipError_t scratch_address(int device, void** out, hipStream_t capturing = nullptr, bool check_capture = false) {
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device] == nullptr) {
        if (check_capture && capturing != nullptr) {
            hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(capturing, &status) == hipSuccess && status != hipStreamCaptureStatusNone)
                return hipErrorStreamCaptureUnsupported;
        }
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (ptr == nullptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// Set-up: resolves the CURRENT device's address of `scratch` ahead of use (mpe_create does this for its device), so
// that a later mpe_world_step_static may be the first one and still run under stream capture.
hipError_t prepare_static_scratch(int device) {
    void* ptr = nullptr;
    return scratch_address(device, &ptr);
}

// The table travels as kernel arguments (copied at launch, also under graph capture), a piece at a time:
// kernel arguments are limited to 4 KiB and the table is a little larger.
constexpr int PIECE_WORDS = 512;
struct TablePiece { uint32_t w[PIECE_WORDS]; };

__global__ __launch_bounds__(256) void table_piece_kernel(uint32_t* __restrict__ dst, const TablePiece piece, int n_words) {
    for (int k = threadIdx.x; k < n_words; k += 256) dst[k] = piece.w[k];
}

hipError_t launch_world_step_static(const HostSpec& s, const DevTable& table, int device, long long B, int dtype, const void* u, const void* c,
                                    void* pos, void* vel, void* comm, int n_speak_rows, hipStream_t st) {
    void* ptr = nullptr;
    hipError_t e = scratch_address(device, &ptr, st, true);
    if (e != hipSuccess) return e;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&table);
    const int total = (int)(sizeof(DevTable) / 4);
    for (int w0 = 0; w0 < total; w0 += PIECE_WORDS) {
        TablePiece piece;
        const int n = total - w0 < PIECE_WORDS ? total - w0 : PIECE_WORDS;
        memcpy(piece.w, src + w0, (size_t)n * 4);
        hipLaunchKernelGGL(table_piece_kernel, dim3(1), dim3(256), 0, st, static_cast<uint32_t*>(ptr) + w0, piece, n);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return launch_world_step(s, static_cast<const DevTable*>(ptr), B, dtype, u, c, pos, vel, comm, n_speak_rows, st);
}

}  // namespace mpe
