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
//
// wave_kernel serves every launch of one step, the physics-only step, double worlds and worlds with noise.  Launches of
// several steps of float simple_spread worlds run on wave_stream_kernel (further down), in which the two waves of a world
// have different jobs: one steps, the other writes observations.
#include <string.h>

#include <mutex>

#include "mpe_launch.hpp"
#include "mpe_math.hpp"

namespace mpe {

enum { EPI_NONE = 0, EPI_SPREAD = 1 };
#ifndef MPE_WAVE_F64_WAVES
#define MPE_WAVE_F64_WAVES 4        // waves per SIMD the double simple_spread instantiations are compiled for
#endif
enum { WMODE_STEP = 0, WMODE_OBSERVE = 2 };
constexpr int NT = 128;     // threads per world

template <typename T>
struct WaveParams {
    const DevTable* table;
    T dt, one_minus_damping, contact_force, contact_margin;
    long long n_envs;
    int A, L, dim_c, n_speak_rows;
    T done_bound;       // > 0: a world is done when an agent leaves [-done_bound, done_bound]^2 (mpe_set_done_bound)
    XcdDeal deal;       // wave_stream_kernel: worlds dealt unevenly to even and odd XCDs (phase == nullptr: workgroup i is world i)
};

template <typename T> __device__ __forceinline__ T spread_round(T v) { return (T)(float)v; }

#ifdef MPE_WSTAMPS      // diagnostic build: 100 MHz timestamps of the first 2048 worlds' two waves, four per step for 32 steps
__device__ long long g_wstamps[2048 * 2 * 32 * 4];
}  // namespace mpe
extern "C" int mpe_debug_wstamps(long long* host, long long count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mpe::g_wstamps), (size_t)count * sizeof(long long));
}
namespace mpe {
#define WSTAMP(slot) do { if ((lane & 63) == 0 && b < 2048 && t < 32) \
    g_wstamps[((b * 2 + (lane >> 6)) * 32 + t) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

// Contact forces on one agent, added to `f` in ascending partner order (core.py:170-181).  CP holds (x, y, size, mass of
// a movable partner or 0) of the n_coll colliding entities in ascending entity order; my_rank is the agent's own place in
// that list (-1: not in it).  Pass 1 marks the partners within reach (squared distances only, one LDS read each); pass 2
// evaluates the softplus force for those, every lane walking its own set bits in ascending order.  A NaN distance counts
// as within reach, so NaNs spread exactly as in the reference.
template <typename T>
__device__ __forceinline__ V2<T> contact_sum(const T (*CP)[4], int n_coll, int my_rank, V2<T> pi, T my_size, T my_mass, T cut, T contact_f,
                                             T contact_margin, bool new_core, V2<T> f) {
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
            V2<T> g = contact_force<T>(pi, pj, my_size + c[2], contact_f, contact_margin);
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
    return f;
}

// Collisions of agent `self` with the other agents of S[0..A) (simple_spread.py:66-72,104-107), branch-free so that the
// LDS reads of eight partners are in flight together (with a branch per partner every iteration waited for its own read:
// 13 of a step's 33 us at 64 agents).  float: the reference's own comparison sqrt(s) < dist_min, one instruction; double:
// the squared distance decides unless it lies within a few ulps of dist_min^2 (touching_many's rule, same truth value).
template <typename T>
__device__ __forceinline__ int count_collisions(const V2<T>* S, const T* size_s, int A, int self, T my_size) {
    const V2<T> pi = S[self];
    int c = 0;
#pragma unroll 8
    for (int a = 0; a < A; ++a) {
        const V2<T> pa = S[a];
        const T dist_min = size_s[a] + my_size;
        const T s = sqdist<T>(pa.x - pi.x, pa.y - pi.y);
        bool touch;
        if constexpr (sizeof(T) == 4) {
            touch = root<T>(s) < dist_min;
        } else {
            const T d2 = dist_min * dist_min, slack = T(32) * T(2.220446049250313e-16);
            touch = s < d2 * (T(1) - slack);
            if (__builtin_expect(!touch && s < d2 * (T(1) + slack), 0)) touch = root<T>(s) < dist_min;
        }
        c += (touch && a != self) ? 1 : 0;
    }
    return c;
}

// Table indices of one coordinate pair of a simple_spread observation (simple_spread.py:110-130): the row of agent i is
// its velocity, its position, the L landmarks and the other agents relative to it, RP = A + L + 1 pairs.  Low byte: index
// of the minuend in the world's table of pairs (entity positions at 0.., velocities at MAX_E..); next byte: the
// subtrahend (agent i, or ZERO for the row's own velocity and position).
__device__ __forceinline__ unsigned spread_row_entry(int i, int r, int A, int L, int zero_slot) {      // pair r of agent i's row
    const int m = r - 2 - L;
    const int src = (m < 0) ? A + r - 2 : m + (m >= i ? 1 : 0);
    const int first = (r == 0) ? MAX_E + i : (r == 1 ? i : src);
    return (unsigned)first | ((unsigned)(r >= 2 ? i : zero_slot) << 8);
}

// NOISE: agents with u_noise (its own instantiation: the generator is an out-of-line call).
template <typename T, int EPI, int MODE, bool CLAMP, bool NOISE = false>
__global__ __launch_bounds__(NT, (MPE_WAVE_F64_WAVES == 4 && sizeof(T) == 8 && EPI == EPI_SPREAD) ? 4 : 1) void wave_kernel(const WaveParams<T> p, const ResetArgs rs, const AutoArgs au, const int* __restrict__ actions,
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
    // The next ACT_CHUNK steps' actions of every agent, a byte each (column `lane` is written and read by agent `lane`
    // only).  On this hardware a wave's loads and stores retire through one in-order counter, so an action loaded inside
    // a step would wait for every observation row the wave stored before it; fetched a chunk ahead, the step loop has no
    // load left and a world's physics runs while its rows drain.
    constexpr int ACT_CHUNK = 32;
    __shared__ uint8_t act_s[EPI == EPI_SPREAD ? ACT_CHUNK : 1][MPE_MAX_AGENTS];

    const int lane = threadIdx.x;      // thread index within the world's workgroup
    const int A = p.A, L = p.L, E = A + L;
    const long long B = p.n_envs;
    const long long b = blockIdx.x;
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
#ifdef MPE_ABLATE_OBSMATH   // diagnostic build: the rows' stores without their index arithmetic and LDS reads
            { T2 w; w.x = (T)q; w.y = (T)lane; *dst = w; continue; }
#endif
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
        WSTAMP(0);
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
        if (EPI == EPI_SPREAD && MODE != WMODE_OBSERVE && (t % ACT_CHUNK) == 0 && is_agent) {
            const int n = n_steps - t < ACT_CHUNK ? n_steps - t : ACT_CHUNK;
            for (int s0 = 0; s0 < n; s0 += 16) {
                int av[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) av[j] = (s0 + j < n) ? actions[((long long)(t + s0 + j) * B + b) * A + lane] : 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) act_s[s0 + j][lane] = (uint8_t)((unsigned)av[j] <= 4u ? av[j] : 0);     // anything else moves nothing
            }
        }
        if (MODE != WMODE_OBSERVE) {
            V2<T> pi = P[me], vi = Vel[me];
            if (is_agent && i_move) {   // a wave without agents goes straight to the barrier
                // ---- action force ----
                V2<T> f;
                if (EPI == EPI_SPREAD) {
                    const int a = act_s[t % ACT_CHUNK][lane];
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
#ifdef MPE_ABLATE_PHYS      // diagnostic build: no contact loops, no reward loops
                if (n_steps == 0x7fffffff)
#endif
                if (i_collide) f = contact_sum<T>(CP, n_coll, my_rank, pi, my_size, my_mass, cut, p.contact_force, p.contact_margin, new_core, f);
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
            WSTAMP(1);
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
#ifdef MPE_ABLATE_OBS       // diagnostic build: the launch without its observation rows
            if (n_steps == 0x7fffffff)
#endif
            if (obs_first) emit_obs(obs_g + row * (long long)(A * OD));
#ifdef MPE_ABLATE_PHYS
            if (n_steps == 0x7fffffff)
#endif
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
                // collisions of this agent, counted while the other wave takes the minima
                const int c = is_agent ? count_collisions<T>(P, size_s, A, lane, my_size) : 0;
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
            WSTAMP(2);
#ifdef MPE_ABLATE_OBS
            if (n_steps == 0x7fffffff)
#endif
            if (!obs_first) emit_obs(obs_g + row * (long long)(A * OD));
            WSTAMP(3);
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
// Launches of several steps of float simple_spread worlds: one wave steps the world, the other streams observations.
//
// In wave_kernel both waves of a world write the observation block, then both wait while the agents' wave runs the
// physics and the rewards; the worlds of a CU, served in turn by one store path, fall into step with one another, all
// store together and all compute together, and the store path idles while they compute (the launch then takes store
// time PLUS part of the compute time).  Here the two waves never take turns: wave 0 (one lane per agent) produces state
// S_{j+1} from S_j -- a step with its rewards, or the next episode's reset -- while wave 1 writes the observation of
// S_j as 16-byte vectors, and the two meet at one barrier per state.  The states alternate between two copies of the
// world's table of pairs in LDS, so the stepping wave never writes what the streaming wave reads.  Every CU thus has
// eight waves that do nothing but store, and the rows leave without gaps.  Results are those of wave_kernel bit for bit
// (same operations in the same order).
//
// State sequence of a launch: K episodes of up to P steps (P = rs.period, or n_steps; a plain rollout is one episode
// whose first state is the loaded one and is not written): state j is (episode e, k) with j = e (P + 1) + k; k = 0 is
// the reset state (-> obs0[e] if asked for), k >= 1 the result of step t = e P + k - 1 (-> obs[t]).
constexpr int STREAM_QUADS_PER_LANE = (MPE_MAX_AGENTS * (MPE_MAX_AGENTS + MPE_MAX_LANDMARKS + 1) / 2 + 63) / 64;

#ifdef MPE_WSTAMPS
#define SSTAMP(t, slot) do { if (lane == 0 && b < 2048 && (t) < 32) \
    g_wstamps[((b * 2 + wave) * 32 + (t)) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SSTAMP(t, slot) do { } while (0)
#endif

template <typename T, bool CLAMP>
__global__ __launch_bounds__(NT, 4) void wave_stream_kernel(const WaveParams<T> p, const ResetArgs rs, const int* __restrict__ actions,
                                                            T* __restrict__ pos_g, T* __restrict__ vel_g, T* __restrict__ obs_g,
                                                            T* __restrict__ rew_g, uint8_t* __restrict__ done_g, int* __restrict__ info_i_g,
                                                            T* __restrict__ info_f_g, T* __restrict__ obs0_g, int n_steps) {
    static_assert(sizeof(T) == 4, "16-byte vectors of two float pairs");
    constexpr int SLOTS = MAX_E + MPE_MAX_AGENTS + 1, ZERO_SLOT = SLOTS - 1;
    __shared__ V2<T> PVb[2][SLOTS];     // entity positions, agent velocities, one pair of zeros: states j even / odd
    __shared__ T size_s[MAX_E];
    __shared__ int coll_idx[MAX_E];
    __shared__ T mind_s[MPE_MAX_LANDMARKS];
    __shared__ int n_coll_s[2];
    __shared__ __attribute__((aligned(16))) T CP[MAX_E][4];
    constexpr int ACT_CHUNK = 32;
    __shared__ uint8_t act_s[ACT_CHUNK][MPE_MAX_AGENTS];     // see wave_kernel

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int A = p.A, L = p.L, E = A + L;
    const long long B = p.n_envs;
    long long b = blockIdx.x;
    if (p.deal.phase != nullptr) {
        b = xcd_unit(p.deal, blockIdx.x);
        if (b < 0) return;
    }
    const DevTable* tb = p.table;
    const int OD = 4 + 2 * L + 2 * (A - 1);
    const int P = (rs.enabled && rs.period > 0) ? rs.period : n_steps;
    const int K = (n_steps + P - 1) / P;
    const int J = n_steps + K - 1;      // index of the last state
    const bool emit0 = rs.enabled && obs0_g != nullptr;

    // Scenario.reset_world (same draws as reset_kernel) into one copy of the table; the landmarks, which no step writes
    // back, go to pos_g at once
    auto draw_world = [&](V2<T>* S, const ResetArgs& r, int first, int stride) {
        for (int e = first; e < E; e += stride) {
            const V2<double> d = reset_draw(r, (unsigned long long)b, e, e < A);
            S[e].x = (T)d.x;
            S[e].y = (T)d.y;
            if (e >= A) {
                pos_g[b * (2 * E) + 2 * e] = S[e].x;
                pos_g[b * (2 * E) + 2 * e + 1] = S[e].y;
            }
        }
        for (int a = first; a < A; a += stride) { S[MAX_E + a].x = T(0); S[MAX_E + a].y = T(0); }
    };
    // ---- stage the world (both waves) ----
    for (int e = tid; e < E; e += NT) size_s[e] = (T)tb->size[e];
    if (tid < 2) { PVb[tid][ZERO_SLOT].x = T(0); PVb[tid][ZERO_SLOT].y = T(0); }
    if (rs.enabled) {
        draw_world(PVb[0], rs, tid, NT);
    } else {
        for (int e = tid; e < E; e += NT) {
            PVb[0][e].x = pos_g[b * (2 * E) + 2 * e];
            PVb[0][e].y = pos_g[b * (2 * E) + 2 * e + 1];
        }
        if (tid < A) {
            PVb[0][MAX_E + tid].x = vel_g[b * (2 * A) + 2 * tid];
            PVb[0][MAX_E + tid].y = vel_g[b * (2 * A) + 2 * tid + 1];
        }
    }
    // the list of colliding entities, ascending (thread e speaks for entity e, as in wave_kernel)
    const bool listed = tid < E && tb->collide[tid] != 0;
    const unsigned long long wave_listed = __ballot(listed);
    if (lane == 0) n_coll_s[wave] = __popcll(wave_listed);
    __syncthreads();
    const int first_wave_count = n_coll_s[0];
    const int n_coll = first_wave_count + n_coll_s[1];
    const int list_pos = (wave == 0 ? 0 : first_wave_count) + __popcll(wave_listed & ((1ull << lane) - 1ull));
    if (listed) coll_idx[list_pos] = tid;
    __syncthreads();
    for (int k = tid; k < n_coll; k += NT) {
        const int j = coll_idx[k];
        CP[k][0] = PVb[0][j].x; CP[k][1] = PVb[0][j].y; CP[k][2] = size_s[j];
        CP[k][3] = T(0);        // masses matter to new_core worlds only
    }
    __syncthreads();

    if (wave == 1) {
        // ================= the streaming wave =================
        // vector k of a lane covers pairs 2q and 2q + 1, q = 64 k + lane: its four table indices, a byte each
        const int n_quads = (A * (OD / 2)) / 2;
        // (row and place of a pair advance by 128 pairs from one vector of a lane to its next: two divisions in all)
        unsigned quad_tab[STREAM_QUADS_PER_LANE];
        {
            const int RP = OD / 2;
            const int step_i = (int)(128.5f / (float)RP), step_r = 128 - step_i * RP;
            int i = (int)(((float)(2 * lane) + 0.5f) / (float)RP), r = 2 * lane - i * RP;
#pragma unroll
            for (int k = 0; k < STREAM_QUADS_PER_LANE; ++k) {
                const bool wrap = r + 1 == RP;      // the vector's second pair begins the next row
                quad_tab[k] = (k * 64 + lane < n_quads)
                                  ? (spread_row_entry(i, r, A, L, ZERO_SLOT) | (spread_row_entry(wrap ? i + 1 : i, wrap ? 0 : r + 1, A, L, ZERO_SLOT) << 16))
                                  : 0u;
                i += step_i; r += step_r;
                if (r >= RP) { r -= RP; ++i; }
            }
        }
        typedef T T4 __attribute__((ext_vector_type(4)));
        int e = 0, k = 0;       // state j = (episode e, k)
        for (int j = 0; j <= J; ++j) {
            T* dest = nullptr;
            if (k == 0) {
                if (emit0) dest = obs0_g + ((long long)e * B + b) * (long long)(A * OD);
            } else {
                dest = obs_g + ((long long)(e * P + k - 1) * B + b) * (long long)(A * OD);
            }
            SSTAMP(j, 0);
            if (dest != nullptr) {
                const V2<T>* S = PVb[j & 1];
                T4* dst = reinterpret_cast<T4*>(dest) + lane;
#pragma unroll
                for (int kq = 0; kq < STREAM_QUADS_PER_LANE; ++kq) {
                    if (kq * 64 >= n_quads) break;
                    if (kq * 64 + lane < n_quads) {
                        unsigned en = quad_tab[kq];
                        asm volatile("" : "+v"(en));        // unpacked here, every time: hoisted out of the loop the addresses are 260 registers
                        const V2<T> a0 = S[en & 255u], b0 = S[(en >> 8) & 255u], a1 = S[(en >> 16) & 255u], b1 = S[en >> 24];
                        T4 w;
                        w.x = spread_round<T>(a0.x - b0.x);
                        w.y = spread_round<T>(a0.y - b0.y);
                        w.z = spread_round<T>(a1.x - b1.x);
                        w.w = spread_round<T>(a1.y - b1.y);
                        dst[kq * 64] = w;
                    }
                    if ((kq & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // four vectors' LDS reads in flight at a time
                }
            }
            SSTAMP(j, 3);
            __syncthreads();
            if (++k > P) { k = 0; ++e; }
        }
        return;
    }

    // ================= the stepping wave: lane i is agent i =================
    const bool is_agent = lane < A;
    const int me = is_agent ? lane : 0;
    const int my_rank = (is_agent && listed) ? list_pos : -1;
    const T cut = T(18) * p.contact_margin;      // float: softplus is exactly 0 beyond 16.7 margins (see wave_kernel)
    const bool i_collide = tb->collide[me] != 0;
    const bool i_move = tb->movable[me] != 0;
    const T my_size = size_s[me];
    const T my_mass = (T)tb->mass[me];
    const T my_sens = (T)tb->sens[me];
    const T my_max_speed = (T)tb->max_speed[me];
    const bool want_info = (info_i_g != nullptr);

    int e = 0, k = 1;       // the state being produced, j + 1 = (episode e, k)
    for (int j = 0; j < J; ++j) {
        const V2<T>* src = PVb[j & 1];
        V2<T>* dst = PVb[(j + 1) & 1];
        SSTAMP(j, 0);
        if (k == 0) {
            // ---- the next episode's reset_world ----
            ResetArgs nr = rs;
            nr.episode = rs.episode + (unsigned long long)e;
            draw_world(dst, nr, lane, 64);
            __builtin_amdgcn_wave_barrier();
            for (int kk = lane; kk < n_coll; kk += 64) {
                const int jj = coll_idx[kk];
                CP[kk][0] = dst[jj].x; CP[kk][1] = dst[jj].y;
            }
        } else {
            const int t = e * P + k - 1;
            const long long row = (long long)t * B + b;
            if (k == 1)     // the landmarks of this episode, for the other copy of the table
                for (int en = A + lane; en < E; en += 64) dst[en] = src[en];
            if ((t % ACT_CHUNK) == 0 && is_agent) {
                const int n = n_steps - t < ACT_CHUNK ? n_steps - t : ACT_CHUNK;
                for (int s0 = 0; s0 < n; s0 += 16) {
                    int av[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) av[q] = (s0 + q < n) ? actions[((long long)(t + s0 + q) * B + b) * A + lane] : 0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) act_s[s0 + q][lane] = (uint8_t)((unsigned)av[q] <= 4u ? av[q] : 0);
                }
            }
            // ---- World.step for agent `lane` (core.py:150-204) ----
            V2<T> pi = src[me], vi = src[MAX_E + me];
            if (is_agent && i_move) {
                V2<T> f = decode_move<T>(act_s[t % ACT_CHUNK][lane], my_sens);      // environment.py:176-195
                if (i_collide) f = contact_sum<T>(CP, n_coll, my_rank, pi, my_size, my_mass, cut, p.contact_force, p.contact_margin, false, f);
                integrate<T, CLAMP>(pi, vi, f, my_mass, my_max_speed, p.one_minus_damping, p.dt);
            }
            // one wave: its LDS accesses are carried out in program order, every lane has read the old CP by now
            __builtin_amdgcn_wave_barrier();
            if (is_agent) {
                dst[lane] = pi; dst[MAX_E + lane] = vi;
                if (my_rank >= 0) { CP[my_rank][0] = pi.x; CP[my_rank][1] = pi.y; }
            }
            __builtin_amdgcn_wave_barrier();
            SSTAMP(j, 1);
            // done_callback (environment.py:98,110-113,143-146), any(done_n): an agent beyond the bound
            bool outside = false;
            if (p.done_bound > T(0))
                outside = __ballot(is_agent && (t_abs<T>(pi.x) > p.done_bound || t_abs<T>(pi.y) > p.done_bound)) != 0ull;
            // ---- reward() simple_spread.py:83-108 and benchmark_data() :58-74 on the new positions ----
            if (lane < L) {
                const V2<T> pl = dst[A + lane];
                T m = sqdist<T>(dst[0].x - pl.x, dst[0].y - pl.y);      // min of the squared distances, rooted once
#pragma unroll 8
                for (int a = 1; a < A; ++a) m = py_min<T>(m, sqdist<T>(dst[a].x - pl.x, dst[a].y - pl.y));
                mind_s[lane] = root<T>(m);
            }
            const int c = is_agent ? count_collisions<T>(dst, size_s, A, lane, my_size) : 0;
            __builtin_amdgcn_wave_barrier();
            if (is_agent) {
                T prox = T(0), min_dists = T(0);
                int occupied = 0;
#pragma unroll 8
                for (int l = 0; l < L; ++l) {
                    const T m = mind_s[l];
                    prox = prox - m;
                    min_dists = min_dists + m;
                    occupied += (m < T(0.1)) ? 1 : 0;
                }
                T bench = prox;
                for (int n = 0; n < c; ++n) bench = bench - T(1);      // rew -= 1 per collision (:106)
                rew_g[row * A + lane] = (lane == 0 ? prox : T(0)) + (T)(-c);
                if (want_info) {
                    info_i_g[row * (A + 1) + lane] = c;
                    if (info_f_g) info_f_g[row * (A + 1) + 1 + lane] = bench;
                    if (lane == 0) {
                        info_i_g[row * (A + 1) + A] = occupied;
                        if (info_f_g) info_f_g[row * (A + 1)] = min_dists;
                    }
                }
            }
            // done: a world's own flag when a bound is set; otherwise always 0, written in whole 128-byte lines by the
            // world that holds the first flag of a 128-flag block
            if (p.done_bound > T(0)) {
                if (lane == 0) done_g[row] = outside ? 1 : 0;
            } else if ((b & 127) == 0) {
                uint8_t* d = done_g + row;
                const long long n = B - b < 128 ? B - b : 128;
                if ((((unsigned long long)(uintptr_t)d) & 3u) == 0 && n == 128) {
                    if (lane < 32) reinterpret_cast<unsigned*>(d)[lane] = 0u;
                } else {
                    for (int q = lane; q < n; q += 64) d[q] = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();        // mind_s is rewritten by the next step
            SSTAMP(j, 2);
        }
        __syncthreads();
        if (++k > P) { k = 0; ++e; }
    }
    __syncthreads();        // the streaming wave writes the last state's observation
    if (is_agent) {
        const V2<T>* S = PVb[J & 1];
        pos_g[b * (2 * E) + 2 * lane] = S[lane].x;
        pos_g[b * (2 * E) + 2 * lane + 1] = S[lane].y;
        vel_g[b * (2 * A) + 2 * lane] = S[MAX_E + lane].x;
        vel_g[b * (2 * A) + 2 * lane + 1] = S[MAX_E + lane].y;
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
    p.deal = XcdDeal{nullptr, {0, 0}, B};
    dim3 grid((unsigned)B), block(NT);
    if constexpr (EPI == EPI_SPREAD && MODE == WMODE_STEP && sizeof(T) == 4) {
        // a world's block is A * (A + L + 1) coordinate pairs: 16-byte vectors when that is even and the arrays are aligned
        const bool aligned = (((uintptr_t)a.obs | (uintptr_t)a.obs0) & 15u) == 0;
        const bool quads = aligned && ((s.A * (s.A + s.L + 1)) & 1) == 0;
#ifndef MPE_WAVE_NO_STREAM   // (defined in diagnostic builds: every launch on wave_kernel)
        // launches of several steps: one wave steps, the other streams (wave_stream_kernel)
        if (quads && !a.noise.enabled && a.au.step_count == nullptr && a.n_steps > 1) {
            // A launch that streams large blocks ends with the odd XCDs (XcdDeal, mpe_launch.hpp): the worlds are dealt about
            // 9 : 7 between even and odd XCDs, in whole worlds per CU on both sides (2 048 worlds of 64 + 64: 8 : 8 per CU
            // 0.698 ms, 8.5 : 7.5 0.689, 9 : 7 0.664, 9.25 : 6.75 0.696; 1 536 worlds 0.559 -> 0.513 ms at 7 : 5; 4 096 worlds
            // 1.402 -> 1.323 ms at 18 : 14); below six worlds per CU a split costs more than it gains (1 024 worlds:
            // 0.369 -> 0.382 ms), and small blocks are not bound by their stores.
            if ((B + 255) / 256 >= 6 && (long long)s.A * (s.A + s.L + 1) * 2 * (long long)sizeof(T) >= 32768) {
                p.deal = make_xcd_deal(a.xcd_phase, B, 32);
                if (p.deal.phase != nullptr) {
                    grid = dim3((unsigned)(8 * p.deal.share[0]));
                    const hipError_t e = launch_xcd_phase(a.xcd_phase, st);
                    if (e != hipSuccess) return e;
                }
            }
            if (s.any_speed_limit())
                hipLaunchKernelGGL((wave_stream_kernel<T, true>), grid, block, 0, st, p, a.reset, a.actions, (T*)a.pos, (T*)a.vel, (T*)a.obs, (T*)a.rew,
                                   a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.n_steps);
            else
                hipLaunchKernelGGL((wave_stream_kernel<T, false>), grid, block, 0, st, p, a.reset, a.actions, (T*)a.pos, (T*)a.vel, (T*)a.obs, (T*)a.rew,
                                   a.done, a.info_i, (T*)a.info_f, (T*)a.obs0, a.n_steps);
            return hipGetLastError();
        }
#endif
    }
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
