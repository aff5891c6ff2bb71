// Fused step of the consensus scenario (multiagent/scenarios/consensus.py on multiagent/new_core.py):
// six non-colliding agents in the walled arena whose observations are the last ten positions and
// velocities of their two ring neighbours relative to their own.
//
// One thread owns one agent of one world, a workgroup of 192 threads 32 worlds.  The history (240 values per
// world) lives between launches in device memory as a ring of ten slots,
// hist T[(slot * A + agent) * 4 + {px, py, vx, vy}][B], whose oldest slot is aux[0][b]: a step
// overwrites the oldest slot with the new state and advances the head, which is what
// World.integrate_state's vstack([his[1:], new]) does to the ordered history (new_core.py:252-254).
// Inside a launch a thread keeps what its agent observes -- the neighbours' history relative to its own --
// in registers and only appends to the ring; see consensus_kernel.
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
    T done_bound;       // > 0: a world is done when an agent leaves [-done_bound, done_bound]^2 (mpe_set_done_bound)
    XcdDeal deal;       // groups of 32 worlds dealt unevenly to even and odd XCDs (phase == nullptr: equal shares)
};

// neighbors(), consensus.py:79-83: the ring neighbours of agent i in ascending order
__host__ __device__ constexpr int nb_lo(int i) { return i == 0 ? 1 : (i == CA - 1 ? 0 : i - 1); }
__host__ __device__ constexpr int nb_hi(int i) { return i == 0 ? CA - 1 : (i == CA - 1 ? CA - 2 : i + 1); }

template <typename T> __device__ __forceinline__ T f32r(T v) { return (T)(float)v; }     // .astype(np.float32), consensus.py:140

constexpr int WPG = 32;          // worlds per workgroup
constexpr int NT = WPG * CA;     // its threads: thread = world * 6 + agent, three full waves
constexpr int XS = WPG + 1;      // row stride of the exchange area (odd: a wave's writes spread over the banks)
constexpr int ROW = CA * COBS;   // observation values of one world-step
#ifndef MPE_NT_STEP_DEFAULT
#define MPE_NT_STEP_DEFAULT true
#endif

// MODE 0: mpe_step / mpe_step_auto, 1: mpe_rollout / mpe_episode(s), 2: mpe_observe
// NOISE: agents with u_noise (its own instantiation: the generator is an out-of-line call)
//
// One thread owns one agent of one world, a workgroup of three waves 32 worlds: every store of a step then
// covers whole 128-byte lines -- 32 worlds' rewards are 768 bytes, one history row of 32 worlds is 128 bytes, a
// wave's 64 observation rows are 160 lines -- where ten worlds per wave left the rewards and the history slot in
// 40-byte pieces that cost a seventh of the launch.  What an agent observes -- its two ring neighbours' last ten
// velocities and positions relative to its own at the same instants (80 values) -- is the thread's register state
// R: a step shifts it by one instant and appends the new differences.  The threads of a world may sit in two waves,
// so they exchange their fresh position and velocity (and their reward terms) through LDS.  The history ring in
// device memory (the form the state has between launches) is read once at the start of a launch; a step stores
// its slot only if no later step of the launch overwrites it.  Each wave writes its lanes' 80 values into its own
// LDS tile (all 64 lanes in float32, 32 at a time in float64), which leaves as 16-byte vectors in the order of those
// 20 480 bytes.
template <typename T, int MODE, bool NOISE = false>
__global__ __launch_bounds__(NT) void consensus_kernel(const ConsensusParams<T> p, const ResetArgs rs, const AutoArgs au, const NoiseArgs nz, const int* __restrict__ actions,
                                                       T* __restrict__ pos_g, T* __restrict__ vel_g, T* __restrict__ hist_g,
                                                       int* __restrict__ aux_g, T* __restrict__ obs_g, T* __restrict__ rew_g,
                                                       uint8_t* __restrict__ done_g, T* __restrict__ obs0_g, int n_steps) {
    // Every lane's 80 values sit PAD apart: with the rows back to back (80 dwords) the lanes' 16-byte writes fall on
    // 2 of the 8 four-bank groups, a 32-way conflict; at 84 they spread over all of them.
    constexpr bool NT_STORES = MODE == 0 && MPE_NT_STEP_DEFAULT;     // a launch of one step stores non-temporally (mpe_math.hpp: put)
    constexpr int PAD = COBS + 16 / (int)sizeof(T);
    // agent-rows a wave's tile holds: all 64 of its lanes' rows in float32 (21 KB a wave, 66 KB a workgroup: two
    // workgroups per CU, which is what the registers allow anyway), half of them at a time in float64
    constexpr int TROWS = sizeof(T) == 4 ? 64 : 32;
    __shared__ __attribute__((aligned(32))) T tiles[NT / 64][TROWS * PAD];
    __shared__ T xch[CA * 4 * XS];       // [agent][px, py, vx, vy][world]: the newest state, also the history slot's image
    __shared__ T rsum[NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long B = p.n_envs;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: workgroups that share an XCD take a
    // contiguous range of the batch (a speed matter for the stores that are pieces of lines -- single steps' history
    // rows -- only: any placement gives the same result).
    const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    unsigned group = xcd * (G >> 3) + (xcd < (G & 7u) ? xcd : (G & 7u)) + idx;
    if (p.deal.phase != nullptr) {
        const long long unit = xcd_unit(p.deal, blockIdx.x);
        if (unit < 0) return;
        group = (unsigned)unit;
    }
    const long long env0 = (long long)group * WPG;
    const long long left = B - env0;
    const int nvalid = left < WPG ? (int)left : WPG;
    // the threads of worlds beyond the batch shadow the workgroup's first world: their loads are valid, they
    // compute what that world's threads compute, and they store nothing
    const int w_raw = tid / CA;
    const bool valid = w_raw < nvalid;
    const int w = valid ? w_raw : 0;
    const int a = tid - w_raw * CA;
    const long long b = env0 + w;
    const int a_lo = nb_lo(a), a_hi = nb_hi(a);

    auto hrow = [&](int slot, int c) -> long long { return ((long long)((slot * CA + a) * 4 + c)) * B + b; };

    V2<T> pos, vel;
    int head = 0;
    bool same_heads = (MODE == 1) && rs.enabled;       // every world reset by this launch at the same steps
    // R[g][k][d]: g 0 lower neighbour's velocity, 1 upper neighbour's velocity, 2 lower's position, 3 upper's
    // position, each minus the agent's own, at instant k (oldest first), rounded as the observation is
    // (.astype(np.float32), consensus.py:140): exactly the agent's 80 observation values, in order
    T R[4][CH][2];
    V2<T> dlo, dhi;             // unrounded offsets to the neighbours' current positions (reward)

    // the thread's four values into the exchange area; readers wait for the barrier that follows
    auto publish = [&](T px, T py, T vx, T vy) {
        if (valid) {
            xch[(a * 4 + 0) * XS + w] = px; xch[(a * 4 + 1) * XS + w] = py;
            xch[(a * 4 + 2) * XS + w] = vx; xch[(a * 4 + 3) * XS + w] = vy;
        }
    };
    // the neighbours' current state (published, barrier passed), and the differences the newest instant holds
    auto newest = [&](T (&out)[4][2]) {
        const T lpx = xch[(a_lo * 4 + 0) * XS + w], lpy = xch[(a_lo * 4 + 1) * XS + w];
        const T lvx = xch[(a_lo * 4 + 2) * XS + w], lvy = xch[(a_lo * 4 + 3) * XS + w];
        const T hpx = xch[(a_hi * 4 + 0) * XS + w], hpy = xch[(a_hi * 4 + 1) * XS + w];
        const T hvx = xch[(a_hi * 4 + 2) * XS + w], hvy = xch[(a_hi * 4 + 3) * XS + w];
        dlo.x = lpx - pos.x; dlo.y = lpy - pos.y;
        dhi.x = hpx - pos.x; dhi.y = hpy - pos.y;
        out[0][0] = f32r(lvx - vel.x); out[0][1] = f32r(lvy - vel.y);
        out[1][0] = f32r(hvx - vel.x); out[1][1] = f32r(hvy - vel.y);
        out[2][0] = f32r(dlo.x); out[2][1] = f32r(dlo.y);
        out[3][0] = f32r(dhi.x); out[3][1] = f32r(dhi.y);
    };
    // reset_world, consensus.py:46-59: agents U(-1,1)^2 at rest, every history row equal to the start.
    // Every thread of the workgroup must call it (it holds barriers); `mine` selects the worlds that are reset.
    // The ring in device memory is the state BETWEEN launches: slot k is rewritten by the k-th step after the
    // reset, so only the slots from `keep_from` on -- those the rest of the launch does not reach -- are stored.
    auto reset_world = [&](const ResetArgs& r, bool mine, int keep_from) {
        if (mine) {
            const V2<double> d = reset_draw(r, (unsigned long long)b, a, true);
            pos.x = (T)d.x; pos.y = (T)d.y;
            vel.x = T(0); vel.y = T(0);
            head = 0;
            if (valid) {
                for (int s = (keep_from > 0 ? keep_from : 0); s < CH; ++s) {
                    hist_g[hrow(s, 0)] = pos.x; hist_g[hrow(s, 1)] = pos.y;
                    hist_g[hrow(s, 2)] = T(0); hist_g[hrow(s, 3)] = T(0);
                }
            }
        }
        publish(pos.x, pos.y, vel.x, vel.y);
        __syncthreads();
        T nw[4][2];
        newest(nw);
        __syncthreads();        // the exchange area is rewritten by the next step
        if (mine) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < CH; ++k) { R[g][k][0] = nw[g][0]; R[g][k][1] = nw[g][1]; }
        }
    };

    // a single step's action is fetched with the state, not after it
    const unsigned act0 = (MODE == 0 && n_steps > 0) ? (unsigned)actions[b * CA + a] : 0u;
    if (MODE == 1 && rs.enabled) {
        reset_world(rs, true, n_steps);
    } else {
        pos.x = pos_g[(long long)(2 * a) * B + b]; pos.y = pos_g[(long long)(2 * a + 1) * B + b];
        vel.x = vel_g[(long long)(2 * a) * B + b]; vel.y = vel_g[(long long)(2 * a + 1) * B + b];
        head = aux_g[b];
        head = (head < 0 || head >= CH) ? 0 : head;
        // own history, oldest first, and from it and the neighbours' the relative history
        // Do the workgroup's worlds agree on the head (they do unless some were reset on their own)?  Then a step's
        // history slot leaves as whole lines, as in an episode launch.
        same_heads = __syncthreads_and(head == aux_g[env0]) != 0;
        // The threads exchange their histories through the observation tiles, idle until the first observation:
        // [instant][agent][px, py, vx, vy][world], all ten instants' loads in flight at once.
        T* const xarea = &tiles[0][0];
        constexpr int AXS = (CH * CA * 4 * XS <= (NT / 64) * TROWS * PAD) ? XS : WPG;       // float64: no room for the padding
        static_assert(CH * CA * 4 * AXS <= (NT / 64) * TROWS * PAD, "ten instants of 32 worlds must fit the tiles");
        if (same_heads) {
            // the ten slots are 240 rows of 32 consecutive worlds: whole lines, two per wave and instruction
#pragma unroll
            for (int i = 0; i < CH * CA * 4 * WPG / NT; ++i) {
                const int j = i * NT + tid, kk = j / (CA * 4 * WPG), q = (j - kk * (CA * 4 * WPG)) / WPG, wj = j % WPG;
                int slot = head + kk;
                slot = slot >= CH ? slot - CH : slot;
                xarea[(kk * CA * 4 + q) * AXS + wj] = hist_g[(long long)(slot * CA * 4 + q) * B + env0 + (wj < nvalid ? wj : 0)];
            }
        } else {
            T own[CH][4];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                int slot = head + k;
                slot = slot >= CH ? slot - CH : slot;
#pragma unroll
                for (int c = 0; c < 4; ++c) own[k][c] = hist_g[hrow(slot, c)];
            }
            if (valid) {
#pragma unroll
                for (int k = 0; k < CH; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xarea[((k * CA + a) * 4 + c) * AXS + w] = own[k][c];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const T own = xarea[((k * CA + a) * 4 + c) * AXS + w];
                const T lo = xarea[((k * CA + a_lo) * 4 + c) * AXS + w], hi = xarea[((k * CA + a_hi) * 4 + c) * AXS + w];
                const int g = (c < 2) ? 2 : 0;         // history rows hold (px, py, vx, vy)
                R[g][k][c & 1] = f32r(lo - own);
                R[g + 1][k][c & 1] = f32r(hi - own);
            }
        __syncthreads();
        publish(pos.x, pos.y, vel.x, vel.y);
        __syncthreads();
        dlo.x = xch[(a_lo * 4 + 0) * XS + w] - pos.x; dlo.y = xch[(a_lo * 4 + 1) * XS + w] - pos.y;
        dhi.x = xch[(a_hi * 4 + 0) * XS + w] - pos.x; dhi.y = xch[(a_hi * 4 + 1) * XS + w] - pos.y;
        __syncthreads();
    }

    // Episode launches with obs0: the observation of every freshly reset state (what MultiAgentEnv.reset returns,
    // environment.py:122-127) is produced by a pass through the loop body that skips the step and the rewards
    // ("pre") before the episode's first step, so the observation code exists once.
    const bool want_obs0 = (MODE == 1) && rs.enabled && obs0_g != nullptr;
    bool pre = want_obs0;
    int t = 0, last_reset = 0;
    // The agent's actions of the next NQ steps wait in a 4-bit queue (an int outside 1..4 is a no-op whatever its
    // value, environment.py:176-181, so ints above 15 are held as 15): memory operations retire in issue
    // order, and a load inside the step loop would wait for every store of the steps before it.
    constexpr int NQ = (MODE == 1) ? 32 : 1;
    unsigned long long q_lo = 0, q_hi = 0;
    int q_left = 0;
    auto refill = [&](int t0) {
        unsigned raw[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const long long tt = (t0 + k < n_steps) ? t0 + k : n_steps - 1;     // past the end: re-read the last step
            raw[k] = (unsigned)actions[(tt * B + b) * CA + a];
        }
        q_lo = q_hi = 0;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const unsigned long long v = raw[k] < 15u ? raw[k] : 15u;
            if (k < 16) q_lo |= v << (4 * k);
            else q_hi |= v << (4 * (k - 16));
        }
        q_left = NQ;
    };
    while (t < n_steps) {
        const long long row = (long long)t * B + env0;
        if (MODE == 1 && rs.enabled && rs.period > 0 && t % rs.period == 0 && t != last_reset) {
            // next episode of a launch of several (mpe_episodes)
            ResetArgs nr = rs;
            nr.episode = rs.episode + (unsigned long long)(t / rs.period);
            reset_world(nr, true, n_steps - t);
            last_reset = t;
            pre = want_obs0;
        }
        bool finished = false;
        if (MODE != 2 && !pre) {
            // ---- _set_action (environment.py:176-195) and World.step (new_core.py:188-205) ----
            int act;
            if (MODE == 0 && t == 0) {
                act = (int)(act0 < 15u ? act0 : 15u);
            } else {
                if (q_left == 0) refill(t);
                act = (int)(q_lo & 15u);
                q_lo = (q_lo >> 4) | (q_hi << 60);
                q_hi >>= 4;
                --q_left;
            }
            V2<T> f = decode_move<T>(act, p.sens);
            f.x = p.force_scale * f.x; f.y = p.force_scale * f.y;           // new_core.py:214
            if constexpr (NOISE) {                                          // + randn(2) * u_noise, new_core.py:213
                const V2<T> z = noise_pair<T>(nz, t, B, b, a);
                f.x = f.x + z.x * p.u_noise; f.y = f.y + z.y * p.u_noise;
            }
            // nothing collides (consensus.py:30); the walls act on every agent (:35-41, new_core.py:230-236)
#pragma unroll
            for (int wl = 0; wl < CW; ++wl) {
                const T perp = p.wall_vertical[wl] ? pos.x : pos.y;
                const T cut = (sizeof(T) == 4 ? T(18) : T(60)) * p.contact_margin;     // beyond it the softplus is 0 (float32) or < e^-60
                if (t_abs<T>(perp - p.wall_axis_pos[wl]) - (p.size + T(0.5) * p.wall_width[wl]) > cut) continue;
                V2<T> wf;
                if (wall_force_fast<T>(pos, p.size, p.ghost != 0, p.wall_vertical[wl] != 0, p.wall_hard[wl] != 0, p.wall_axis_pos[wl],
                                       p.wall_end0[wl], p.wall_end1[wl], p.wall_width[wl], p.contact_force, p.contact_margin, wf)) {
                    f.x = f.x + wf.x; f.y = f.y + wf.y;
                }
            }
            integrate<T, false>(pos, vel, f, p.mass, T(0), p.one_minus_damping, p.dt);
            publish(pos.x, pos.y, vel.x, vel.y);
            __syncthreads();
            // the new state replaces the oldest history slot (new_core.py:252-254); a slot that a later step of this
            // launch overwrites again is not stored ...
            if (t + CH >= n_steps) {
                if (same_heads) {
                    // the heads agree: the exchange area is the slot's image, [agent][px, py, vx, vy] x 32 worlds, and
                    // leaves as 24 whole lines
#pragma unroll
                    for (int i = 0; i < CA * 4 * WPG / NT; ++i) {
                        const int j = i * NT + tid, r = j / WPG, wj = j - r * WPG;
                        if (wj < nvalid) put<NT_STORES>(&hist_g[(long long)(head * CA * 4 + r) * B + env0 + wj], xch[r * XS + wj]);
                    }
                } else if (valid) {
                    hist_g[hrow(head, 0)] = pos.x; hist_g[hrow(head, 1)] = pos.y;
                    hist_g[hrow(head, 2)] = vel.x; hist_g[hrow(head, 3)] = vel.y;
                }
            }
            head = (head + 1 == CH) ? 0 : head + 1;
            // ... and the relative history moves on by one instant
            T nw[4][2];
            newest(nw);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k + 1 < CH; ++k) { R[g][k][0] = R[g][k + 1][0]; R[g][k][1] = R[g][k + 1][1]; }
                R[g][CH - 1][0] = nw[g][0]; R[g][CH - 1][1] = nw[g][1];
            }
        }

        // ---- done_callback (environment.py:98,110-113,143-146), any(done_n) over the world's agents: an agent beyond
        //      the bound.  The exchange area holds every agent's new position (published, barrier passed). ----
        bool outside = false;
        if (MODE != 2 && !pre && p.done_bound > T(0)) {
#pragma unroll
            for (int i = 0; i < CA; ++i)
                outside = outside || t_abs<T>(xch[(i * 4 + 0) * XS + w]) > p.done_bound || t_abs<T>(xch[(i * 4 + 1) * XS + w]) > p.done_bound;
        }
        // ---- reward(), consensus.py:91-113; shared by np.sum over the six agents (environment.py:103-105) ----
        const bool want_score = !pre && (MODE != 2 || rew_g != nullptr);
        if (want_score) {
            const T d0 = norm2<T>(dlo.x, dlo.y), d1 = norm2<T>(dhi.x, dhi.y);
            T r = T(0) - (t_abs<T>(d0 - T(0.8)) + t_abs<T>(d1 - T(0.8))) * T(0.1);
            const T r0 = r - t_abs<T>(norm2<T>(pos.x, pos.y) - T(0.8)) * T(0.1);
            r = (a == 0) ? r0 : r;
            r = r - T(2) * arena_bound<T>(t_abs<T>(pos.x));
            r = r - T(2) * arena_bound<T>(t_abs<T>(pos.y));
            rsum[tid] = r;
            __syncthreads();        // (also: every thread has read the exchange area of this step)
            T total = rsum[w_raw * CA];
#pragma unroll
            for (int i = 1; i < CA; ++i) total = total + rsum[w_raw * CA + i];
            const T out = (MODE == 2) ? r : total;        // observe: what the callback returns
            if (MODE == 0 && au.step_count != nullptr) {
                // ---- episode bookkeeping (mpe_step_auto): returns, step counter, time-limit reset ----
                T* ep = (T*)au.ep_return + b * CA;
                const int cnt = au.step_count[b] + 1;
                finished = cnt >= au.max_steps || outside;       // the time limit, or the world's own done
                const int epi = au.episode_count[b] + 1;
                // every thread of the world has read the counters before its first thread rewrites them
                __syncthreads();
                if (valid) {
                    const T sum = ep[a] + total;
                    ep[a] = finished ? T(0) : sum;
                    if (finished) ((T*)au.last_return)[b * CA + a] = sum;
                    if (a == 0) {
                        au.step_count[b] = finished ? 0 : cnt;
                        if (finished) au.episode_count[b] = epi;
                    }
                }
                // Scenario.reset_world with the next episode's draws for the worlds that finished; obs below is the fresh one
                reset_world(ResetArgs{1, au.seed, (unsigned long long)epi, au.landmark_lim, 0, au.env_base}, finished, 0);
            }
            if (valid) {
                put<NT_STORES>(&rew_g[row * CA + tid], out);        // 32 worlds x 6 rewards: six whole lines
                // no done callback (environment.py:143-146); mpe_step_auto flags the step that reaches its time limit
                if (MODE == 0 && a == 0) done_g[row + w] = (finished || outside) ? 1 : 0;
                if (MODE == 1 && a == 0 && p.done_bound > T(0)) done_g[row + w] = outside ? 1 : 0;
            }
            if (MODE == 1 && p.done_bound <= T(0) && (env0 & 127) == 0) {
                // An episode launch has no time limit: done is 0 throughout.  A workgroup's 32 flags are a quarter of
                // a 128-byte line: the workgroup that holds the first flag of a 128-flag block writes the whole block.
                uint8_t* d = done_g + (long long)t * B + env0;
                const long long n = B - env0 < 128 ? B - env0 : 128;
                if ((((unsigned long long)(uintptr_t)d) & 3u) == 0 && n == 128) {
                    if (tid < 32) reinterpret_cast<unsigned*>(d)[tid] = 0u;
                } else if (tid < n) {
                    d[tid] = 0;
                }
            }
        }

        // ---- observation(), consensus.py:122-140: the lanes' 80 values into the image of the wave's rows ----
        {
            typedef T V4 __attribute__((ext_vector_type(4)));
            constexpr int VEC = 16 / (int)sizeof(T), NVA = COBS / VEC, NVT = TROWS * NVA;      // vectors per agent-row, per tile
            typedef T VT __attribute__((ext_vector_type(VEC)));
            T* const my_tile = tiles[wave];
            T* const block = pre ? obs0_g + ((long long)(rs.period > 0 ? t / rs.period : 0) * B + env0) * ROW : obs_g + row * ROW;
            const int rows_valid = nvalid * CA;          // agent-rows of the workgroup that exist
#pragma unroll 1
            for (int half = 0; half < 64 / TROWS; ++half) {
                if ((lane / TROWS) == half) {
                    V4* mine = reinterpret_cast<V4*>(my_tile + (lane % TROWS) * PAD);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int k = 0; k < CH; k += 2) {
                            V4 v;
                            v[0] = R[g][k][0]; v[1] = R[g][k][1]; v[2] = R[g][k + 1][0]; v[3] = R[g][k + 1][1];
                            mine[(g * CH + k) / 2] = v;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // vector g of the tile's 32 contiguous agent-rows: values [VEC * j, + VEC) of agent-row g / NVA
                const int row0 = wave * 64 + half * TROWS;
                VT* dst = reinterpret_cast<VT*>(block + (long long)row0 * COBS);
#pragma unroll 5
                for (int i = 0; i < NVT / 64; ++i) {
                    const int g = i * 64 + lane, ar = g / NVA, j = g - ar * NVA;
                    if (row0 + ar < rows_valid) put<NT_STORES>(&dst[g], *reinterpret_cast<const VT*>(my_tile + ar * PAD + j * VEC));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (pre) pre = false;
        else ++t;
    }

    if (MODE != 2) {
        if (same_heads) {
            // The exchange area still holds what the last step (or reset) published, the final state as
            // [agent][px, py, vx, vy] x 32 worlds: rows 2a, 2a+1 of pos and of vel, whole lines.
#pragma unroll
            for (int i = 0; i < CA * 4 * WPG / NT; ++i) {
                const int j = i * NT + tid, r = j / WPG, wj = j - r * WPG, c = r & 3;
                T* const dst = (c < 2) ? pos_g : vel_g;
                if (wj < nvalid) put<NT_STORES>(&dst[(long long)(2 * (r >> 2) + (c & 1)) * B + env0 + wj], xch[r * XS + wj]);
            }
            // every world's own head: a time-limit reset inside the launch (mpe_step_auto) puts the worlds it resets
            // back to slot 0 while their neighbours advance, so the heads may differ by now
            if (valid && a == 0) aux_g[b] = head;
        } else if (valid) {
            pos_g[(long long)(2 * a) * B + b] = pos.x; pos_g[(long long)(2 * a + 1) * B + b] = pos.y;
            vel_g[(long long)(2 * a) * B + b] = vel.x; vel_g[(long long)(2 * a + 1) * B + b] = vel.y;
            if (a == 0) aux_g[b] = head;
        }
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
    p.done_bound = T(0);
    return p;
}

template <typename T>
hipError_t launch_consensus_t(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    auto p = make_consensus_params<T>(s, B);
    p.done_bound = (T)a.done_bound;
    dim3 grid((unsigned)((B + WPG - 1) / WPG)), block(NT);
    p.deal = XcdDeal{nullptr, {0, 0}, (long long)grid.x};
    // Launches of several steps run as four or more rounds of resident workgroups and end on the odd XCDs (XcdDeal,
    // mpe_launch.hpp): the groups of 32 worlds are dealt 9 : 7 between even and odd XCDs (65 536 worlds 0.670 -> 0.652 ms,
    // 262 144 worlds 2.599 -> 2.537 ms; 8.5 : 7.5 gains nothing).
    if (mode == 1 && grid.x >= 2048u) {
        p.deal = make_xcd_deal(a.xcd_phase, (long long)grid.x, 1);
        if (p.deal.phase != nullptr) {
            grid = dim3((unsigned)(8 * p.deal.share[0]));
            const hipError_t e = launch_xcd_phase(a.xcd_phase, st);
            if (e != hipSuccess) return e;
        }
    }
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
                                  unsigned long long env_base, const uint8_t* mask, void* pos, void* vel, void* hist, int32_t* aux, hipStream_t st) {
    const ResetArgs rs{1, seed, episode, 1.0, 0, env_base};
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    if (dtype == MPE_F64)
        hipLaunchKernelGGL(consensus_reset_kernel<double>, grid, block, 0, st, B, rs, mask, (double*)pos, (double*)vel, (double*)hist, aux);
    else
        hipLaunchKernelGGL(consensus_reset_kernel<float>, grid, block, 0, st, B, rs, mask, (float*)pos, (float*)vel, (float*)hist, aux);
    return hipGetLastError();
}

}  // namespace mpe
