// Compile-time descriptions of the small in-register scenarios and their callbacks
// (observation / reward / benchmark_data), evaluated by one lane for one world.
//
// Every loop below has constant bounds and is fully unrolled, so pos[]/vel[] live in registers and
// the observation row is written to LDS with constant offsets.  The structure of a scenario (who
// collides, who is an adversary, which landmark is a forest) is fixed here; magnitudes (sizes,
// accelerations, speed limits, masses, world constants) arrive at run time in SmallParams.
#pragma once
#include "mpe_math.hpp"

namespace mpe {

constexpr int SMALL_MAX_WALLS = 4;

#ifndef MPE_NT_STEP_DEFAULT
#define MPE_NT_STEP_DEFAULT true       // (a diagnostic build may switch the single steps' non-temporal stores off: -DMPE_NT_STEP_DEFAULT=false)
#endif

template <typename T, int E, int A>
struct SmallParams {
    T size[E];
    T mass[A];
    T sens[A];
    T max_speed[A];
    T dt, one_minus_damping, contact_force, contact_margin;
    long long n_envs;
    unsigned long long env_base;                // global index of the batch's first env (counter of the respawn draws)
    T done_bound;                               // > 0: an env is done when an agent leaves [-done_bound, done_bound]^2; 0: never
    int skip_post_step;                         // scenarios with a post_step: 1 = the step ends after the callbacks (post_step_callback=None)
    XcdDeal deal;        // tiles dealt unevenly to even and odd XCDs (phase == nullptr: workgroup i is tile i)
    // read by scenarios on the new_core physics with walls only
    T force_scale[A];                           // mass (* accel): new_core.py:214
    T wall_axis_pos[SMALL_MAX_WALLS], wall_end0[SMALL_MAX_WALLS], wall_end1[SMALL_MAX_WALLS], wall_width[SMALL_MAX_WALLS];
    int wall_vertical[SMALL_MAX_WALLS], wall_hard[SMALL_MAX_WALLS], ghost[A];      // dwords: scalar loads cannot fetch single bytes of the kernel arguments
    double size_d[E];                           // float64 sizes for the reset's placement test
    T u_noise[A], c_noise[A];                   // Agent.u_noise / c_noise, 0 when None; read when the launch's NoiseArgs is enabled
};

// Defaults of a scenario description; a scenario overrides what differs.  The constexpr helpers are force-inlined:
// left to the inliner's budget, rep() was seen compiled as a real call in the largest kernel, which turned every
// p.mass[rep(i)] into a run-time indexed vector load of the kernel arguments (and a wait for the wave's older stores).
struct CfgBase {
    static constexpr int BITS = 3;          // bits kept of every action int (moves 0..4, comm words 0..dim_c-1)
    static constexpr bool SHARED = false;   // World.collaborative: step returns the summed reward to every agent
    static constexpr int N_AUX = 0;         // int32 rows of scenario state: landmark indices reset_world draws ...
    static constexpr bool DYNAMIC = false;  // ... or state post_step rewrites every step (then landmarks may move too)
    static constexpr bool NEW_CORE = false; // World.step of multiagent/new_core.py (mass-scaled action, mass-ratio contacts)
    static constexpr int N_WALLS = 0;
    // > 1: the observation row is produced and streamed out in this many equal column chunks (one LDS
    // tile of a chunk's width instead of the whole row), through observe_prepare / observe_chunk<C>
    static constexpr int OBS_CHUNKS = 1;
    static constexpr int CHUNK_MAX = 0, CHUNK_MIN = 0;     // widest / narrowest chunk; chunk c covers columns [chunk_offset(c), + chunk_width(c))
    // OBS_LOOP: the chunks are produced by ONE copy of the code, observe_agent(i, ...), run in a loop over
    // the agents with the wave's positions and velocities staged in LDS (Staged below) so that they can be
    // indexed at run time; keeps a long per-agent observation routine out of the instruction stream A times
    static constexpr bool OBS_LOOP = false;
    // OBS_ROW_PASSES = R > 0: the observation is evaluated R times per step through observe_pass(..., row, mine);
    // in pass k the lanes of worlds [k * 64 / R, (k + 1) * 64 / R) write their whole rows into a tile of 64 / R
    // rows, which leaves as contiguous 16-byte vectors (for rows too long to hold 64 of them in LDS)
    static constexpr int OBS_ROW_PASSES = 0;
    // OBS_COOP: for the observation the wave regroups: in each of 8 passes its lanes are (8 worlds) x (A agents, A <= 8),
    //   every lane runs observe_agent(its agent, its pass's world) once, and the pass's 8 whole rows leave as
    //   contiguous 16-byte vectors.  The arithmetic is that of one observe_agent per agent and world, the stores are
    //   whole lines.  The world's state is staged in LDS (StagedRows); what observe_prepare returns and the aux words
    //   travel from the world's own lane by lane reads (ctx_from_lane).
    static constexpr bool OBS_COOP = false;
    // NT_STEP: a launch of ONE step writes its outputs with non-temporal stores (mpe_math.hpp: put).  Measured per
    // scenario (one launch per step under a hipGraph): 6.0 -> 5.1 us for simple_spread at 65 536 worlds and 6-18 % for the
    // other scenarios whose row leaves in one flush; the scenarios of the cooperative passes, whose stores are spread
    // over the step, lose with them (collector 44 -> 52 us, simple_predator_prey 54 -> 60 us) and keep plain stores.
    static constexpr bool NT_STEP = MPE_NT_STEP_DEFAULT;
    // stage_packed(): the cooperative passes' staging area without its padding word (row stride 2E + 2A, a multiple of 4
    //   values), written as 16-byte vectors; for a scenario that needs the 256 bytes to fit eight workgroups a CU
    template <typename T> __host__ __device__ static constexpr bool stage_packed() { return false; }
    // Occupancy the compiler is to allow for in episode launches of full tiles (waves per SIMD, i.e. a register budget
    // of 512 / waves_per_simd(); the guarded instantiations for ragged tails, odd batch sizes and noise, and those with
    // benchmark outputs, keep the whole register file: held to half of it they spill inside the steps), and the
    // action words a lane holds at once between a fetch and its packing in an episode launch (whole steps; more words
    // mean fewer mid-launch fetches, each of which waits for the wave's older stores, but more registers)
    template <typename T> __host__ __device__ static constexpr int waves_per_simd() { return 1; }
    template <typename T> __host__ __device__ static constexpr int action_words() { return 84; }
    template <class Ctx> __device__ __forceinline__ static Ctx ctx_from_lane(const Ctx& c, int) { return c; }
    __host__ __device__ __forceinline__ static constexpr bool movable(int) { return true; }    // Agent.movable
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }    // not Agent.silent
    // Entity whose magnitudes (size, mass, sensitivity, speed limit ...) entity e shares: a scenario whose
    // roles have uniform properties names one representative per role, so that the kernel reads a handful
    // of parameters instead of one set per entity (mpe_create checks that the spec is uniform per role).
    __host__ __device__ __forceinline__ static constexpr int rep(int e) { return e; }
    // envs per wave: fewer when a 64-row observation tile does not fit the LDS
    template <typename T> __host__ __device__ __forceinline__ static constexpr int epw() { return 64; }
    // hooks of scenarios with DYNAMIC state
    template <class RS> __host__ __device__ static int initial_aux(const RS&, unsigned long long, int) { return 0; }
    template <class RS, class Pos, class P> __host__ __device__ static void fix_reset(const RS&, unsigned long long, Pos&, const P&) {}
    template <class Pos, class P> __device__ static void post_step(Pos&, int*, const P&, unsigned long long) {}
};

// A wave's entity positions and agent velocities in LDS, one column per lane (value v of lane l at
// v * 64 + l: conflict-free), for code that indexes entities at run time.
template <typename T, int E, int A>
struct Staged {
    static constexpr int SIZE = (2 * E + 2 * A) * 64;
    T* base;
    int lane;
    __device__ __forceinline__ V2<T> pos(int e) const { V2<T> v; v.x = base[(2 * e) * 64 + lane]; v.y = base[(2 * e + 1) * 64 + lane]; return v; }
    __device__ __forceinline__ V2<T> vel(int i) const { V2<T> v; v.x = base[(2 * E + 2 * i) * 64 + lane]; v.y = base[(2 * E + 2 * i + 1) * 64 + lane]; return v; }
    __device__ __forceinline__ void put(const V2<T> (&p)[E], const V2<T> (&w)[A]) const {
#pragma unroll
        for (int e = 0; e < E; ++e) { base[(2 * e) * 64 + lane] = p[e].x; base[(2 * e + 1) * 64 + lane] = p[e].y; }
#pragma unroll
        for (int i = 0; i < A; ++i) { base[(2 * E + 2 * i) * 64 + lane] = w[i].x; base[(2 * E + 2 * i + 1) * 64 + lane] = w[i].y; }
    }
};

// The same staging world-major: world (lane) l keeps its values at l * STRIDE + v.  STRIDE is odd, so that lanes
// reading different values of ONE world (the cooperative observation passes) and lanes writing their own world are
// both conflict-free; PACKED drops the padding word when 2E + 2A is a multiple of the 16-byte vector and writes the
// row as vectors (8 consecutive lanes' vectors of a 40-word stride fall on distinct banks; the passes' reads -- 8
// worlds x at most 2 columns per instruction -- stay conflict-free except the agents' own positions, 2-way).
template <typename T, int E, int A, bool PACKED = false>
struct StagedRows {
    static constexpr int N = 2 * E + 2 * A, VEC = 16 / (int)sizeof(T);
    static_assert(!PACKED || N % VEC == 0, "a packed row is a whole number of 16-byte vectors");
    static constexpr int STRIDE = PACKED ? N : (N | 1);
    static constexpr int SIZE = STRIDE * 64;
    T* base;
    int lane;       // the world's column
    __device__ __forceinline__ V2<T> pos(int e) const { V2<T> v; v.x = base[lane * STRIDE + 2 * e]; v.y = base[lane * STRIDE + 2 * e + 1]; return v; }
    __device__ __forceinline__ V2<T> vel(int i) const { V2<T> v; v.x = base[lane * STRIDE + 2 * E + 2 * i]; v.y = base[lane * STRIDE + 2 * E + 2 * i + 1]; return v; }
    __device__ __forceinline__ void put(const V2<T> (&p)[E], const V2<T> (&w)[A]) const {
        if constexpr (PACKED) {
            typedef T VT __attribute__((ext_vector_type(VEC)));
            T flat[N];
#pragma unroll
            for (int e = 0; e < E; ++e) { flat[2 * e] = p[e].x; flat[2 * e + 1] = p[e].y; }
#pragma unroll
            for (int i = 0; i < A; ++i) { flat[2 * E + 2 * i] = w[i].x; flat[2 * E + 2 * i + 1] = w[i].y; }
            VT* row = reinterpret_cast<VT*>(base + lane * STRIDE);
#pragma unroll
            for (int q = 0; q < N / VEC; ++q) {
                VT v;
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = flat[q * VEC + k];
                row[q] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) { base[lane * STRIDE + 2 * e] = p[e].x; base[lane * STRIDE + 2 * e + 1] = p[e].y; }
#pragma unroll
            for (int i = 0; i < A; ++i) { base[lane * STRIDE + 2 * E + 2 * i] = w[i].x; base[lane * STRIDE + 2 * E + 2 * i + 1] = w[i].y; }
        }
    }
};

// simple_spread casts its observation to float32 (simple_spread.py:130); the float64 kernel keeps
// that rounding so it can be compared bit for bit.
template <typename T> __device__ __forceinline__ T round_f32(T v) { return (T)(float)v; }

// ------------------------------------------------------------------------------------------
// simple_spread with A agents and L landmarks (multiagent/scenarios/simple_spread.py)
// ------------------------------------------------------------------------------------------
template <int A_, int L_>
struct SpreadCfg : CfgBase {
    static constexpr int A = A_, L = L_, E = A_ + L_;
    static constexpr int DIM_C = 0, S = 0, N_ACT = A_;
    static constexpr int OBS_DIM = 4 + 2 * L_ + 2 * (A_ - 1);
    static constexpr int OBS_TOTAL = A_ * OBS_DIM;
    static constexpr int N_AUX = 0;
    static constexpr int NI = A_ + 1;   // collisions per agent, occupied_landmarks
    static constexpr int NF = A_ + 1;   // min_dists, benchmark reward per agent
    // landmarks have collide = False (simple_spread.py:34)
    __host__ __device__ __forceinline__ static constexpr bool collides(int e) { return e < A_; }
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }

    // observation(), simple_spread.py:110-130
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int*, const P&, T* row) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            row[k++] = round_f32(vel[i].x); row[k++] = round_f32(vel[i].y);
            row[k++] = round_f32(pos[i].x); row[k++] = round_f32(pos[i].y);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                row[k++] = round_f32(pos[A + l].x - pos[i].x);
                row[k++] = round_f32(pos[A + l].y - pos[i].y);
            }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                row[k++] = round_f32(pos[j].x - pos[i].x);
                row[k++] = round_f32(pos[j].y - pos[i].y);
            }
        }
    }

    // reward() :83-108 and benchmark_data() :58-74
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int*, const P& p, T* rew, bool want_info, int* info_i, T* info_f) {
        T prox = T(0), min_dists = T(0);
        int occupied = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            T m = norm2<T>(pos[0].x - pos[A + l].x, pos[0].y - pos[A + l].y);
#pragma unroll
            for (int a = 1; a < A; ++a) m = py_min<T>(m, norm2<T>(pos[a].x - pos[A + l].x, pos[a].y - pos[A + l].y));
            prox = prox - m;
            min_dists = min_dists + m;
            occupied += (m < T(0.1)) ? 1 : 0;
        }
        bool hit[A][A];
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = i + 1; j < A; ++j) {
                hit[i][j] = touching<T>(pos[i], pos[j], p.size[i] + p.size[j]);
                hit[j][i] = hit[i][j];
            }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            int c = 0;
            T bench = prox;
#pragma unroll
            for (int a = 0; a < A; ++a) {
                if (a == i) continue;
                if (hit[a][i]) { c += 1; bench = bench - T(1); }
            }
            rew[i] = (i == 0 ? prox : T(0)) + (T)(-c);
            if (want_info) { info_i[i] = c; info_f[1 + i] = bench; }
        }
        if (want_info) { info_i[A] = occupied; info_f[0] = min_dists; }
    }
};

// ------------------------------------------------------------------------------------------
// simple_tag: 3 adversaries, 1 good agent, 2 obstacles (multiagent/scenarios/simple_tag.py)
// ------------------------------------------------------------------------------------------
struct TagCfg : CfgBase {
    static constexpr int N_ADV = 3, A = 4, L = 2, E = 6;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 4;
    static constexpr int OBS_ADV = 4 + 2 * L + 2 * (A - 1) + 2, OBS_GOOD = 4 + 2 * L + 2 * (A - 1);
    static constexpr int OBS_TOTAL = N_ADV * OBS_ADV + (A - N_ADV) * OBS_GOOD;
    static constexpr int NI = A, NF = 0, N_AUX = 0;
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return true; }   // simple_tag.py:19,30
    // a 25-step episode's 100 action words in one fetch (three shift registers): no refill in mid-launch
    template <typename T> __host__ __device__ static constexpr int action_words() { return 120; }
    template <typename T> __host__ __device__ static constexpr int waves_per_simd() { return sizeof(T) == 4 ? 2 : 1; }
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }
    __host__ __device__ __forceinline__ static constexpr bool adversary(int i) { return i < N_ADV; }

    // observation(), simple_tag.py:131-147
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int*, const P&, T* row) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            row[k++] = vel[i].x; row[k++] = vel[i].y;
            row[k++] = pos[i].x; row[k++] = pos[i].y;
#pragma unroll
            for (int l = 0; l < L; ++l) { row[k++] = pos[A + l].x - pos[i].x; row[k++] = pos[A + l].y - pos[i].y; }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                row[k++] = pos[j].x - pos[i].x; row[k++] = pos[j].y - pos[i].y;
            }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i || adversary(j)) continue;
                row[k++] = vel[j].x; row[k++] = vel[j].y;
            }
        }
    }

    // reward() :84-129 and benchmark_data() :57-66
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int*, const P& p, T* rew, bool want_info, int* info_i, T*) {
        bool hit[A][N_ADV];   // hit[g][a]: good agent g touches adversary a
        T adv_rew = T(0);
#pragma unroll
        for (int g = N_ADV; g < A; ++g)
#pragma unroll
            for (int a = 0; a < N_ADV; ++a) {
                hit[g][a] = touching<T>(pos[g], pos[a], p.size[g] + p.size[a]);
                if (hit[g][a]) adv_rew = adv_rew + T(10);
            }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if (adversary(i)) {
                rew[i] = adv_rew;
                if (want_info) {
                    int c = 0;
#pragma unroll
                    for (int g = N_ADV; g < A; ++g) c += hit[g][i] ? 1 : 0;
                    info_i[i] = c;
                }
            } else {
                T r = T(0);
#pragma unroll
                for (int a = 0; a < N_ADV; ++a) if (hit[i][a]) r = r - T(10);
                r = r - arena_bound<T>(t_abs<T>(pos[i].x));
                r = r - arena_bound<T>(t_abs<T>(pos[i].y));
                rew[i] = r;
                if (want_info) info_i[i] = 0;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// simple_world_comm: 4 adversaries (agent 0 leads and speaks), 2 good agents; landmarks
// [obstacle, food0, food1, forest0, forest1] (multiagent/scenarios/simple_world_comm.py:52-53)
// ------------------------------------------------------------------------------------------
struct WorldCommCfg : CfgBase {
    static constexpr int N_ADV = 4, A = 6, L = 5, E = 11;
    static constexpr int DIM_C = 4, S = 1, N_ACT = 7;
    static constexpr int FOOD0 = A + 1, N_FOOD = 2, FOREST0 = A + 3;
    static constexpr int OBS_ADV = 4 + 2 * L + 2 * (A - 1) + 2 * (A - N_ADV) + 2 + DIM_C;       // 34
    static constexpr int OBS_GOOD = 4 + 2 * L + 2 * (A - 1) + 2 + 2 * (A - N_ADV - 1);          // 28
    static constexpr int OBS_TOTAL = N_ADV * OBS_ADV + (A - N_ADV) * OBS_GOOD;                  // 192
    static constexpr int NI = A + 2, NF = 0, N_AUX = 0;
    // only agents and the obstacle collide (simple_world_comm.py:22,34,41,48)
    __host__ __device__ __forceinline__ static constexpr bool collides(int e) { return e <= A; }
    __host__ __device__ __forceinline__ static constexpr bool speaks(int i) { return i == 0; }
    __host__ __device__ __forceinline__ static constexpr bool adversary(int i) { return i < N_ADV; }
    __host__ __device__ __forceinline__ static constexpr bool leader(int i) { return i == 0; }

    template <typename T, class P>
    __device__ static __forceinline__ void forest_bits(const V2<T> (&pos)[E], const P& p, bool (&inf)[A][2]) {
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                inf[i][f] = touching<T>(pos[i], pos[FOREST0 + f], p.size[i] + p.size[FOREST0 + f]);
    }

    // visibility test of simple_world_comm.py:254
    __device__ static __forceinline__ bool sees(int i, bool i1, bool i2, bool o1, bool o2) {
        return (i1 && o1) || (i2 && o2) || (!i1 && !o1 && !i2 && !o2) || leader(i);
    }

    // The 192-value row leaves in three chunks of 64 columns (two full 128-byte lines of a float32 row each)
    // through a tile of a third of the size: a 64 x 193 float32 tile is 49 KB of LDS, room for three of these
    // one-wave workgroups per CU; two halves (25 KB) for six; a third (16.6 KB, 18 KB with the reward tile) for the
    // eight that two waves per SIMD need.  The episode kernel fits 256 registers with 6 steps' action words per fetch
    // (2 registers spilled) and runs two rounds of 2 048 resident waves at 262 144 worlds instead of four of 1 024.
    static constexpr int OBS_CHUNKS = 3, CHUNK_MAX = OBS_TOTAL / OBS_CHUNKS, CHUNK_MIN = OBS_TOTAL / OBS_CHUNKS;
    __host__ __device__ __forceinline__ static constexpr int chunk_width(int) { return OBS_TOTAL / OBS_CHUNKS; }
    __host__ __device__ __forceinline__ static constexpr int chunk_offset(int c) { return c * (OBS_TOTAL / OBS_CHUNKS); }
    template <typename T> __host__ __device__ static constexpr int waves_per_simd() { return sizeof(T) == 4 ? 2 : 1; }
    template <typename T> __host__ __device__ static constexpr int action_words() { return sizeof(T) == 4 ? 42 : 84; }
    __host__ __device__ __forceinline__ static constexpr int obs_offset(int i) { return i < N_ADV ? i * OBS_ADV : N_ADV * OBS_ADV + (i - N_ADV) * OBS_GOOD; }

    // in_forest bits of every agent (bit 2i+f), shared by the chunks of one step's observation
    template <typename T, class P>
    __device__ static __forceinline__ int observe_prepare(const V2<T> (&pos)[E], const P& p) {
        bool inf[A][2];
        forest_bits<T>(pos, p, inf);
        int bits = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) bits |= (inf[i][0] ? 1 : 0) << (2 * i) | (inf[i][1] ? 1 : 0) << (2 * i + 1);
        return bits;
    }

    // observation(), simple_world_comm.py:224-287: columns [LO, HI) of the world's row, written to row[0 .. HI - LO).
    // Every column index is a constant once the loops are unrolled, so values outside the window cost nothing.
    template <typename T, int LO, int HI>
    __device__ static __forceinline__ void observe_cols(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T* comm, int forest, T* row) {
        bool inf[A][2];
#pragma unroll
        for (int i = 0; i < A; ++i) { inf[i][0] = (forest >> (2 * i)) & 1; inf[i][1] = (forest >> (2 * i + 1)) & 1; }
        int k = 0;
        auto put = [&](T v) {
            if (k >= LO && k < HI) row[k - LO] = v;
            ++k;
        };
#pragma unroll
        for (int i = 0; i < A; ++i) {
            V2<T> me = pos[i];
            // A chunk is computed between the barriers that bracket its tile; pinning the agent's position keeps
            // the compiler from hoisting the second chunk's arithmetic (96 live values) above the first.
            // (float32 only: the float64 parity build does not need it, and a 64-bit operand pinned this way was
            // seen to come back corrupted in the instantiation that spills most.)
            if constexpr (sizeof(T) == 4)
                if (HI - LO < OBS_TOTAL && obs_offset(i) < HI && obs_offset(i) + (adversary(i) ? OBS_ADV : OBS_GOOD) > LO)
                    asm volatile("" : "+v"(me.x), "+v"(me.y));
            put(vel[i].x); put(vel[i].y);
            put(me.x); put(me.y);
#pragma unroll
            for (int l = 0; l < L; ++l) { put(pos[A + l].x - me.x); put(pos[A + l].y - me.y); }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                const bool vis = sees(i, inf[i][0], inf[i][1], inf[j][0], inf[j][1]);
                put(vis ? pos[j].x - me.x : T(0));
                put(vis ? pos[j].y - me.y : T(0));
            }
            if (!adversary(i)) {   // good agents put in_forest before other_vel (:287)
                put(inf[i][0] ? T(1) : T(-1));
                put(inf[i][1] ? T(1) : T(-1));
            }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i || adversary(j)) continue;
                const bool vis = sees(i, inf[i][0], inf[i][1], inf[j][0], inf[j][1]);
                put(vis ? vel[j].x : T(0));
                put(vis ? vel[j].y : T(0));
            }
            if (adversary(i)) {    // :281-285
                put(inf[i][0] ? T(1) : T(-1));
                put(inf[i][1] ? T(1) : T(-1));
#pragma unroll
                for (int c = 0; c < DIM_C; ++c) put(comm[c]);
            }
        }
    }

    template <typename T, int C>
    __device__ static __forceinline__ void observe_chunk(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T* comm, const int*, int forest, T* row) {
        observe_cols<T, C * (OBS_TOTAL / OBS_CHUNKS), (C + 1) * (OBS_TOTAL / OBS_CHUNKS)>(pos, vel, comm, forest, row);
    }

    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T* comm, const int*, const P& p, T* row) {
        observe_cols<T, 0, OBS_TOTAL>(pos, vel, comm, observe_prepare<T>(pos, p), row);
    }

    // reward() :142-198 and benchmark_data() :115-123; info also carries the integer masks.
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int*, const P& p, T* rew, bool want_info, int* info_i, T*) {
        bool hit[A][N_ADV];
#pragma unroll
        for (int g = N_ADV; g < A; ++g)
#pragma unroll
            for (int a = 0; a < N_ADV; ++a) hit[g][a] = touching<T>(pos[g], pos[a], p.size[g] + p.size[a]);
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if (adversary(i)) {
                T dmin = norm2<T>(pos[N_ADV].x - pos[i].x, pos[N_ADV].y - pos[i].y);
#pragma unroll
                for (int g = N_ADV + 1; g < A; ++g) dmin = py_min<T>(dmin, norm2<T>(pos[g].x - pos[i].x, pos[g].y - pos[i].y));
                T r = T(0) - T(0.1) * dmin;
#pragma unroll
                for (int g = N_ADV; g < A; ++g)
#pragma unroll
                    for (int a = 0; a < N_ADV; ++a) if (hit[g][a]) r = r + T(5);
                rew[i] = r;
                if (want_info) {
                    int c = 0;
#pragma unroll
                    for (int g = N_ADV; g < A; ++g) c += hit[g][i] ? 1 : 0;
                    info_i[i] = c;
                }
            } else {
                T r = T(0);
#pragma unroll
                for (int a = 0; a < N_ADV; ++a) if (hit[i][a]) r = r - T(5);
                r = r - T(2) * arena_bound<T>(t_abs<T>(pos[i].x));
                r = r - T(2) * arena_bound<T>(t_abs<T>(pos[i].y));
                T fmin = T(0);
#pragma unroll
                for (int f = 0; f < N_FOOD; ++f) {
                    const T d = norm2<T>(pos[FOOD0 + f].x - pos[i].x, pos[FOOD0 + f].y - pos[i].y);
                    // is_collision(agent, food) uses agent - food: the squares are identical
                    if (d < p.size[i] + p.size[FOOD0 + f]) r = r + T(2);
                    fmin = (f == 0) ? d : py_min<T>(fmin, d);
                }
                r = r + T(0.05) * fmin;
                rew[i] = r;
                if (want_info) info_i[i] = 0;
            }
        }
        if (want_info) {
            bool inf[A][2];
            forest_bits<T>(pos, p, inf);
            int fb = 0, vb = 0;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                fb |= (inf[i][0] ? 1 : 0) << (2 * i);
                fb |= (inf[i][1] ? 1 : 0) << (2 * i + 1);
#pragma unroll
                for (int j = 0; j < A; ++j)
                    if (j != i && sees(i, inf[i][0], inf[i][1], inf[j][0], inf[j][1])) vb |= 1 << ((A - 1) * i + (j > i ? j - 1 : j));
            }
            info_i[A] = fb;
            info_i[A + 1] = vb;
        }
    }
};

// Position of the goal landmark (agent.goal_a): a select chain keeps pos[] in registers.
template <typename T, int E, int A, int L>
__device__ __forceinline__ V2<T> goal_of(const V2<T> (&pos)[E], int goal) {
    V2<T> g = pos[A];
#pragma unroll
    for (int l = 1; l < L; ++l) {
        g.x = (goal == l) ? pos[A + l].x : g.x;
        g.y = (goal == l) ? pos[A + l].y : g.y;
    }
    return g;
}

// np.sum(np.square(d)) of a 2-vector
template <typename T> __device__ __forceinline__ T sq2(T dx, T dy) { return dx * dx + dy * dy; }

// ------------------------------------------------------------------------------------------
// simple: one agent, one landmark, nothing collides (multiagent/scenarios/simple.py)
// ------------------------------------------------------------------------------------------
struct SimpleCfg : CfgBase {
    static constexpr int A = 1, L = 1, E = 2;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 1;
    static constexpr int OBS_TOTAL = 2 + 2 * L;
    static constexpr int NI = 0, NF = 0, N_AUX = 0;    // no benchmark_data
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return false; }   // simple.py:12,18
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }

    // observation(), simple.py:45-50
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int*, const P&, T* row) {
        row[0] = vel[0].x; row[1] = vel[0].y;
        row[2] = pos[1].x - pos[0].x; row[3] = pos[1].y - pos[0].y;
    }

    // reward(), simple.py:41-43
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int*, const P&, T* rew, bool, int*, T*) {
        rew[0] = -sq2<T>(pos[0].x - pos[1].x, pos[0].y - pos[1].y);
    }
};

// ------------------------------------------------------------------------------------------
// simple_adversary: agent 0 is the adversary, 2 good agents, 2 landmarks of which aux[0] is the
// goal; nothing collides (multiagent/scenarios/simple_adversary.py)
// ------------------------------------------------------------------------------------------
struct AdversaryCfg : CfgBase {
    static constexpr int N_ADV = 1, A = 3, L = 2, E = 5;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 3;            // world.dim_c = 2 but every agent is silent
    static constexpr int OBS_ADV = 2 * L + 2 * (A - 1), OBS_GOOD = 2 + OBS_ADV;
    static constexpr int OBS_TOTAL = N_ADV * OBS_ADV + (A - N_ADV) * OBS_GOOD;
    static constexpr int NI = 1, NF = N_ADV + (A - N_ADV) * (L + 1), N_AUX = 1;
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return false; }   // simple_adversary.py:19,26
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }
    __host__ __device__ __forceinline__ static constexpr bool adversary(int i) { return i < N_ADV; }

    // observation(), simple_adversary.py:119-138
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&)[A], const T*, const int* aux, const P&, T* row) {
        const V2<T> g = goal_of<T, E, A, L>(pos, aux[0]);
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if (!adversary(i)) { row[k++] = g.x - pos[i].x; row[k++] = g.y - pos[i].y; }
#pragma unroll
            for (int l = 0; l < L; ++l) { row[k++] = pos[A + l].x - pos[i].x; row[k++] = pos[A + l].y - pos[i].y; }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                row[k++] = pos[j].x - pos[i].x; row[k++] = pos[j].y - pos[i].y;
            }
        }
    }

    // reward() :76-116 (shaped branches) and benchmark_data() :56-65
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P&, T* rew, bool want_info, int* info_i, T* info_f) {
        const V2<T> g = goal_of<T, E, A, L>(pos, aux[0]);
        T sq[A];
#pragma unroll
        for (int i = 0; i < A; ++i) sq[i] = sq2<T>(pos[i].x - g.x, pos[i].y - g.y);
        T adv_rew = T(0);                       // sum([...]) starts from int 0
#pragma unroll
        for (int a = 0; a < N_ADV; ++a) adv_rew = adv_rew + t_sqrt<T>(sq[a]);
        T dmin = t_sqrt<T>(sq[N_ADV]);
#pragma unroll
        for (int a = N_ADV + 1; a < A; ++a) dmin = py_min<T>(dmin, t_sqrt<T>(sq[a]));
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if (adversary(i)) {
                rew[i] = -sq[i];
                if (want_info) info_f[k++] = sq[i];
            } else {
                rew[i] = -dmin + adv_rew;
                if (want_info) {
#pragma unroll
                    for (int l = 0; l < L; ++l) info_f[k++] = sq2<T>(pos[i].x - pos[A + l].x, pos[i].y - pos[A + l].y);
                    info_f[k++] = sq[i];
                }
            }
        }
        if (want_info) info_i[0] = aux[0];
    }
};

// ------------------------------------------------------------------------------------------
// simple_push: agent 0 is the adversary, agent 1 the good agent (they collide), 2 landmarks of which
// aux[0] is the goal (multiagent/scenarios/simple_push.py)
// ------------------------------------------------------------------------------------------
struct PushCfg : CfgBase {
    static constexpr int N_ADV = 1, A = 2, L = 2, E = 4;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 2;
    static constexpr int OBS_ADV = 2 + 2 * L + 2 * (A - 1), OBS_GOOD = 2 + 2 + 3 + 2 * L + 3 * L + 2 * (A - 1);
    static constexpr int OBS_TOTAL = N_ADV * OBS_ADV + (A - N_ADV) * OBS_GOOD;
    static constexpr int NI = 0, NF = 0, N_AUX = 1;     // no benchmark_data
    __host__ __device__ __forceinline__ static constexpr bool collides(int e) { return e < A; }   // simple_push.py:17,28
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return false; }
    __host__ __device__ __forceinline__ static constexpr bool adversary(int i) { return i < N_ADV; }

    // observation(), simple_push.py:75-95.  Colours (reset_world :35-48): landmark l is 0.1 grey with
    // 0.8 added to channel l + 1; the good agent is 0.25 grey with 0.5 added to channel goal + 1.
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int* aux, const P&, T* row) {
        const int goal = aux[0];
        const V2<T> g = goal_of<T, E, A, L>(pos, goal);
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            row[k++] = vel[i].x; row[k++] = vel[i].y;
            if (!adversary(i)) {
                row[k++] = g.x - pos[i].x; row[k++] = g.y - pos[i].y;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) row[k++] = (ch == goal + 1) ? (T)(0.25 + 0.5) : (T)0.25;
            }
#pragma unroll
            for (int l = 0; l < L; ++l) { row[k++] = pos[A + l].x - pos[i].x; row[k++] = pos[A + l].y - pos[i].y; }
            if (!adversary(i)) {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) row[k++] = (ch == l + 1) ? (T)(0.1 + 0.8) : (T)0.1;
            }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                row[k++] = pos[j].x - pos[i].x; row[k++] = pos[j].y - pos[i].y;
            }
        }
    }

    // reward() :55-73
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P&, T* rew, bool, int*, T*) {
        const V2<T> g = goal_of<T, E, A, L>(pos, aux[0]);
        T d[A];
#pragma unroll
        for (int i = 0; i < A; ++i) d[i] = norm2<T>(pos[i].x - g.x, pos[i].y - g.y);
        T dmin = d[N_ADV];
#pragma unroll
        for (int a = N_ADV + 1; a < A; ++a) dmin = py_min<T>(dmin, d[a]);
#pragma unroll
        for (int i = 0; i < A; ++i) {
            // adversary: goal - agent, same squares as agent - goal
            rew[i] = adversary(i) ? dmin - d[i] : -d[i];
        }
    }
};

// ------------------------------------------------------------------------------------------
// simple_speaker_listener: agent 0 speaks (dim_c 3) and cannot move, agent 1 moves and is silent;
// aux[0] is the landmark the listener should reach (agents[0].goal_b); rewards are shared
// (multiagent/scenarios/simple_speaker_listener.py)
// ------------------------------------------------------------------------------------------
struct SpeakerListenerCfg : CfgBase {
    static constexpr int A = 2, L = 3, E = 5;
    static constexpr int DIM_C = 3, S = 1, N_ACT = 2;       // [speaker's word, listener's move]
    static constexpr int OBS_TOTAL = 3 + (2 + 2 * L + DIM_C);
    static constexpr int NI = 1, NF = A, N_AUX = 1;          // info: goal index; the agents' own rewards
    static constexpr bool SHARED = true;                     // :11
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return false; }   // :16,26
    __host__ __device__ __forceinline__ static constexpr bool movable(int i) { return i == 1; } // :19
    __host__ __device__ __forceinline__ static constexpr bool speaks(int i) { return i == 0; }  // :21

    // observation() :69-91.  Landmark l is 0.15 grey with channel l at 0.65 (reset_world :45-47).
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T* comm, const int* aux, const P&, T* row) {
        int k = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) row[k++] = (ch == aux[0]) ? (T)0.65 : (T)0.15;      // speaker: goal colour
        row[k++] = vel[1].x; row[k++] = vel[1].y;                                          // listener
#pragma unroll
        for (int l = 0; l < L; ++l) { row[k++] = pos[A + l].x - pos[1].x; row[k++] = pos[A + l].y - pos[1].y; }
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = comm[c];
    }

    // reward() :63-67: both agents get minus the squared listener-goal distance
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P&, T* rew, bool want_info, int* info_i, T* info_f) {
        const V2<T> g = goal_of<T, E, A, L>(pos, aux[0]);
        const T r = -sq2<T>(pos[1].x - g.x, pos[1].y - g.y);
        rew[0] = r; rew[1] = r;
        if (want_info) { info_i[0] = aux[0]; info_f[0] = r; info_f[1] = r; }
    }
};

// ------------------------------------------------------------------------------------------
// simple_reference: two agents that both move and speak (dim_c 10); aux[i] is the landmark agent i
// wants the other agent to reach (goal_b); rewards are shared (multiagent/scenarios/simple_reference.py)
// ------------------------------------------------------------------------------------------
struct ReferenceCfg : CfgBase {
    static constexpr int A = 2, L = 3, E = 5;
    static constexpr int DIM_C = 10, S = 2, N_ACT = 4;      // [move 0, word 0, move 1, word 1]
    static constexpr int BITS = 4;
    // a 25-step episode's 100 action words in one fetch (four shift registers of 8 steps): no refill in mid-launch
    template <typename T> __host__ __device__ static constexpr int action_words() { return 128; }
    static constexpr int OBS_DIM = 2 + 2 * L + 3 + DIM_C, OBS_TOTAL = A * OBS_DIM;
    static constexpr int NI = 2, NF = A, N_AUX = 2;
    static constexpr bool SHARED = true;                     // :10
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return false; }   // :15,20
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return true; }

    // observation() :61-80.  Landmark l is 0.25 grey with channel l at 0.75 (reset_world :40-42).
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T* comm, const int* aux, const P&, T* row) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            row[k++] = vel[i].x; row[k++] = vel[i].y;
#pragma unroll
            for (int l = 0; l < L; ++l) { row[k++] = pos[A + l].x - pos[i].x; row[k++] = pos[A + l].y - pos[i].y; }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) row[k++] = (ch == aux[i]) ? (T)0.75 : (T)0.25;
#pragma unroll
            for (int c = 0; c < DIM_C; ++c) row[k++] = comm[(1 - i) * DIM_C + c];
        }
    }

    // reward() :55-59: minus the squared distance of the other agent (goal_a) to this agent's landmark
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P&, T* rew, bool want_info, int* info_i, T* info_f) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const V2<T> g = goal_of<T, E, A, L>(pos, aux[i]);
            rew[i] = -sq2<T>(pos[1 - i].x - g.x, pos[1 - i].y - g.y);
            if (want_info) { info_i[i] = aux[i]; info_f[i] = rew[i]; }
        }
    }
};

// ------------------------------------------------------------------------------------------
// simple_crypto: adversary Eve (agent 0), listener Bob (1), speaker Alice (2); nobody moves,
// everybody speaks (dim_c 4); aux[0] is the goal landmark, aux[1] the landmark whose colour is
// Alice's key; landmark l's colour is the one-hot e_l (multiagent/scenarios/simple_crypto.py)
// ------------------------------------------------------------------------------------------
struct CryptoCfg : CfgBase {
    static constexpr int A = 3, L = 2, E = 5;
    static constexpr int DIM_C = 4, S = 3, N_ACT = 3;
    static constexpr int OBS_TOTAL = DIM_C + 2 * DIM_C + 2 * DIM_C;
    static constexpr int NI = 2, NF = A * 2 * DIM_C, N_AUX = 2;
    __host__ __device__ __forceinline__ static constexpr bool collides(int) { return false; }   // :32,40
    __host__ __device__ __forceinline__ static constexpr bool movable(int) { return false; }    // :35
    __host__ __device__ __forceinline__ static constexpr bool speaks(int) { return true; }
    static constexpr int SPEAKER = 2;                                           // :34

    // observation() :124-169: Eve hears Alice; Bob gets the key and hears Alice; Alice gets goal and key
    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&)[E], const V2<T> (&)[A], const T* comm, const int* aux, const P&, T* row) {
        int k = 0;
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = comm[SPEAKER * DIM_C + c];
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = (c == aux[1]) ? T(1) : T(0);
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = comm[SPEAKER * DIM_C + c];
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = (c == aux[0]) ? T(1) : T(0);
#pragma unroll
        for (int c = 0; c < DIM_C; ++c) row[k++] = (c == aux[1]) ? T(1) : T(0);
    }

    // reward() :94-121 and benchmark_data() :78-80 (per agent: state.c, then the goal colour)
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&)[E], const T* comm, const int* aux, const P&, T* rew, bool want_info, int* info_i, T* info_f) {
        T err[2];
        bool spoke[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {       // Eve and Bob
            err[i] = T(0);
            spoke[i] = false;
#pragma unroll
            for (int c = 0; c < DIM_C; ++c) {
                const T d = comm[i * DIM_C + c] - ((c == aux[0]) ? T(1) : T(0));
                err[i] = err[i] + d * d;
                spoke[i] = spoke[i] || (comm[i * DIM_C + c] != T(0));
            }
        }
        const T good_rew = spoke[1] ? T(0) - err[1] : T(0);
        const T adv_rew = spoke[0] ? T(0) + err[0] : T(0);
        rew[0] = spoke[0] ? T(0) - err[0] : T(0);
        rew[1] = adv_rew + good_rew;
        rew[2] = adv_rew + good_rew;
        if (want_info) {
            info_i[0] = aux[0]; info_i[1] = aux[1];
#pragma unroll
            for (int i = 0; i < A; ++i)
#pragma unroll
                for (int c = 0; c < DIM_C; ++c) {
                    info_f[i * 2 * DIM_C + c] = comm[i * DIM_C + c];
                    info_f[i * 2 * DIM_C + DIM_C + c] = (c == aux[0]) ? T(1) : T(0);
                }
        }
    }
};

// ------------------------------------------------------------------------------------------
// simple_predator_prey on the new_core physics: adversaries 0-2 (0 leads), good agents 3-6 (3 leads,
// 4-6 are the members that carry food to it); landmarks [barrier, food x3, forest x2]; 4 walls
// (multiagent/scenarios/simple_predator_prey.py).  aux[0] holds the flags post_step rewrites: bit f
// = food f is alive, bit 3+k = member k holds food; aux[1] / aux[2] are the key and the position of
// the world's respawn stream.  The scenario reads the distance cache World.step fills
// (new_core.py:147-170), i.e. distances of the current positions.
// ------------------------------------------------------------------------------------------
struct PredatorPreyCfg : CfgBase {
    static constexpr int N_ADV = 3, A = 7, L = 6, E = 13;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 7;             // world.dim_c = 3 but every agent is silent
    static constexpr int BARRIER = A, FOOD0 = A + 1, N_FOOD = 3, FOREST0 = A + 4;
    static constexpr int GOOD_LEADER = 3, MEMBER0 = 4, N_MEMBERS = 3;
    static constexpr int OBS_DIM = 4 + 4 * (A - 1) + N_MEMBERS + 2 * L + 2;      // 45 for every role
    static constexpr int OBS_TOTAL = A * OBS_DIM;
    static constexpr int NI = A + 1, NF = 0, N_AUX = 3;           // info: collisions per agent, the flags
    static constexpr bool DYNAMIC = true, NEW_CORE = true;
    static constexpr int N_WALLS = 4;
    __host__ __device__ __forceinline__ static constexpr bool collides(int e) { return e <= A; }      // agents and the barrier (:34,47,56,65)
    __host__ __device__ __forceinline__ static constexpr bool adversary(int i) { return i < N_ADV; }
    __host__ __device__ __forceinline__ static constexpr bool leader(int i) { return i == 0 || i == GOOD_LEADER; }
    __host__ __device__ __forceinline__ static constexpr bool member(int i) { return i >= MEMBER0; }
    // adversaries, good agents, the barrier, foods and forests each share their properties (:39-41,49,58,67)
    __host__ __device__ __forceinline__ static constexpr int rep(int e) { return e < N_ADV ? 0 : (e < A ? N_ADV : (e == BARRIER ? BARRIER : (e < FOREST0 ? FOOD0 : FOREST0))); }
    // a 64 x 315 tile (80 KB in float32) would leave one wave per CU: 8 whole rows at a time, the lanes regrouped
    // as 8 worlds x 7 agents (8 lanes idle during the observation)
    static constexpr bool NT_STEP = false;      // see CfgBase
    static constexpr bool OBS_COOP = true;
    // Two waves per SIMD (all 2 048 waves of a 131 072-world launch resident at once): 256 registers, eight 20 KB
    // workgroups a CU.  The kernel fits the registers with 6 steps' action words per fetch; what still spills does so
    // around the fetch and the reset draws, outside the steps.
    // (float32; the float64 build keeps one wave per SIMD)
    template <typename T> __host__ __device__ static constexpr bool stage_packed() { return sizeof(T) == 4; }
    template <typename T> __host__ __device__ static constexpr int waves_per_simd() { return sizeof(T) == 4 ? 2 : 1; }
    template <typename T> __host__ __device__ static constexpr int action_words() { return sizeof(T) == 4 ? 42 : 84; }
    __host__ __device__ __forceinline__ static constexpr int chunk_width(int) { return OBS_DIM; }
    __host__ __device__ __forceinline__ static constexpr int chunk_offset(int c) { return c * OBS_DIM; }

    __device__ static __forceinline__ bool holds(int flags, int i) { return (flags >> (3 + i - MEMBER0)) & 1; }

    template <typename T, class P>
    __device__ static __forceinline__ void forest_bits(const V2<T> (&pos)[E], const P& p, bool (&inf)[A][2]) {
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                inf[i][f] = touching<T>(pos[i], pos[FOREST0 + f], p.size[rep(i)] + p.size[rep(FOREST0 + f)]);
    }

    // visibility test of simple_predator_prey.py:325
    __device__ static __forceinline__ bool sees(int i, bool i1, bool i2, bool o1, bool o2) {
        return (i1 && o1) || (i2 && o2) || (!i1 && !o1 && !i2 && !o2) || leader(i);
    }

    // What one step's observations share: in_forest bits of every agent (bit 2i+f) and who sees whom (vis bit
    // A*i+j: agent i sees agent j), worked out once per step, not once per agent and pass
    struct ObsCtx { int forest; unsigned vis_lo, vis_hi; };
    template <typename T, class P>
    __device__ static __forceinline__ ObsCtx observe_prepare(const V2<T> (&pos)[E], const P& p) {
        bool inf[A][2];
        forest_bits<T>(pos, p, inf);
        int bits = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) bits |= (inf[i][0] ? 1 : 0) << (2 * i) | (inf[i][1] ? 1 : 0) << (2 * i + 1);
        unsigned long long vis = 0;
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j)
                if (j != i && sees(i, inf[i][0], inf[i][1], inf[j][0], inf[j][1])) vis |= 1ull << (A * i + j);
        return ObsCtx{bits, (unsigned)vis, (unsigned)(vis >> 32)};
    }

    __device__ __forceinline__ static ObsCtx ctx_from_lane(const ObsCtx& c, int src) {
        return ObsCtx{__shfl(c.forest, src, 64), (unsigned)__shfl((int)c.vis_lo, src, 64), (unsigned)__shfl((int)c.vis_hi, src, 64)};
    }

    // observation() :300-349 of agent i (a run-time index) of the world staged in `w`; offsets are
    // cached_dist_vect[agent, other] = p_agent - p_other; cast to float32
    template <typename T, class Staging>
    __device__ static __forceinline__ void observe_agent(int i, const Staging& w, const int* aux, const ObsCtx& ctx, T* row) {
        const int flags = aux[0];
        const unsigned long long visbits = ((unsigned long long)ctx.vis_hi << 32) | ctx.vis_lo;
        const unsigned seen = (unsigned)(visbits >> (A * i)) & ((1u << A) - 1u);      // bit j: agent i sees agent j
        const V2<T> me = w.pos(i), mv = w.vel(i);
        row[0] = round_f32(mv.x); row[1] = round_f32(mv.y);
        row[2] = round_f32(me.x); row[3] = round_f32(me.y);
        int k = 4;
        if (i >= MEMBER0) { row[4] = holds(flags, i) ? T(1) : T(-1); k = 5; }
#pragma unroll
        for (int n = 0; n < A - 1; ++n) {
            const int j = n + (n >= i ? 1 : 0);
            const bool vis = (seen >> j) & 1u;
            const V2<T> pj = w.pos(j), vj = w.vel(j);
            row[k + 2 * n] = vis ? round_f32(me.x - pj.x) : T(0);
            row[k + 2 * n + 1] = vis ? round_f32(me.y - pj.y) : T(0);
            row[k + 2 * (A - 1) + 2 * n] = vis ? round_f32(vj.x) : T(0);
            row[k + 2 * (A - 1) + 2 * n + 1] = vis ? round_f32(vj.y) : T(0);
        }
        k += 4 * (A - 1);
#pragma unroll
        for (int j = MEMBER0; j < A; ++j) {
            if (j == i) continue;
            row[k++] = ((seen >> j) & 1u) ? (holds(flags, j) ? T(1) : T(-1)) : T(0);
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const V2<T> pl = w.pos(A + l);
            row[k + 2 * l] = round_f32(me.x - pl.x); row[k + 2 * l + 1] = round_f32(me.y - pl.y);
        }
        k += 2 * L;
        row[k] = ((ctx.forest >> (2 * i)) & 1) ? T(1) : T(-1);
        row[k + 1] = ((ctx.forest >> (2 * i + 1)) & 1) ? T(1) : T(-1);
    }

    // observation() :300-349 of agents [I0, I1) with compile-time roles (mpe_observe's straight-line form)
    template <typename T, int I0, int I1>
    __device__ static __forceinline__ void observe_agents(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const int* aux, const ObsCtx& ctx, T* row) {
        int k = 0;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            // One agent's values are computed and stored at a time; pinning the agent's own position, the
            // visibility bits and the flags here keeps the compiler from evaluating every agent's arithmetic and
            // selects (315 live values) up front, or once for all passes over the rows.
            V2<T> me = pos[i];
            int fbits = ctx.forest, flags = aux[0];
            unsigned vlo = ctx.vis_lo, vhi = ctx.vis_hi;
            // (float32 only, see simple_world_comm's observe_cols)
            if constexpr (sizeof(T) == 4)
                if (I1 - I0 == 1) asm volatile("" : "+v"(me.x), "+v"(me.y), "+v"(fbits), "+v"(flags), "+v"(vlo), "+v"(vhi));
            const unsigned long long visbits = ((unsigned long long)vhi << 32) | vlo;
            bool inf[A][2], seen[A];
#pragma unroll
            for (int j = 0; j < A; ++j) { inf[j][0] = (fbits >> (2 * j)) & 1; inf[j][1] = (fbits >> (2 * j + 1)) & 1; seen[j] = (visbits >> (A * i + j)) & 1; }
            row[k++] = round_f32(vel[i].x); row[k++] = round_f32(vel[i].y);
            row[k++] = round_f32(me.x); row[k++] = round_f32(me.y);
            if (member(i)) row[k++] = holds(flags, i) ? T(1) : T(-1);
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                const bool vis = seen[j];
                row[k++] = vis ? round_f32(me.x - pos[j].x) : T(0);
                row[k++] = vis ? round_f32(me.y - pos[j].y) : T(0);
            }
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (j == i) continue;
                const bool vis = seen[j];
                row[k++] = vis ? round_f32(vel[j].x) : T(0);
                row[k++] = vis ? round_f32(vel[j].y) : T(0);
            }
#pragma unroll
            for (int j = MEMBER0; j < A; ++j) {
                if (j == i) continue;
                const bool vis = seen[j];
                row[k++] = vis ? (holds(flags, j) ? T(1) : T(-1)) : T(0);
            }
#pragma unroll
            for (int l = 0; l < L; ++l) { row[k++] = round_f32(me.x - pos[A + l].x); row[k++] = round_f32(me.y - pos[A + l].y); }
            row[k++] = inf[i][0] ? T(1) : T(-1);
            row[k++] = inf[i][1] ? T(1) : T(-1);
        }
    }

    // one pass of the row-pass scheme: agent by agent (each pinned, see observe_agents), stored by the pass's lanes
    template <typename T, int I>
    __device__ static __forceinline__ void pass_agents(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const int* aux, const ObsCtx& ctx, T* row, bool mine) {
        T vals[OBS_DIM];
        observe_agents<T, I, I + 1>(pos, vel, aux, ctx, vals);
        if (mine) {
#pragma unroll
            for (int k = 0; k < OBS_DIM; ++k) row[I * OBS_DIM + k] = vals[k];
        }
        if constexpr (I + 1 < A) pass_agents<T, I + 1>(pos, vel, aux, ctx, row, mine);
    }
    template <typename T, class P>
    __device__ static __forceinline__ void observe_pass(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int* aux, const P& p, T* row, bool mine) {
        pass_agents<T, 0>(pos, vel, aux, observe_prepare<T>(pos, p), row, mine);
    }

    template <typename T, class P>
    __device__ static __forceinline__ void observe(const V2<T> (&pos)[E], const V2<T> (&vel)[A], const T*, const int* aux, const P& p, T* row) {
        observe_agents<T, 0, A>(pos, vel, aux, observe_prepare<T>(pos, p), row);
    }

    // reward() :171-275 and benchmark_data() :148-156
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P& p, T* rew, bool want_info, int* info_i, T*) {
        const int flags = aux[0];
        bool hit[A][N_ADV];     // hit[g][a]: good agent g touches adversary a
        T dist[A][N_ADV];
#pragma unroll
        for (int g = N_ADV; g < A; ++g)
#pragma unroll
            for (int a = 0; a < N_ADV; ++a) {
                dist[g][a] = norm2<T>(pos[a].x - pos[g].x, pos[a].y - pos[g].y);
                hit[g][a] = dist[g][a] < p.size[rep(a)] + p.size[rep(g)];
            }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            T r;
            if (adversary(i)) {                     // :243-275
                T dmin = dist[N_ADV][i];
                T n = T(0);
#pragma unroll
                for (int g = N_ADV; g < A; ++g) {
                    if (g > N_ADV) dmin = py_min<T>(dmin, dist[g][i]);
                    n = n + (hit[g][i] ? T(1) : T(0));
                }
                r = T(0) - T(0.1) * dmin;
                r = r + T(5) * n;
                if (leader(i)) {
#pragma unroll
                    for (int g = MEMBER0; g < A; ++g) {     // only members ever hold food
                        T m = T(0);
#pragma unroll
                        for (int a = 0; a < N_ADV; ++a) m = m + (hit[g][a] ? T(1) : T(0));
                        if (holds(flags, g)) r = r + T(10) * m;
                    }
                }
                if (want_info) {
                    int c = 0;
#pragma unroll
                    for (int g = N_ADV; g < A; ++g) c += hit[g][i] ? 1 : 0;
                    info_i[i] = c;
                }
            } else {                                // :183-241
                T s = dist[i][0], n = T(0);
#pragma unroll
                for (int a = 0; a < N_ADV; ++a) {
                    if (a > 0) s = s + dist[i][a];
                    n = n + (hit[i][a] ? T(1) : T(0));
                }
                r = T(0.1) * s;
                r = r - T(5) * n;
                if (member(i)) {
                    if (!holds(flags, i)) {
                        T fmin = norm2<T>(pos[FOOD0].x - pos[i].x, pos[FOOD0].y - pos[i].y);
#pragma unroll
                        for (int f = 1; f < N_FOOD; ++f) fmin = py_min<T>(fmin, norm2<T>(pos[FOOD0 + f].x - pos[i].x, pos[FOOD0 + f].y - pos[i].y));
                        r = r - T(0.1) * fmin;
                    } else {
                        r = r - T(0.1) * norm2<T>(pos[GOOD_LEADER].x - pos[i].x, pos[GOOD_LEADER].y - pos[i].y);
                    }
                } else {
                    bool any = false;
                    T dmin = T(0);
#pragma unroll
                    for (int m = MEMBER0; m < A; ++m) {
                        if (!holds(flags, m)) continue;
                        const T d = norm2<T>(pos[i].x - pos[m].x, pos[i].y - pos[m].y);
                        dmin = any ? py_min<T>(dmin, d) : d;
                        any = true;
                    }
                    if (any) r = r - T(0.1) * dmin;
#pragma unroll
                    for (int m = MEMBER0; m < A; ++m) {
                        T c = T(0);
                        if (!holds(flags, m)) {
#pragma unroll
                            for (int f = 0; f < N_FOOD; ++f)
                                c = c + (touching<T>(pos[FOOD0 + f], pos[m], p.size[rep(FOOD0 + f)] + p.size[rep(m)]) ? T(1) : T(0));
                        } else {
                            c = touching<T>(pos[i], pos[m], p.size[rep(i)] + p.size[rep(m)]) ? T(1) : T(0);
                        }
                        r = r + T(5) * c;
                    }
                }
                if (want_info) info_i[i] = 0;
            }
            rew[i] = r;
        }
        if (want_info) info_i[A] = flags;
    }

    // post_step() :83-107, after the step's observations and rewards are taken: a member that does
    // not hold food picks up a live food it touches (the food is parked at -999), a dead food
    // respawns at U(-0.95, 0.95)^2 (respawn_prob is 1), a member touching the good leader hands over.
    template <class Pos, class P>
    __device__ static __forceinline__ void post_step(Pos& pos, int* aux, const P& p, unsigned long long b) {
        typedef decltype(pos[0].x) T;
        int flags = aux[0];
        bool at_leader[N_MEMBERS];
#pragma unroll
        for (int k = 0; k < N_MEMBERS; ++k)
            at_leader[k] = touching<T>(pos[MEMBER0 + k], pos[GOOD_LEADER], p.size[rep(MEMBER0 + k)] + p.size[rep(GOOD_LEADER)]);
#pragma unroll
        for (int f = 0; f < N_FOOD; ++f) {
            if ((flags >> f) & 1) {
                bool taken = false;
#pragma unroll
                for (int k = 0; k < N_MEMBERS; ++k) {
                    const bool pick = !taken && !((flags >> (3 + k)) & 1) &&
                                      touching<T>(pos[FOOD0 + f], pos[MEMBER0 + k], p.size[rep(FOOD0 + f)] + p.size[rep(MEMBER0 + k)]);
                    if (pick) { flags |= 1 << (3 + k); taken = true; }
                }
                if (taken) {
                    flags &= ~(1 << f);
                    pos[FOOD0 + f].x = T(-999); pos[FOOD0 + f].y = T(-999);
                }
            } else {
                const V2<double> d = respawn_draw(aux[1], b, aux[2], 0.95);
                aux[2] += 1;
                pos[FOOD0 + f].x = (T)d.x; pos[FOOD0 + f].y = (T)d.y;
                flags |= 1 << f;
            }
        }
#pragma unroll
        for (int k = 0; k < N_MEMBERS; ++k)
            if (((flags >> (3 + k)) & 1) && at_leader[k]) flags &= ~(1 << (3 + k));
        aux[0] = flags;
    }

    // reset_world :109-146: every food alive, nobody holds; a fresh respawn stream
    template <class RS>
    __host__ __device__ static int initial_aux(const RS& rs, unsigned long long b, int k) {
        return k == 0 ? 7 : (k == 1 ? reset_stream_key(rs, b) : 0);
    }

    // reset_world :139-144: a food is redrawn until it touches neither the barrier nor a forest.
    // d holds the float64 draws of all entities; p.size_d their float64 sizes.
    template <class RS, class Pos, class P>
    __host__ __device__ static void fix_reset(const RS& rs, unsigned long long b, Pos& d, const P& p) {
        const int obst[3] = {BARRIER, FOREST0, FOREST0 + 1};
        for (int f = FOOD0; f < FOOD0 + N_FOOD; ++f) {
            for (int attempt = 1; attempt <= 64; ++attempt) {
                bool bad = false;
                for (int o = 0; o < 3; ++o) {
                    const double dx = d[f].x - d[obst[o]].x, dy = d[f].y - d[obst[o]].y;
                    bad = bad || (sqrt(dx * dx + dy * dy) < p.size_d[rep(f)] + p.size_d[rep(obst[o])]);
                }
                if (!bad) break;
                d[f] = reset_draw(rs, b, f, false, attempt);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// collector on the new_core physics: collectors 0-5 pick up treasures of two types and carry them to
// the deposit of that type (agents 6, 7); every agent collides; 6 treasures; no walls
// (multiagent/scenarios/collector.py).  aux[0] holds the state post_step rewrites: bits 2a, 2a+1 =
// what collector a holds (0 nothing, 1 + type), bit 12+l = type of treasure l, bit 18+l = treasure l
// is alive; aux[1] / aux[2] are the key and the position of the world's respawn stream.  Distances
// are those of the cache World.step fills (new_core.py:147-170): vect[a, b] = p_a - p_b.
// ------------------------------------------------------------------------------------------
struct CollectorCfg : CfgBase {
    static constexpr int NC = 6, A = 8, L = 6, E = 14;
    static constexpr int DIM_C = 0, S = 0, N_ACT = 8;
    static constexpr int OBS_COL = 4 + 2 + (A - 1) * 8 + L * 4, OBS_DEP = 4 + (A - 1) * 8 + L * 4;      // 86, 84
    static constexpr int OBS_TOTAL = NC * OBS_COL + (A - NC) * OBS_DEP;
    static constexpr int NI = 3 * A + 1, NF = 0, N_AUX = 3;       // info: (collected, deposited, collisions) per agent, the state word
    static constexpr bool DYNAMIC = true, NEW_CORE = true;
    static constexpr int OBS_CHUNKS = A, CHUNK_MAX = OBS_COL, CHUNK_MIN = OBS_DEP;
    __host__ __device__ __forceinline__ static constexpr int chunk_width(int c) { return c < NC ? OBS_COL : OBS_DEP; }
    __host__ __device__ __forceinline__ static constexpr int chunk_offset(int c) { return c < NC ? c * OBS_COL : NC * OBS_COL + (c - NC) * OBS_DEP; }
    __host__ __device__ __forceinline__ static constexpr bool collides(int e) { return e < A; }       // :40,57
    // collectors, deposits and treasures each share their properties (:44-47,59)
    __host__ __device__ __forceinline__ static constexpr int rep(int e) { return e < NC ? 0 : (e < A ? NC : A); }

    __device__ static __forceinline__ int held(int st, int a) { return ((st >> (2 * a)) & 3) - 1; }      // -1: nothing, else the type
    __device__ static __forceinline__ int ttype(int st, int l) { return (st >> (12 + l)) & 1; }
    __device__ static __forceinline__ bool alive(int st, int l) { return (st >> (18 + l)) & 1; }

    template <typename T, class P>
    __device__ static __forceinline__ int observe_prepare(const V2<T> (&)[E], const P&) { return 0; }

    // sorted(zip(distance, index)): ascending distance, ties by index (collector.py:272-278)
    template <typename T> __device__ static __forceinline__ bool after(T ka, int ia, T kb, int ib) { return ka > kb || (ka == kb && ia > ib); }

    static constexpr bool NT_STEP = false;      // see CfgBase
    static constexpr bool OBS_COOP = true;

    // observation() of agent i (a run-time index), collector.py:268-295; cast to float32
    template <typename T, class Staging>
    __device__ static __forceinline__ void observe_agent(int i, const Staging& w, const int* aux, int, T* row) {
        const int st = aux[0];
        const V2<T> me = w.pos(i), mv = w.vel(i);
        row[0] = round_f32(me.x); row[1] = round_f32(me.y);
        row[2] = round_f32(mv.x); row[3] = round_f32(mv.y);
        int k = 4;
        if (i < NC) {
            const int h = held(st, i);
            row[4] = (h == 0) ? T(1) : T(0); row[5] = (h == 1) ? T(1) : T(0);
            k = 6;
        }
        // ---- the 7 other agents, nearest first: offset, velocity, encoding (:257-266) ----
        {
            constexpr int N = A - 1;
            T key[N], dx[N], dy[N], vx[N], vy[N];
            int idx[N], code[N];        // code: 0 collector holding nothing, 1 / 2 holding type 0 / 1, 3 / 4 deposit of type 0 / 1
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int j = n + (n >= i ? 1 : 0);
                const V2<T> pj = w.pos(j), vj = w.vel(j);
                dx[n] = pj.x - me.x; dy[n] = pj.y - me.y;
                key[n] = norm2<T>(dx[n], dy[n]);
                vx[n] = vj.x; vy[n] = vj.y;
                idx[n] = j;
                code[n] = j < NC ? held(st, j) + 1 : 3 + (j - NC);
            }
            auto ce = [&](int a, int b) {
                const bool sw = after<T>(key[a], idx[a], key[b], idx[b]);
                T t;
                int u;
                t = key[a]; key[a] = sw ? key[b] : t; key[b] = sw ? t : key[b];
                t = dx[a]; dx[a] = sw ? dx[b] : t; dx[b] = sw ? t : dx[b];
                t = dy[a]; dy[a] = sw ? dy[b] : t; dy[b] = sw ? t : dy[b];
                t = vx[a]; vx[a] = sw ? vx[b] : t; vx[b] = sw ? t : vx[b];
                t = vy[a]; vy[a] = sw ? vy[b] : t; vy[b] = sw ? t : vy[b];
                u = idx[a]; idx[a] = sw ? idx[b] : u; idx[b] = sw ? u : idx[b];
                u = code[a]; code[a] = sw ? code[b] : u; code[b] = sw ? u : code[b];
            };
            // 16-comparator network for 7 keys
            ce(0, 6); ce(2, 3); ce(4, 5); ce(0, 2); ce(1, 4); ce(3, 6); ce(0, 1); ce(2, 5); ce(3, 4);
            ce(1, 2); ce(4, 6); ce(2, 3); ce(4, 5); ce(1, 2); ce(3, 4); ce(5, 6);
#pragma unroll
            for (int q = 0; q < N; ++q) {
                T* o = row + k + 8 * q;
                o[0] = round_f32(dx[q]); o[1] = round_f32(dy[q]);
                o[2] = round_f32(vx[q]); o[3] = round_f32(vy[q]);
                o[4] = (code[q] == 3) ? T(1) : T(0); o[5] = (code[q] == 4) ? T(1) : T(0);
                o[6] = (code[q] == 1) ? T(1) : T(0); o[7] = (code[q] == 2) ? T(1) : T(0);
            }
            k += 8 * N;
        }
        // ---- the 6 treasures, nearest first: offset, type (:275-278,290-293) ----
        {
            T key[L], dx[L], dy[L];
            int idx[L], ty[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const V2<T> pl = w.pos(A + l);
                dx[l] = pl.x - me.x; dy[l] = pl.y - me.y;
                key[l] = norm2<T>(dx[l], dy[l]);
                idx[l] = l;
                ty[l] = ttype(st, l);
            }
            auto ce = [&](int a, int b) {
                const bool sw = after<T>(key[a], idx[a], key[b], idx[b]);
                T t;
                int u;
                t = key[a]; key[a] = sw ? key[b] : t; key[b] = sw ? t : key[b];
                t = dx[a]; dx[a] = sw ? dx[b] : t; dx[b] = sw ? t : dx[b];
                t = dy[a]; dy[a] = sw ? dy[b] : t; dy[b] = sw ? t : dy[b];
                u = idx[a]; idx[a] = sw ? idx[b] : u; idx[b] = sw ? u : idx[b];
                u = ty[a]; ty[a] = sw ? ty[b] : u; ty[b] = sw ? u : ty[b];
            };
            // 12-comparator network for 6 keys
            ce(0, 5); ce(1, 3); ce(2, 4); ce(1, 2); ce(3, 4); ce(0, 3); ce(2, 5); ce(0, 1); ce(2, 3); ce(4, 5); ce(1, 2); ce(3, 4);
#pragma unroll
            for (int q = 0; q < L; ++q) {
                T* o = row + k + 4 * q;
                o[0] = round_f32(dx[q]); o[1] = round_f32(dy[q]);
                o[2] = (ty[q] == 0) ? T(1) : T(0); o[3] = (ty[q] == 1) ? T(1) : T(0);
            }
        }
    }

    // reward() :165-255 (everything is paid to collector 0) and benchmark_data() :132-155
    template <typename T, class P>
    __device__ static __forceinline__ void score(const V2<T> (&pos)[E], const T*, const int* aux, const P& p, T* rew, bool want_info, int* info_i, T*) {
        const int st = aux[0];
        const T s_cc = p.size[rep(0)] + p.size[rep(0)], s_cd = p.size[rep(0)] + p.size[rep(NC)], s_ct = p.size[rep(0)] + p.size[rep(A)];
        T d_cd[NC][A - NC];         // collector - deposit distances
        bool hit_cd[NC][A - NC];
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int d = 0; d < A - NC; ++d) {
                d_cd[a][d] = norm2<T>(pos[NC + d].x - pos[a].x, pos[NC + d].y - pos[a].y);
                hit_cd[a][d] = d_cd[a][d] < s_cd;
            }
        // global rewards :239-255
        int gdr = 0, gcr = 0;
#pragma unroll
        for (int d = 0; d < A - NC; ++d)
#pragma unroll
            for (int a = 0; a < NC; ++a) gdr += (held(st, a) == d && hit_cd[a][d]) ? 5 : 0;
        T tmin[NC];                 // distance of collector a to its nearest live treasure
        bool tany = false;
        int n_touch[NC];            // treasures (dead ones at the origin included) collector a touches
#pragma unroll
        for (int a = 0; a < NC; ++a) { tmin[a] = T(0); n_touch[a] = 0; }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const bool live = alive(st, l);
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                const T d = norm2<T>(pos[A + l].x - pos[a].x, pos[A + l].y - pos[a].y);
                const bool h = d < s_ct;
                n_touch[a] += h ? 1 : 0;
                if (live) {
                    gcr += (h && held(st, a) < 0) ? 5 : 0;
                    tmin[a] = (tany && !(d < tmin[a])) ? tmin[a] : d;       // Python min: first wins
                }
            }
            tany = tany || live;
        }
        const T glob = (T)(gdr + gcr);
        // deposit_reward :177-213 of both deposits
        T main = T(0);
#pragma unroll
        for (int d = 0; d < A - NC; ++d) {
            T r;
            bool any = false;
            T dmin = T(0);
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                if (held(st, a) != d) continue;
                dmin = (any && !(d_cd[a][d] < dmin)) ? dmin : d_cd[a][d];
                any = true;
            }
            if (any) {
                r = T(0) - T(0.1) * dmin;
            } else {
                // nobody carries this type: norm of the mean offset of the collectors, summed nearest first
                T key[NC], dx[NC], dy[NC];
                int idx[NC];
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    key[a] = d_cd[a][d]; dx[a] = pos[a].x - pos[NC + d].x; dy[a] = pos[a].y - pos[NC + d].y; idx[a] = a;
                }
                auto ce = [&](int a, int b) {
                    const bool sw = after<T>(key[a], idx[a], key[b], idx[b]);
                    T t;
                    int u;
                    t = key[a]; key[a] = sw ? key[b] : t; key[b] = sw ? t : key[b];
                    t = dx[a]; dx[a] = sw ? dx[b] : t; dx[b] = sw ? t : dx[b];
                    t = dy[a]; dy[a] = sw ? dy[b] : t; dy[b] = sw ? t : dy[b];
                    u = idx[a]; idx[a] = sw ? idx[b] : u; idx[b] = sw ? u : idx[b];
                };
                ce(0, 5); ce(1, 3); ce(2, 4); ce(1, 2); ce(3, 4); ce(0, 3); ce(2, 5); ce(0, 1); ce(2, 3); ce(4, 5); ce(1, 2); ce(3, 4);
                T sx = dx[0], sy = dy[0];
#pragma unroll
                for (int q = 1; q < NC; ++q) { sx = sx + dx[q]; sy = sy + dy[q]; }
                sx = sx / T(NC); sy = sy / T(NC);
                r = T(0) - T(0.1) * norm2<T>(sx, sy);
            }
            r = r + (T)gdr;
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                const int h = held(st, a);
                if (h < 0) {
                    if (tany) r = r - T(0.1) * tmin[a];
                } else {
                    r = r - T(0.1) * (h == 0 ? d_cd[a][0] : d_cd[a][1]);
                }
                r = r + glob;
            }
            main = (d == 0) ? r : main + r;
        }
        // collector_reward :215-229 and the collision counts
        int n_coll[NC];
        int total = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a) n_coll[a] = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int b = a + 1; b < NC; ++b) {
                const bool h = norm2<T>(pos[a].x - pos[b].x, pos[a].y - pos[b].y) < s_cc;
                n_coll[a] += h ? 1 : 0;
                n_coll[b] += h ? 1 : 0;
            }
#pragma unroll
        for (int a = 0; a < NC; ++a) total += n_coll[a];
        rew[0] = T(0) + main + (T)(-5 * total);
#pragma unroll
        for (int i = 1; i < A; ++i) rew[i] = T(0);
        if (want_info) {
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                info_i[3 * a] = held(st, a) < 0 ? n_touch[a] : 0;
                info_i[3 * a + 1] = 0;
                info_i[3 * a + 2] = n_coll[a];
            }
#pragma unroll
            for (int d = 0; d < A - NC; ++d) {
                int c = 0;
#pragma unroll
                for (int a = 0; a < NC; ++a) c += (held(st, a) == d && hit_cd[a][d]) ? 1 : 0;
                info_i[3 * (NC + d)] = 0;
                info_i[3 * (NC + d) + 1] = c;
                info_i[3 * (NC + d) + 2] = 0;
            }
            info_i[3 * A] = st;
        }
    }

    // post_step() :84-110, after the step's observations and rewards are taken
    template <class Pos, class P>
    __device__ static __forceinline__ void post_step(Pos& pos, int* aux, const P& p, unsigned long long b) {
        typedef decltype(pos[0].x) T;
        int st = aux[0];
        const T s_cd = p.size[rep(0)] + p.size[rep(NC)], s_ct = p.size[rep(0)] + p.size[rep(A)];
        bool at_deposit[NC][A - NC];
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int d = 0; d < A - NC; ++d) at_deposit[a][d] = touching<T>(pos[a], pos[NC + d], s_cd);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (alive(st, l)) {
                bool taken = false;
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    const bool pick = !taken && held(st, a) < 0 && touching<T>(pos[A + l], pos[a], s_ct);
                    if (pick) { st |= (1 + ttype(st, l)) << (2 * a); taken = true; }
                }
                if (taken) {
                    st &= ~(1 << (18 + l));
                    pos[A + l].x = T(-0.0); pos[A + l].y = T(-0.0);
                }
            } else {
                int spare;
                const V2<double> d = respawn_draw(aux[1], b, aux[2], 0.95, &spare);
                aux[2] += 1;
                pos[A + l].x = (T)d.x; pos[A + l].y = (T)d.y;
                st = (st & ~(1 << (12 + l))) | ((spare & 1) << (12 + l)) | (1 << (18 + l));
            }
        }
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const int h = held(st, a);
            if (h >= 0 && (h == 0 ? at_deposit[a][0] : at_deposit[a][1])) st &= ~(3 << (2 * a));
        }
        aux[0] = st;
    }

    // reset_world :112-130: nobody holds anything, every treasure alive with a fresh type; a fresh respawn stream
    template <class RS>
    __host__ __device__ static int initial_aux(const RS& rs, unsigned long long b, int k) {
        return k == 0 ? (((reset_stream_key(rs, b, 1) & 63) << 12) | (63 << 18)) : (k == 1 ? reset_stream_key(rs, b) : 0);
    }
};

}  // namespace mpe
