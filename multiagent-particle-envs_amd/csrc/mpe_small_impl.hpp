// Fused step kernels for the small scenarios (simple_spread 3/3, simple_tag, simple_world_comm and the
// other classic scenarios with a handful of entities).
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
#pragma once
#include "mpe_scenarios.hpp"
#include "mpe_launch.hpp"

namespace mpe {

enum { MODE_STEP = 0, MODE_ROLLOUT = 1, MODE_OBSERVE = 2 };


// ---- wave-private LDS tile holding W values for each of the wave's ROWS consecutive envs --------
// Lane l writes its env's row, then the wave streams the tile to the env-major global array.
//
// LINEAR: rows are stored back to back (row stride W), i.e. the tile is a byte image of the wave's
//   contiguous 64*W-element block of the global array, which goes out as 16-byte vectors with no
//   index arithmetic.  Row writes then land with lane stride W: fine when W*sizeof(V)/4 is not a
//   multiple of a large power of two (simple_spread 42, simple_tag 62: 2-way, free for ds_write).
// PADDED: row stride W|1 (odd) keeps the row writes conflict-free for any W (a row of 96 or 192 values
//   would otherwise be a 32-way conflict); full tiles still leave as 16-byte vectors, each gathered from
//   four padded LDS words (4 % fewer cycles than dword stores for simple_world_comm at 262 144 worlds).
template <typename V, int W, bool LINEAR, int ROWS>
struct Tile {
    static constexpr int STRIDE = LINEAR ? W : (W | 1);
    static constexpr int SIZE = ROWS * STRIDE;
    static constexpr int VEC = 16 / (int)sizeof(V);

    __device__ static __forceinline__ V* row(V* tile, int lane) { return tile + lane * STRIDE; }

    // FULL: all 64 envs valid and gdst 16-byte aligned (fixed per kernel instantiation, so the
    // fast path is straight-line code and the compiler can count outstanding stores)
    // NT: non-temporal stores (see put<> below): the launches of a single step use them
    template <bool FULL, bool NT = false>
    __device__ static __forceinline__ void flush(const V* tile, V* gdst, int lane, int nvalid) {
        if (LINEAR && FULL) {
            typedef V VT __attribute__((ext_vector_type(VEC)));
            static_assert((ROWS * W) % VEC == 0, "tile must be a whole number of 16-byte vectors");
            constexpr int NV = ROWS * W / VEC, FULL_IT = NV / 64, REM = NV % 64;
            const VT* src = reinterpret_cast<const VT*>(tile);
            VT* dst = reinterpret_cast<VT*>(gdst);
#pragma unroll
            for (int it = 0; it < FULL_IT; ++it) put<NT>(&dst[it * 64 + lane], src[it * 64 + lane]);
            if (REM > 0 && lane < REM) put<NT>(&dst[FULL_IT * 64 + lane], src[FULL_IT * 64 + lane]);
        } else if (!LINEAR && FULL && W % VEC == 0) {
            // padded rows, 16-byte stores: vector g covers values [VEC * (g % NVR), + VEC) of row g / NVR
            typedef V VT __attribute__((ext_vector_type(VEC)));
            constexpr int NVR = W / VEC, NV = ROWS * NVR;
#pragma unroll
            for (int it = 0; it < (NV + 63) / 64; ++it) {
                const int g = it * 64 + lane;
                const int r = g / NVR, j = (g - r * NVR) * VEC;
                VT v;
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = tile[r * STRIDE + j + q];
                if (NV % 64 == 0 || g < NV) put<NT>(reinterpret_cast<VT*>(gdst + r * W + j), v);
            }
        } else {
            const int n = nvalid * W;
#pragma unroll
            for (int it = 0; it < (ROWS * W + 63) / 64; ++it) {
                const int g = it * 64 + lane;
                if (LINEAR) {
                    if (g < n) put<NT>(&gdst[g], tile[g]);
                } else {
                    const int r = g / W;
                    if (g < n) put<NT>(&gdst[g], tile[r * STRIDE + (g - r * W)]);
                }
            }
        }
    }
};

// Observation rows of a scenario that produces them in OBS_CHUNKS column chunks: every chunk is
// written into the wave's (rows x chunk width) tile and streamed to its columns of the env-major
// block at gbase (row stride OT).  The tile's row stride is the widest chunk, made odd when that
// width would put every lane's row on the same few LDS banks; chunks of such a tile whose width is
// a whole number of 16-byte vectors leave as vectors, the others 4 bytes per lane.
template <typename T, class Cfg>
struct ChunkTile {
    static constexpr bool LINEAR = ((Cfg::CHUNK_MAX * (int)sizeof(T) / 4) % 16) != 0;
    static constexpr int STRIDE = LINEAR ? Cfg::CHUNK_MAX : (Cfg::CHUNK_MAX | 1);
};

template <typename T, class Cfg, int C, int EPW, bool FULL, bool NT = false>
struct ObsChunks {
    static constexpr int W = Cfg::chunk_width(C), S = ChunkTile<T, Cfg>::STRIDE, COL0 = Cfg::chunk_offset(C);
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr bool VECTORS = FULL && EPW == 64 && !ChunkTile<T, Cfg>::LINEAR && W % VEC == 0 && COL0 % VEC == 0 && Cfg::OBS_TOTAL % VEC == 0;
    template <class Pos, class Vel, class Ctx>
    static __device__ __forceinline__ void run(const Pos& pos, const Vel& vel, const T* comm, const int* aux, const Ctx& ctx, T* tile, T* gbase,
                                               int lane, int tile_lane, bool active, int nvalid) {
        if (active) Cfg::template observe_chunk<T, C>(pos, vel, comm, aux, ctx, tile + tile_lane * S);
        __syncthreads();
        if constexpr (VECTORS) {
            // vector g covers values [VEC * (g % NVR), + VEC) of row g / NVR, gathered from the padded tile
            typedef T VT __attribute__((ext_vector_type(VEC)));
            constexpr int NVR = W / VEC, NV = EPW * NVR;
#pragma unroll
            for (int it = 0; it < (NV + 63) / 64; ++it) {
                const int g = it * 64 + lane;
                const int r = g / NVR, j = (g - r * NVR) * VEC;
                VT v;
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = tile[r * S + j + q];
                if (NV % 64 == 0 || g < NV) put<NT>(reinterpret_cast<VT*>(gbase + (long long)r * Cfg::OBS_TOTAL + COL0 + j), v);
            }
        } else {
            const int n = nvalid * W;
#pragma unroll
            for (int it = 0; it < (EPW * W + 63) / 64; ++it) {
                const int g = it * 64 + lane;
                const int r = g / W;
                if (g < n) put<NT>(&gbase[(long long)r * Cfg::OBS_TOTAL + COL0 + (g - r * W)], tile[r * S + (g - r * W)]);
            }
        }
        __syncthreads();
        if constexpr (C + 1 < Cfg::OBS_CHUNKS)
            ObsChunks<T, Cfg, C + 1, EPW, FULL, NT>::run(pos, vel, comm, aux, ctx, tile, gbase, lane, tile_lane, active, nvalid);
    }
};

#ifdef MPE_STAMPS
// Diagnostic build only: 100 MHz timestamps per workgroup (slot 0 start, 1 state and actions ready,
// 2+t step t computed and its rows issued, 60 all steps done, 61 the final state's stores issued, 62 every store of
// the wave acknowledged; within step 10: 30 physics done, 31 rewards done, 32 state staged, 33.. cooperative pass k
// flushed, 41 post_step done).  tools/probes/episode_stamps.py
// and tools/probes/step_phases.py read them.
__device__ long long g_stamps[4096 * 64];
}  // namespace mpe
// copies the timestamps to host memory; the one translation unit built with -DMPE_STAMPS defines it
extern "C" int mpe_debug_stamps(long long* host, long long count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mpe::g_stamps), (size_t)count * sizeof(long long));
}
namespace mpe {
#define STAMP(slot) do { if (lane == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_T(slot) do { if (t == 10) STAMP(slot); } while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP_T(slot) do { } while (0)
#endif

// EPW: envs per wave (tile rows); 64 uses every lane.
// FULL: every wave of the launch owns EPW valid envs and all output blocks are 16-byte aligned.
// INFO: info_i / info_f are written.  env_begin / env_end: the env range this launch covers.
// rs.enabled (rollout launches only): instead of loading the state, draw a fresh one exactly as
//   reset_kernel would (Scenario.reset_world), so an episode is a single launch.
// NOISE: the world has agents with u_noise / c_noise (its own instantiation: the generator is an out-of-line
//   call, and a kernel with a call site pays for the calling convention in registers on every path).
template <typename T, class Cfg, bool CLAMP, int MODE, bool FULL, bool INFO, int EPW, bool NOISE = false>
__global__ __launch_bounds__(64, (MODE == MODE_ROLLOUT && FULL && !INFO && !NOISE) ? Cfg::template waves_per_simd<T>() : 1) void small_kernel(const SmallParams<T, Cfg::E, Cfg::A> p, const ResetArgs rs, const AutoArgs au,
                                                   const long long env_begin, const long long env_end,
                                                   const int* __restrict__ actions, T* __restrict__ pos_g,
                                                   T* __restrict__ vel_g, T* __restrict__ comm_g,
                                                   T* __restrict__ obs_g, T* __restrict__ rew_g,
                                                   uint8_t* __restrict__ done_g, int* __restrict__ info_i_g,
                                                   T* __restrict__ info_f_g, int* __restrict__ aux_g, T* __restrict__ obs0_g, const NoiseArgs nz,
                                                   int n_steps) {
    constexpr int A = Cfg::A, E = Cfg::E, OT = Cfg::OBS_TOTAL, NA = Cfg::N_ACT;
    constexpr bool NT = MODE == MODE_STEP && Cfg::NT_STEP;      // a single step's outputs leave as non-temporal stores (see put<>)
    constexpr int NI = Cfg::NI > 0 ? Cfg::NI : 1, NF = Cfg::NF > 0 ? Cfg::NF : 1, DC = Cfg::DIM_C, NX = Cfg::N_AUX;
    constexpr int NC = Cfg::S * Cfg::DIM_C;      // comm state values: the speakers' utterances in agent order
    constexpr int OW = Cfg::OBS_CHUNKS > 1 ? Cfg::CHUNK_MAX : OT;        // columns held by the observation tile
    constexpr bool LIN = ((OW * (int)sizeof(T) / 4) % 16) != 0;
    constexpr int RP = Cfg::OBS_ROW_PASSES;                              // > 0: the tile holds EPW / RP whole rows at a time
    static_assert(RP == 0 || (Cfg::OBS_CHUNKS == 1 && !Cfg::OBS_LOOP && EPW % (RP > 0 ? RP : 1) == 0 && LIN), "row passes stream whole rows through a linear tile");
    constexpr bool COOP = Cfg::OBS_COOP;                                 // whole rows of 8 worlds per pass, lanes = worlds x agents
    static_assert(!COOP || (EPW == 64 && A <= 8 && !Cfg::OBS_LOOP && RP == 0), "cooperative passes regroup a full wave as 8 worlds x up to 8 agents");
    constexpr int TROWS = COOP ? 8 : (RP > 0 ? EPW / RP : EPW);
    using ObsTile = Tile<T, (COOP ? OT : OW), (COOP ? true : LIN), TROWS>;
    using CoopStage = StagedRows<T, E, A, Cfg::template stage_packed<T>()>;
    using RewTile = Tile<T, A, true, EPW>;
    using IiTile = Tile<int, NI, true, EPW>;
    using IfTile = Tile<T, NF, true, EPW>;
    __shared__ __attribute__((aligned(16))) T tile_obs[ObsTile::SIZE];
    __shared__ __attribute__((aligned(16))) T stage_lds[Cfg::OBS_LOOP ? Staged<T, E, A>::SIZE : (COOP ? CoopStage::SIZE : 1)];
    // Under cooperative passes the reward and info tiles live in the observation tile: they leave right after the
    // rewards are known, before the first pass writes rows there (20 KB of LDS a workgroup: eight fit a CU).
    constexpr int REW_BYTES = (RewTile::SIZE * (int)sizeof(T) + 15) / 16 * 16, II_BYTES = INFO ? (IiTile::SIZE * 4 + 15) / 16 * 16 : 0;
    static_assert(!COOP || REW_BYTES + II_BYTES + (INFO ? IfTile::SIZE : 0) * (int)sizeof(T) <= ObsTile::SIZE * (int)sizeof(T), "small tiles must fit the observation tile");
    __shared__ __attribute__((aligned(16))) T tile_rew_own[COOP ? 4 : RewTile::SIZE];
    __shared__ __attribute__((aligned(16))) int tile_ii_own[INFO && !COOP ? IiTile::SIZE : 4];
    __shared__ __attribute__((aligned(16))) T tile_if_own[INFO && !COOP ? IfTile::SIZE : 4];
    T* const tile_rew = COOP ? tile_obs : tile_rew_own;
    int* const tile_ii = COOP ? reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(tile_obs) + REW_BYTES) : tile_ii_own;
    T* const tile_if = COOP ? reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(tile_obs) + REW_BYTES + II_BYTES) : tile_if_own;

    const int lane = threadIdx.x;
    const long long B = p.n_envs;
    unsigned tile = blockIdx.x;
    if (p.deal.phase != nullptr) {     // tiles dealt unevenly to even and odd XCDs (launch_range)
        const long long unit = xcd_unit(p.deal, blockIdx.x);
        if (unit < 0) return;
        tile = (unsigned)unit;
    }
    const long long env0 = env_begin + (long long)tile * EPW;
    const long long left = env_end - env0;
    const int nvalid = FULL ? EPW : (left < EPW ? (int)left : EPW);
    const bool valid = lane < nvalid;
    const int tile_lane = (EPW == 64 || lane < EPW) ? lane : 0;   // lanes beyond the tile write row 0 (masked below)
    const long long b = valid ? env0 + lane : env0;   // idle tail lanes shadow the wave's first env

    STAMP(0);
    // Actions are fetched NP * CH steps at a time and packed BITS (3 or 4) bits each into NP 128-bit shift
    // registers, so the step loop itself issues no loads: memory operations retire in issue order,
    // and a load issued between two steps' store bursts is only served after every older store has
    // drained (about 4 us in mid-launch).  NP is as many registers as the usual 25-step episode needs,
    // within a budget of 84 loaded words held at once (more costs occupancy in large batches).  The
    // first fetch is issued before the state is set up, so that its latency overlaps the reset
    // draws, and all loads of a fetch are issued before any is consumed.  Ints beyond the kept bits
    // saturate: a move outside 1..4 is a no-op anyway (environment.py:178-181) and a comm int outside
    // 0..dim_c-1 selects no word.  Columns follow the reference's action_n: per agent its move (if it is
    // movable), then its comm word (if it speaks).
    constexpr int BITS = Cfg::BITS, ABITS = BITS * NA;
    constexpr unsigned AMASK = (1u << BITS) - 1u;
    constexpr int CH = (MODE == MODE_ROLLOUT) ? (128 / ABITS) : 1;
    constexpr int NP_EPISODE = (25 + CH - 1) / CH, NP_BUDGET = Cfg::template action_words<T>() / (CH * NA) > 0 ? Cfg::template action_words<T>() / (CH * NA) : 1;
    constexpr int NP = (MODE == MODE_ROLLOUT) ? (NP_EPISODE < NP_BUDGET ? NP_EPISODE : NP_BUDGET) : 1;
    static_assert(ABITS <= 64, "one step's actions must fit 64 bits");
    unsigned __int128 fetched[NP] = {};
    unsigned raw[NP * CH][NA];
    auto load_actions = [&](int t0) {
#pragma unroll
        for (int k = 0; k < NP * CH; ++k) {
            const long long tt = (t0 + k < n_steps) ? t0 + k : n_steps - 1;     // past the end: re-read the last step
#pragma unroll
            for (int j = 0; j < NA; ++j) raw[k][j] = (unsigned)actions[(tt * B + b) * NA + j];
        }
    };
    auto pack_actions = [&]() {
        // pin the first use of the loaded words here, so that nothing waits on them earlier
#pragma unroll
        for (int k = 0; k < NP * CH; ++k) {
#pragma unroll
            for (int j = 0; j < NA; ++j) asm volatile("" : "+v"(raw[k][j]));
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            fetched[q] = 0;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int col = 0; col < NA; ++col) {
                    const unsigned v = raw[q * CH + k][col];
                    fetched[q] |= (unsigned __int128)(v < AMASK ? v : AMASK) << (k * ABITS + BITS * col);
                }
            }
        }
    };
    if (MODE != MODE_OBSERVE && n_steps > 0) load_actions(0);
    V2<T> pos[E];
    V2<T> vel[A];
    T comm[NC > 0 ? NC : 1];
    int aux[NX > 0 ? NX : 1] = {0};
#pragma unroll
    for (int k = 0; k < NX; ++k) {      // goal landmark(s): drawn with the reset, otherwise the caller's
        if (MODE == MODE_ROLLOUT && rs.enabled) {
            aux[k] = Cfg::DYNAMIC ? Cfg::initial_aux(rs, (unsigned long long)b, k) : reset_aux(rs, (unsigned long long)b, k, Cfg::L);
            if (valid) aux_g[(long long)k * B + b] = aux[k];
        } else {
            aux[k] = aux_g[(long long)k * B + b];
        }
    }
    // Scenario.reset_world positions; a DYNAMIC scenario may redraw some until they are free
    auto draw_world = [&](const ResetArgs& r) {
        if (Cfg::DYNAMIC) {
            V2<double> d[E];
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = reset_draw(r, (unsigned long long)b, e, e < A);
            Cfg::fix_reset(r, (unsigned long long)b, d, p);
#pragma unroll
            for (int e = 0; e < E; ++e) { pos[e].x = (T)d[e].x; pos[e].y = (T)d[e].y; }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const V2<double> d = reset_draw(r, (unsigned long long)b, e, e < A);
                pos[e].x = (T)d.x;
                pos[e].y = (T)d.y;
            }
        }
    };
    if (MODE == MODE_ROLLOUT && rs.enabled) {
        draw_world(rs);
#pragma unroll
        for (int i = 0; i < A; ++i) { vel[i].x = T(0); vel[i].y = T(0); }
        if (valid) {     // landmarks are not written back by the step loop: publish them now
#pragma unroll
            for (int e = A; e < E; ++e) {
                pos_g[(long long)(2 * e) * B + b] = pos[e].x;
                pos_g[(long long)(2 * e + 1) * B + b] = pos[e].y;
            }
        }
    } else {
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
    }
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) comm[c] = (MODE == MODE_OBSERVE) ? comm_g[(long long)c * B + b] : T(0);
    }
    constexpr bool want_info = INFO;

    int to_reset = (MODE == MODE_ROLLOUT && rs.enabled && rs.period > 0) ? rs.period : 0x7fffffff;     // steps until the next episode starts
    unsigned long long episode = rs.episode;
    // obs0_g (episode launches, optional): the observation of every freshly reset state, what
    // MultiAgentEnv.reset returns (environment.py:122-127).  It is produced by a pass through the step body
    // that skips the physics, the rewards and post_step ("pre"), before the episode's first step.
    const bool want_obs0 = (MODE == MODE_ROLLOUT) && rs.enabled && obs0_g != nullptr;
    bool fresh = want_obs0;
    long long ep_idx = 0;           // episodes started so far in this launch, minus one
    for (int t0 = 0; t0 < n_steps;) {
      if (MODE != MODE_OBSERVE) {
          if (t0 > 0) load_actions(t0);
          if (MODE == MODE_ROLLOUT && to_reset == 0) {
              // next episode of a multi-episode launch: Scenario.reset_world again, while the action words arrive
              episode += 1;
              ep_idx += 1;
              fresh = want_obs0;
              to_reset = rs.period;
              ResetArgs nr = rs;
              nr.episode = episode;
              draw_world(nr);
#pragma unroll
              for (int i = 0; i < A; ++i) { vel[i].x = T(0); vel[i].y = T(0); }
              if (NC > 0) {
#pragma unroll
                  for (int c = 0; c < NC; ++c) comm[c] = T(0);
              }
#pragma unroll
              for (int k = 0; k < NX; ++k) {
                  aux[k] = Cfg::DYNAMIC ? Cfg::initial_aux(nr, (unsigned long long)b, k) : reset_aux(nr, (unsigned long long)b, k, Cfg::L);
                  if (valid) aux_g[(long long)k * B + b] = aux[k];
              }
              if (valid) {
#pragma unroll
                  for (int e = A; e < E; ++e) {
                      pos_g[(long long)(2 * e) * B + b] = pos[e].x;
                      pos_g[(long long)(2 * e + 1) * B + b] = pos[e].y;
                  }
              }
          }
          pack_actions();
      }
      int nc = (n_steps - t0 < NP * CH) ? n_steps - t0 : NP * CH;
      nc = nc < to_reset ? nc : to_reset;       // a chunk of fetched actions ends with its episode
      to_reset = (to_reset == 0x7fffffff) ? to_reset : to_reset - nc;
      if (t0 == 0) STAMP(1);
      unsigned __int128 packed = 0;
      for (int kc = fresh ? -1 : 0; kc < nc; ++kc) {
        const bool pre = (MODE == MODE_ROLLOUT) && kc < 0;      // the reset state's observation only
        const int t = t0 + kc;
        if (!pre && kc % CH == 0) {          // next shift register (uniform select, NP is small)
            const int q = kc / CH;
            packed = fetched[0];
#pragma unroll
            for (int j = 1; j < NP; ++j) if (q == j) packed = fetched[j];
        }
        if (MODE != MODE_OBSERVE && !pre) {
            const unsigned long long cur = (unsigned long long)packed;
            packed >>= ABITS;
            int act[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) act[j] = (int)((cur >> (BITS * j)) & AMASK);
            // ---- MultiAgentEnv._set_action: int actions -> forces and comm one-hot ----
            V2<T> force[A];
            int col = 0, slot = 0;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                if (Cfg::movable(i)) {
                    force[i] = decode_move<T>(act[col++], p.sens[Cfg::rep(i)]);     // apply_action_force: u + 0
                    if (Cfg::NEW_CORE) { force[i].x = p.force_scale[Cfg::rep(i)] * force[i].x; force[i].y = p.force_scale[Cfg::rep(i)] * force[i].y; }
                } else {
                    force[i].x = T(0); force[i].y = T(0);
                }
                if (Cfg::speaks(i)) {
                    const int word = act[col++];
#pragma unroll
                    for (int c = 0; c < DC; ++c) comm[slot * DC + c] = (word == c) ? T(1) : T(0);   // update_agent_state
                    ++slot;
                }
            }
            if constexpr (NOISE) {
                // Agent.u_noise: p_force = action.u (scaled) + randn(2) * u_noise (core.py:165, new_core.py:213-214);
                // Agent.c_noise: state.c = action.c + randn(dim_c) * c_noise (core.py:211).  An agent whose level is
                // 0 (None in the reference) takes no draw.
                int cslot = 0;
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    if (Cfg::movable(i) && p.u_noise[i] != T(0)) {
                        const V2<T> z = noise_pair<T>(nz, t, B, b, i);
                        force[i].x = force[i].x + z.x * p.u_noise[i];
                        force[i].y = force[i].y + z.y * p.u_noise[i];
                    }
                    if (Cfg::speaks(i)) {
                        if (p.c_noise[i] != T(0)) {
#pragma unroll
                            for (int c = 0; c < DC; ++c) {
                                const int j = cslot * DC + c;           // value j of the row's comm part: pair A + j / 2
                                const V2<T> z = noise_pair<T>(nz, t, B, b, A + j / 2);
                                comm[j] = comm[j] + ((j & 1) ? z.y : z.x) * p.c_noise[i];
                            }
                        }
                        ++cslot;
                    }
                }
            }
            // ---- World.apply_environment_force: pairs in the reference's (a, b) order ----
#ifdef MPE_ABLATE_SMALL_PHYS        // diagnostic build: no contact forces
            if (n_steps == 0x7fffffff)
#endif
#pragma unroll
            for (int a = 0; a < E; ++a) {
#pragma unroll
                for (int c = a + 1; c < E; ++c) {
                    if (!Cfg::collides(a) || !Cfg::collides(c)) continue;
                    const bool ma = a < A && Cfg::movable(a), mc = c < A && Cfg::movable(c);
                    if (!ma && !mc) continue;         // neither entity moves
                    const V2<T> f = contact_force<T>(pos[a], pos[c], p.size[Cfg::rep(a)] + p.size[Cfg::rep(c)], p.contact_force, p.contact_margin);
                    if (Cfg::NEW_CORE && ma && mc) {
                        // two movable entities share the force by their masses (new_core.py:290-294)
                        const T ratio = p.mass[Cfg::rep(c)] / p.mass[Cfg::rep(a)], back = -(T(1) / ratio);
                        force[a].x = ratio * f.x + force[a].x; force[a].y = ratio * f.y + force[a].y;
                        force[c].x = back * f.x + force[c].x; force[c].y = back * f.y + force[c].y;
                    } else {
                        if (ma) { force[a].x = f.x + force[a].x; force[a].y = f.y + force[a].y; }
                        if (mc) { force[c].x = -f.x + force[c].x; force[c].y = -f.y + force[c].y; }
                    }
                }
            }
            // ---- wall contact for every movable entity, after its entity contacts (core.py:182-188) ----
#ifdef MPE_ABLATE_SMALL_PHYS        // ... and no wall forces
            if (n_steps == 0x7fffffff)
#endif
            if (Cfg::N_WALLS > 0) {
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    if (!Cfg::movable(i)) continue;
#pragma unroll
                    for (int w = 0; w < Cfg::N_WALLS; ++w) {
                        // A wall further away than this contributes nothing: in float32 the softplus of
                        // -(dist - dist_min) / k is exactly 0 beyond 16.7 k, in float64 it is below exp(-60),
                        // i.e. a force under 1e-27 (the same reach as the contact loop of the large worlds)
                        const T perp = p.wall_vertical[w] ? pos[i].x : pos[i].y;
                        const T cut = (sizeof(T) == 4 ? T(18) : T(60)) * p.contact_margin;
                        if (t_abs<T>(perp - p.wall_axis_pos[w]) - (p.size[Cfg::rep(i)] + T(0.5) * p.wall_width[w]) > cut) continue;
                        V2<T> wf;
                        if (wall_force_fast<T>(pos[i], p.size[Cfg::rep(i)], p.ghost[Cfg::rep(i)] != 0, p.wall_vertical[w] != 0, p.wall_hard[w] != 0, p.wall_axis_pos[w],
                                          p.wall_end0[w], p.wall_end1[w], p.wall_width[w], p.contact_force, p.contact_margin, wf)) {
                            force[i].x = force[i].x + wf.x;
                            force[i].y = force[i].y + wf.y;
                        }
                    }
                }
            }
            // ---- World.integrate_state ----
#pragma unroll
            for (int i = 0; i < A; ++i)
                if (Cfg::movable(i)) integrate<T, CLAMP>(pos[i], vel[i], force[i], p.mass[Cfg::rep(i)], p.max_speed[Cfg::rep(i)], p.one_minus_damping, p.dt);
        }

        STAMP_T(30);
        // ---- done_callback (environment.py:98,110-113,143-146): any(done_n) over the agents, for the one predicate a
        //      batch can carry -- an agent beyond the arena bound (mpe_set_done_bound); a NaN position compares false ----
        bool outside = false;
        if (MODE != MODE_OBSERVE && !pre && p.done_bound > T(0)) {
#pragma unroll
            for (int i = 0; i < A; ++i) outside = outside || t_abs<T>(pos[i].x) > p.done_bound || t_abs<T>(pos[i].y) > p.done_bound;
        }
        // ---- scenario callbacks into the wave's LDS tiles ----
        const bool want_score = !pre && ((MODE != MODE_OBSERVE) || (rew_g != nullptr));   // observe: rewards on request
        bool finished = false;
        if (EPW == 64 || lane < EPW) {
            T* const rrow = RewTile::row(tile_rew, tile_lane);
            if (want_score)
                Cfg::template score<T>(pos, comm, aux, p, rrow, want_info, IiTile::row(tile_ii, tile_lane), IfTile::row(tile_if, tile_lane));
            if (Cfg::SHARED && MODE != MODE_OBSERVE && !pre) {
                // collaborative world: step hands every agent np.sum(reward_n) (environment.py:103-105);
                // numpy adds fewer than 8 values left to right
                static_assert(!Cfg::SHARED || A < 8, "np.sum switches to pairwise blocks at 8 values");
                T total = rrow[0];
#pragma unroll
                for (int i = 1; i < A; ++i) total = total + rrow[i];
#pragma unroll
                for (int i = 0; i < A; ++i) rrow[i] = total;
            }
            if (MODE == MODE_STEP && au.step_count != nullptr) {
                // ---- episode bookkeeping (mpe_step_auto): returns, step counter, time-limit reset ----
                T* ep = (T*)au.ep_return + b * A;
                const int cnt = au.step_count[b] + 1;
                finished = cnt >= au.max_steps || outside;       // the time limit, or the env's own done
                T total[A];
#pragma unroll
                for (int i = 0; i < A; ++i) total[i] = ep[i] + rrow[i];
                if (valid) {
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        ep[i] = finished ? T(0) : total[i];
                        if (finished) ((T*)au.last_return)[b * A + i] = total[i];
                    }
                    au.step_count[b] = finished ? 0 : cnt;
                }
                if (finished) {       // Scenario.reset_world with the next episode's draws; obs below is the fresh one
                    const int epi = au.episode_count[b] + 1;
                    if (valid) au.episode_count[b] = epi;
                    const ResetArgs nr{1, au.seed, (unsigned long long)epi, au.landmark_lim, 0, au.env_base};
                    draw_world(nr);
#pragma unroll
                    for (int i = 0; i < A; ++i) { vel[i].x = T(0); vel[i].y = T(0); }
                    if (NC > 0) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) comm[c] = T(0);
                    }
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        aux[k] = Cfg::DYNAMIC ? Cfg::initial_aux(nr, (unsigned long long)b, k) : reset_aux(nr, (unsigned long long)b, k, Cfg::L);
                        if (valid) aux_g[(long long)k * B + b] = aux[k];
                    }
                    if (valid) {
#pragma unroll
                        for (int e = A; e < E; ++e) {
                            pos_g[(long long)(2 * e) * B + b] = pos[e].x;
                            pos_g[(long long)(2 * e + 1) * B + b] = pos[e].y;
                        }
                    }
                }
            }
            if constexpr (Cfg::OBS_CHUNKS == 1 && RP == 0 && !COOP) Cfg::template observe<T>(pos, vel, comm, aux, p, ObsTile::row(tile_obs, tile_lane));
        }
        STAMP_T(31);
        const long long row0 = (long long)t * B + env0;
        T* const obs_dst = pre ? obs0_g + (ep_idx * B + env0) * OT : obs_g + row0 * OT;
        auto flush_small = [&]() {
            // (a kernel on a tight register budget: an opaque copy of the lane keeps the flushes' addresses per-step
            // work; hoisted out of the step loop they are spilled, and every reload waits for the wave's older stores)
            int lane_f = lane;
            if constexpr (MODE == MODE_ROLLOUT && FULL && !INFO && !NOISE && Cfg::template waves_per_simd<T>() > 1) asm volatile("" : "+v"(lane_f));
            if (want_score) {
#ifdef MPE_ABLATE_REW
                // Diagnostic build only (tools/pmc_headline.py): the reward rows stay computed but are not stored, behind
                // a condition the compiler cannot fold, to price that output stream with counters.
                if (n_steps == 0x7fffffff)
#endif
                RewTile::template flush<FULL, NT>(tile_rew, rew_g + row0 * A, lane_f, nvalid);
                if (MODE == MODE_ROLLOUT && FULL && EPW == 64 && p.done_bound > T(0)) {
                    done_g[row0 + lane_f] = outside ? 1 : 0;         // a batch with a done predicate: every env its own flag
                } else if (MODE == MODE_ROLLOUT && FULL && EPW == 64) {
                    // No scenario ends an episode (environment.py:143-146) and a rollout has no time limit, so
                    // done is 0 throughout.  A wave's 64 flags are half a 128-byte line: neighbouring waves take
                    // turns (even steps the even wave, odd steps the odd one) to write both halves as one
                    // full line; a last wave without a neighbour writes its own half.
                    const long long pair0 = env_begin + (long long)(tile & ~1u) * 64;
                    const bool paired = pair0 + 128 <= env_end;
                    if (paired) {
                        if (((unsigned)t & 1u) == (tile & 1u) && lane_f < 32)
                            reinterpret_cast<unsigned*>(done_g + (long long)t * B + pair0)[lane_f] = 0u;
                    } else {
                        done_g[row0 + lane_f] = 0;
                    }
                } else if (MODE != MODE_OBSERVE && valid) {
                    put<NT>(&done_g[row0 + lane_f], (uint8_t)((finished || outside) ? 1 : 0));
                }
                if (want_info) {
                    IiTile::template flush<FULL, NT>(tile_ii, info_i_g + row0 * NI, lane_f, nvalid);
                    if (Cfg::NF > 0 && info_f_g != nullptr) IfTile::template flush<FULL, NT>(tile_if, info_f_g + row0 * NF, lane_f, nvalid);
                }
            }
        };
        if constexpr (COOP) {
            __syncthreads();
            flush_small();
            __syncthreads();    // the passes below write their rows over these tiles
        }
        if constexpr (Cfg::OBS_LOOP) {
            // one copy of the per-agent observation code, looped over the agents; every lane reads back only
            // the column of the staging area it wrote itself, so no barrier is needed before the first read
            const bool active = EPW == 64 || lane < EPW;
            const Staged<T, E, A> staged{stage_lds, lane};
            if (active) staged.put(pos, vel);
            const auto ctx = Cfg::template observe_prepare<T>(pos, p);
#pragma unroll 1
            for (int i = 0; i < A; ++i) {
                const int W = Cfg::chunk_width(i), S = ChunkTile<T, Cfg>::STRIDE;
                if (active) Cfg::template observe_agent<T>(i, staged, aux, ctx, tile_obs + tile_lane * S);
                __syncthreads();
                // pairs of values: every chunk starts on an even column of an even-length row, so a pair is an
                // 8-byte (float32) or 16-byte (float64) aligned access in the tile and in the array alike, and the
                // flush takes half the instructions of a value-per-lane one
                typedef T Pair __attribute__((ext_vector_type(2)));
                static_assert(!Cfg::OBS_LOOP || (OT % 2 == 0 && S % 2 == 0 && Cfg::CHUNK_MIN % 2 == 0 && Cfg::CHUNK_MAX % 2 == 0), "pairs need even widths");
                static_assert(!Cfg::OBS_LOOP || (Cfg::CHUNK_MIN / 2 > 32 && Cfg::CHUNK_MAX / 2 < 64), "a 64-pair stride crosses one or two row ends");
                Pair* gbase = reinterpret_cast<Pair*>(obs_dst + Cfg::chunk_offset(i));
                const Pair* tile2 = reinterpret_cast<const Pair*>(tile_obs);
                const int W2 = W / 2, n = nvalid * W2;
                int r = 0, j = lane;
                if (j >= W2) { j -= W2; ++r; }
#pragma unroll 4
                for (int g = lane; g < n; g += 64) {
                    put<NT>(&gbase[r * (OT / 2) + j], tile2[r * (S / 2) + j]);
                    j += 64 - W2;
                    ++r;
                    if (j >= W2) { j -= W2; ++r; }
                }
                __syncthreads();
            }
        } else if constexpr (COOP) {
            // Whole rows, TROWS = 8 worlds at a time, with the lanes regrouped as (world of the pass, agent): lane
            // ag * 8 + k observes for agent ag in world pass * 8 + k, whose state it reads from the staging area and
            // whose int state and per-step context it fetches from that world's lane (lanes beyond 8 * A idle).  One
            // observe_agent per lane and pass: the same arithmetic as a loop over the agents, but the tile is a byte
            // image of 8 consecutive rows and leaves as 16-byte vectors, whole 128-byte lines.
            const CoopStage own{stage_lds, lane};
            own.put(pos, vel);
            const auto ctx = Cfg::template observe_prepare<T>(pos, p);
            __syncthreads();
            STAMP_T(32);
            const int k_in = lane % TROWS, ag = lane / TROWS;
#pragma unroll 1
            for (int pass = 0; pass < 64 / TROWS; ++pass) {
                const int wl = pass * TROWS + k_in;
                int auxw[NX > 0 ? NX : 1];
#pragma unroll
                for (int k = 0; k < NX; ++k) auxw[k] = __shfl(aux[k], wl, 64);
                const auto ctxw = Cfg::ctx_from_lane(ctx, wl);
                const CoopStage theirs{stage_lds, wl};
                if (A == 8 || ag < A) Cfg::template observe_agent<T>(ag, theirs, auxw, ctxw, tile_obs + k_in * OT + Cfg::chunk_offset(ag));
                __syncthreads();
                int nv = nvalid - pass * TROWS;
                nv = nv < 0 ? 0 : (nv > TROWS ? TROWS : nv);
#ifdef MPE_ABLATE_COOP_FLUSH        // diagnostic build: the passes' arithmetic without their stores
                if (n_steps == 0x7fffffff)
#endif
                ObsTile::template flush<FULL, NT>(tile_obs, obs_dst + (long long)(pass * TROWS) * OT, lane, nv);
                __syncthreads();
                STAMP_T(33 + pass);
            }
        } else if constexpr (RP > 0) {
            // Whole rows, TROWS worlds at a time: every lane evaluates the observation in each pass, the lanes of
            // the pass's worlds write their rows, and the tile -- a byte image of TROWS consecutive rows of the
            // env-major array -- leaves as 16-byte vectors.  The arithmetic is repeated RP times, the stores are
            // full 128-byte lines: a row that leaves in per-agent column pieces writes every line in several
            // partial accesses, which alone takes as long as the whole launch (simple_predator_prey, 131 072
            // worlds: pieces 1.36-1.46 ms, whole lines 0.73 ms with no arithmetic at all).
#pragma unroll 1
            for (int pass = 0; pass < RP; ++pass) {
                const bool mine = (EPW == 64 || lane < EPW) && (lane / TROWS) == pass;
                Cfg::template observe_pass<T>(pos, vel, comm, aux, p, tile_obs + (lane % TROWS) * OT, mine);
                __syncthreads();
                int nv = nvalid - pass * TROWS;
                nv = nv < 0 ? 0 : (nv > TROWS ? TROWS : nv);
                ObsTile::template flush<FULL, NT>(tile_obs, obs_dst + (long long)(pass * TROWS) * OT, lane, nv);
                __syncthreads();
            }
        } else if constexpr (Cfg::OBS_CHUNKS > 1)
        {
            // The flushes' index arithmetic depends on the lane alone, so the optimiser would compute it once
            // before the step loop and keep every offset live throughout; an opaque copy of the lane makes it
            // per-step work and frees those registers.
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));
            ObsChunks<T, Cfg, 0, EPW, FULL, NT>::run(pos, vel, comm, aux, Cfg::template observe_prepare<T>(pos, p), tile_obs, obs_dst, lane_s,
                                                 tile_lane, EPW == 64 || lane < EPW, nvalid);
        }
        // post_step_callback runs after the callbacks (environment.py:107-108); a world that was just
        // reset starts its episode untouched
        if (Cfg::DYNAMIC && MODE != MODE_OBSERVE && !pre && !finished && p.skip_post_step == 0 && (EPW == 64 || lane < EPW))
            Cfg::post_step(pos, aux, p, p.env_base + (unsigned long long)b);
        __syncthreads();
        STAMP_T(41);
        if constexpr (Cfg::OBS_CHUNKS == 1 && RP == 0 && !COOP) ObsTile::template flush<FULL, NT>(tile_obs, obs_dst, lane, nvalid);
        if constexpr (!COOP) flush_small();
        if (MODE == MODE_ROLLOUT) {
            if (t >= 0 && t < 28) STAMP(2 + t);
            __syncthreads();   // tiles are rewritten by the next step
        }
      }
      fresh = false;
      t0 += nc;
    }

    STAMP(60);
    if (MODE != MODE_OBSERVE && valid) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            put<NT>(&pos_g[(long long)(2 * i) * B + b], pos[i].x);
            put<NT>(&pos_g[(long long)(2 * i + 1) * B + b], pos[i].y);
            put<NT>(&vel_g[(long long)(2 * i) * B + b], vel[i].x);
            put<NT>(&vel_g[(long long)(2 * i + 1) * B + b], vel[i].y);
        }
        if (NC > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) put<NT>(&comm_g[(long long)c * B + b], comm[c]);
        }
        if (Cfg::DYNAMIC) {     // post_step moves landmarks and rewrites the scenario's int state
#pragma unroll
            for (int e = A; e < E; ++e) {
                put<NT>(&pos_g[(long long)(2 * e) * B + b], pos[e].x);
                put<NT>(&pos_g[(long long)(2 * e + 1) * B + b], pos[e].y);
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) put<NT>(&aux_g[(long long)k * B + b], aux[k]);
        }
    }
#ifdef MPE_STAMPS
    STAMP(61);                                      // every store of the wave issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(62);                                      // ... and acknowledged
#endif
}

// Scenario.post_step alone, for a caller that runs it as MultiAgentEnv's post_step_callback between its
// own steps (environment.py:107-108): one lane per world applies the scenario's pick-up / hand-over /
// respawn rules to the landmark positions and the aux state, exactly as the fused step does.
template <typename T, class Cfg>
__global__ __launch_bounds__(64) void post_step_kernel(const SmallParams<T, Cfg::E, Cfg::A> p, T* __restrict__ pos_g, int* __restrict__ aux_g) {
    constexpr int A = Cfg::A, E = Cfg::E, NX = Cfg::N_AUX;
    const long long B = p.n_envs;
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    V2<T> pos[E];
    int aux[NX > 0 ? NX : 1] = {0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
        pos[e].x = pos_g[(long long)(2 * e) * B + b];
        pos[e].y = pos_g[(long long)(2 * e + 1) * B + b];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) aux[k] = aux_g[(long long)k * B + b];
    Cfg::post_step(pos, aux, p, p.env_base + (unsigned long long)b);
#pragma unroll
    for (int e = A; e < E; ++e) {
        pos_g[(long long)(2 * e) * B + b] = pos[e].x;
        pos_g[(long long)(2 * e + 1) * B + b] = pos[e].y;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) aux_g[(long long)k * B + b] = aux[k];
}

// ------------------------------------------------------------------------------------------
template <typename T, class Cfg>
static SmallParams<T, Cfg::E, Cfg::A> make_params(const HostSpec& s, long long n_envs, unsigned long long env_base) {
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
    p.env_base = env_base;
    p.done_bound = T(0);
    p.skip_post_step = 0;
    p.deal = XcdDeal{nullptr, {0, 0}, 0};
    for (int e = 0; e < Cfg::E; ++e) p.size_d[e] = s.size[e];
    for (int i = 0; i < Cfg::A; ++i) { p.u_noise[i] = (T)s.u_noise[i]; p.c_noise[i] = (T)s.c_noise[i]; }
    for (int i = 0; i < Cfg::A; ++i) { p.force_scale[i] = (T)s.force_scale[i]; p.ghost[i] = s.ghost[i]; }
    for (int w = 0; w < SMALL_MAX_WALLS; ++w) {
        const bool on = w < s.n_walls;
        p.wall_axis_pos[w] = on ? (T)s.wall_axis_pos[w] : T(0); p.wall_end0[w] = on ? (T)s.wall_end0[w] : T(0);
        p.wall_end1[w] = on ? (T)s.wall_end1[w] : T(0); p.wall_width[w] = on ? (T)s.wall_width[w] : T(0);
        p.wall_vertical[w] = on ? s.wall_vertical[w] : 0; p.wall_hard[w] = on ? s.wall_hard[w] : 0;
    }
    return p;
}

template <typename T, class Cfg, bool CLAMP, int MODE, bool FULL, bool INFO, int EPW, bool NOISE = false>
static void launch_range(const SmallParams<T, Cfg::E, Cfg::A>& p, long long env_begin, long long env_end, const StepArgs& a, hipStream_t st) {
    if (env_end <= env_begin) return;
    dim3 grid((unsigned)((env_end - env_begin + EPW - 1) / EPW)), block(64);
    SmallParams<T, Cfg::E, Cfg::A> q = p;
    // A launch of many rounds of waves ends on the odd XCDs alone (XcdDeal, mpe_launch.hpp); dealing the tiles 9 : 7 lets
    // the XCDs end together: simple_spread 1.050 -> 1.008 ms at 1 048 576 worlds, 4.235 -> 4.078 ms at 4 194 304,
    // simple_tag 1.567 -> 1.458 ms at 1 048 576, float64 2.011 -> 1.932 ms.  Not below four rounds (the BASELINE sizes
    // lose: 65 536 worlds 59.8 -> 65.2 us, simple_tag at 131 072 worlds 184 -> 196 us; those launches half-fit the
    // Infinity Cache and are not bound the same way) and not for the cooperative-pass scenarios (collector at 262 144
    // worlds 3.6 -> 4.2 ms).  Even shares: the done flags' pairs of tiles stay on one side.
    if (MODE == MODE_ROLLOUT && FULL && !Cfg::OBS_COOP && grid.x >= 8192u) {
        q.deal = make_xcd_deal(a.xcd_phase, (long long)grid.x, 2);
        if (q.deal.phase != nullptr) {
            grid = dim3((unsigned)(8 * q.deal.share[0]));
            if (launch_xcd_phase(a.xcd_phase, st) != hipSuccess) return;        // (hipGetLastError reports it to the caller)
        }
    }
    hipLaunchKernelGGL((small_kernel<T, Cfg, CLAMP, MODE, FULL, INFO, EPW, NOISE>), grid, block, 0, st, q, a.reset, a.au, env_begin, env_end, a.actions, (T*)a.pos,
                       (T*)a.vel, (T*)a.comm, (T*)a.obs, (T*)a.rew, a.done, a.info_i, (T*)a.info_f, a.aux, (T*)a.obs0, a.noise, a.n_steps);
}

// Full tiles run the straight-line FULL instantiation; the ragged tail (and every wave when the
// per-step output blocks are not 16-byte aligned, i.e. B % 4 != 0) runs the guarded 64-wide one.
template <typename T, class Cfg, bool CLAMP, int MODE, bool INFO, int EPW>
static void launch_epw(const SmallParams<T, Cfg::E, Cfg::A>& p, long long B, const StepArgs& a, hipStream_t st) {
    if constexpr (MODE != MODE_OBSERVE) {
        if (a.noise.enabled) {      // a rare configuration: every wave runs the guarded instantiation with the noise terms
            launch_range<T, Cfg, CLAMP, MODE, false, INFO, (EPW < 64 ? EPW : 64), true>(p, 0, B, a, st);
            return;
        }
    }
    const long long n_full = ((B & 3) == 0) ? (B / EPW) * EPW : 0;
    launch_range<T, Cfg, CLAMP, MODE, true, INFO, EPW>(p, 0, n_full, a, st);
    launch_range<T, Cfg, CLAMP, MODE, false, INFO, (EPW < 64 ? EPW : 64)>(p, n_full, B, a, st);
}

// Every launch uses 64-env tiles and one-wave workgroups.  Measured alternatives at 65 536 envs
// (one wave per SIMD): 32- / 16-env tiles lose 15 % / 50 % (a wave's instruction stream is as long
// whatever its tile, so narrower tiles only multiply the instructions issued); a two-wave workgroup
// whose second wave does nothing but stream finished tiles out of a double-buffered LDS tile, and
// issuing step t-1's stores in slices between the phases of step t, both change nothing: per-wave
// timestamps show every wave store-bound from step 2 on (1.4-2.8 us per step against 0.9 us of
// arithmetic), i.e. the launch is limited by how fast the fabric takes the rows, not by a wave
// stalling on its own burst.  Weighting the tiles per XCD (odd XCDs finish 15-20 % later) only moved
// the slow spot to other XCDs.
template <typename T, class Cfg, bool CLAMP, int MODE, bool INFO>
static void launch_mode(const SmallParams<T, Cfg::E, Cfg::A>& p, long long B, const StepArgs& a, hipStream_t st) {
    launch_epw<T, Cfg, CLAMP, MODE, INFO, Cfg::template epw<T>()>(p, B, a, st);
}

template <typename T, class Cfg, bool CLAMP>
static hipError_t launch_cfg(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    auto p = make_params<T, Cfg>(s, B, a.env_base);
    p.done_bound = (T)a.done_bound;
    p.skip_post_step = a.skip_post_step;
    const bool info = a.info_i != nullptr;
    switch (mode) {
        case MODE_STEP:
            info ? launch_mode<T, Cfg, CLAMP, MODE_STEP, true>(p, B, a, st) : launch_mode<T, Cfg, CLAMP, MODE_STEP, false>(p, B, a, st);
            break;
        case MODE_ROLLOUT:
            info ? launch_mode<T, Cfg, CLAMP, MODE_ROLLOUT, true>(p, B, a, st) : launch_mode<T, Cfg, CLAMP, MODE_ROLLOUT, false>(p, B, a, st);
            break;
        default:
            info ? launch_mode<T, Cfg, CLAMP, MODE_OBSERVE, true>(p, B, a, st) : launch_mode<T, Cfg, CLAMP, MODE_OBSERVE, false>(p, B, a, st);
            break;
    }
    return hipGetLastError();
}

template <typename T, class Cfg>
static hipError_t launch_post_step(const HostSpec& s, long long B, const StepArgs& a, hipStream_t st) {
    const auto p = make_params<T, Cfg>(s, B, a.env_base);
    hipLaunchKernelGGL((post_step_kernel<T, Cfg>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, p, (T*)a.pos, a.aux);
    return hipGetLastError();
}

enum { MODE_POST_STEP = 3 };     // launch_small only: Scenario.post_step alone (a.pos, a.aux)

// The instantiations are spread over several translation units, (scenario group) x (float type), so that they
// compile in parallel: mpe_small_<type>_<group>.hip each define one launch_small_<type>_<group>().
enum { GROUP_CORE = 0,          // simple_spread 3/3, simple_tag, simple_world_comm
       GROUP_CLASSIC = 1,       // simple, simple_adversary, simple_push, simple_speaker_listener, simple_reference, simple_crypto
       GROUP_PREDATOR_PREY = 2,
       GROUP_COLLECTOR = 3 };

template <typename T, int GROUP>
hipError_t launch_small_group(const HostSpec& s, long long B, int mode, const StepArgs& a, hipStream_t st) {
    if (mode == MODE_POST_STEP) {
        if constexpr (GROUP == GROUP_PREDATOR_PREY) return launch_post_step<T, PredatorPreyCfg>(s, B, a, st);
        if constexpr (GROUP == GROUP_COLLECTOR) return launch_post_step<T, CollectorCfg>(s, B, a, st);
        return hipErrorInvalidValue;
    }
    if constexpr (GROUP == GROUP_CORE) {
        const bool clamp = s.any_speed_limit();
        switch (s.scenario) {
            case MPE_SCN_SPREAD:
                return clamp ? launch_cfg<T, SpreadCfg<3, 3>, true>(s, B, mode, a, st) : launch_cfg<T, SpreadCfg<3, 3>, false>(s, B, mode, a, st);
            case MPE_SCN_TAG:
                return launch_cfg<T, TagCfg, true>(s, B, mode, a, st);
            case MPE_SCN_WORLD_COMM:
                return launch_cfg<T, WorldCommCfg, true>(s, B, mode, a, st);
        }
    } else if constexpr (GROUP == GROUP_CLASSIC) {
        // no speed limit in these scenarios; an infinite max_speed makes the clamp a no-op
        switch (s.scenario) {
            case MPE_SCN_SIMPLE:
                return launch_cfg<T, SimpleCfg, true>(s, B, mode, a, st);
            case MPE_SCN_ADVERSARY:
                return launch_cfg<T, AdversaryCfg, true>(s, B, mode, a, st);
            case MPE_SCN_PUSH:
                return launch_cfg<T, PushCfg, true>(s, B, mode, a, st);
            case MPE_SCN_SPEAKER_LISTENER:
                return launch_cfg<T, SpeakerListenerCfg, true>(s, B, mode, a, st);
            case MPE_SCN_REFERENCE:
                return launch_cfg<T, ReferenceCfg, true>(s, B, mode, a, st);
            case MPE_SCN_CRYPTO:
                return launch_cfg<T, CryptoCfg, true>(s, B, mode, a, st);
        }
    } else if constexpr (GROUP == GROUP_PREDATOR_PREY) {
        if (s.scenario == MPE_SCN_PREDATOR_PREY) return launch_cfg<T, PredatorPreyCfg, true>(s, B, mode, a, st);
    } else {
        if (s.scenario == MPE_SCN_COLLECTOR) return launch_cfg<T, CollectorCfg, true>(s, B, mode, a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace mpe
