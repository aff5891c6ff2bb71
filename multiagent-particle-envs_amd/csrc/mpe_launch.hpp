// Host-side description of a world and the launcher entry points shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mpe_hip.h"

namespace mpe {

constexpr int MAX_E = MPE_MAX_AGENTS + MPE_MAX_LANDMARKS;

// Validated copy of MpeSpec (host memory).
struct HostSpec {
    int scenario, A, L, dim_c, collaborative, physics;
    double dt, damping, contact_force, contact_margin;
    double size[MAX_E], mass[MAX_E];
    uint8_t movable[MAX_E], collide[MAX_E];
    double sens[MPE_MAX_AGENTS], max_speed[MPE_MAX_AGENTS];
    uint8_t silent[MPE_MAX_AGENTS], adversary[MPE_MAX_AGENTS], leader[MPE_MAX_AGENTS], ghost[MPE_MAX_AGENTS];
    double force_scale[MPE_MAX_AGENTS];     // action force = force_scale * u: 1 (core.py:166) or mass (* accel) (new_core.py:214)
    int n_walls;
    uint8_t wall_vertical[MPE_MAX_WALLS], wall_hard[MPE_MAX_WALLS];
    double wall_axis_pos[MPE_MAX_WALLS], wall_end0[MPE_MAX_WALLS], wall_end1[MPE_MAX_WALLS], wall_width[MPE_MAX_WALLS];
    double u_noise[MPE_MAX_AGENTS], c_noise[MPE_MAX_AGENTS];     // Agent.u_noise / c_noise, 0 when None (core.py:165,211)
    bool any_noise() const {
        for (int i = 0; i < A; ++i) if (u_noise[i] != 0.0 || c_noise[i] != 0.0) return true;
        return false;
    }
    int noise_stride() const {      // values per (step, env) row of a caller-supplied draw array
        int n = 2 * A;
        for (int i = 0; i < A; ++i) n += silent[i] ? 0 : dim_c;
        return (n + 1) & ~1;
    }
    int E() const { return A + L; }
    // rows of per-episode scenario state: the landmark indices reset_world draws
    int n_aux() const {
        switch (scenario) {
            case MPE_SCN_ADVERSARY: case MPE_SCN_PUSH: case MPE_SCN_SPEAKER_LISTENER: case MPE_SCN_CONSENSUS: return 1;
            case MPE_SCN_REFERENCE: case MPE_SCN_CRYPTO: return 2;
            case MPE_SCN_PREDATOR_PREY: case MPE_SCN_COLLECTOR: return 3;
            default: return 0;
        }
    }
    bool any_speed_limit() const {
        for (int i = 0; i < A; ++i) if (isfinite(max_speed[i])) return true;
        return false;
    }
};

// Per-entity table in device memory, read by the run-time driven (wave-per-world) kernels.
struct DevTable {
    double size[MAX_E], mass[MAX_E];
    double sens[MPE_MAX_AGENTS], max_speed[MPE_MAX_AGENTS], force_scale[MPE_MAX_AGENTS];
    double wall_axis_pos[MPE_MAX_WALLS], wall_end0[MPE_MAX_WALLS], wall_end1[MPE_MAX_WALLS], wall_width[MPE_MAX_WALLS];
    double u_noise[MPE_MAX_AGENTS];     // Agent.u_noise, 0 when None
    uint8_t collide[MAX_E], movable[MAX_E], silent[MPE_MAX_AGENTS], ghost[MPE_MAX_AGENTS];
    uint8_t wall_vertical[MPE_MAX_WALLS], wall_hard[MPE_MAX_WALLS];
    int n_walls;
    int new_core;       // contacts between movable entities are split by their mass ratio (new_core.py:290-294)
};

// On-device Scenario.reset_world fused into the start of a rollout launch (mpe_episode).
struct ResetArgs {
    int enabled;
    unsigned long long seed, episode;
    double landmark_lim;      // landmarks ~ U(-lim, lim)^2; agents always U(-1, 1)^2
    int period;               // > 0: the launch holds several episodes, reset again every `period` steps (episode + 1, + 2, ...)
    unsigned long long env_base;      // global index of the batch's first env: env b draws as env env_base + b of an unsharded batch
};

// Episode bookkeeping of mpe_step_auto (all pointers NULL = plain mpe_step).
struct AutoArgs {
    int32_t* step_count;      // [B] steps taken in the running episode
    int32_t* episode_count;   // [B] number of the running episode (keys the reset draws)
    void* ep_return;          // T[B][A] running return
    void* last_return;        // T[B][A] return of the last finished episode
    int max_steps;
    unsigned long long seed;
    double landmark_lim;
    unsigned long long env_base;      // as ResetArgs::env_base
};

// Where the standard-normal draws of Agent.u_noise / Agent.c_noise come from (mpe_math.hpp: noise_pair).
struct NoiseArgs {
    int enabled;                        // the world has an agent with u_noise or c_noise
    int stride;                         // values per (step, env) row of `draws`: 2A + S * dim_c, rounded up to even
    const void* draws;                  // T[n_steps][B][stride]: [2i + d] force noise of agent i, [2A + j] comm value j; or NULL
    unsigned long long seed, step0;     // generator key and the step number of the launch's first step
    unsigned long long env_base;        // the generator's env counter is env_base + b (the draws array is indexed by b)
};

struct StepArgs {
    const int32_t* actions;
    void *pos, *vel, *comm, *obs, *rew;
    uint8_t* done;
    int32_t* info_i;
    void* info_f;
    int n_steps;
    ResetArgs reset;
    AutoArgs au;
    int32_t* aux;             // [n_aux][B] per-episode scenario state (landmark indices), or NULL
    void* obs0;               // episode launches: T[n_episodes][B][obs_total] observation of each reset state, or NULL
    NoiseArgs noise;          // enabled only for worlds with u_noise / c_noise
    unsigned long long env_base;      // global index of the batch's first env (mpe_set_env_base): keys every on-device draw
    double done_bound;                // > 0: done_callback "an agent is outside [-bound, bound]^2" (mpe_set_done_bound); 0: none
    int skip_post_step;               // 1: a scenario's post_step is not applied by the stepping launches (mpe_set_post_step(h, 0))
    int* xcd_phase;           // the handle's phase word for an uneven deal over the XCDs (XcdDeal), or nullptr
};

// simple_spread.py:53 draws landmarks from U(-1,1); simple_tag.py:53 and simple_world_comm.py:106-112
// from U(-0.9,0.9).
inline double landmark_limit(const HostSpec& s) {
    if (s.scenario == MPE_SCN_COLLECTOR) return 0.95;        // collector.py:123-127
    return (s.scenario == MPE_SCN_TAG || s.scenario == MPE_SCN_WORLD_COMM || s.scenario == MPE_SCN_PREDATOR_PREY) ? 0.9 : 1.0;
}

// small in-register scenarios (mpe_small.hip); mode: 0 step, 1 rollout, 2 observe, 3 Scenario.post_step alone (pos, aux)
bool small_supported(const HostSpec& s);
hipError_t launch_small(const HostSpec& s, long long n_envs, int dtype, int mode, const StepArgs& a, hipStream_t st);

}  // namespace mpe

namespace mpe {
// on-device reset_world (mpe_reset.hip); env_major selects the state layout
hipError_t launch_reset(const HostSpec& s, long long n_envs, int dtype, int comm_rows, bool env_major, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_base, const uint8_t* mask, void* pos, void* vel, void* comm, int32_t* aux,
                        hipStream_t st);

// consensus (mpe_consensus.hip): a.comm is the history ring, a.aux its head; mode: 0 step, 1 rollout, 2 observe
hipError_t launch_consensus(const HostSpec& s, long long n_envs, int dtype, int mode, const StepArgs& a, hipStream_t st);
hipError_t launch_consensus_reset(const HostSpec& s, long long n_envs, int dtype, unsigned long long seed, unsigned long long episode,
                                  unsigned long long env_base, const uint8_t* mask, void* pos, void* vel, void* hist, int32_t* aux, hipStream_t st);

// advantage scan (mpe_gae.hip)
hipError_t launch_gae(int mode, long long n_envs, int n_agents, int n_steps, const float* rew, const float* val, const uint8_t* dones,
                      const float* last_val, const uint8_t* last_dones, double gamma, double lam, float* adv, float* ret, hipStream_t st);

// wave-per-world kernels (mpe_wave.hip), env-major state; mode: 0 step, 1 rollout, 2 observe
bool wave_supported(const HostSpec& s);
hipError_t launch_wave(const HostSpec& s, const DevTable* table, long long n_envs, int dtype, int mode, const StepArgs& a, hipStream_t st);
hipError_t launch_world_step(const HostSpec& s, const DevTable* table, long long n_envs, int dtype, const void* u, const void* c,
                             void* pos, void* vel, void* comm, int n_speak_rows, hipStream_t st);
// World.step with the per-entity table staged in the library's static device array instead of a handle's
// allocation (mpe_wave.hip); `device` must be current
hipError_t launch_world_step_static(const HostSpec& s, const DevTable& table, int device, long long n_envs, int dtype, const void* u, const void* c,
                                    void* pos, void* vel, void* comm, int n_speak_rows, hipStream_t st);
// resolves the static array's address on `device` (which must be current) ahead of the first mpe_world_step_static
hipError_t prepare_static_scratch(int device);

// ---- an uneven deal of a launch's units (worlds, tiles, workgroups' worth of worlds) to the eight XCDs ----
//
// While all eight XCDs stream rows, the odd ones write 15-20 % more slowly than the even ones (per-XCD timestamps and
// the L2's DRAM write-credit stalls), so a launch that is bound by its stores from beginning to end finishes on the odd
// XCDs alone.  Giving an even XCD `share[0]` units and an odd one `share[1]` lets them finish together.
// The dispatcher deals a launch's workgroups to the XCDs round-robin, but the XCD of workgroup 0 varies with the queue's
// history (0 in a fresh process, something else after other streams or graphs have run).  So a one-workgroup launch on
// the same stream, right before the kernel, stores the XCC_ID it runs on -- the phase s -- in the handle's `phase` word,
// and the kernel takes workgroup i for slot (i + s) / 8 of XCD (i + s) % 8 (less one for the XCDs before s, whose first row
// is missing).  For any value of s that is a bijection between workgroups and (XCD, slot) pairs, so every unit is
// taken exactly once whatever the hardware does; only the speed depends on the guess being right.  (Counters per XCD
// drawn with atomics would not need the guess, but 2 304 device-scope atomics on one line cost the launch 140 us.)
struct XcdDeal {
    const int* phase;       // nullptr: workgroup i takes unit i
    int share[2];           // units of an even / an odd XCD
    long long units;        // units of the launch; 4 * (share[0] + share[1]) >= units
};

// share[0] : share[1] about 9 : 7, both multiples of `quantum` (whole worlds per CU matter where a launch is one round of
// resident workgroups: quantum 32 there); the grid to launch is 8 * share[0] workgroups.
inline XcdDeal make_xcd_deal(const int* phase, long long units, long long quantum) {
    XcdDeal d{nullptr, {0, 0}, units};
    if (phase == nullptr || units <= 0) return d;
    const long long per = (units + 7) / 8;
    const long long even = ((per * 9 + 7) / 8 + quantum - 1) / quantum * quantum, left = units - 4 * even;
    const long long odd = left > 0 ? ((left + 3) / 4 + quantum - 1) / quantum * quantum : 0;
    if (odd <= 0 || 8 * even >= 0x7fffffffLL) return d;
    d.phase = phase; d.share[0] = (int)even; d.share[1] = (int)odd;
    return d;
}

// Switched off for the process by MPE_XCD_SPLIT=0 in the environment, read once.
inline bool xcd_split_enabled() {
    static const bool on = [] { const char* e = getenv("MPE_XCD_SPLIT"); return !(e != nullptr && e[0] == '0'); }();
    return on;
}
// the one-workgroup launch that stores the phase (mpe_reset.hip)
hipError_t launch_xcd_phase(int* phase, hipStream_t st);
}  // namespace mpe
