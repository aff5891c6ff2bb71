// extern "C" boundary of libmpe_hip.so: handle life cycle, argument validation and dispatch.
// Declarations and the reference functions each entry point replaces: include/mpe_hip.h.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "mpe_launch.hpp"

using namespace mpe;

struct MpeHandle {
    HostSpec spec;
    MpeLayout layout;
    int dtype;
    int device;
    bool use_wave;          // wave-per-world kernels, env-major state
    DevTable* table;        // device copy of the per-entity table (wave kernels only)
    int* xcd_phase;         // device word of the uneven deal over the XCDs (XcdDeal): the XCD of a launch's workgroup 0
    // Agent.u_noise / c_noise draws: generator key, steps run since mpe_set_noise, and the draw array of the next launch
    unsigned long long noise_seed, noise_step;
    const void* noise_draws;
    unsigned long long env_base;    // global index of the handle's first env (mpe_set_env_base), 0 by default
    double done_bound;              // > 0: the batch's done predicate (mpe_set_done_bound); 0: no done callback
    int skip_post_step;             // 1: stepping launches leave the scenario's post_step out (mpe_set_post_step(h, 0))
    char err[512];
};

static thread_local char g_create_err[512] = "";

static int fail(MpeHandle* h, int code, const char* fmt, ...) {
    char* dst = h ? h->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(MpeHandle* h, hipError_t e, const char* what) {
    return fail(h, MPE_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// Makes `device` current for the launches of one entry point and puts the caller's device back on return.
struct DeviceScope {
    int prev = -1;
    hipError_t status;
    explicit DeviceScope(int device) {
        status = hipGetDevice(&prev);
        if (status != hipSuccess) { prev = -1; return; }
        if (prev != device) status = hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// The specialised kernels fix a scenario's structure at compile time; refuse a spec that
// disagrees with it instead of silently simulating something else.
static const char* check_structure(const HostSpec& s) {
    const int A = s.A, L = s.L;
    for (int l = 0; l < L; ++l)
        if (s.movable[A + l]) return "landmarks must not be movable";
    const bool shared = s.scenario == MPE_SCN_SPEAKER_LISTENER || s.scenario == MPE_SCN_REFERENCE || s.scenario == MPE_SCN_CONSENSUS;
    if (s.scenario != MPE_SCN_NONE && (s.collaborative != 0) != shared)
        return shared ? "the scenario shares rewards: collaborative must be 1" : "the scenario does not share rewards: collaborative must be 0";
    switch (s.scenario) {
        case MPE_SCN_SPEAKER_LISTENER:
            if (A != 2 || L != 3 || s.dim_c != 3) return "simple_speaker_listener has 2 agents, 3 landmarks, dim_c 3";
            if (s.movable[0] || s.silent[0] || !s.movable[1] || !s.silent[1]) return "simple_speaker_listener: agent 0 speaks and cannot move, agent 1 moves and is silent";
            for (int e = 0; e < A + L; ++e)
                if (s.collide[e]) return "simple_speaker_listener: nothing collides";
            return nullptr;
        case MPE_SCN_CRYPTO:
            if (A != 3 || L != 2 || s.dim_c != 4) return "simple_crypto has 3 agents, 2 landmarks, dim_c 4";
            for (int i = 0; i < A; ++i)
                if (s.movable[i] || s.silent[i] || s.adversary[i] != (i < 1)) return "simple_crypto agents speak and cannot move, agent 0 is the adversary";
            return nullptr;
        default:
            break;
    }
    for (int i = 0; i < A; ++i)
        if (!s.movable[i] && s.scenario != MPE_SCN_NONE) return "agents must be movable";
    switch (s.scenario) {
        case MPE_SCN_NONE:
            return nullptr;
        case MPE_SCN_CONSENSUS:
            if (A != 6 || L != 0 || s.n_walls != 4) return "consensus has 6 agents, no landmarks and 4 walls";
            for (int i = 0; i < A; ++i) {
                if (s.collide[i] || !s.silent[i] || !s.movable[i]) return "consensus agents move, are silent and do not collide";
                if (s.size[i] != s.size[0] || s.mass[i] != s.mass[0] || s.sens[i] != s.sens[0] || s.force_scale[i] != s.force_scale[0] ||
                    s.ghost[i] != s.ghost[0] || isfinite(s.max_speed[i]) || s.u_noise[i] != s.u_noise[0])
                    return "consensus agents share size, mass, accel, ghost and u_noise and have no speed limit";
            }
            return nullptr;
        case MPE_SCN_COLLECTOR:
            if (A != 8 || L != 6 || s.n_walls != 0) return "collector has 8 agents, 6 treasures and no walls";
            for (int i = 0; i < A; ++i)
                if (!s.collide[i] || !s.silent[i]) return "collector agents collide and are silent";
            for (int l = 0; l < L; ++l)
                if (s.collide[A + l]) return "collector treasures do not collide";
            {   // the kernel reads one set of magnitudes per role
                const int rep[14] = {0, 0, 0, 0, 0, 0, 6, 6, 8, 8, 8, 8, 8, 8};
                for (int e = 0; e < A + L; ++e)
                    if (s.size[e] != s.size[rep[e]] || s.mass[e] != s.mass[rep[e]]) return "collector: size and mass must be uniform per role";
                for (int i = 0; i < A; ++i)
                    if (s.sens[i] != s.sens[rep[i]] || s.max_speed[i] != s.max_speed[rep[i]] || s.force_scale[i] != s.force_scale[rep[i]])
                        return "collector: accel and max_speed must be uniform per role";
            }
            return nullptr;
        case MPE_SCN_PREDATOR_PREY:
            if (A != 7 || L != 6 || s.n_walls != 4) return "simple_predator_prey has 7 agents, 6 landmarks and 4 walls";
            for (int i = 0; i < A; ++i)
                if (!s.collide[i] || !s.silent[i] || s.adversary[i] != (i < 3) || s.leader[i] != (i == 0 || i == 3))
                    return "simple_predator_prey: 3 adversaries then 4 good agents, agents 0 and 3 lead, all collide and are silent";
            for (int l = 0; l < L; ++l)
                if (s.collide[A + l] != (l == 0)) return "simple_predator_prey: only the barrier (first landmark) collides";
            {   // the kernel reads one set of magnitudes per role
                const int rep[13] = {0, 0, 0, 3, 3, 3, 3, 7, 8, 8, 8, 11, 11};
                for (int e = 0; e < A + L; ++e)
                    if (s.size[e] != s.size[rep[e]] || s.mass[e] != s.mass[rep[e]]) return "simple_predator_prey: size and mass must be uniform per role";
                for (int i = 0; i < A; ++i)
                    if (s.sens[i] != s.sens[rep[i]] || s.max_speed[i] != s.max_speed[rep[i]] || s.force_scale[i] != s.force_scale[rep[i]] || s.ghost[i] != s.ghost[rep[i]])
                        return "simple_predator_prey: accel, max_speed and ghost must be uniform per role";
            }
            return nullptr;
        case MPE_SCN_REFERENCE:
            if (A != 2 || L != 3 || s.dim_c != 10) return "simple_reference has 2 agents, 3 landmarks, dim_c 10";
            for (int i = 0; i < A; ++i)
                if (s.silent[i]) return "simple_reference agents speak";
            for (int e = 0; e < A + L; ++e)
                if (s.collide[e]) return "simple_reference: nothing collides";
            return nullptr;
        case MPE_SCN_SPREAD:
            for (int i = 0; i < A; ++i)
                if (!s.collide[i] || !s.silent[i]) return "simple_spread agents collide and are silent";
            for (int l = 0; l < L; ++l)
                if (s.collide[A + l]) return "simple_spread landmarks do not collide";
            return nullptr;
        case MPE_SCN_TAG:
            if (A != 4 || L != 2) return "simple_tag has 4 agents and 2 landmarks";
            for (int e = 0; e < A + L; ++e)
                if (!s.collide[e]) return "simple_tag entities all collide";
            for (int i = 0; i < A; ++i)
                if (!s.silent[i] || s.adversary[i] != (i < 3)) return "simple_tag agents are silent, the first 3 are adversaries";
            return nullptr;
        case MPE_SCN_WORLD_COMM:
            if (A != 6 || L != 5 || s.dim_c != 4) return "simple_world_comm has 6 agents, 5 landmarks, dim_c 4";
            for (int i = 0; i < A; ++i)
                if (!s.collide[i] || s.adversary[i] != (i < 4) || s.leader[i] != (i == 0) || s.silent[i] != (i != 0))
                    return "simple_world_comm: 4 adversaries then 2 good agents, agent 0 leads and speaks";
            for (int l = 0; l < L; ++l)
                if (s.collide[A + l] != (l == 0)) return "simple_world_comm: only the first landmark collides";
            return nullptr;
        case MPE_SCN_SIMPLE:
            if (A != 1 || L != 1) return "simple has 1 agent and 1 landmark";
            if (s.collide[0] || s.collide[1] || !s.silent[0]) return "simple: nothing collides, the agent is silent";
            return nullptr;
        case MPE_SCN_ADVERSARY:
            if (A != 3 || L != 2) return "simple_adversary has 3 agents and 2 landmarks";
            for (int e = 0; e < A + L; ++e)
                if (s.collide[e]) return "simple_adversary: nothing collides";
            for (int i = 0; i < A; ++i)
                if (!s.silent[i] || s.adversary[i] != (i < 1)) return "simple_adversary agents are silent, agent 0 is the adversary";
            return nullptr;
        case MPE_SCN_PUSH:
            if (A != 2 || L != 2) return "simple_push has 2 agents and 2 landmarks";
            for (int i = 0; i < A; ++i)
                if (!s.collide[i] || !s.silent[i] || s.adversary[i] != (i < 1)) return "simple_push agents collide and are silent, agent 0 is the adversary";
            for (int l = 0; l < L; ++l)
                if (s.collide[A + l]) return "simple_push landmarks do not collide";
            return nullptr;
    }
    return "unknown scenario id";
}

static void fill_layout(const HostSpec& s, long long n_envs, MpeLayout* lo) {
    memset(lo, 0, sizeof(*lo));
    const int A = s.A, L = s.L;
    lo->n_envs = n_envs;
    lo->n_agents = A;
    lo->n_landmarks = L;
    lo->dim_c = s.dim_c;
    int speakers = 0, n_good = 0, movers = 0;
    for (int i = 0; i < A; ++i) {
        speakers += s.silent[i] ? 0 : 1;
        n_good += s.adversary[i] ? 0 : 1;
        movers += s.movable[i] ? 1 : 0;
    }
    lo->n_speakers = speakers;
    // MPE_SCN_NONE takes forces for every agent (mpe_world_step), not int actions
    lo->n_act = (s.scenario == MPE_SCN_NONE ? A : movers) + speakers;
    int off = 0;
    for (int i = 0; i < A; ++i) {
        int d = 0;
        const int good_others = n_good - (s.adversary[i] ? 0 : 1);
        switch (s.scenario) {
            case MPE_SCN_SPREAD: d = 4 + 2 * L + 2 * (A - 1); break;
            case MPE_SCN_TAG: d = 4 + 2 * L + 2 * (A - 1) + 2 * good_others; break;
            case MPE_SCN_WORLD_COMM: d = 4 + 2 * L + 2 * (A - 1) + 2 * good_others + 2 + (s.adversary[i] ? s.dim_c : 0); break;
            case MPE_SCN_SIMPLE: d = 2 + 2 * L; break;
            case MPE_SCN_ADVERSARY: d = (s.adversary[i] ? 0 : 2) + 2 * L + 2 * (A - 1); break;
            case MPE_SCN_PUSH: d = s.adversary[i] ? 2 + 2 * L + 2 * (A - 1) : 2 + 2 + 3 + 2 * L + 3 * L + 2 * (A - 1); break;
            case MPE_SCN_SPEAKER_LISTENER: d = (i == 0) ? 3 : 2 + 2 * L + s.dim_c; break;
            case MPE_SCN_REFERENCE: d = 2 + 2 * L + 3 + s.dim_c; break;
            case MPE_SCN_CRYPTO: d = (i == 0) ? s.dim_c : 2 * s.dim_c; break;
            case MPE_SCN_PREDATOR_PREY: d = 4 + 4 * (A - 1) + 3 + 2 * L + 2; break;
            case MPE_SCN_CONSENSUS: d = 2 * 2 * 10 * 2; break;
            case MPE_SCN_COLLECTOR: d = 4 + (i < 6 ? 2 : 0) + (A - 1) * 8 + L * 4; break;
            default: d = 0;
        }
        lo->obs_dim[i] = d;
        lo->obs_offset[i] = off;
        off += d;
    }
    lo->obs_total = off;
    switch (s.scenario) {
        case MPE_SCN_SPREAD: lo->n_info_i = A + 1; lo->n_info_f = A + 1; break;
        case MPE_SCN_TAG: lo->n_info_i = A; break;
        case MPE_SCN_WORLD_COMM: lo->n_info_i = A + 2; break;
        case MPE_SCN_ADVERSARY: lo->n_info_i = 1; lo->n_info_f = 1 + (A - 1) * (L + 1); break;
        case MPE_SCN_SPEAKER_LISTENER: lo->n_info_i = 1; lo->n_info_f = A; break;
        case MPE_SCN_REFERENCE: lo->n_info_i = 2; lo->n_info_f = A; break;
        case MPE_SCN_CRYPTO: lo->n_info_i = 2; lo->n_info_f = A * 2 * s.dim_c; break;
        case MPE_SCN_PREDATOR_PREY: lo->n_info_i = A + 1; break;
        case MPE_SCN_COLLECTOR: lo->n_info_i = 3 * A + 1; break;
        default: break;
    }
    lo->n_aux = s.n_aux();
    lo->n_hist_rows = (s.scenario == MPE_SCN_CONSENSUS) ? 10 * A * 4 : 0;
}

// Validates an MpeSpec and copies it into `s`; errors are recorded for mpe_last_error(NULL).
static int read_spec(const MpeSpec* spec, int64_t n_envs, int dtype, int device, HostSpec& s, const char* who) {
    if (!spec) return fail(nullptr, MPE_ERR_INVALID, "%s: spec is NULL", who);
    if (n_envs <= 0) return fail(nullptr, MPE_ERR_INVALID, "%s: n_envs must be positive (got %lld)", who, (long long)n_envs);
    if (dtype != MPE_F32 && dtype != MPE_F64) return fail(nullptr, MPE_ERR_INVALID, "%s: dtype must be MPE_F32 or MPE_F64", who);
    const int A = spec->n_agents, L = spec->n_landmarks;
    if (A < 1 || A > MPE_MAX_AGENTS || L < 0 || L > MPE_MAX_LANDMARKS)
        return fail(nullptr, MPE_ERR_INVALID, "%s: need 1..%d agents and 0..%d landmarks (got %d, %d)", who, MPE_MAX_AGENTS, MPE_MAX_LANDMARKS, A, L);
    if (!spec->size || !spec->mass || !spec->movable || !spec->collide || !spec->sensitivity || !spec->max_speed || !spec->silent)
        return fail(nullptr, MPE_ERR_INVALID, "%s: a required per-entity array is NULL", who);
    if (spec->dim_c < 0 || spec->dim_c > 16) return fail(nullptr, MPE_ERR_INVALID, "%s: dim_c out of range", who);
    if (!(spec->contact_margin > 0) || !(spec->dt > 0)) return fail(nullptr, MPE_ERR_INVALID, "%s: dt and contact_margin must be positive", who);

    s.scenario = spec->scenario; s.A = A; s.L = L; s.dim_c = spec->dim_c; s.collaborative = spec->collaborative ? 1 : 0;
    s.physics = spec->physics;
    if (s.physics != MPE_PHYSICS_CORE && s.physics != MPE_PHYSICS_NEW_CORE) return fail(nullptr, MPE_ERR_INVALID, "%s: unknown physics %d", who, s.physics);
    const bool new_core_scn = spec->scenario == MPE_SCN_PREDATOR_PREY || spec->scenario == MPE_SCN_CONSENSUS || spec->scenario == MPE_SCN_COLLECTOR;
    if (s.physics == MPE_PHYSICS_NEW_CORE && spec->scenario != MPE_SCN_NONE && !new_core_scn)
        return fail(nullptr, MPE_ERR_UNSUPPORTED, "%s: new_core physics is simulated by mpe_world_step (MPE_SCN_NONE), simple_predator_prey, consensus and collector only", who);
    if (new_core_scn && s.physics != MPE_PHYSICS_NEW_CORE) return fail(nullptr, MPE_ERR_INVALID, "%s: this scenario runs on the new_core physics", who);
    if (s.physics == MPE_PHYSICS_NEW_CORE && !spec->accel) return fail(nullptr, MPE_ERR_INVALID, "%s: new_core physics needs the accel array", who);
    s.dt = spec->dt; s.damping = spec->damping; s.contact_force = spec->contact_force; s.contact_margin = spec->contact_margin;
    for (int e = 0; e < A + L; ++e) {
        s.size[e] = spec->size[e]; s.mass[e] = spec->mass[e];
        s.movable[e] = spec->movable[e] ? 1 : 0; s.collide[e] = spec->collide[e] ? 1 : 0;
        if (!(s.mass[e] > 0)) return fail(nullptr, MPE_ERR_INVALID, "%s: entity %d has non-positive mass", who, e);
    }
    for (int i = 0; i < A; ++i) {
        s.sens[i] = spec->sensitivity[i]; s.max_speed[i] = spec->max_speed[i];
        s.silent[i] = spec->silent[i] ? 1 : 0;
        s.adversary[i] = (spec->adversary && spec->adversary[i]) ? 1 : 0;
        s.leader[i] = (spec->leader && spec->leader[i]) ? 1 : 0;
        s.ghost[i] = (spec->ghost && spec->ghost[i]) ? 1 : 0;
        if (isnan(s.max_speed[i])) return fail(nullptr, MPE_ERR_INVALID, "%s: max_speed[%d] is NaN (use +inf for no limit)", who, i);
        s.u_noise[i] = spec->u_noise ? spec->u_noise[i] : 0.0;
        s.c_noise[i] = spec->c_noise ? spec->c_noise[i] : 0.0;
        if (!(s.u_noise[i] >= 0.0) || !(s.c_noise[i] >= 0.0) || isinf(s.u_noise[i]) || isinf(s.c_noise[i]))
            return fail(nullptr, MPE_ERR_INVALID, "%s: u_noise / c_noise of agent %d must be finite and >= 0", who, i);
        s.force_scale[i] = 1.0;
        if (s.physics == MPE_PHYSICS_NEW_CORE) s.force_scale[i] = isnan(spec->accel[i]) ? s.mass[i] : s.mass[i] * spec->accel[i];
    }
    s.n_walls = spec->n_walls;
    if (s.n_walls < 0 || s.n_walls > MPE_MAX_WALLS) return fail(nullptr, MPE_ERR_INVALID, "%s: 0..%d walls (got %d)", who, MPE_MAX_WALLS, spec->n_walls);
    if (s.n_walls > 0) {
        if (spec->scenario != MPE_SCN_NONE && !new_core_scn)
            return fail(nullptr, MPE_ERR_UNSUPPORTED, "%s: walls are simulated by mpe_world_step (MPE_SCN_NONE), simple_predator_prey, consensus and collector only", who);
        if (!spec->wall_vertical || !spec->wall_axis_pos || !spec->wall_end0 || !spec->wall_end1 || !spec->wall_width || !spec->wall_hard)
            return fail(nullptr, MPE_ERR_INVALID, "%s: a wall array is NULL", who);
        for (int w = 0; w < s.n_walls; ++w) {
            s.wall_vertical[w] = spec->wall_vertical[w] ? 1 : 0; s.wall_hard[w] = spec->wall_hard[w] ? 1 : 0;
            s.wall_axis_pos[w] = spec->wall_axis_pos[w]; s.wall_end0[w] = spec->wall_end0[w];
            s.wall_end1[w] = spec->wall_end1[w]; s.wall_width[w] = spec->wall_width[w];
        }
    }
    if (const char* why = check_structure(s)) return fail(nullptr, MPE_ERR_INVALID, "%s: spec does not match scenario %d: %s", who, spec->scenario, why);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= n_dev) return fail(nullptr, MPE_ERR_INVALID, "%s: device %d out of range (%d visible)", who, device, n_dev);
    return MPE_OK;
}

// The per-entity table the wave-per-world kernels read from device memory; unused entries are zero.
static void fill_table(const HostSpec& s, DevTable& t) {
    memset(&t, 0, sizeof(t));
    const int A = s.A, L = s.L;
    for (int k = 0; k < A + L; ++k) { t.size[k] = s.size[k]; t.mass[k] = s.mass[k]; t.collide[k] = s.collide[k]; t.movable[k] = s.movable[k]; }
    for (int i = 0; i < A; ++i) { t.sens[i] = s.sens[i]; t.max_speed[i] = s.max_speed[i]; t.silent[i] = s.silent[i]; t.ghost[i] = s.ghost[i]; t.force_scale[i] = s.force_scale[i]; t.u_noise[i] = s.u_noise[i]; }
    t.n_walls = s.n_walls;
    t.new_core = s.physics == MPE_PHYSICS_NEW_CORE ? 1 : 0;
    for (int w = 0; w < s.n_walls; ++w) {
        t.wall_vertical[w] = s.wall_vertical[w]; t.wall_hard[w] = s.wall_hard[w]; t.wall_axis_pos[w] = s.wall_axis_pos[w];
        t.wall_end0[w] = s.wall_end0[w]; t.wall_end1[w] = s.wall_end1[w]; t.wall_width[w] = s.wall_width[w];
    }
}

extern "C" {

int mpe_create(const MpeSpec* spec, int64_t n_envs, int dtype, int device, MpeHandle** out) {
    if (!out) return fail(nullptr, MPE_ERR_INVALID, "mpe_create: out is NULL");
    *out = nullptr;
    MpeHandle* h = new (std::nothrow) MpeHandle();
    if (!h) return fail(nullptr, MPE_ERR_INVALID, "mpe_create: out of host memory");
    memset(h, 0, sizeof(*h));
    HostSpec& s = h->spec;
    int rc = read_spec(spec, n_envs, dtype, device, s, "mpe_create");
    if (rc != MPE_OK) { delete h; return rc; }
    h->dtype = dtype;
    h->device = device;
    fill_layout(s, n_envs, &h->layout);
    h->use_wave = !small_supported(s) && s.scenario != MPE_SCN_CONSENSUS;
    if (h->use_wave && !wave_supported(s)) {
        delete h;
        return fail(nullptr, MPE_ERR_UNSUPPORTED, "mpe_create: no kernel for scenario %d with %d agents / %d landmarks", s.scenario, s.A, s.L);
    }
    h->layout.state_env_major = h->use_wave ? 1 : 0;
    {
        // the library's set-up on this device: the address of mpe_world_step_static's static array, so that a first
        // call of that entry point may come under stream capture (a failure here is reported by that call, not this one)
        DeviceScope scope(device);
        if (scope.status == hipSuccess) (void)prepare_static_scratch(device);
    }
    if (h->use_wave) {
        if (n_envs > 0x7fffffffLL) { delete h; return fail(nullptr, MPE_ERR_INVALID, "mpe_create: at most 2^31-1 envs for this scenario size"); }
        DevTable t;
        fill_table(s, t);
        DeviceScope scope(device);
        hipError_t e;
        if ((e = scope.status) != hipSuccess || (e = hipMalloc((void**)&h->table, sizeof(DevTable))) != hipSuccess ||
            (e = hipMemcpy(h->table, &t, sizeof(DevTable), hipMemcpyHostToDevice)) != hipSuccess) {
            if (h->table) (void)hipFree(h->table);
            delete h;
            return hip_fail(nullptr, e, "mpe_create: device table");
        }
    }
    {
        DeviceScope scope(device);
        hipError_t e;
        if ((e = scope.status) != hipSuccess || (e = hipMalloc((void**)&h->xcd_phase, sizeof(int))) != hipSuccess ||
            (e = hipMemset(h->xcd_phase, 0, sizeof(int))) != hipSuccess) {
            if (h->xcd_phase) (void)hipFree(h->xcd_phase);
            if (h->table) (void)hipFree(h->table);
            delete h;
            return hip_fail(nullptr, e, "mpe_create: phase word");
        }
    }
    *out = h;
    return MPE_OK;
}

int mpe_destroy(MpeHandle* h) {
    if (h && (h->table || h->xcd_phase)) {
        DeviceScope scope(h->device);
        if (h->table) (void)hipFree(h->table);
        if (h->xcd_phase) (void)hipFree(h->xcd_phase);
    }
    delete h;
    return MPE_OK;
}

int mpe_layout(const MpeHandle* h, MpeLayout* out) {
    if (!h || !out) return MPE_ERR_INVALID;
    *out = h->layout;
    return MPE_OK;
}

const char* mpe_last_error(const MpeHandle* h) { return h ? h->err : g_create_err; }

static int run(MpeHandle* h, int mode, const StepArgs& args, mpe_stream_t stream, const char* who) {
    if (!h) return MPE_ERR_INVALID;
    StepArgs a = args;
    a.env_base = a.reset.env_base = a.au.env_base = h->env_base;
    a.done_bound = h->done_bound;
    a.skip_post_step = h->skip_post_step;
    a.xcd_phase = xcd_split_enabled() ? h->xcd_phase : nullptr;
    if (mode != 2 && h->spec.any_noise()) {
        a.noise = NoiseArgs{1, h->spec.noise_stride(), h->noise_draws, h->noise_seed, h->noise_step, h->env_base};
        if (((uintptr_t)h->noise_draws & (h->dtype == MPE_F64 ? 7u : 3u)) != 0) return fail(h, MPE_ERR_INVALID, "%s: misaligned noise draws", who);
    }
    if (h->spec.scenario == MPE_SCN_NONE) return fail(h, MPE_ERR_INVALID, "%s: handle was created with MPE_SCN_NONE; use mpe_world_step", who);
    if (!a.pos || !a.vel || !a.obs) return fail(h, MPE_ERR_INVALID, "%s: pos, vel and obs must not be NULL", who);
    if ((h->layout.n_speakers > 0 || h->layout.n_hist_rows > 0) && !a.comm)
        return fail(h, MPE_ERR_INVALID, "%s: comm must not be NULL for a scenario with speaking agents or a history", who);
    if (h->spec.scenario == MPE_SCN_CONSENSUS && a.info_i != nullptr)
        return fail(h, MPE_ERR_UNSUPPORTED, "%s: consensus has no benchmark outputs (its benchmark_data is constant)", who);
    if (mode != 2 && (!a.actions || !a.rew || !a.done)) return fail(h, MPE_ERR_INVALID, "%s: actions, rew and done must not be NULL", who);
    if (mode == 1 && a.n_steps < 1) return fail(h, MPE_ERR_INVALID, "%s: n_steps must be >= 1", who);
    if (a.info_f && !a.info_i) return fail(h, MPE_ERR_INVALID, "%s: info_f requires info_i", who);
    if (a.info_i && h->layout.n_info_i == 0) return fail(h, MPE_ERR_INVALID, "%s: this scenario has no benchmark_data; info_i must be NULL", who);
    if (h->layout.n_aux > 0 && !a.aux) return fail(h, MPE_ERR_INVALID, "%s: aux must not be NULL for a scenario with a goal landmark", who);
    // output rows leave the kernels as 16-byte vectors; actions and state are read and written one element at a time
    const void* outs[] = {a.obs, a.rew, a.done, a.info_i, a.info_f, a.obs0};
    for (const void* ptr : outs)
        if (((uintptr_t)ptr & 15u) != 0) return fail(h, MPE_ERR_INVALID, "%s: every output buffer must be 16-byte aligned (got %p)", who, ptr);
    const uintptr_t elem = h->dtype == MPE_F64 ? 8u : 4u;
    const void* state[] = {a.pos, a.vel, a.comm};
    for (const void* ptr : state)
        if (((uintptr_t)ptr & (elem - 1u)) != 0) return fail(h, MPE_ERR_INVALID, "%s: state buffers must be aligned to their element size (got %p)", who, ptr);
    if ((((uintptr_t)a.actions | (uintptr_t)a.aux) & 3u) != 0) return fail(h, MPE_ERR_INVALID, "%s: actions and aux must be 4-byte aligned", who);
    DeviceScope scope(h->device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(h, e, "hipSetDevice");
    e = h->spec.scenario == MPE_SCN_CONSENSUS ? launch_consensus(h->spec, h->layout.n_envs, h->dtype, mode, a, (hipStream_t)stream)
        : h->use_wave                           ? launch_wave(h->spec, h->table, h->layout.n_envs, h->dtype, mode, a, (hipStream_t)stream)
                                                : launch_small(h->spec, h->layout.n_envs, h->dtype, mode, a, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(h, e, who);
    if (a.noise.enabled) {      // the launch used its draws: move on
        h->noise_step += (unsigned long long)(mode == 1 ? a.n_steps : 1);
        h->noise_draws = nullptr;
    }
    return MPE_OK;
}

int mpe_set_env_base(MpeHandle* h, uint64_t env_base) {
    if (!h) return MPE_ERR_INVALID;
    if (env_base > 0xffffffffffffull) return fail(h, MPE_ERR_INVALID, "mpe_set_env_base: env_base must be below 2^48");
    h->env_base = env_base;
    return MPE_OK;
}

int mpe_set_post_step(MpeHandle* h, int enabled) {
    if (!h) return MPE_ERR_INVALID;
    if (h->spec.scenario != MPE_SCN_PREDATOR_PREY && h->spec.scenario != MPE_SCN_COLLECTOR)
        return fail(h, MPE_ERR_UNSUPPORTED, "mpe_set_post_step: only simple_predator_prey and collector define post_step");
    h->skip_post_step = enabled ? 0 : 1;
    return MPE_OK;
}

int mpe_set_done_bound(MpeHandle* h, double bound) {
    if (!h) return MPE_ERR_INVALID;
    if (h->spec.scenario == MPE_SCN_NONE) return fail(h, MPE_ERR_INVALID, "mpe_set_done_bound: a physics-only handle evaluates no callbacks");
    if (isnan(bound)) return fail(h, MPE_ERR_INVALID, "mpe_set_done_bound: bound is NaN");
    h->done_bound = (bound > 0.0 && isfinite(bound)) ? bound : 0.0;
    return MPE_OK;
}

int mpe_set_noise(MpeHandle* h, uint64_t seed, const void* draws) {
    if (!h) return MPE_ERR_INVALID;
    if (!h->spec.any_noise()) return fail(h, MPE_ERR_INVALID, "mpe_set_noise: no agent of this world has u_noise or c_noise");
    if (h->spec.scenario == MPE_SCN_NONE) return fail(h, MPE_ERR_INVALID, "mpe_set_noise: mpe_world_step takes its forces as given and adds no noise");
    h->noise_seed = seed;
    h->noise_step = 0;
    h->noise_draws = draws;
    return MPE_OK;
}

int mpe_noise_state(MpeHandle* h, int set, uint64_t* seed, uint64_t* step) {
    if (!h) return MPE_ERR_INVALID;
    if (!seed || !step) return fail(h, MPE_ERR_INVALID, "mpe_noise_state: seed and step must not be NULL");
    if (set) {
        h->noise_seed = *seed;
        h->noise_step = *step;
        h->noise_draws = nullptr;
    } else {
        *seed = h->noise_seed;
        *step = h->noise_step;
    }
    return MPE_OK;
}

int mpe_step(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux, void* obs, void* rew,
             uint8_t* done, int32_t* info_i, void* info_f, mpe_stream_t stream) {
    StepArgs a{actions, pos, vel, comm, obs, rew, done, info_i, info_f, 1};
    a.aux = aux;
    return run(h, 0, a, stream, "mpe_step");
}

int mpe_step_auto(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux, void* obs, void* rew, uint8_t* done,
                  int32_t* info_i, void* info_f, int32_t* step_count, int32_t* episode_count, void* ep_return, void* last_return,
                  int32_t max_steps, uint64_t seed, mpe_stream_t stream) {
    if (!h) return MPE_ERR_INVALID;
    if (!step_count || !episode_count || !ep_return || !last_return)
        return fail(h, MPE_ERR_INVALID, "mpe_step_auto: step_count, episode_count, ep_return and last_return must not be NULL");
    if (max_steps < 1) return fail(h, MPE_ERR_INVALID, "mpe_step_auto: max_steps must be >= 1");
    StepArgs a{actions, pos, vel, comm, obs, rew, done, info_i, info_f, 1};
    a.au = AutoArgs{step_count, episode_count, ep_return, last_return, max_steps, seed, landmark_limit(h->spec)};
    a.aux = aux;
    return run(h, 0, a, stream, "mpe_step_auto");
}

int mpe_rollout(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux, void* obs, void* rew,
                uint8_t* done, int32_t* info_i, void* info_f, int32_t n_steps, mpe_stream_t stream) {
    StepArgs a{actions, pos, vel, comm, obs, rew, done, info_i, info_f, n_steps};
    a.aux = aux;
    return run(h, 1, a, stream, "mpe_rollout");
}

int mpe_episode(MpeHandle* h, uint64_t seed, uint64_t episode, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux,
                void* obs, void* rew, uint8_t* done, int32_t* info_i, void* info_f, void* obs0, int32_t n_steps, mpe_stream_t stream) {
    StepArgs a{actions, pos, vel, comm, obs, rew, done, info_i, info_f, n_steps};
    a.aux = aux;
    a.obs0 = obs0;
    if (h) a.reset = ResetArgs{1, seed, episode, landmark_limit(h->spec)};
    return run(h, 1, a, stream, "mpe_episode");
}

int mpe_episodes(MpeHandle* h, uint64_t seed, uint64_t first_episode, int32_t n_episodes, int32_t steps_per_episode, const int32_t* actions,
                 void* pos, void* vel, void* comm, int32_t* aux, void* obs, void* rew, uint8_t* done, int32_t* info_i, void* info_f,
                 void* obs0, mpe_stream_t stream) {
    if (!h) return MPE_ERR_INVALID;
    if (n_episodes < 1 || steps_per_episode < 1) return fail(h, MPE_ERR_INVALID, "mpe_episodes: n_episodes and steps_per_episode must be >= 1");
    if ((long long)n_episodes * steps_per_episode > 0x7fffffffLL) return fail(h, MPE_ERR_INVALID, "mpe_episodes: too many steps");
    StepArgs a{actions, pos, vel, comm, obs, rew, done, info_i, info_f, n_episodes * steps_per_episode};
    a.aux = aux;
    a.obs0 = obs0;
    a.reset = ResetArgs{1, seed, first_episode, landmark_limit(h->spec), n_episodes > 1 ? steps_per_episode : 0};
    return run(h, 1, a, stream, "mpe_episodes");
}

int mpe_observe(MpeHandle* h, const void* pos, const void* vel, const void* comm, const int32_t* aux, void* obs, void* rew,
                int32_t* info_i, void* info_f, mpe_stream_t stream) {
    if (h && !rew && (info_i || info_f)) return fail(h, MPE_ERR_INVALID, "mpe_observe: info_i / info_f require rew");
    StepArgs a{nullptr, (void*)pos, (void*)vel, (void*)comm, obs, rew, nullptr, info_i, info_f, 1};
    a.aux = (int32_t*)aux;
    return run(h, 2, a, stream, "mpe_observe");
}

int mpe_post_step(MpeHandle* h, void* pos, int32_t* aux, mpe_stream_t stream) {
    if (!h) return MPE_ERR_INVALID;
    if (h->spec.scenario != MPE_SCN_PREDATOR_PREY && h->spec.scenario != MPE_SCN_COLLECTOR)
        return fail(h, MPE_ERR_UNSUPPORTED, "mpe_post_step: only simple_predator_prey and collector define post_step");
    if (!pos || !aux) return fail(h, MPE_ERR_INVALID, "mpe_post_step: pos and aux must not be NULL");
    const uintptr_t elem = h->dtype == MPE_F64 ? 8u : 4u;
    if (((uintptr_t)pos & (elem - 1u)) != 0 || ((uintptr_t)aux & 3u) != 0) return fail(h, MPE_ERR_INVALID, "mpe_post_step: misaligned buffer");
    DeviceScope scope(h->device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(h, e, "hipSetDevice");
    StepArgs a{};
    a.pos = pos;
    a.aux = aux;
    a.env_base = h->env_base;
    e = launch_small(h->spec, h->layout.n_envs, h->dtype, 3, a, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(h, e, "mpe_post_step");
    return MPE_OK;
}

int mpe_reset(MpeHandle* h, uint64_t seed, uint64_t episode, const uint8_t* mask, void* pos, void* vel, void* comm,
              int32_t* aux, mpe_stream_t stream) {
    if (!h) return MPE_ERR_INVALID;
    if (!pos || !vel) return fail(h, MPE_ERR_INVALID, "mpe_reset: pos and vel must not be NULL");
    const int comm_rows = h->layout.n_speakers * h->spec.dim_c;
    if ((comm_rows > 0 || h->layout.n_hist_rows > 0) && !comm)
        return fail(h, MPE_ERR_INVALID, "mpe_reset: comm must not be NULL for a scenario with speaking agents or a history");
    if (h->layout.n_aux > 0 && !aux) return fail(h, MPE_ERR_INVALID, "mpe_reset: aux must not be NULL for a scenario with a goal landmark");
    DeviceScope scope(h->device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(h, e, "hipSetDevice");
    e = h->spec.scenario == MPE_SCN_CONSENSUS
            ? launch_consensus_reset(h->spec, h->layout.n_envs, h->dtype, seed, episode, h->env_base, mask, pos, vel, comm, aux, (hipStream_t)stream)
            : launch_reset(h->spec, h->layout.n_envs, h->dtype, comm_rows, h->use_wave, seed, episode, h->env_base, mask, pos, vel, comm, aux, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(h, e, "mpe_reset");
    return MPE_OK;
}

int mpe_world_step(MpeHandle* h, const void* u, const void* c, void* pos, void* vel, void* comm, mpe_stream_t stream) {
    if (!h) return MPE_ERR_INVALID;
    if (h->spec.scenario != MPE_SCN_NONE) return fail(h, MPE_ERR_INVALID, "mpe_world_step: handle must be created with MPE_SCN_NONE");
    if (!u || !pos || !vel) return fail(h, MPE_ERR_INVALID, "mpe_world_step: u, pos and vel must not be NULL");
    const int rows = h->layout.n_speakers * h->spec.dim_c;
    if (rows > 0 && (!c || !comm)) return fail(h, MPE_ERR_INVALID, "mpe_world_step: c and comm must not be NULL for a world with speaking agents");
    DeviceScope scope(h->device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(h, e, "hipSetDevice");
    e = launch_world_step(h->spec, h->table, h->layout.n_envs, h->dtype, u, c, pos, vel, comm, rows, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(h, e, "mpe_world_step");
    return MPE_OK;
}

int mpe_world_step_static(const MpeSpec* spec, int64_t n_envs, int dtype, int device, const void* u, const void* c, void* pos, void* vel,
                          void* comm, mpe_stream_t stream) {
    HostSpec s;
    memset(&s, 0, sizeof(s));
    const int rc = read_spec(spec, n_envs, dtype, device, s, "mpe_world_step_static");
    if (rc != MPE_OK) return rc;
    if (s.scenario != MPE_SCN_NONE) return fail(nullptr, MPE_ERR_INVALID, "mpe_world_step_static: the spec's scenario must be MPE_SCN_NONE");
    if (n_envs > 0x7fffffffLL) return fail(nullptr, MPE_ERR_INVALID, "mpe_world_step_static: at most 2^31-1 envs");
    if (!u || !pos || !vel) return fail(nullptr, MPE_ERR_INVALID, "mpe_world_step_static: u, pos and vel must not be NULL");
    int rows = 0;
    for (int i = 0; i < s.A; ++i) rows += s.silent[i] ? 0 : s.dim_c;
    if (rows > 0 && (!c || !comm)) return fail(nullptr, MPE_ERR_INVALID, "mpe_world_step_static: c and comm must not be NULL for a world with speaking agents");
    DevTable t;
    fill_table(s, t);
    DeviceScope scope(device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
    e = launch_world_step_static(s, t, device, n_envs, dtype, u, c, pos, vel, comm, rows, (hipStream_t)stream);
    if (e == hipErrorStreamCaptureUnsupported)
        return fail(nullptr, MPE_ERR_INVALID, "mpe_world_step_static: the first call on device %d came while its stream is being captured, and the "
                    "static array's address is not known yet; make one call outside capture first (mpe_create on the device also resolves it). "
                    "Nothing was enqueued", device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "mpe_world_step_static");
    return MPE_OK;
}

int mpe_gae(int device, int mode, int64_t n_envs, int32_t n_agents, int32_t n_steps, const float* rewards, const float* values,
            const uint8_t* dones, const float* last_values, const uint8_t* last_dones, double gamma, double lam, float* adv, float* ret,
            mpe_stream_t stream) {
    if (mode != MPE_GAE_REFERENCE && mode != MPE_GAE_STANDARD) return fail(nullptr, MPE_ERR_INVALID, "mpe_gae: unknown mode %d", mode);
    if (n_envs <= 0 || n_agents < 1 || n_agents > MPE_MAX_AGENTS || n_steps < 1)
        return fail(nullptr, MPE_ERR_INVALID, "mpe_gae: need n_envs > 0, 1..%d agents, n_steps > 0", MPE_MAX_AGENTS);
    if (!rewards || !values || !dones || !last_values || !last_dones || !adv || !ret) return fail(nullptr, MPE_ERR_INVALID, "mpe_gae: NULL buffer");
    if (adv == ret || (const float*)adv == values || (const float*)adv == rewards) return fail(nullptr, MPE_ERR_INVALID, "mpe_gae: adv must not alias another buffer");
    DeviceScope scope(device);
    hipError_t e = scope.status;
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
    e = launch_gae(mode, n_envs, n_agents, n_steps, rewards, values, dones, last_values, last_dones, gamma, lam, adv, ret, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(nullptr, e, "mpe_gae");
    return MPE_OK;
}

}  // extern "C"
