/*
 * mpe_hip -- C ABI of the MI355X (gfx950) batched particle-world simulator.
 *
 * The reference simulator (hepengli/multiagent-particle-envs) is pure Python and has no FFI;
 * each entry point below names the Python function(s) it replaces for a batch of B independent
 * worlds.  A Python binding loads this library with ctypes and passes raw device pointers
 * (tensor.data_ptr()) and the caller's HIP stream.
 *
 * Conventions
 *   - Every function returns 0 on success and a negative MPE_ERR_* code on failure; nothing is
 *     thrown across the boundary.  mpe_last_error() returns a description of the last failure
 *     on a handle (or of the last failed mpe_create when called with NULL).
 *   - The caller allocates and owns every buffer.  All data pointers are DEVICE pointers on the
 *     handle's device.  Output buffers (obs, rew, done, info_i, info_f) are 16-byte aligned: their
 *     rows leave the kernels as 16-byte vectors.  Inputs and state (actions, pos, vel, comm, aux)
 *     need the alignment of their element type only, so a step may read a slice actions[t] of a
 *     pre-generated [T][B][n_act] array whatever B is.  A misaligned pointer is refused with
 *     MPE_ERR_INVALID.  The library only holds the immutable world description and a small
 *     per-entity table in device memory.
 *   - Kernels are enqueued on the caller's stream and the call returns without synchronising.
 *     A call makes the handle's device current for its launches and puts the caller's current
 *     device back before it returns.
 *   - A handle is not thread-safe; distinct handles are independent.  There is no global state.
 *
 * Memory layout (T = float for MPE_F32, double for MPE_F64; B = n_envs; E = A + L entities,
 * agents first, then landmarks -- the order of World.entities, multiagent/core.py:127-129)
 *   pos      T[2E][B]        row 2e+d holds coordinate d of entity e for every env (env-minor)
 *   vel      T[2A][B]        same for agent velocities (landmarks never move)
 *   comm     T[S*dim_c][B]   state.c of the S non-silent agents, in agent order (may be NULL if S == 0).
 *                            consensus (no speakers) keeps its history here instead, MpeLayout.n_hist_rows
 *                            = 10 * A * 4 rows: state.p_his_pos / p_his_vel (new_core.py:252-254) as a ring of
 *                            ten slots, row (slot * A + agent) * 4 + {0: x, 1: y, 2: vx, 3: vy}; aux row 0 is
 *                            the slot holding the oldest entry (history entry k, oldest first, is slot
 *                            (aux + k) % 10)
 *   aux      int32[n_aux][B] landmark indices reset_world draws besides positions and keeps for the
 *                            episode (may be NULL if n_aux == 0): row 0 is the goal landmark
 *                            (agent.goal_a) in simple_adversary and simple_push and the listener's
 *                            target (agents[0].goal_b) in simple_speaker_listener; rows 0 and 1 are
 *                            goal_b of agent 0 and of agent 1 in simple_reference, and the goal landmark
 *                            and the landmark whose colour is the speaker's key in simple_crypto.
 *                            simple_predator_prey keeps the state its post_step rewrites every step
 *                            here: row 0 = flags (bit f: food f is alive; bit 3+k: member k, agent 4+k,
 *                            holds food), rows 1 and 2 = key and position of the world's respawn
 *                            stream (set by reset; any values give a valid stream).  collector likewise:
 *                            row 0 = bits 2a, 2a+1: what collector a holds (0 nothing, 1 + treasure type),
 *                            bit 12+l: type of treasure l, bit 18+l: treasure l is alive; rows 1-2 the stream
 *     Small worlds (one lane simulates one world) keep this env-minor layout so that a wave's 64
 *     worlds load as coalesced rows.  Worlds simulated by a whole wave (MpeLayout.state_env_major
 *     == 1: simple_spread other than 3/3, and MPE_SCN_NONE) store the state arrays transposed,
 *     pos T[B][2E], vel T[B][2A], comm T[B][S*dim_c], so that one world is contiguous.
 *   actions  int32[B][n_act] per env, the reference's action_n list flattened: for each agent its
 *                            move int 0..4 if it is movable, followed by its comm int 0..dim_c-1 if
 *                            it is not silent (environment.py:173-204)
 *   obs      T[B][obs_total] per env, the agents' observation vectors concatenated in agent order;
 *                            agent i owns columns [obs_offset[i], obs_offset[i] + obs_dim[i])
 *   rew      T[B][A]         what step returns: in a collaborative world (World.collaborative:
 *                            simple_speaker_listener, simple_reference) every agent's entry is the
 *                            sum of the agents' own rewards (environment.py:103-105); mpe_observe
 *                            writes the own rewards, which is what the reward callback returns
 *   done     uint8[B]        0: no in-scope scenario has a done callback (environment.py:143-146), unless the batch
 *                            was given one with mpe_set_done_bound; mpe_step_auto also sets it to 1 on the step
 *                            that reaches its time limit
 *   info_i   int32[B][n_info_i]   optional (NULL to skip), see mpe_layout
 *   info_f   T[B][n_info_f]       optional (NULL to skip)
 */
#ifndef MPE_HIP_H
#define MPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_OK 0
#define MPE_ERR_INVALID (-1)     /* bad argument / spec does not match the scenario's structure */
#define MPE_ERR_UNSUPPORTED (-2) /* valid request the library has no kernel for */
#define MPE_ERR_HIP (-3)         /* a HIP runtime call failed */

#define MPE_F32 0
#define MPE_F64 1

/* scenario ids */
#define MPE_SCN_NONE 0       /* physics only: World.step without scenario callbacks */
#define MPE_SCN_SPREAD 1     /* multiagent/scenarios/simple_spread.py */
#define MPE_SCN_TAG 2        /* multiagent/scenarios/simple_tag.py */
#define MPE_SCN_WORLD_COMM 3 /* multiagent/scenarios/simple_world_comm.py */
#define MPE_SCN_SIMPLE 4     /* multiagent/scenarios/simple.py */
#define MPE_SCN_ADVERSARY 5  /* multiagent/scenarios/simple_adversary.py */
#define MPE_SCN_PUSH 6       /* multiagent/scenarios/simple_push.py */
#define MPE_SCN_SPEAKER_LISTENER 7 /* multiagent/scenarios/simple_speaker_listener.py */
#define MPE_SCN_REFERENCE 8  /* multiagent/scenarios/simple_reference.py */
#define MPE_SCN_CRYPTO 9     /* multiagent/scenarios/simple_crypto.py */
#define MPE_SCN_PREDATOR_PREY 10 /* multiagent/scenarios/simple_predator_prey.py (on multiagent/new_core.py) */
#define MPE_SCN_CONSENSUS 11 /* multiagent/scenarios/consensus.py (on multiagent/new_core.py) */
#define MPE_SCN_COLLECTOR 12 /* multiagent/scenarios/collector.py (on multiagent/new_core.py) */

/* World.step variants (MpeSpec.physics) */
#define MPE_PHYSICS_CORE 0     /* multiagent/core.py:142-233 */
#define MPE_PHYSICS_NEW_CORE 1 /* multiagent/new_core.py:188-298: action force scaled by mass (* accel) (:214),
                                  contact between two movable entities split by their mass ratio (:290-294);
                                  MPE_SCN_NONE (mpe_world_step), MPE_SCN_PREDATOR_PREY, MPE_SCN_CONSENSUS,
                                  MPE_SCN_COLLECTOR */

#define MPE_MAX_AGENTS 64
#define MPE_MAX_LANDMARKS 64
#define MPE_MAX_WALLS 16

typedef void* mpe_stream_t; /* hipStream_t */

/*
 * What Scenario.make_world() builds: World scalars (multiagent/core.py:117-124) and the static
 * properties of every entity (core.py:45-101).  Arrays are HOST pointers, copied by mpe_create.
 */
typedef struct MpeSpec {
    int32_t scenario;    /* MPE_SCN_* */
    int32_t n_agents;    /* A */
    int32_t n_landmarks; /* L */
    int32_t dim_c;       /* World.dim_c */
    double dt;             /* World.dt             (0.1)  */
    double damping;        /* World.damping        (0.25) */
    double contact_force;  /* World.contact_force  (1e2)  */
    double contact_margin; /* World.contact_margin (1e-3) */
    const double* size;       /* [E] Entity.size */
    const double* mass;       /* [E] Entity.mass */
    const uint8_t* movable;   /* [E] Entity.movable; landmarks must be 0, agents 1 unless the scenario says otherwise */
    const uint8_t* collide;   /* [E] Entity.collide */
    const double* sensitivity; /* [A] Agent.accel, or 5.0 when accel is None (environment.py:192-195) */
    const double* max_speed;  /* [A] Entity.max_speed, or +inf when None (core.py:199) */
    const uint8_t* silent;    /* [A] Agent.silent */
    const uint8_t* adversary; /* [A] agent.adversary (simple_tag.py:21); zeros otherwise */
    const uint8_t* leader;    /* [A] agent.leader (simple_world_comm.py:23); zeros otherwise */
    /* World.walls (core.py:26-42; new_core.py walls of the arena scenarios).  mpe_world_step
     * (MPE_SCN_NONE) simulates any 0..MPE_MAX_WALLS walls; simple_predator_prey and consensus need
     * exactly their 4 arena walls (mpe_create checks); every other built-in scenario has none
     * (n_walls == 0, NULL pointers). */
    int32_t n_walls;
    int32_t collaborative; /* World.collaborative (environment.py:38): must match the scenario (1 for
                              simple_speaker_listener and simple_reference, 0 for the others) */
    const uint8_t* wall_vertical;  /* [W] 0: orient 'H' (lies along x at y = axis_pos), 1: orient 'V' */
    const double* wall_axis_pos;   /* [W] Wall.axis_pos */
    const double* wall_end0;       /* [W] Wall.endpoints[0] */
    const double* wall_end1;       /* [W] Wall.endpoints[1] */
    const double* wall_width;      /* [W] Wall.width */
    const uint8_t* wall_hard;      /* [W] Wall.hard */
    const uint8_t* ghost;          /* [A] Agent.ghost: passes through walls that are not hard (NULL = none) */
    int32_t physics;               /* MPE_PHYSICS_* */
    int32_t reserved;
    const double* accel;           /* [A] Agent.accel, NaN when None; read when physics == MPE_PHYSICS_NEW_CORE
                                      (the action force is (mass * accel, or mass when None) * u, new_core.py:214) */
    const double* u_noise;         /* [A] Agent.u_noise, 0 when None (NULL = none): standard deviation of the normal noise
                                      added to the action force of a movable agent, core.py:165 / new_core.py:213-214 */
    const double* c_noise;         /* [A] Agent.c_noise, 0 when None (NULL = none): the same for the comm state of a
                                      speaking agent, core.py:211 */
} MpeSpec;

/* Derived sizes of the caller-allocated buffers. */
typedef struct MpeLayout {
    int64_t n_envs;
    int32_t n_agents, n_landmarks, dim_c;
    int32_t n_speakers;  /* S: agents with silent == 0 */
    int32_t n_act;       /* movable agents + S */
    int32_t obs_total;   /* sum of obs_dim */
    int32_t n_info_i;    /* simple_spread: A collision counts + occupied_landmarks;
                            simple_tag: A collision counts;
                            simple_world_comm: A collision counts + in_forest bits (bit 2i+f)
                            + visibility bits (agent i sees its k-th other agent, others in
                            ascending order skipping i: bit (A-1)*i+k);
                            simple_adversary: the goal landmark index;
                            simple_speaker_listener, simple_reference, simple_crypto: the aux rows;
                            simple_predator_prey: A collision counts (adversary with good agents) + the
                            flags of aux row 0 as the callbacks saw them (before post_step);
                            collector: benchmark_data (collector.py:132-155) per agent: treasures touched,
                            treasures being deposited, collisions with collectors; then aux row 0 likewise;
                            simple, simple_push: none (no benchmark_data), info_i must be NULL */
    int32_t n_info_f;    /* simple_spread: min_dists + A benchmark rewards;
                            simple_adversary: benchmark_data (simple_adversary.py:56-65) flattened in
                            agent order: the adversary's squared goal distance, then per good agent
                            its squared distance to each landmark and to the goal;
                            simple_speaker_listener, simple_reference: the A own rewards before sharing
                            (info_n['r<i>'], environment.py:100);
                            simple_crypto: benchmark_data (simple_crypto.py:78-80) per agent: its
                            utterance state.c[dim_c], then the goal landmark's colour [dim_c];
                            otherwise 0 */
    int32_t state_env_major; /* 0: pos/vel/comm are [rows][B]; 1: [B][rows] (see "Memory layout") */
    int32_t n_aux;       /* rows of aux */
    int32_t obs_dim[MPE_MAX_AGENTS];
    int32_t obs_offset[MPE_MAX_AGENTS];
    int32_t n_hist_rows; /* rows of history state kept in the comm buffer (consensus), else 0 */
    int32_t reserved;
} MpeLayout;

typedef struct MpeHandle MpeHandle;

/* Scenario.make_world() + MultiAgentEnv.__init__ (make_env.py:36-43, environment.py:15-69)
 * for n_envs worlds of one scenario on HIP device `device`.  The handle owns a few bytes of device
 * memory (one word that the large multi-step launches use to deal their workgroups unevenly to the
 * GPU's XCDs, and the per-entity table of the scaled simple_spread / physics-only worlds), allocated
 * here and freed by mpe_destroy; the stepping entry points allocate nothing.  Launches on one handle
 * must be ordered (one stream, or events between streams): they share that word as they share the
 * caller's state arrays.  MPE_XCD_SPLIT=0 in the environment switches the uneven deal off. */
int mpe_create(const MpeSpec* spec, int64_t n_envs, int dtype, int device, MpeHandle** out);
int mpe_destroy(MpeHandle* h);
int mpe_layout(const MpeHandle* h, MpeLayout* out);
const char* mpe_last_error(const MpeHandle* h);

/*
 * The position of the handle's first env in the whole job's env index (0 unless set).  The reference seeds env k
 * of a vectorised run from its global position (multiagent/backup/mappo.py:97 `env.seed(seed + subrank)`, :113
 * `seed + 10000 * mpi_rank`); here every on-device draw -- the reset distributions of mpe_reset / mpe_episode(s) /
 * mpe_step_auto, the respawn streams of simple_predator_prey and collector, the generator of mpe_set_noise -- is
 * keyed by env_base + b for env b of the handle.  A batch sharded over several handles (one per GPU, shard g with
 * env_base = first env of the shard) therefore draws, bit for bit, the worlds of the unsharded batch under the same
 * seed.  Buffers stay indexed by the local b.  env_base < 2^48.
 */
int mpe_set_env_base(MpeHandle* h, uint64_t env_base);

/*
 * MultiAgentEnv's done_callback (environment.py:15-18,143-146) for a batch.  A Python callable cannot run in a kernel,
 * so the batch carries the predicate the reference's own comment has in mind ("agents are allowed to go beyond the
 * viewing screen", :142): done_callback(agent, world) = any(|agent.state.p_pos| > bound), an agent outside
 * [-bound, bound]^2.  step combines the agents' values with any(done_n) (:98,110-113), evaluated on the state after
 * World.step like the other callbacks.  With a bound set, mpe_step / mpe_rollout / mpe_episode(s) write that flag to
 * done, and mpe_step_auto ends the episode on it exactly as on its time limit (done = 1, returns closed, on-device
 * reset; what the runner does on a done env, backup/runner_old.py:56).  bound <= 0 or +inf switches it off (the
 * default: done_callback=None, done always 0).  A NaN coordinate compares false, as in numpy.
 */
int mpe_set_done_bound(MpeHandle* h, double bound);

/*
 * MultiAgentEnv.step (environment.py:83-115) for every env of the batch:
 *   _set_action (environment.py:155-206, discrete int input) ->
 *   World.step (core.py:142-158: apply_action_force :161-167, apply_environment_force :170-181 with
 *   get_collision_force :215-233, integrate_state :193-204, update_agent_state :206-212) ->
 *   the scenario's observation / reward / benchmark_data callbacks.
 * pos, vel and comm are updated in place; aux is read (mpe_step_auto rewrites it when it resets).
 * simple_predator_prey and collector also run the scenario's post_step (simple_predator_prey.py:83-107,
 * collector.py:84-110; a taken treasure of collector parks at the origin, a respawn also draws its type) as
 * post_step_callback, after the callbacks (environment.py:107-108): it rewrites aux and moves the food
 * landmarks (pick-up parks a food at -999; a dead food respawns at U(-0.95,0.95)^2 from the world's
 * counter-based stream, where the reference draws from numpy's global generator).
 */
int mpe_step(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux,
             void* obs, void* rew, uint8_t* done, int32_t* info_i, void* info_f, mpe_stream_t stream);

/*
 * mpe_step plus the episode bookkeeping the reference leaves to its caller (the runner loop of
 * backup/runner_old.py:44-56 resets an env when it is done; baselines' Monitor sums the episode
 * return): per env, ep_return T[B][A] accumulates the rewards and step_count int32[B] counts the
 * steps of the running episode.  When step_count reaches max_steps the episode is truncated:
 * done[b] = 1, last_return[b] receives the finished return, ep_return and step_count restart at 0,
 * episode_count int32[B] is incremented and the env is reset on device with the draws of
 * mpe_reset(h, seed, episode_count[b], ...), so obs holds the first observation of the new episode
 * (rew and info still describe the step that ended the old one).  All four bookkeeping arrays are
 * caller-owned device buffers, zero-initialised by the caller before the first episode.
 * (consensus: the reset also refills the history ring with the fresh state.)
 */
int mpe_step_auto(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux,
                  void* obs, void* rew, uint8_t* done, int32_t* info_i, void* info_f, int32_t* step_count,
                  int32_t* episode_count, void* ep_return, void* last_return, int32_t max_steps,
                  uint64_t seed, mpe_stream_t stream);

/*
 * n_steps consecutive MultiAgentEnv.step calls in one launch, for pre-computed actions (the
 * "random actions" benchmark loop and scripted policies).  Buffers gain a leading step axis:
 * actions int32[n_steps][B][n_act], obs T[n_steps][B][obs_total], rew T[n_steps][B][A],
 * done uint8[n_steps][B], info_* likewise.  pos/vel/comm hold the state before the first step on
 * entry and after the last on return.
 */
int mpe_rollout(MpeHandle* h, const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux,
                void* obs, void* rew, uint8_t* done, int32_t* info_i, void* info_f, int32_t n_steps,
                mpe_stream_t stream);

/*
 * One whole episode in one launch: Scenario.reset_world for every env (the draws of
 * mpe_reset(h, seed, episode, NULL, ...)) followed by n_steps MultiAgentEnv.step calls as in
 * mpe_rollout.  pos / vel / comm need not be initialised; they hold the final state on return.
 * obs0 T[B][obs_total] (NULL to skip) receives the observation of the freshly reset state, the list
 * MultiAgentEnv.reset returns (environment.py:122-127), so that the launch does everything the loop
 * "obs = env.reset(); 25 x env.step(a)" does.
 */
int mpe_episode(MpeHandle* h, uint64_t seed, uint64_t episode, const int32_t* actions, void* pos,
                void* vel, void* comm, int32_t* aux, void* obs, void* rew, uint8_t* done, int32_t* info_i,
                void* info_f, void* obs0, int32_t n_steps, mpe_stream_t stream);

/*
 * n_episodes consecutive episodes in one launch: as mpe_episode(h, seed, first_episode + k, ...) for k = 0 ..
 * n_episodes - 1, each of steps_per_episode steps; the step axis of actions and outputs has
 * n_episodes * steps_per_episode entries, episode after episode.  What the benchmark loop "reset, 25 random
 * steps, repeat" issues, without a launch boundary between the episodes (a launch starts with draws and
 * action fetches during which nothing is stored, and ends with a tail of late waves; inside one launch the
 * waves drift apart and fill both).  obs0 T[n_episodes][B][obs_total] (NULL to skip): the observation
 * of every episode's reset state.
 */
int mpe_episodes(MpeHandle* h, uint64_t seed, uint64_t first_episode, int32_t n_episodes, int32_t steps_per_episode,
                 const int32_t* actions, void* pos, void* vel, void* comm, int32_t* aux, void* obs, void* rew, uint8_t* done,
                 int32_t* info_i, void* info_f, void* obs0, mpe_stream_t stream);

/*
 * Where a world with Agent.u_noise / Agent.c_noise (MpeSpec.u_noise / c_noise; no built-in scenario sets them)
 * takes its standard-normal draws, for which the reference calls np.random.randn (core.py:165,211):
 *   draws == NULL  the device's counter-based generator keyed by (seed, env_base + env index, step number, agent); the step
 *                  number counts the steps the handle has run since this call (every mpe_step / mpe_step_auto
 *                  adds 1, a rollout or episode launch its n_steps), so a sequence of launches never reuses a draw
 *   draws != NULL  for the NEXT stepping launch only: T[n_steps][B][stride] device array of the draws themselves,
 *                  stride = 2A + S * dim_c rounded up to even; value 2i + d is the force noise of agent i
 *                  (coordinate d), value 2A + j the j-th comm value of the speakers in agent order.  Lets a test feed
 *                  the reference's own draws.  Later launches go back to the generator.
 * Draws are scaled by the agent's noise level and added as the reference does; an agent whose level is 0 uses
 * none.  A handle that never receives this call draws from seed 0.  mpe_world_step adds no noise: it takes
 * the forces as given.  MPE_ERR_INVALID for a world without noise levels.
 */
int mpe_set_noise(MpeHandle* h, uint64_t seed, const void* draws);

/*
 * The position of the handle's noise generator, for a checkpoint: set == 0 reads (seed, number of steps run since
 * mpe_set_noise) into *seed / *step, set != 0 puts the generator there (and drops a pending draw array), so that a
 * resumed batch continues the stream it left.  Like the generator itself it has no counterpart in the reference,
 * whose noise comes from numpy's global generator (core.py:165,211).
 */
int mpe_noise_state(MpeHandle* h, int set, uint64_t* seed, uint64_t* step);

/*
 * Whether the stepping launches (mpe_step, mpe_step_auto, mpe_rollout, mpe_episode(s)) apply the scenario's post_step
 * after the callbacks: MultiAgentEnv(post_step_callback=scenario.post_step) (enabled = 1, the default) or
 * MultiAgentEnv(post_step_callback=None) (enabled = 0), which is what make_env builds (make_env.py:36-43 passes no
 * post_step_callback, so environment.py:107-108 calls none): the world then steps and is observed but nothing is ever
 * picked up, handed over or respawned.  MPE_ERR_UNSUPPORTED for scenarios without a post_step.
 */
int mpe_set_post_step(MpeHandle* h, int enabled);

/*
 * Scenario.post_step alone (simple_predator_prey.py:83-107, collector.py:84-110), what MultiAgentEnv.step
 * calls as post_step_callback after the observation / reward callbacks (environment.py:107-108): pick-ups,
 * hand-overs and respawns exactly as mpe_step applies them, for a caller that steps the world and
 * evaluates the callbacks itself (mpe_world_step + mpe_observe).  Rewrites the landmark rows of pos and
 * aux (rows 1-2 hold the world's respawn stream, advanced by every respawn).  MPE_ERR_UNSUPPORTED for
 * scenarios without a post_step.
 */
int mpe_post_step(MpeHandle* h, void* pos, int32_t* aux, mpe_stream_t stream);

/* The scenario callbacks on the current state, without stepping: observation for every agent
 * (what MultiAgentEnv.reset returns, environment.py:122-127) and, when rew is not NULL, also
 * reward (environment.py:149-152) and benchmark_data (info_i / info_f, NULL to skip). */
int mpe_observe(MpeHandle* h, const void* pos, const void* vel, const void* comm, const int32_t* aux,
                void* obs, void* rew,
                int32_t* info_i, void* info_f, mpe_stream_t stream);

/*
 * Scenario.reset_world (simple_spread.py:40-56, simple_tag.py:39-54, simple_world_comm.py:88-113)
 * for every env whose mask byte is non-zero (all envs when mask is NULL): agent positions
 * U(-1,1)^2, landmark positions U(-0.9,0.9)^2 (simple_tag, simple_world_comm, simple_predator_prey,
 * whose foods are redrawn until they touch neither the barrier nor a forest, :139-144), U(-0.95,0.95)^2
 * (collector, with a uniform type per treasure, collector.py:122-129) or U(-1,1)^2 (all others), zero
 * velocity and comm; the aux landmark indices (goal of simple_adversary.py:41-45,
 * simple_push.py:40-42, simple_speaker_listener.py:40; both goal_b of simple_reference.py:33-35; goal
 * and key of simple_crypto.py:61-63) uniform over the landmarks.
 * Draws come from a counter-based generator keyed by (seed, env_base + env index, episode), env_base from
 * mpe_set_env_base; they follow the reference's distributions, not numpy's Mersenne-Twister stream.
 */
int mpe_reset(MpeHandle* h, uint64_t seed, uint64_t episode, const uint8_t* mask, void* pos,
              void* vel, void* comm, int32_t* aux, mpe_stream_t stream);

/*
 * World.step alone (core.py:142-158) for scenarios whose callbacks stay in Python (handles created
 * with MPE_SCN_NONE, env-major state): forces u T[B][2A] (already scaled, i.e. agent.action.u) and
 * comm actions c T[B][S*dim_c] (agent.action.c of the speakers, may be NULL if S == 0) are given
 * directly; pos, vel, comm are updated in place.  Wall contact (get_wall_collision_force,
 * core.py:270-307) is applied to every agent after its entity contacts, as in core.py:182-188.
 * Immovable agents keep their state (core.py:195).  With MPE_PHYSICS_NEW_CORE the step is
 * new_core.py:188-298 (its distance cache, :147-170, only stores what the step computes anyway).
 */
int mpe_world_step(MpeHandle* h, const void* u, const void* c, void* pos, void* vel, void* comm,
                   mpe_stream_t stream);

/*
 * mpe_world_step without a handle and without allocating anything.  It has no counterpart in the
 * reference: it exists for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  mpe_create gives a physics-only handle a
 * per-entity table from hipMalloc; here that table lives in a static device array inside the
 * library (MPE_STATIC_SCRATCH_BYTES per GPU, held until the process ends) and is rewritten in full
 * from `spec` by every call, on `stream`, before the step kernel reads it.  `spec` is a HOST pointer,
 * read before the call returns, with scenario MPE_SCN_NONE; buffers as in mpe_world_step (env-major
 * state on `device`).  Limits: MPE_MAX_AGENTS agents, MPE_MAX_LANDMARKS landmarks, MPE_MAX_WALLS
 * walls, 2^31-1 envs; a larger request is refused with MPE_ERR_INVALID.  The call uses no hipMalloc,
 * hipFree or host allocation; errors are reported by mpe_last_error(NULL).  All calls on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (different streams or threads)
 * race for it; calls on one stream, or otherwise ordered, are safe.
 * Set-up: the array's device address is looked up once per device, by the first mpe_create on that device or by the
 * first call of this entry point there.  That lookup must not happen under stream capture: a first call on a device
 * that comes while `stream` is capturing (and no mpe_create has run for the device) is refused with MPE_ERR_INVALID
 * before anything is enqueued -- the capture stays valid -- so make one call outside capture, or create any handle
 * on the device, before capturing.
 */
#define MPE_STATIC_SCRATCH_BYTES 8192
int mpe_world_step_static(const MpeSpec* spec, int64_t n_envs, int dtype, int device, const void* u, const void* c,
                          void* pos, void* vel, void* comm, mpe_stream_t stream);

/*
 * Generalised advantage estimation, the scan the reference's runner applies to a rollout before the
 * policy update (multiagent/backup/runner_old.py:58-80), for all envs at once.  Needs no handle.
 *   rewards, values  float[T][B][A]   what step returned / the value head predicted, per step
 *   dones            uint8[T][B]      the done flag step t returned (mb_dones, runner_old.py:52-54)
 *   last_values      float[B][A]      value of the observation after the last step (:64)
 *   last_dones       uint8[B]         the env's current done flags (self.dones, :72); for a rollout that was
 *                                     just collected this is dones[T-1]
 *   adv, ret         float[T][B][A]   advantages and returns (mb_advs, mb_returns); ret = adv + values
 * The runner's arrays are float32 (:61-62) and so are these; the scan runs in float64 as the runner's does.
 * mode MPE_GAE_REFERENCE reproduces :69-80 as written: a single lastgaelam carried through the agent loop
 * (agent i continues from agent i-1 of the same step, agent 0 from the last agent of step t+1) and the
 * non-terminal mask of step t read from dones[t+1].  MPE_GAE_STANDARD scans every agent independently with
 * the mask of step t read from dones[t].  All pointers are device pointers on `device`.
 */
#define MPE_GAE_REFERENCE 0
#define MPE_GAE_STANDARD 1
int mpe_gae(int device, int mode, int64_t n_envs, int32_t n_agents, int32_t n_steps, const float* rewards, const float* values,
            const uint8_t* dones, const float* last_values, const uint8_t* last_dones, double gamma, double lam, float* adv,
            float* ret, mpe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPE_HIP_H */
