"""Scenario plugin contract (multiagent/scenario.py:4-10, README.md:33-44) and the base class of
the built-in scenarios whose callbacks are evaluated by the HIP kernels."""
import numpy as np

from . import _lib
from .spec import WorldSpec


class BaseScenario:
    """A scenario supplies ``make_world()`` and ``reset_world(world[, np_random])`` plus the
    callbacks ``reward(agent, world)``, ``observation(agent, world)`` and optionally
    ``benchmark_data(agent, world)``, ``done(agent, world)``, ``post_step(world, np_random)``."""

    def make_world(self):
        raise NotImplementedError()

    def reset_world(self, world, np_random=None):
        raise NotImplementedError()


def spec_from_world(world, scenario_id, name):
    """WorldSpec of a World object's current static properties."""
    ents = world.entities
    ag = world.agents
    return WorldSpec(scenario_id, len(ag), len(world.landmarks), world.dim_c,
                     size=np.array([e.size for e in ents], float), mass=np.array([e.mass for e in ents], float),
                     movable=np.array([bool(e.movable) for e in ents]), collide=np.array([bool(e.collide) for e in ents]),
                     sensitivity=np.array([a.accel if a.accel is not None else 5.0 for a in ag], float),
                     max_speed=np.array([a.max_speed if a.max_speed is not None else np.inf for a in ag], float),
                     silent=np.array([bool(a.silent) for a in ag]),
                     adversary=np.array([bool(getattr(a, "adversary", False)) for a in ag]),
                     leader=np.array([bool(getattr(a, "leader", False)) for a in ag]),
                     dt=world.dt, damping=world.damping, contact_force=world.contact_force,
                     contact_margin=world.contact_margin, name=name, collaborative=bool(getattr(world, "collaborative", False)),
                     walls=[dict(orient=w.orient, axis_pos=w.axis_pos, endpoints=tuple(w.endpoints), width=w.width, hard=w.hard)
                            for w in world.walls],
                     ghost=np.array([bool(getattr(a, "ghost", False)) for a in ag]), new_core=bool(getattr(world, "new_core", False)),
                     accel=np.array([np.nan if a.accel is None else a.accel for a in ag], float))


class HipScenario(BaseScenario):
    """Built-in scenario: ``observation`` / ``reward`` / ``benchmark_data`` of all agents are
    computed together by one ``mpe_observe`` launch on a one-world float64 batch and cached until
    the world's state changes, so the per-agent callbacks of ``MultiAgentEnv`` cost one launch."""

    scenario_id = _lib.SCN_NONE
    hip_name = ""
    obs_dtype = np.float64
    has_info = True          # the scenario defines benchmark_data
    fused_info = None        # ask the kernel for info_i / info_f (None: when has_info)
    fused_step = True        # MultiAgentEnv.step may run the whole step as one fused launch
    dynamic = False          # the scenario has a post_step (the kernel applies it: int state in aux, landmarks move)
    rescore_after_step = False   # a step leaves only the shared reward: fetch the agents' own with mpe_observe on the new state

    def __init__(self):
        self._ow = None         # (signature, OneWorld)
        self._ow_bare = None    # the same for the one-world batch whose step leaves post_step out
        self._cache = None      # (state key, obs [obs_total], rew [A], info_i, info_f)
        self._pinned = False

    # -- device plumbing ------------------------------------------------
    def one_world(self, world, post_step=True):
        """The one-world batch (with its pinned host mirror) matching the world's static properties.  ``post_step=False``
        (dynamic scenarios): the batch whose step ends after the callbacks, for an env built without a
        post_step_callback."""
        from .oneworld import OneWorld
        sig = world._signature() + (tuple(bool(getattr(a, "adversary", False)) for a in world.agents),
                                    bool(getattr(world, "collaborative", False)))
        slot = "_ow" if post_step else "_ow_bare"
        have = getattr(self, slot)
        if have is None or have[0] != sig:
            spec = spec_from_world(world, self.scenario_id, self.hip_name)
            want_info = self.has_info if self.fused_info is None else self.fused_info
            have = (sig, OneWorld(spec, device=world.device, info=want_info, post_step=post_step))
            setattr(self, slot, have)
            self._cache = None
        return have[1]

    def fused_env(self, world):
        """The BatchedEnv behind one_world(world) (sizes, layout)."""
        return self.one_world(world).env

    def goal_index(self, world):
        """The scenario's aux rows for this world -- landmark indices reset_world drew for the episode (the goal
        landmark the agents share, ``agent.goal_a``), or the int state a dynamic scenario keeps: an int, a list
        of ints, or None."""
        g = getattr(world.agents[0], "goal_a", None)
        return None if g is None else world.landmarks.index(g)

    def state_rows(self, world):
        """(pos, vel, comm, aux) as the kernel's buffers hold them for this world."""
        pos, vel, comm = world.pack_state()
        return pos, vel, comm, self.goal_index(world)

    @staticmethod
    def _key(pos, vel, comm, goal):
        return pos.tobytes() + vel.tobytes() + (comm.tobytes() if comm is not None else b"") + \
            (b"-" if goal is None else np.asarray(goal, np.int64).tobytes())

    def _outputs(self, ow):
        """(obs, rew, info_i, info_f) of the one world, copied out of the host mirror; rew holds what the reward
        callback returns, so in a collaborative world the agents' own rewards (info_f) rather than the shared sum a
        step leaves in rew."""
        v, env = ow.v, ow.env
        info_i = v["info_i"][0].copy() if "info_i" in v else None
        info_f = v["info_f"][0].copy() if "info_f" in v else None
        rew = info_f[:env.n_agents] if (env.spec.collaborative and info_f is not None) else v["rew"][0].copy()
        return v["obs"][0].copy(), rew, info_i, info_f

    def stage(self, world, ow):
        """Put the world's state into the host mirror; returns the cache key of that state."""
        pos, vel, comm, goal = self.state_rows(world)
        ow.put_state(pos, vel, comm, goal)
        return self._key(pos, vel, comm, goal)

    def store(self, key, ow):
        """Cache the callbacks' outputs ``ow`` holds for the state with this key."""
        self._cache = (key,) + self._outputs(ow)
        return self._cache

    def pinned(self):
        """Context in which evaluate() hands out the cached outputs as they are (those of the step just taken)
        without comparing the world's state with the cache key."""
        sc = self

        class _Pin:
            def __enter__(self):
                sc._pinned = True

            def __exit__(self, *exc):
                sc._pinned = False
        return _Pin()

    def evaluate(self, world):
        if self._pinned:
            return self._cache
        pos, vel, comm, goal = self.state_rows(world)
        key = self._key(pos, vel, comm, goal)
        if self._cache is None or self._cache[0] != key:
            ow = self.one_world(world)
            ow.put_state(pos, vel, comm, goal)
            ow.observe(score=True)
            self.store(key, ow)
        return self._cache

    def split_obs(self, obs, world):
        """The reference's obs_n list from the concatenated row."""
        env = self.fused_env(world)
        return [obs[o:o + d].astype(self.obs_dtype) for o, d in zip(env.obs_offsets, env.obs_dims)]

    # -- post_step on the device ------------------------------------------
    def absorb(self, world, aux, landmark_pos, np_random=None):
        """Map what the kernel's post_step left (aux rows [n_aux], landmark positions [L, 2]) back onto
        the world's objects.  Dynamic scenarios override it."""
        raise NotImplementedError()

    def absorb_from(self, world, ow, np_random):
        A = len(world.agents)
        pos, _, _ = ow.state()
        self.absorb(world, [int(x) for x in ow.v["aux"].reshape(-1)], pos[A:].copy(), np_random)
        self._cache = None

    def post_step(self, world, np_random=None):
        """Scenario.post_step (called by MultiAgentEnv as post_step_callback, environment.py:107-108), run by
        the ``mpe_post_step`` kernel on the one-world batch: the rules themselves live in the kernel only.  A
        generator passed as ``np_random`` supplies the values of a respawn (as the reference's collector draws
        them from its argument); without one they come from the world's counter-based device stream."""
        if not self.dynamic:
            raise AttributeError("scenario %r defines no post_step" % self.hip_name)
        ow = self.one_world(world)
        self.stage(world, ow)
        ow.post_step()
        self.absorb_from(world, ow, np_random)

    # -- the plugin callbacks --------------------------------------------
    def observation(self, agent, world):
        _, obs, _, _, _ = self.evaluate(world)
        env = self.fused_env(world)
        i = world.agents.index(agent)
        o, d = env.obs_offsets[i], env.obs_dims[i]
        return obs[o:o + d].astype(self.obs_dtype)

    def reward(self, agent, world):
        return float(self.evaluate(world)[2][world.agents.index(agent)])

    def benchmark_data(self, agent, world):
        if not self.has_info:
            raise AttributeError("scenario %r defines no benchmark_data" % self.hip_name)
        return int(self.evaluate(world)[3][world.agents.index(agent)])
