"""Drop-in single-world ``MultiAgentEnv`` (multiagent/environment.py:10-206).

Same constructor, ``seed`` / ``reset`` / ``step`` signatures and return shapes as the reference:
``step(action_n) -> (obs_n: list of arrays, reward_n: list of floats, done: bool, info_n: dict with
'n' and 'r<i>' keys)``.  The simulation itself runs on the GPU:

* built-in scenarios (``mpe_hip.scenarios``): one fused ``mpe_step`` launch on a one-world float64
  batch does action decoding, World.step and every agent's observation / reward / benchmark_data;
* any other ``Scenario`` plugin: actions are decoded here into ``agent.action``, ``World.step`` runs
  the ``mpe_world_step`` kernel, and the plugin's Python callbacks are called on the updated world.

Accepted action forms (``discrete_action=True``, the only mode the reference can construct -- its
continuous branch reads the undefined ``agent.c_range``, environment.py:56): one int 0..4 per silent
movable agent; a ``(move, comm)`` pair for an agent that moves and speaks (the reference builds a
``Tuple`` space for it, environment.py:60-63, but its ``_set_action`` cannot split it); one int
0..dim_c-1 for an immovable speaker.
"""
import numpy as np

from . import spaces
from .scenario import HipScenario


def _accepts_two_positionals(fn):
    """Whether ``fn(world, np_random)`` is a valid call (bound methods: self is already taken)."""
    if fn is None:
        return False
    import inspect
    try:
        params = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return True                     # no signature to inspect (a builtin): call it as the reference does
    if any(p.kind == p.VAR_POSITIONAL for p in params):
        return True
    return len([p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]) >= 2


class MultiAgentEnv:
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, world, reset_callback=None, reward_callback=None, observation_callback=None, info_callback=None,
                 done_callback=None, post_step_callback=None, shared_viewer=True, discrete_action=True):
        if not discrete_action:
            raise NotImplementedError("only discrete int actions are supported (the reference's continuous mode "
                                      "cannot be constructed either: Agent has no c_range)")
        self.world = world
        self.agents = world.policy_agents
        self.n = len(self.agents)
        self.reset_callback, self.reward_callback = reset_callback, reward_callback
        self.observation_callback, self.info_callback = observation_callback, info_callback
        self.done_callback, self.post_step_callback = done_callback, post_step_callback
        self.discrete_action_space = True
        self.discrete_action_input = True
        self.force_discrete_action = bool(getattr(world, "discrete_action", False))
        self.shared_reward = bool(getattr(world, "collaborative", False))
        self.shared_viewer = shared_viewer
        self.np_random = None
        # the reference always calls reset_callback(world, np_random) (environment.py:119); scenarios whose
        # reset_world takes only the world (simple_tag.py:39) are called without the generator.  Decided once
        # from the signature, so that an error inside the callback is never mistaken for a wrong arity.
        self._reset_takes_rng = _accepts_two_positionals(reset_callback)
        # a built-in scenario drives the fused kernel when it owns every callback that was passed
        owner = getattr(reward_callback, "__self__", None)
        cbs = [cb for cb in (reset_callback, reward_callback, observation_callback, info_callback) if cb is not None]
        fusable = isinstance(owner, HipScenario) and all(getattr(cb, "__self__", None) is owner for cb in cbs) \
            and observation_callback is not None and done_callback is None and not world.scripted_agents and owner.fused_step
        self._fused_post_step = False
        if fusable and owner.dynamic:
            # the fused kernel of a scenario with a post_step applies it when the scenario's own (device) post_step
            # is the post_step_callback, and leaves it out when there is no post_step_callback at all (what make_env
            # builds, make_env.py:36-43); any other callback is called here, after an unfused step
            own = getattr(post_step_callback, "__self__", None) is owner and \
                getattr(post_step_callback, "__func__", None) is HipScenario.post_step
            fusable = own or post_step_callback is None
            self._fused_post_step = own
        elif fusable:
            fusable = post_step_callback is None
        self._scenario = owner if fusable else None

        self.action_space, self.observation_space = [], []
        for agent in self.agents:
            parts = []
            if agent.movable:
                parts.append(spaces.Discrete(world.dim_p * 2 + 1))            # environment.py:47
            if not agent.silent:
                parts.append(spaces.Discrete(world.dim_c))                    # environment.py:54
            self.action_space.append(spaces.Tuple(parts) if len(parts) > 1 else parts[0])
            obs_dim = len(observation_callback(agent, world)) if observation_callback is not None else 0
            self.observation_space.append(spaces.Box(low=-np.inf, high=+np.inf, shape=(obs_dim,), dtype=np.float32))
            agent.action.c = np.zeros(world.dim_c)

    # ------------------------------------------------------------------
    def seed(self, seed=None):
        """environment.py:79-81 (old gym's seeding.np_random hands back a RandomState)."""
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def reset(self):
        """environment.py:117-127.  Scenarios whose reset_world takes only the world
        (simple_tag.py:39) are called without the generator."""
        if self.np_random is None:
            raise AttributeError("call env.seed(seed) before env.reset() (the reference has no np_random until seeded)")
        if self._reset_takes_rng:
            self.reset_callback(self.world, self.np_random)
        else:
            self.reset_callback(self.world)
        self.agents = self.world.policy_agents
        return [self._get_obs(a) for a in self.agents]

    def step(self, action_n):
        self.agents = self.world.policy_agents
        if len(action_n) != self.n:
            raise ValueError("expected %d actions, got %d" % (self.n, len(action_n)))
        # Agent.u_noise / c_noise (core.py:165,211; new_core.py:213): World.step draws them from numpy's global
        # generator exactly where the reference does, which the fused launch cannot; a world with a noise level
        # takes the unfused path (World.step on the device, callbacks through mpe_observe, post_step_callback below)
        noisy = any(getattr(a, "u_noise", None) or getattr(a, "c_noise", None) for a in self.world.agents)
        fused = self._scenario is not None and not noisy
        if fused:
            obs_n, reward_n, info_list = self._fused_step(action_n)
            done_n = [False] * self.n
        else:
            for i, agent in enumerate(self.agents):
                self._set_action(action_n[i], agent, self.action_space[i])
            self.world.step()
            obs_n = [self._get_obs(a) for a in self.agents]
            reward_n = [self._get_reward(a) for a in self.agents]
            done_n = [self._get_done(a) for a in self.agents]
            info_list = [self._get_info(a) for a in self.agents]
        info_n = {"n": info_list}
        for i, r in enumerate(reward_n):
            info_n["r{}".format(i)] = r                                        # environment.py:100
        if self.shared_reward:
            reward_n = [np.sum(reward_n)] * self.n                             # environment.py:103-105
        if self.post_step_callback is not None and not (fused and self._fused_post_step):
            self.post_step_callback(self.world, self.np_random)         # (the fused step of a dynamic scenario already applied it)
        return obs_n, reward_n, bool(any(done_n)), info_n                      # environment.py:110-115

    # ------------------------------------------------------------------
    def _action_ints(self, action, agent):
        """Flatten one agent's action into [move][, comm] ints."""
        both = agent.movable and not agent.silent
        if both:
            a = np.asarray(action).reshape(-1)
            if a.size != 2:
                raise ValueError("agent %r moves and speaks: its action must be a (move, comm) pair" % agent.name)
            move, word = int(a[0]), int(a[1])
            if not 0 <= word < self.world.dim_c:
                raise IndexError("comm action %d out of range for dim_c %d" % (word, self.world.dim_c))
            return [move, word]
        a = np.asarray(action).reshape(-1)
        if a.size != 1:
            raise ValueError("agent %r takes one int action" % agent.name)
        if not agent.movable and not agent.silent and not 0 <= int(a[0]) < self.world.dim_c:
            raise IndexError("comm action %d out of range for dim_c %d" % (int(a[0]), self.world.dim_c))
        return [int(a[0])] if (agent.movable or not agent.silent) else []

    def _set_action(self, action, agent, action_space=None, time=None):
        """environment.py:155-206 for discrete int input."""
        ints = self._action_ints(action, agent)
        agent.action.u = np.zeros(self.world.dim_p)
        agent.action.c = np.zeros(self.world.dim_c)
        if agent.movable:
            m = ints.pop(0)
            if m == 1: agent.action.u[0] = -1.0
            if m == 2: agent.action.u[0] = +1.0
            if m == 3: agent.action.u[1] = -1.0
            if m == 4: agent.action.u[1] = +1.0
            agent.action.u *= agent.accel if agent.accel is not None else 5.0
        if not agent.silent:
            agent.action.c[ints.pop(0)] = 1.0
        assert not ints

    def _fused_step(self, action_n):
        """One mpe_step launch on the one-world batch: the state goes up and every result comes down in one copy each."""
        sc, world = self._scenario, self.world
        ow = sc.one_world(world, post_step=self._fused_post_step or not sc.dynamic)
        flat = []
        for i, agent in enumerate(self.agents):
            flat += self._action_ints(action_n[i], agent)
        sc.stage(world, ow)
        ow.step(flat)
        if sc.rescore_after_step:
            ow.observe(score=True, upload=False)
        world.t += 1
        pos, vel, comm = ow.state()
        world.unpack_state(pos[None], vel[None], None if comm is None else comm[None])
        after = getattr(world, "after_step", None)      # new_core worlds: history and distance cache of the new state
        if after is not None:
            after()
        _, obs, rew, _, _ = sc.store(None, ow)
        obs_n = sc.split_obs(obs, world)
        reward_n = [float(r) for r in rew]
        info_list = [{}] * self.n
        if self.info_callback is not None:
            with sc.pinned():           # benchmark_data formats the step's own outputs, no second look at the world
                info_list = [sc.benchmark_data(a, world) for a in self.agents]
        if sc.dynamic and self._fused_post_step:
            # the kernel ran post_step after the callbacks: take over its pick-ups, hand-overs and respawns
            sc.absorb_from(world, ow, self.np_random)
        return obs_n, reward_n, info_list

    def _get_obs(self, agent):
        return np.zeros(0) if self.observation_callback is None else self.observation_callback(agent, self.world)

    def _get_reward(self, agent):
        return 0.0 if self.reward_callback is None else self.reward_callback(agent, self.world)

    def _get_done(self, agent):
        return False if self.done_callback is None else self.done_callback(agent, self.world)

    def _get_info(self, agent):
        return {} if self.info_callback is None else self.info_callback(agent, self.world)

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering is outside this package; read entity states from env.world")


def make_env(scenario_name, benchmark=False):
    """make_env.py:15-44: build a scenario by name and wrap it in a MultiAgentEnv."""
    from . import scenarios
    scenario = scenarios.load(scenario_name + ".py").Scenario()
    world = scenario.make_world()
    if benchmark:
        return MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation, scenario.benchmark_data)
    return MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation)
