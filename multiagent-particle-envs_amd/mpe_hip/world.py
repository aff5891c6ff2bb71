"""Single-world object model with the reference's attribute names, stepped on the GPU.

These classes keep the data model of the reference's ``multiagent/core.py`` (``World`` with
``agents`` / ``landmarks`` lists of entities carrying ``state.p_pos``, ``state.p_vel``, ``state.c``,
``action.u``, ``action.c`` and the static properties ``size``, ``movable``, ``collide``, ``mass``,
``accel``, ``max_speed``, ``silent`` ...) so that existing ``Scenario`` plugins keep working, but
``World.step()`` does no arithmetic in Python: it packs the entities into a one-world batch, runs
the ``mpe_world_step`` HIP kernel (action force, pairwise contact force, damped integration with
speed clamp, comm state) and unpacks the result.  There is no CPU physics path.
"""
import numpy as np

__all__ = ["EntityState", "AgentState", "Action", "Entity", "Landmark", "Agent", "Wall", "World"]


class EntityState:
    """Physical state (multiagent/core.py:4-9)."""

    def __init__(self):
        self.p_pos = None
        self.p_vel = None


class AgentState(EntityState):
    """Adds the communication utterance ``c`` (multiagent/core.py:12-16)."""

    def __init__(self):
        super().__init__()
        self.c = None


class Action:
    """Physical action ``u`` and communication action ``c`` (multiagent/core.py:19-24)."""

    def __init__(self):
        self.u = None
        self.c = None


class Wall:
    """Axis-aligned wall segment (multiagent/core.py:26-42); its contact force is part of the GPU step."""

    def __init__(self, orient="H", axis_pos=0.0, endpoints=(-1, 1), width=0.1, hard=True):
        self.orient, self.axis_pos, self.endpoints = orient, axis_pos, np.array(endpoints)
        self.width, self.hard, self.visible = width, hard, False
        self.color = np.zeros(3, dtype=np.float32)


class Entity:
    """Defaults of multiagent/core.py:45-71."""

    def __init__(self):
        self.i = 0                 # index among all entities (multiagent/new_core.py:50)
        self.name = ""
        self.size = 0.050
        self.movable = False
        self.collide = True
        self.density = 25.0
        self.color = None
        self.visible = True
        self.max_speed = None
        self.accel = None
        self.state = EntityState()
        self.initial_mass = 1.0

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    pass


class Agent(Entity):
    """Defaults of multiagent/core.py:79-101."""

    def __init__(self):
        super().__init__()
        self.movable = True
        self.silent = False
        self.blind = False
        self.ghost = False
        self.u_noise = None
        self.c_noise = None
        self.u_range = 1.0
        self.state = AgentState()
        self.action = Action()
        self.action_callback = None


class World:
    """World constants of multiagent/core.py:104-124; ``step()`` runs on the GPU."""

    def __init__(self):
        self.agents = []
        self.landmarks = []
        self.walls = []
        self.dim_c = 0
        self.dim_p = 2
        self.dim_color = 3
        self.dt = 0.1
        self.t = 0
        self.damping = 0.25
        self.contact_force = 1e2
        self.contact_margin = 1e-3
        self.device = "cuda:0"
        self.new_core = False      # World.step of multiagent/new_core.py (mpe_hip.new_core.World sets it)
        self._phys = None          # (signature, BatchedEnv) of the physics-only handle

    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        return [a for a in self.agents if a.action_callback is None]

    @property
    def scripted_agents(self):
        return [a for a in self.agents if a.action_callback is not None]

    # ------------------------------------------------------------------
    def _signature(self):
        ents = self.entities
        return (len(self.agents), len(self.landmarks), self.dim_c, self.dt, self.damping, self.contact_force, self.contact_margin, self.new_core,
                tuple((e.size, e.mass, bool(e.movable), bool(e.collide)) for e in ents),
                tuple((a.max_speed, bool(a.silent), bool(getattr(a, "ghost", False)), a.accel if self.new_core else None) for a in self.agents),
                tuple((w.orient, w.axis_pos, tuple(w.endpoints), w.width, bool(w.hard)) for w in self.walls), str(self.device))

    def _physics_world(self):
        """One-world physics handle (with its pinned host mirror) matching the entities' current static properties."""
        from . import physics_spec
        from .oneworld import OneWorld
        sig = self._signature()
        if self._phys is None or self._phys[0] != sig:
            A = len(self.agents)
            ents = self.entities
            spec = physics_spec(A, len(self.landmarks), self.dim_c,
                                size=[e.size for e in ents], mass=[e.mass for e in ents], collide=[e.collide for e in ents],
                                sensitivity=[a.accel if a.accel is not None else 5.0 for a in self.agents],
                                max_speed=[a.max_speed if a.max_speed is not None else np.inf for a in self.agents],
                                silent=[a.silent for a in self.agents],
                                walls=[dict(orient=w.orient, axis_pos=w.axis_pos, endpoints=tuple(w.endpoints), width=w.width, hard=w.hard)
                                       for w in self.walls],
                                ghost=[getattr(a, "ghost", False) for a in self.agents],
                                movable=[bool(e.movable) for e in ents], new_core=self.new_core,
                                accel=[np.nan if a.accel is None else a.accel for a in self.agents],
                                dt=self.dt, damping=self.damping, contact_force=self.contact_force, contact_margin=self.contact_margin)
            self._phys = (sig, OneWorld(spec, device=self.device))
        return self._phys[1]

    def pack_state(self):
        """(pos [1, E, 2], vel [1, A, 2], comm [1, S*dim_c]) float64 from the entity objects."""
        pos = np.array([e.state.p_pos for e in self.entities], dtype=np.float64)[None]
        vel = np.array([a.state.p_vel for a in self.agents], dtype=np.float64)[None]
        comm = [np.asarray(a.state.c, dtype=np.float64) if a.state.c is not None else np.zeros(self.dim_c)
                for a in self.agents if not a.silent]
        return pos, vel, (np.concatenate(comm)[None] if comm else None)

    def unpack_state(self, pos, vel, comm):
        A = len(self.agents)
        for i, a in enumerate(self.agents):
            a.state.p_pos = np.array(pos[0, i])
            a.state.p_vel = np.array(vel[0, i])
        k = 0
        for a in self.agents:
            if a.silent:
                a.state.c = np.zeros(self.dim_c, dtype=np.float32)      # multiagent/core.py:208-209
            else:
                a.state.c = np.array(comm[0, k * self.dim_c:(k + 1) * self.dim_c])
                k += 1

    def step(self):
        """World.step (multiagent/core.py:142-158): scripted agents act, then the GPU integrates."""
        self.t += 1
        for agent in self.scripted_agents:
            agent.action = agent.action_callback(agent, self)
        ow = self._physics_world()
        A = len(self.agents)
        u = np.zeros((1, 2 * A))
        c_rows = []
        # motor / comm noise is drawn on the host exactly where and in the order the reference draws it from numpy's
        # global generator: every movable agent's force noise first (apply_action_force, multiagent/core.py:163-166),
        # the speakers' comm noise after it (update_agent_state, core.py:157-158,211).  new_core adds the force noise
        # after scaling the action (new_core.py:213-214), so it is divided by the scale the kernel applies.
        for i, a in enumerate(self.agents):
            un = 0.0
            if a.movable and a.u_noise:
                un = np.random.randn(*np.shape(a.action.u)) * a.u_noise
                if self.new_core:
                    un = un / (a.mass * a.accel if a.accel is not None else a.mass)
            u[0, 2 * i:2 * i + 2] = np.asarray(a.action.u, dtype=np.float64) + un
        for a in self.agents:
            if not a.silent:
                cn = np.random.randn(*np.shape(a.action.c)) * a.c_noise if a.c_noise else 0.0
                c_rows.append(np.asarray(a.action.c, dtype=np.float64) + cn)
        pos, vel, comm = self.pack_state()
        ow.put_state(pos, vel, comm)
        ow.world_step(u, np.concatenate(c_rows) if c_rows else None)      # state up, one launch, state down
        pos, vel, comm = ow.state()
        self.unpack_state(pos[None], vel[None], None if comm is None else comm[None])
