"""Static world descriptions (what ``Scenario.make_world`` builds) for the batched simulator."""
from dataclasses import dataclass, field

import numpy as np

from . import _lib


@dataclass
class WorldSpec:
    """Per-entity properties and world constants of one scenario, agents first then landmarks
    (the order of ``World.entities``, multiagent/core.py:127-129)."""

    scenario: int
    n_agents: int
    n_landmarks: int
    dim_c: int
    size: np.ndarray
    mass: np.ndarray
    movable: np.ndarray
    collide: np.ndarray
    sensitivity: np.ndarray      # Agent.accel or 5.0 when None (multiagent/environment.py:192-195)
    max_speed: np.ndarray        # +inf when Entity.max_speed is None (multiagent/core.py:199)
    silent: np.ndarray
    adversary: np.ndarray
    leader: np.ndarray
    # World.__init__ defaults, multiagent/core.py:117-124
    dt: float = 0.1
    damping: float = 0.25
    contact_force: float = 1e2
    contact_margin: float = 1e-3
    name: str = ""
    # World.walls (multiagent/core.py:26-42) as dicts {orient 'H'|'V', axis_pos, endpoints (lo, hi), width, hard};
    # Agent.ghost flags.  The physics-only step (SCN_NONE) simulates any walls; simple_predator_prey and
    # consensus carry their four arena walls; the other built-in scenarios have none.
    walls: list = field(default_factory=list)
    ghost: np.ndarray = None
    # World.collaborative: step returns the summed reward to every agent (multiagent/environment.py:38,103-105)
    collaborative: bool = False
    # World.step of multiagent/new_core.py instead of multiagent/core.py (physics-only step, SCN_NONE);
    # accel [A] is Agent.accel with nan for None (it scales the action force, new_core.py:214)
    new_core: bool = False
    accel: np.ndarray = None
    # Agent.u_noise / Agent.c_noise [A], 0 for None (multiagent/core.py:165,211); no built-in scenario sets them
    u_noise: np.ndarray = None
    c_noise: np.ndarray = None

    def to_c(self):
        """Returns (MpeSpec, keepalive) -- keepalive owns the arrays the struct points into."""
        E, A = self.n_agents + self.n_landmarks, self.n_agents
        f64 = lambda a, n: np.ascontiguousarray(np.asarray(a, np.float64).reshape(n))
        u8 = lambda a, n: np.ascontiguousarray(np.asarray(a, bool).astype(np.uint8).reshape(n))
        arrs = dict(size=f64(self.size, E), mass=f64(self.mass, E), movable=u8(self.movable, E), collide=u8(self.collide, E),
                    sensitivity=f64(self.sensitivity, A), max_speed=f64(self.max_speed, A), silent=u8(self.silent, A),
                    adversary=u8(self.adversary, A), leader=u8(self.leader, A))
        W = len(self.walls)
        if W:
            arrs.update(wall_vertical=u8([w["orient"] == "V" for w in self.walls], W), wall_axis_pos=f64([w["axis_pos"] for w in self.walls], W),
                        wall_end0=f64([w["endpoints"][0] for w in self.walls], W), wall_end1=f64([w["endpoints"][1] for w in self.walls], W),
                        wall_width=f64([w["width"] for w in self.walls], W), wall_hard=u8([w["hard"] for w in self.walls], W))
        if self.ghost is not None:
            arrs["ghost"] = u8(self.ghost, A)
        if self.new_core:
            arrs["accel"] = f64(np.full(A, np.nan) if self.accel is None else self.accel, A)
        for name in ("u_noise", "c_noise"):
            if getattr(self, name) is not None:
                arrs[name] = f64(getattr(self, name), A)
        s = _lib.MpeSpec(scenario=self.scenario, n_agents=A, n_landmarks=self.n_landmarks, dim_c=self.dim_c,
                         dt=self.dt, damping=self.damping, contact_force=self.contact_force, contact_margin=self.contact_margin,
                         n_walls=W, collaborative=int(bool(self.collaborative)),
                         physics=_lib.PHYSICS_NEW_CORE if self.new_core else _lib.PHYSICS_CORE)
        for k, a in arrs.items():
            ptr_t = _lib._p_f64 if a.dtype == np.float64 else _lib._p_u8
            setattr(s, k, a.ctypes.data_as(ptr_t))
        return s, arrs


def spread_spec(n_agents=3, n_landmarks=3):
    """simple_spread (multiagent/scenarios/simple_spread.py:8-38): agents of size 0.15 that collide
    and are silent, default acceleration (sensitivity 5.0), no speed limit; landmarks of the default
    size 0.05 that neither move nor collide.  The reference fixes 3/3; both counts are free here."""
    A, L = n_agents, n_landmarks
    return WorldSpec(_lib.SCN_SPREAD, A, L, 2,
                     size=np.r_[np.full(A, 0.15), np.full(L, 0.05)], mass=np.ones(A + L),
                     movable=np.r_[np.ones(A, bool), np.zeros(L, bool)], collide=np.r_[np.ones(A, bool), np.zeros(L, bool)],
                     sensitivity=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.ones(A, bool),
                     adversary=np.zeros(A, bool), leader=np.zeros(A, bool), name="simple_spread")


def tag_spec():
    """simple_tag (multiagent/scenarios/simple_tag.py:7-36): 3 adversaries (size .075, accel 3,
    max_speed 1) chase 1 good agent (size .05, accel 4, max_speed 1.3) around 2 obstacles (size .2)."""
    adv = np.array([True, True, True, False])
    return WorldSpec(_lib.SCN_TAG, 4, 2, 2,
                     size=np.r_[np.where(adv, 0.075, 0.05), [0.2, 0.2]], mass=np.ones(6),
                     movable=np.r_[np.ones(4, bool), np.zeros(2, bool)], collide=np.ones(6, bool),
                     sensitivity=np.where(adv, 3.0, 4.0), max_speed=np.where(adv, 1.0, 1.3), silent=np.ones(4, bool),
                     adversary=adv, leader=np.zeros(4, bool), name="simple_tag")


def world_comm_spec():
    """simple_world_comm (multiagent/scenarios/simple_world_comm.py:7-57): 4 adversaries (agent 0
    is the leader and the only speaker, dim_c 4) and 2 good agents (size .045); landmarks are
    [obstacle .2 (collides), food .03 x2, forest .3 x2] in that order (:52-53)."""
    adv = np.array([True] * 4 + [False] * 2)
    leader = np.arange(6) == 0
    return WorldSpec(_lib.SCN_WORLD_COMM, 6, 5, 4,
                     size=np.r_[np.where(adv, 0.075, 0.045), [0.2, 0.03, 0.03, 0.3, 0.3]], mass=np.ones(11),
                     movable=np.r_[np.ones(6, bool), np.zeros(5, bool)],
                     collide=np.r_[np.ones(6, bool), [True, False, False, False, False]],
                     sensitivity=np.where(adv, 3.0, 4.0), max_speed=np.where(adv, 1.0, 1.3), silent=~leader,
                     adversary=adv, leader=leader, name="simple_world_comm")


def simple_spec():
    """simple (multiagent/scenarios/simple.py:6-22): one silent agent and one landmark of the default
    size 0.05; nothing collides; default acceleration, no speed limit, World.dim_c 0."""
    return WorldSpec(_lib.SCN_SIMPLE, 1, 1, 0, size=np.full(2, 0.05), mass=np.ones(2),
                     movable=np.array([True, False]), collide=np.zeros(2, bool),
                     sensitivity=np.full(1, 5.0), max_speed=np.full(1, np.inf), silent=np.ones(1, bool),
                     adversary=np.zeros(1, bool), leader=np.zeros(1, bool), name="simple")


def adversary_spec():
    """simple_adversary (multiagent/scenarios/simple_adversary.py:8-33): agent 0 is the adversary,
    agents 1-2 are good, all of size 0.15, silent and non-colliding; 2 landmarks of size 0.08, one
    of which is drawn as the goal at every reset."""
    adv = np.array([True, False, False])
    return WorldSpec(_lib.SCN_ADVERSARY, 3, 2, 2, size=np.r_[np.full(3, 0.15), np.full(2, 0.08)], mass=np.ones(5),
                     movable=np.r_[np.ones(3, bool), np.zeros(2, bool)], collide=np.zeros(5, bool),
                     sensitivity=np.full(3, 5.0), max_speed=np.full(3, np.inf), silent=np.ones(3, bool),
                     adversary=adv, leader=np.zeros(3, bool), name="simple_adversary")


def push_spec():
    """simple_push (multiagent/scenarios/simple_push.py:6-32): adversary agent 0 and good agent 1 of
    the default size 0.05 that collide with each other; 2 non-colliding landmarks, one drawn as the
    goal at every reset."""
    adv = np.array([True, False])
    return WorldSpec(_lib.SCN_PUSH, 2, 2, 2, size=np.full(4, 0.05), mass=np.ones(4),
                     movable=np.r_[np.ones(2, bool), np.zeros(2, bool)], collide=np.r_[np.ones(2, bool), np.zeros(2, bool)],
                     sensitivity=np.full(2, 5.0), max_speed=np.full(2, np.inf), silent=np.ones(2, bool),
                     adversary=adv, leader=np.zeros(2, bool), name="simple_push")


def speaker_listener_spec():
    """simple_speaker_listener (multiagent/scenarios/simple_speaker_listener.py:6-31): agent 0 speaks
    (dim_c 3) and cannot move, agent 1 moves and is silent, both of size 0.075; 3 landmarks of size
    0.04; nothing collides; rewards are shared."""
    return WorldSpec(_lib.SCN_SPEAKER_LISTENER, 2, 3, 3, size=np.r_[np.full(2, 0.075), np.full(3, 0.04)], mass=np.ones(5),
                     movable=np.array([False, True, False, False, False]), collide=np.zeros(5, bool),
                     sensitivity=np.full(2, 5.0), max_speed=np.full(2, np.inf), silent=np.array([False, True]),
                     adversary=np.zeros(2, bool), leader=np.zeros(2, bool), name="simple_speaker_listener", collaborative=True)


def reference_spec():
    """simple_reference (multiagent/scenarios/simple_reference.py:6-24): two agents that both move and
    speak (dim_c 10), 3 landmarks, default sizes, nothing collides; rewards are shared."""
    return WorldSpec(_lib.SCN_REFERENCE, 2, 3, 10, size=np.full(5, 0.05), mass=np.ones(5),
                     movable=np.r_[np.ones(2, bool), np.zeros(3, bool)], collide=np.zeros(5, bool),
                     sensitivity=np.full(2, 5.0), max_speed=np.full(2, np.inf), silent=np.zeros(2, bool),
                     adversary=np.zeros(2, bool), leader=np.zeros(2, bool), name="simple_reference", collaborative=True)


def crypto_spec():
    """simple_crypto (multiagent/scenarios/simple_crypto.py:21-44): adversary Eve (agent 0), listener
    Bob (1) and speaker Alice (2); nobody moves, everybody speaks (dim_c 4); 2 landmarks."""
    return WorldSpec(_lib.SCN_CRYPTO, 3, 2, 4, size=np.full(5, 0.05), mass=np.ones(5),
                     movable=np.zeros(5, bool), collide=np.zeros(5, bool),
                     sensitivity=np.full(3, 5.0), max_speed=np.full(3, np.inf), silent=np.zeros(3, bool),
                     adversary=np.array([True, False, False]), leader=np.zeros(3, bool), name="simple_crypto")


ARENA_WALLS = tuple(dict(orient="H" if i % 2 == 0 else "V", axis_pos=-1.2 if i < 2 else 1.2, endpoints=(-1.2, 1.2), width=0.4, hard=True)
                    for i in range(4))


def predator_prey_spec():
    """simple_predator_prey (multiagent/scenarios/simple_predator_prey.py:7-81) on the new_core physics:
    3 adversaries (size .065, mass 2.25, max_speed 1; agent 0 leads) and 4 good agents (size .045, mass 1,
    max_speed 1.3; agent 3 leads, 4-6 carry food); landmarks [barrier .2 (collides), food .03 x3, forest
    .25 x2]; 4 hard walls of width .4 on the lines +-1.2; accel is None (default sensitivity 5)."""
    adv = np.arange(7) < 3
    return WorldSpec(_lib.SCN_PREDATOR_PREY, 7, 6, 3, size=np.r_[np.where(adv, 0.065, 0.045), [0.2, 0.03, 0.03, 0.03, 0.25, 0.25]],
                     mass=np.r_[np.where(adv, 2.25, 1.0), np.ones(6)], movable=np.r_[np.ones(7, bool), np.zeros(6, bool)],
                     collide=np.r_[np.ones(7, bool), [True, False, False, False, False, False]],
                     sensitivity=np.full(7, 5.0), max_speed=np.where(adv, 1.0, 1.3), silent=np.ones(7, bool), adversary=adv,
                     leader=(np.arange(7) == 0) | (np.arange(7) == 3), name="simple_predator_prey",
                     walls=[dict(w) for w in ARENA_WALLS], ghost=np.zeros(7, bool), new_core=True, accel=np.full(7, np.nan))


def consensus_spec():
    """consensus (multiagent/scenarios/consensus.py:8-44) on the new_core physics: 6 silent, non-colliding
    ghost agents of size .03 inside the 4 hard arena walls, no landmarks, shared rewards."""
    return WorldSpec(_lib.SCN_CONSENSUS, 6, 0, 2, size=np.full(6, 0.03), mass=np.ones(6), movable=np.ones(6, bool),
                     collide=np.zeros(6, bool), sensitivity=np.full(6, 5.0), max_speed=np.full(6, np.inf), silent=np.ones(6, bool),
                     adversary=np.zeros(6, bool), leader=np.zeros(6, bool), name="consensus", walls=[dict(w) for w in ARENA_WALLS],
                     ghost=np.ones(6, bool), new_core=True, accel=np.full(6, np.nan), collaborative=True)


def collector_spec():
    """collector (multiagent/scenarios/collector.py:8-71) on the new_core physics: 6 collectors (size .05,
    mass 1) and 2 deposits (size .075, mass 2.25), colliding ghosts with accel 1.5 and max_speed 1; 6
    treasures (size .025) that neither collide nor move; no walls."""
    col = np.arange(8) < 6
    return WorldSpec(_lib.SCN_COLLECTOR, 8, 6, 2, size=np.r_[np.where(col, 0.05, 0.075), np.full(6, 0.025)],
                     mass=np.r_[np.where(col, 1.0, 2.25), np.ones(6)], movable=np.r_[np.ones(8, bool), np.zeros(6, bool)],
                     collide=np.r_[np.ones(8, bool), np.zeros(6, bool)], sensitivity=np.full(8, 1.5), max_speed=np.ones(8),
                     silent=np.ones(8, bool), adversary=np.zeros(8, bool), leader=np.zeros(8, bool), name="collector",
                     ghost=np.ones(8, bool), new_core=True, accel=np.full(8, 1.5))


def physics_spec(n_agents, n_landmarks, dim_c, size, mass, collide, sensitivity, max_speed, silent, walls=(), ghost=None,
                 movable=None, **world):
    """A world stepped by ``mpe_world_step`` only (World.step, multiagent/core.py:142-158, or with
    ``new_core=True, accel=[...]`` multiagent/new_core.py:188-205); the scenario's observation /
    reward callbacks stay with the caller.  ``movable`` [E] defaults to agents yes, landmarks no."""
    A, L = n_agents, n_landmarks
    return WorldSpec(_lib.SCN_NONE, A, L, dim_c, size=np.asarray(size, float), mass=np.asarray(mass, float),
                     movable=np.r_[np.ones(A, bool), np.zeros(L, bool)] if movable is None else np.asarray(movable, bool),
                     collide=np.asarray(collide, bool),
                     sensitivity=np.asarray(sensitivity, float), max_speed=np.asarray(max_speed, float),
                     silent=np.asarray(silent, bool), adversary=np.zeros(A, bool), leader=np.zeros(A, bool),
                     name="physics", walls=list(walls), ghost=None if ghost is None else np.asarray(ghost, bool), **world)


def make_spec(name, **kw):
    if name == "simple_spread":
        return spread_spec(**kw)
    if name == "simple_tag":
        return tag_spec()
    if name == "simple_world_comm":
        return world_comm_spec()
    if name == "simple":
        return simple_spec()
    if name == "simple_adversary":
        return adversary_spec()
    if name == "simple_push":
        return push_spec()
    if name == "simple_speaker_listener":
        return speaker_listener_spec()
    if name == "simple_reference":
        return reference_spec()
    if name == "simple_crypto":
        return crypto_spec()
    if name == "simple_predator_prey":
        return predator_prey_spec()
    if name == "consensus":
        return consensus_spec()
    if name == "collector":
        return collector_spec()
    raise ValueError("no batched kernel for scenario %r" % (name,))
