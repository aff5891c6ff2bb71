"""CPU oracle for the particle-world step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, in plain numpy float64, the arithmetic of the reference
simulator's hot path so that the HIP kernels can be checked against it.  Only
``tests/``, ``__graft_entry__.smoke()``, ``tools/gen_golden.py`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under the product
package does.  Parity of this oracle itself is pinned by the golden fixtures in
``tests/golden/`` which ``tools/gen_golden.py`` captured by running the real
reference code (see ``tests/test_oracle_golden.py``).

Every array carries a leading batch axis ``B`` so the same code serves one env
(``B == 1``, the CPU-baseline timing) and thousands (parity sweeps).  All
elementwise operations are written in the *same order* as the reference's
scalar code so that float64 results are bit-identical, not merely close:

* ``pos``  float64 ``[B, E, 2]``   entity positions, agents first then landmarks
                                   (reference ``World.entities``, multiagent/core.py:127-129)
* ``vel``  float64 ``[B, A, 2]``   agent velocities (landmarks never move in scope)
* ``comm`` float64 ``[B, A, dim_c]`` agent communication state ``state.c``
* ``goal`` int ``[B]``             index of the goal landmark (``agent.goal_a``) in
                                   simple_adversary / simple_push / simple_speaker_listener
                                   (``agents[0].goal_b``), drawn by reset_world; ``[B, 2]`` for
                                   simple_reference (``goal_b`` of agent 0 and of agent 1) and
                                   simple_crypto (goal landmark, key landmark)
* ``actions`` int ``[B, n_act]``   flattened reference ``action_n`` list: for each agent
                                   its move int 0..4 if it is movable, followed by its comm
                                   int if it is not silent (multiagent/environment.py:173-204)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

SCN_SPREAD = "simple_spread"
SCN_TAG = "simple_tag"
SCN_WORLD_COMM = "simple_world_comm"
SCN_SIMPLE = "simple"
SCN_ADVERSARY = "simple_adversary"
SCN_PUSH = "simple_push"
SCN_SPEAKER_LISTENER = "simple_speaker_listener"
SCN_REFERENCE = "simple_reference"
SCN_CRYPTO = "simple_crypto"
SCN_PREDATOR_PREY = "simple_predator_prey"
SCN_CONSENSUS = "consensus"


@dataclass
class Spec:
    """Static description of one scenario's world (what ``make_world`` builds)."""

    scenario: str
    A: int
    L: int
    dim_c: int
    size: np.ndarray          # [E]
    movable: np.ndarray       # [E] bool
    collide: np.ndarray       # [E] bool
    mass: np.ndarray          # [E]
    sens: np.ndarray          # [A] accel if set else 5.0 (environment.py:192-195)
    max_speed: np.ndarray     # [A] inf when the reference has None
    silent: np.ndarray        # [A] bool
    adversary: np.ndarray     # [A] bool
    leader: np.ndarray        # [A] bool
    # World.__init__ defaults, multiagent/core.py:117-124
    dt: float = 0.1
    damping: float = 0.25
    contact_force: float = 1e2
    contact_margin: float = 1e-3
    # landmark roles for simple_world_comm (indices into the landmark list)
    food: List[int] = field(default_factory=list)
    forests: List[int] = field(default_factory=list)
    # World.walls as dicts {orient, axis_pos, endpoints, width, hard} (core.py:26-42) and Agent.ghost
    walls: List[dict] = field(default_factory=list)
    ghost: Optional[np.ndarray] = None
    # World.collaborative: every agent receives the sum of the rewards (environment.py:103-105)
    shared: bool = False
    # physics of multiagent/new_core.py instead of multiagent/core.py: the action force is scaled by
    # mass (* accel) (:214) and a contact between two movable entities is split by their mass ratio
    # (:290-294); accel [A] holds Agent.accel with nan for None
    new_core: bool = False
    accel: Optional[np.ndarray] = None
    # Agent.u_noise / Agent.c_noise [A] (core.py:85-87; None in the reference = 0 here): the standard deviation of
    # the normal draws added to the action force (core.py:165) and to the comm state (core.py:211)
    u_noise: Optional[np.ndarray] = None
    c_noise: Optional[np.ndarray] = None

    @property
    def E(self) -> int:
        return self.A + self.L

    @property
    def n_act(self) -> int:
        return int(np.count_nonzero(self.movable[:self.A]) + np.count_nonzero(~self.silent))

    @property
    def obs_dims(self) -> List[int]:
        A, L = self.A, self.L
        if self.scenario == SCN_SPREAD:
            return [4 + 2 * L + 2 * (A - 1)] * A
        if self.scenario == SCN_TAG:
            out = []
            for i in range(A):
                n_good_others = sum(1 for j in range(A) if j != i and not self.adversary[j])
                out.append(4 + 2 * L + 2 * (A - 1) + 2 * n_good_others)
            return out
        if self.scenario == SCN_WORLD_COMM:
            out = []
            for i in range(A):
                n_good_others = sum(1 for j in range(A) if j != i and not self.adversary[j])
                d = 4 + 2 * L + 2 * (A - 1) + 2 * n_good_others + 2
                if self.adversary[i]:
                    d += self.dim_c
                out.append(d)
            return out
        if self.scenario == SCN_SIMPLE:
            return [2 + 2 * L]
        if self.scenario == SCN_ADVERSARY:
            return [(0 if self.adversary[i] else 2) + 2 * L + 2 * (A - 1) for i in range(A)]
        if self.scenario == SCN_PUSH:
            return [2 + 2 * L + 2 * (A - 1) if self.adversary[i] else 2 + 2 + 3 + 2 * L + 3 * L + 2 * (A - 1) for i in range(A)]
        if self.scenario == SCN_SPEAKER_LISTENER:
            return [3, 2 + 2 * L + self.dim_c]
        if self.scenario == SCN_REFERENCE:
            return [2 + 2 * L + 3 + self.dim_c] * A
        if self.scenario == SCN_CRYPTO:
            return [self.dim_c, 2 * self.dim_c, 2 * self.dim_c]
        if self.scenario == "collector":
            # pos, vel, own holding (collectors), 7 other agents x (offset, vel, encoding), 6 treasures x (offset, type)
            return [4 + (2 if i < 6 else 0) + 7 * 8 + 6 * 4 for i in range(A)]
        if self.scenario == SCN_CONSENSUS:
            return [2 * 2 * CONSENSUS_HISTORY * 2] * A      # 2 neighbours x (velocity, position) history x 2 coordinates
        if self.scenario == SCN_PREDATOR_PREY:
            # vel, pos, own holding (members), others' pos and vel, members' holding, landmarks, in_forest
            return [4 + 2 * (A - 1) * 2 + 3 + 2 * L + 2] * A
        raise ValueError(self.scenario)


def spec_spread(A: int = 3, L: int = 3) -> Spec:
    """simple_spread.make_world, multiagent/scenarios/simple_spread.py:8-38.

    The reference hard-codes 3 agents / 3 landmarks (:12-13); A and L are
    parameters here for the scaled 64/64 configuration.
    """
    E = A + L
    size = np.array([0.15] * A + [0.05] * L)                  # :29, core.py:50
    movable = np.array([True] * A + [False] * L)              # core.py:83, :35
    collide = np.array([True] * A + [False] * L)              # :27, :34
    return Spec(SCN_SPREAD, A, L, 2, size, movable, collide, np.ones(E),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf),
                silent=np.ones(A, bool), adversary=np.zeros(A, bool),
                leader=np.zeros(A, bool))


def spec_tag() -> Spec:
    """simple_tag.make_world, multiagent/scenarios/simple_tag.py:7-36."""
    n_adv, n_good, L = 3, 1, 2
    A = n_adv + n_good
    adv = np.array([True] * n_adv + [False] * n_good)
    size = np.concatenate([np.where(adv, 0.075, 0.05), np.full(L, 0.2)])   # :22, :32
    movable = np.array([True] * A + [False] * L)
    collide = np.ones(A + L, bool)                                          # :19, :30
    return Spec(SCN_TAG, A, L, 2, size, movable, collide, np.ones(A + L),
                sens=np.where(adv, 3.0, 4.0),                               # :23
                max_speed=np.where(adv, 1.0, 1.3),                          # :25
                silent=np.ones(A, bool), adversary=adv, leader=np.zeros(A, bool))


def spec_world_comm() -> Spec:
    """simple_world_comm.make_world, multiagent/scenarios/simple_world_comm.py:7-57.

    Landmark order is [obstacle, food0, food1, forest0, forest1] (:52-53).
    """
    n_adv, n_good = 4, 2
    A = n_adv + n_good
    L = 5
    adv = np.array([True] * n_adv + [False] * n_good)
    size = np.concatenate([np.where(adv, 0.075, 0.045), [0.2, 0.03, 0.03, 0.3, 0.3]])  # :26,:36,:43,:50
    movable = np.array([True] * A + [False] * L)
    collide = np.array([True] * A + [True, False, False, False, False])                # :22,:34,:41,:48
    leader = np.array([i == 0 for i in range(A)])                                       # :23
    silent = ~leader                                                                    # :24
    return Spec(SCN_WORLD_COMM, A, L, 4, size, movable, collide, np.ones(A + L),
                sens=np.where(adv, 3.0, 4.0), max_speed=np.where(adv, 1.0, 1.3),        # :27,:29
                silent=silent, adversary=adv, leader=leader, food=[1, 2], forests=[3, 4])


def spec_simple() -> Spec:
    """simple.make_world, multiagent/scenarios/simple.py:6-22 (World.dim_c stays 0)."""
    return Spec(SCN_SIMPLE, 1, 1, 0, np.full(2, 0.05), np.array([True, False]), np.zeros(2, bool), np.ones(2),
                sens=np.full(1, 5.0), max_speed=np.full(1, np.inf), silent=np.ones(1, bool),
                adversary=np.zeros(1, bool), leader=np.zeros(1, bool))


def spec_adversary() -> Spec:
    """simple_adversary.make_world, multiagent/scenarios/simple_adversary.py:8-33."""
    A, L = 3, 2
    adv = np.array([True, False, False])                                   # :22
    size = np.array([0.15] * A + [0.08] * L)                               # :23, :30
    return Spec(SCN_ADVERSARY, A, L, 2, size, np.array([True] * A + [False] * L), np.zeros(A + L, bool), np.ones(A + L),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.ones(A, bool),
                adversary=adv, leader=np.zeros(A, bool))


def spec_push() -> Spec:
    """simple_push.make_world, multiagent/scenarios/simple_push.py:6-32."""
    A, L = 2, 2
    adv = np.array([True, False])                                          # :19-22
    collide = np.array([True] * A + [False] * L)                           # :17, :28
    return Spec(SCN_PUSH, A, L, 2, np.full(A + L, 0.05), np.array([True] * A + [False] * L), collide, np.ones(A + L),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.ones(A, bool),
                adversary=adv, leader=np.zeros(A, bool))


def spec_speaker_listener() -> Spec:
    """simple_speaker_listener.make_world, multiagent/scenarios/simple_speaker_listener.py:6-31:
    agent 0 speaks and cannot move, agent 1 moves and is silent; rewards are shared (:11)."""
    A, L = 2, 3
    size = np.array([0.075] * A + [0.04] * L)                              # :17, :28
    movable = np.array([False, True] + [False] * L)                        # :19, :27
    return Spec(SCN_SPEAKER_LISTENER, A, L, 3, size, movable, np.zeros(A + L, bool), np.ones(A + L),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.array([False, True]),   # :21
                adversary=np.zeros(A, bool), leader=np.zeros(A, bool), shared=True)


def spec_reference() -> Spec:
    """simple_reference.make_world, multiagent/scenarios/simple_reference.py:6-24: two agents that
    both move and speak (dim_c 10), default sizes, nothing collides, shared rewards (:10)."""
    A, L = 2, 3
    return Spec(SCN_REFERENCE, A, L, 10, np.full(A + L, 0.05), np.array([True] * A + [False] * L), np.zeros(A + L, bool),
                np.ones(A + L), sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.zeros(A, bool),
                adversary=np.zeros(A, bool), leader=np.zeros(A, bool), shared=True)


def spec_crypto() -> Spec:
    """simple_crypto.make_world, multiagent/scenarios/simple_crypto.py:21-44: adversary (Eve),
    listener (Bob) and speaker (Alice, agent 2); nobody moves, everybody speaks (dim_c 4)."""
    A, L = 3, 2
    return Spec(SCN_CRYPTO, A, L, 4, np.full(A + L, 0.05), np.zeros(A + L, bool), np.zeros(A + L, bool), np.ones(A + L),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.zeros(A, bool),
                adversary=np.array([True, False, False]), leader=np.zeros(A, bool))


ARENA_WALLS = [dict(orient="H" if i % 2 == 0 else "V", axis_pos=-1.2 if i < 2 else 1.2, endpoints=(-1.2, 1.2), width=0.4, hard=True)
               for i in range(4)]


def spec_predator_prey() -> Spec:
    """simple_predator_prey.make_world, multiagent/scenarios/simple_predator_prey.py:7-81, on the
    new_core physics: 3 adversaries (agent 0 leads) and 4 good agents (agent 3 leads; 4-6 are the
    members that carry food); landmarks [barrier, food x3, forest x2]; 4 walls around the arena."""
    A, L = 7, 6
    adv = np.arange(A) < 3
    size = np.concatenate([np.where(adv, 0.065, 0.045), [0.2, 0.03, 0.03, 0.03, 0.25, 0.25]])   # :39,49,58,67
    mass = np.concatenate([np.where(adv, 2.25, 1.0), np.ones(L)])                              # :40
    collide = np.array([True] * A + [True] + [False] * 5)                                       # :34,47,56,65
    leader = (np.arange(A) == 0) | (np.arange(A) == 3)                                          # :35
    return Spec(SCN_PREDATOR_PREY, A, L, 3, size, np.array([True] * A + [False] * L), collide, mass,
                sens=np.full(A, 5.0), max_speed=np.where(adv, 1.0, 1.3), silent=np.ones(A, bool), adversary=adv, leader=leader,
                food=[1, 2, 3], forests=[4, 5], walls=[dict(w) for w in ARENA_WALLS], ghost=np.zeros(A, bool),
                new_core=True, accel=np.full(A, np.nan))


CONSENSUS_HISTORY = 10


def spec_consensus() -> Spec:
    """consensus.make_world, multiagent/scenarios/consensus.py:8-44, on the new_core physics: 6 silent,
    non-colliding ghost agents of size 0.03 in the walled arena (the walls are hard, so they still act
    on ghosts, new_core.py:302); no landmarks; shared rewards (:24)."""
    A = 6
    return Spec(SCN_CONSENSUS, A, 0, 2, np.full(A, 0.03), np.ones(A, bool), np.zeros(A, bool), np.ones(A),
                sens=np.full(A, 5.0), max_speed=np.full(A, np.inf), silent=np.ones(A, bool), adversary=np.zeros(A, bool),
                leader=np.zeros(A, bool), walls=[dict(w) for w in ARENA_WALLS], ghost=np.ones(A, bool), shared=True,
                new_core=True, accel=np.full(A, np.nan))


def make_spec(name: str, **kw) -> Spec:
    if name == "collector":
        return spec_collector()
    if name == SCN_CONSENSUS:
        return spec_consensus()
    if name == SCN_PREDATOR_PREY:
        return spec_predator_prey()
    if name == SCN_SPEAKER_LISTENER:
        return spec_speaker_listener()
    if name == SCN_REFERENCE:
        return spec_reference()
    if name == SCN_CRYPTO:
        return spec_crypto()
    if name == SCN_SIMPLE:
        return spec_simple()
    if name == SCN_ADVERSARY:
        return spec_adversary()
    if name == SCN_PUSH:
        return spec_push()
    if name == SCN_SPREAD:
        return spec_spread(**kw)
    if name == SCN_TAG:
        return spec_tag()
    if name == SCN_WORLD_COMM:
        return spec_world_comm()
    raise ValueError(name)


# --------------------------------------------------------------------------
# MultiAgentEnv._set_action, multiagent/environment.py:155-206 (discrete int input)
# --------------------------------------------------------------------------
def decode_actions(spec: Spec, actions: np.ndarray):
    """int actions [B, n_act] -> (u [B, A, 2], c [B, A, dim_c])."""
    actions = np.asarray(actions)
    B = actions.shape[0]
    u = np.zeros((B, spec.A, 2))
    c = np.zeros((B, spec.A, spec.dim_c))
    col = 0
    for i in range(spec.A):
        if spec.movable[i]:                                 # :173
            a = actions[:, col]
            col += 1
            # :178-181 ; ints outside 1..4 leave u at zero
            u[:, i, 0] = np.where(a == 1, -1.0, np.where(a == 2, 1.0, 0.0))
            u[:, i, 1] = np.where(a == 3, -1.0, np.where(a == 4, 1.0, 0.0))
            u[:, i, :] *= spec.sens[i]                      # :192-195
        if not spec.silent[i]:
            k = actions[:, col]
            col += 1
            c[np.arange(B), i, k] = 1.0                 # :199-201
    assert col == spec.n_act
    return u, c


# --------------------------------------------------------------------------
# World.step, multiagent/core.py:142-158
# --------------------------------------------------------------------------
def _dist(d):
    """sqrt(sum(square(delta))) for a length-2 delta, core.py:224."""
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])


def collision_force(spec: Spec, pa, pb, size_a, size_b):
    """World.get_collision_force arithmetic, multiagent/core.py:222-230."""
    delta = pa - pb
    dist = _dist(delta)
    dist_min = size_a + size_b
    k = spec.contact_margin
    with np.errstate(all="ignore"):
        pen = np.logaddexp(0, -(dist - dist_min) / k) * k
        force = spec.contact_force * delta / dist[..., None] * pen[..., None]
    return force


def wall_force(spec: Spec, p, size, ghost, wall):
    """World.get_wall_collision_force, multiagent/core.py:270-307, for positions p [B, 2].
    Returns (force [B, 2], hit [B] bool); hit is None when the wall never acts (ghost at a soft wall)."""
    if ghost and not wall["hard"]:
        return None, None
    prll_dim, perp_dim = (0, 1) if wall["orient"] == "H" else (1, 0)
    e0, e1 = wall["endpoints"]
    prll, perp = p[:, prll_dim], p[:, perp_dim]
    hit = ~((prll < e0 - size) | (prll > e1 + size))
    beyond = (prll < e0) | (prll > e1)
    past = np.where(prll < e0, prll - e0, prll - e1)
    with np.errstate(all="ignore"):
        theta = np.where(beyond, np.arcsin(past / size), 0.0)
        dist_min = np.where(beyond, np.cos(theta) * size + 0.5 * wall["width"], size + 0.5 * wall["width"])
        delta = perp - wall["axis_pos"]
        dist = np.abs(delta)
        k = spec.contact_margin
        pen = np.logaddexp(0, -(dist - dist_min) / k) * k
        mag = spec.contact_force * delta / dist * pen
        f = np.zeros_like(p)
        f[:, perp_dim] = np.cos(theta) * mag
        f[:, prll_dim] = np.sin(theta) * np.abs(mag)
    return f, hit


def world_step(spec: Spec, pos, vel, u, c, noise_u=None, noise_c=None):
    """One World.step.  Returns new (pos, vel, comm); inputs are not modified.

    noise_u [B, A, 2] / noise_c [B, A, dim_c]: the standard-normal draws the reference takes with
    np.random.randn for an agent whose u_noise / c_noise is set (core.py:165,211), passed in so that the
    step stays a function of its arguments; they are scaled by spec.u_noise / spec.c_noise here."""
    A, E = spec.A, spec.E
    pos = np.array(pos, dtype=np.float64)
    vel = np.array(vel, dtype=np.float64)
    B = pos.shape[0]
    # apply_action_force, core.py:161-167: noise = randn(2) * u_noise if u_noise else 0.0
    force: List[Optional[np.ndarray]] = [None] * E
    for i in range(A):
        if spec.movable[i]:
            un = spec.u_noise[i] if spec.u_noise is not None else 0.0
            noise = noise_u[:, i, :] * un if un else 0.0
            if spec.new_core:                           # new_core.py:214
                acc = spec.accel[i] if spec.accel is not None else np.nan
                scale = spec.mass[i] * acc if not np.isnan(acc) else spec.mass[i]
                force[i] = scale * u[:, i, :] + noise
            else:
                force[i] = u[:, i, :] + noise
    # apply_environment_force, core.py:170-181
    for a in range(E):
        for b in range(a + 1, E):
            if (not spec.collide[a]) or (not spec.collide[b]):
                continue                                # core.py:216-217
            if (not spec.movable[a]) and (not spec.movable[b]):
                continue                                # core.py:218-219
            f = collision_force(spec, pos[:, a, :], pos[:, b, :], spec.size[a], spec.size[b])
            fa, fb = +f, -f
            if spec.new_core and spec.movable[a] and spec.movable[b]:
                ratio = spec.mass[b] / spec.mass[a]     # new_core.py:290-294
                fa = ratio * f
                fb = -(1 / ratio) * f
            if spec.movable[a]:
                if force[a] is None:
                    force[a] = np.zeros((B, 2))
                force[a] = fa + force[a]                # core.py:178
            if spec.movable[b]:
                if force[b] is None:
                    force[b] = np.zeros((B, 2))
                force[b] = fb + force[b]                # core.py:181
    # wall contact for every movable entity, after its entity contacts (core.py:182-188)
    for i in range(A):
        if not spec.movable[i]:
            continue
        for wall in spec.walls:
            wf, hit = wall_force(spec, pos[:, i, :], spec.size[i], bool(spec.ghost[i]) if spec.ghost is not None else False, wall)
            if hit is None:
                continue
            if force[i] is None:
                force[i] = np.zeros((B, 2))
            force[i] = np.where(hit[:, None], force[i] + wf, force[i])
    # integrate_state, core.py:193-204
    for i in range(A):
        if not spec.movable[i]:
            continue
        v = vel[:, i, :] * (1 - spec.damping)
        if force[i] is not None:
            v = v + (force[i] / spec.mass[i]) * spec.dt
        if np.isfinite(spec.max_speed[i]):
            with np.errstate(all="ignore"):
                speed = np.sqrt(np.square(v[:, 0]) + np.square(v[:, 1]))
                over = speed > spec.max_speed[i]
                clamped = v / speed[:, None] * spec.max_speed[i]
            v = np.where(over[:, None], clamped, v)
        vel[:, i, :] = v
        pos[:, i, :] = pos[:, i, :] + v * spec.dt
    # update_agent_state, core.py:206-212
    comm = np.zeros((B, A, spec.dim_c))
    for i in range(A):
        if not spec.silent[i]:
            cn = spec.c_noise[i] if spec.c_noise is not None else 0.0
            comm[:, i, :] = c[:, i, :] + (noise_c[:, i, :] * cn if cn else 0.0)      # core.py:211
    return pos, vel, comm


# --------------------------------------------------------------------------
# scenario callbacks
# --------------------------------------------------------------------------
def _is_collision(pos, i, j, size):
    """Scenario.is_collision (simple_spread.py:77-81, simple_tag.py:69-73,
    simple_world_comm.py:126-130): strict ``dist < size_i + size_j``."""
    d = pos[:, i, :] - pos[:, j, :]
    return _dist(d) < (size[i] + size[j])


def _bound(x):
    """bound(x) of simple_tag.py:103-108 / simple_world_comm.py:167-172."""
    with np.errstate(all="ignore"):
        big = np.minimum(np.exp(2 * x - 2), 10)
    return np.where(x < 0.9, 0.0, np.where(x < 1.0, (x - 0.9) * 10, big))


def _goal_pos(spec: Spec, pos, goal):
    """Position of each env's goal landmark, [B, 2]."""
    return pos[np.arange(pos.shape[0]), spec.A + np.asarray(goal)]


def observe(spec: Spec, pos, vel, comm, goal=None) -> List[np.ndarray]:
    """Per-agent observations.  Returns a list of A arrays ``[B, D_i]``."""
    A, L = spec.A, spec.L
    B = pos.shape[0]
    out = []
    if spec.scenario == SCN_SIMPLE:
        # simple.py:45-50
        return [np.concatenate([vel[:, 0, :]] + [pos[:, A + l, :] - pos[:, 0, :] for l in range(L)], axis=1)]
    if spec.scenario == SCN_ADVERSARY:
        # simple_adversary.py:119-138
        g = _goal_pos(spec, pos, goal)
        for i in range(A):
            parts = [] if spec.adversary[i] else [g - pos[:, i, :]]
            parts += [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            parts += [pos[:, j, :] - pos[:, i, :] for j in range(A) if j != i]
            out.append(np.concatenate(parts, axis=1))
        return out
    if spec.scenario == SCN_PUSH:
        # simple_push.py:75-95; colours of reset_world :35-48
        g = _goal_pos(spec, pos, goal)
        lm_color = []
        for l in range(L):
            c = np.array([0.1, 0.1, 0.1])
            c[l + 1] += 0.8
            lm_color.append(np.broadcast_to(c, (B, 3)))
        for i in range(A):
            ent = [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            oth = [pos[:, j, :] - pos[:, i, :] for j in range(A) if j != i]
            if spec.adversary[i]:
                out.append(np.concatenate([vel[:, i, :]] + ent + oth, axis=1))
            else:
                color = np.full((B, 3), 0.25)
                color[np.arange(B), np.asarray(goal) + 1] += 0.5
                out.append(np.concatenate([vel[:, i, :], g - pos[:, i, :], color] + ent + lm_color + oth, axis=1))
        return out
    if spec.scenario == SCN_SPEAKER_LISTENER:
        # simple_speaker_listener.py:69-91; landmark colours of reset_world :45-47
        color = np.full((B, 3), 0.15)
        color[np.arange(B), np.asarray(goal)] = 0.65
        ent = [pos[:, A + l, :] - pos[:, 1, :] for l in range(L)]
        return [color, np.concatenate([vel[:, 1, :]] + ent + [comm[:, 0, :]], axis=1)]
    if spec.scenario == SCN_REFERENCE:
        # simple_reference.py:61-80; landmark colours of reset_world :40-42
        goal = np.asarray(goal)
        for i in range(A):
            color = np.full((B, 3), 0.25)
            color[np.arange(B), goal[:, i]] = 0.75
            ent = [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            out.append(np.concatenate([vel[:, i, :]] + ent + [color] + [comm[:, j, :] for j in range(A) if j != i], axis=1))
        return out
    if spec.scenario == SCN_CRYPTO:
        # simple_crypto.py:124-169; one-hot landmark colours of reset_world :55-59
        goal = np.asarray(goal)
        goal_color = np.zeros((B, spec.dim_c))
        goal_color[np.arange(B), goal[:, 0]] = 1.0
        key = np.zeros((B, spec.dim_c))
        key[np.arange(B), goal[:, 1]] = 1.0
        said = comm[:, 2, :]                                 # the speaker's utterance
        return [said.copy(), np.concatenate([key, said], axis=1), np.concatenate([goal_color, key], axis=1)]
    if spec.scenario == SCN_SPREAD:
        # simple_spread.py:110-130 ; result cast to float32 (:130)
        for i in range(A):
            parts = [vel[:, i, :], pos[:, i, :]]
            parts += [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            parts += [pos[:, j, :] - pos[:, i, :] for j in range(A) if j != i]
            out.append(np.concatenate(parts, axis=1).astype(np.float32))
        return out
    if spec.scenario == SCN_TAG:
        # simple_tag.py:131-147 ; float64
        for i in range(A):
            parts = [vel[:, i, :], pos[:, i, :]]
            parts += [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            parts += [pos[:, j, :] - pos[:, i, :] for j in range(A) if j != i]
            parts += [vel[:, j, :] for j in range(A) if j != i and not spec.adversary[j]]
            out.append(np.concatenate(parts, axis=1))
        return out
    if spec.scenario == SCN_WORLD_COMM:
        # simple_world_comm.py:224-287
        f1, f2 = A + spec.forests[0], A + spec.forests[1]
        inf = np.stack([np.stack([_is_collision(pos, i, f, spec.size) for f in (f1, f2)], axis=1)
                        for i in range(A)], axis=1)            # [B, A, 2] bool
        for i in range(A):
            parts = [vel[:, i, :], pos[:, i, :]]
            parts += [pos[:, A + l, :] - pos[:, i, :] for l in range(L)]
            inf1, inf2 = inf[:, i, 0], inf[:, i, 1]
            in_forest = np.where(inf[:, i, :], 1.0, -1.0)      # :231-239
            other_pos, other_vel = [], []
            for j in range(A):
                if j == i:
                    continue
                of1, of2 = inf[:, j, 0], inf[:, j, 1]
                vis = (inf1 & of1) | (inf2 & of2) | (~inf1 & ~of1 & ~inf2 & ~of2)
                if spec.leader[i]:
                    vis = np.ones(B, bool)                      # :254
                other_pos.append(np.where(vis[:, None], pos[:, j, :] - pos[:, i, :], 0.0))
                if not spec.adversary[j]:
                    other_vel.append(np.where(vis[:, None], vel[:, j, :], 0.0))
            c0 = comm[:, 0, :]                                  # :279
            if spec.adversary[i]:
                parts = parts + other_pos + other_vel + [in_forest, c0]      # :281-285
            else:
                parts = parts + other_pos + [in_forest] + other_vel          # :287
            out.append(np.concatenate(parts, axis=1))
        return out
    raise ValueError(spec.scenario)


def visibility_masks(spec: Spec, pos):
    """Integer masks of simple_world_comm.observation: (in_forest [B, A, 2] bool,
    visible [B, A, A] bool with the diagonal False)."""
    A = spec.A
    B = pos.shape[0]
    f1, f2 = A + spec.forests[0], A + spec.forests[1]
    inf = np.stack([np.stack([_is_collision(pos, i, f, spec.size) for f in (f1, f2)], axis=1)
                    for i in range(A)], axis=1)
    vis = np.zeros((B, A, A), bool)
    for i in range(A):
        for j in range(A):
            if i == j:
                continue
            v = (inf[:, i, 0] & inf[:, j, 0]) | (inf[:, i, 1] & inf[:, j, 1]) | \
                (~inf[:, i, 0] & ~inf[:, j, 0] & ~inf[:, i, 1] & ~inf[:, j, 1])
            vis[:, i, j] = True if spec.leader[i] else v
    return inf, vis


def _sq(d):
    """np.sum(np.square(d)) over the last axis of a 2-vector."""
    sq = np.square(d)
    return sq[..., 0] + sq[..., 1]


def reward(spec: Spec, pos, goal=None, comm=None) -> np.ndarray:
    """Per-agent rewards ``[B, A]`` float64 as the scenario's reward callback returns them, i.e.
    before MultiAgentEnv shares them (reference mixes Python ints and numpy floats; the values
    are equal)."""
    A, L = spec.A, spec.L
    B = pos.shape[0]
    rew = np.zeros((B, A))
    if spec.scenario == SCN_SPEAKER_LISTENER:
        # simple_speaker_listener.py:63-67: listener (agents[0].goal_a) to the goal landmark
        r = -_sq(pos[:, 1, :] - _goal_pos(spec, pos, goal))
        rew[:, 0] = r
        rew[:, 1] = r
        return rew
    if spec.scenario == SCN_REFERENCE:
        # simple_reference.py:55-59: the other agent (goal_a) to this agent's goal landmark (goal_b)
        goal = np.asarray(goal)
        for i in range(A):
            rew[:, i] = -_sq(pos[:, 1 - i, :] - _goal_pos(spec, pos, goal[:, i]))
        return rew
    if spec.scenario == SCN_CRYPTO:
        # simple_crypto.py:94-121: squared error of Bob's / Eve's utterance against the goal colour;
        # an all-zero utterance (only before the first step) is skipped
        goal_color = np.zeros((B, spec.dim_c))
        goal_color[np.arange(B), np.asarray(goal)[:, 0]] = 1.0
        err = [np.square(comm[:, i, :] - goal_color).sum(axis=1) for i in range(A)]
        spoke = [comm[:, i, :].any(axis=1) for i in range(A)]
        good_rew = np.where(spoke[1], 0 - err[1], 0.0)       # good_listeners = [agent 1]
        adv_rew = np.where(spoke[0], 0 + err[0], 0.0)        # adversaries = [agent 0]
        rew[:, 0] = np.where(spoke[0], 0 - err[0], 0.0)
        rew[:, 1] = adv_rew + good_rew
        rew[:, 2] = adv_rew + good_rew
        return rew
    if spec.scenario == SCN_SIMPLE:
        rew[:, 0] = -_sq(pos[:, 0, :] - pos[:, A, :])            # simple.py:41-43
        return rew
    if spec.scenario in (SCN_ADVERSARY, SCN_PUSH):
        g = _goal_pos(spec, pos, goal)
        good_i = [i for i in range(A) if not spec.adversary[i]]
        adv_i = [i for i in range(A) if spec.adversary[i]]
        d = [np.sqrt(_sq(pos[:, i, :] - g)) for i in range(A)]
        dmin = np.stack([d[i] for i in good_i], axis=0).min(axis=0)
        if spec.scenario == SCN_ADVERSARY:
            # simple_adversary.py:76-116, shaped branches
            adv_rew = np.zeros(B)
            for a in adv_i:
                adv_rew = adv_rew + d[a]
            for i in range(A):
                rew[:, i] = -_sq(pos[:, i, :] - g) if spec.adversary[i] else -dmin + adv_rew
        else:
            # simple_push.py:55-73
            for i in range(A):
                rew[:, i] = dmin - np.sqrt(_sq(g - pos[:, i, :])) if spec.adversary[i] else -d[i]
        return rew
    if spec.scenario == SCN_SPREAD:
        # simple_spread.py:83-108
        for i in range(A):
            r = np.zeros(B)
            if i == 0:
                for l in range(L):
                    dists = np.stack([_dist(pos[:, a, :] - pos[:, A + l, :]) for a in range(A)], axis=0)
                    r = r - dists.min(axis=0)
            pen = np.zeros(B)
            for a in range(A):
                if a != i:
                    pen = pen - _is_collision(pos, a, i, spec.size)
            rew[:, i] = r + pen
        return rew
    good = [i for i in range(A) if not spec.adversary[i]]
    advs = [i for i in range(A) if spec.adversary[i]]
    if spec.scenario == SCN_TAG:
        # simple_tag.py:84-129
        n_pairs = np.zeros(B)
        for g in good:
            for a in advs:
                n_pairs = n_pairs + 10 * _is_collision(pos, g, a, spec.size)
        for i in range(A):
            if spec.adversary[i]:
                rew[:, i] = n_pairs
            else:
                r = np.zeros(B)
                for a in advs:
                    r = r - 10 * _is_collision(pos, a, i, spec.size)
                for p in range(2):
                    r = r - _bound(np.abs(pos[:, i, p]))
                rew[:, i] = r
        return rew
    if spec.scenario == SCN_WORLD_COMM:
        # simple_world_comm.py:142-198
        hits = [(g, a, _is_collision(pos, g, a, spec.size)) for g in good for a in advs]
        for i in range(A):
            if spec.adversary[i]:
                dmin = np.stack([_dist(pos[:, g, :] - pos[:, i, :]) for g in good], axis=0).min(axis=0)
                r = 0 - 0.1 * dmin                                            # :192
                for _, _, hit in hits:                                        # :194-197, one add per pair
                    r = r + 5 * hit
                rew[:, i] = r
            else:
                r = np.zeros(B)
                for a in advs:
                    r = r - 5 * _is_collision(pos, a, i, spec.size)           # :164-166
                for p in range(2):
                    r = r - 2 * _bound(np.abs(pos[:, i, p]))                  # :174-176
                for f in spec.food:
                    r = r + 2 * _is_collision(pos, i, A + f, spec.size)       # :178-180
                fmin = np.stack([_dist(pos[:, A + f, :] - pos[:, i, :]) for f in spec.food],
                                axis=0).min(axis=0)
                r = r + 0.05 * fmin                                           # :181
                rew[:, i] = r
        return rew
    raise ValueError(spec.scenario)


def info(spec: Spec, pos, goal=None, comm=None):
    """benchmark_data.  Returns dict of arrays.

    simple_crypto (simple_crypto.py:78-80): ``c_goal`` [B, A, 2 * dim_c] -- per agent its utterance
    state.c followed by the goal landmark's one-hot colour.

    simple_adversary (simple_adversary.py:56-65): ``sq`` -- per agent the list of
    squared distances it reports (adversary: to the goal; good: to every landmark,
    then to the goal), concatenated in agent order [B, 1 + 2 * (L + 1)].

    simple_spread (simple_spread.py:58-74): ``rew`` [B, A] float, ``collisions``
    [B, A] int, ``min_dists`` [B] float, ``occupied`` [B] int.
    simple_tag / simple_world_comm (simple_tag.py:57-66,
    simple_world_comm.py:115-123): ``collisions`` [B, A] int (0 for good agents).
    """
    A, L = spec.A, spec.L
    B = pos.shape[0]
    if spec.scenario == SCN_CRYPTO:
        goal_color = np.zeros((B, spec.dim_c))
        goal_color[np.arange(B), np.asarray(goal)[:, 0]] = 1.0
        return {"c_goal": np.concatenate([comm, np.repeat(goal_color[:, None, :], A, axis=1)], axis=2)}
    if spec.scenario == SCN_ADVERSARY:
        g = _goal_pos(spec, pos, goal)
        cols = []
        for i in range(A):
            if not spec.adversary[i]:
                cols += [_sq(pos[:, i, :] - pos[:, A + l, :]) for l in range(L)]
            cols.append(_sq(pos[:, i, :] - g))
        return {"sq": np.stack(cols, axis=1)}
    if spec.scenario == SCN_SPREAD:
        min_dists = np.zeros(B)
        r0 = np.zeros(B)
        occupied = np.zeros(B, np.int64)
        for l in range(L):
            m = np.stack([_dist(pos[:, a, :] - pos[:, A + l, :]) for a in range(A)], axis=0).min(axis=0)
            min_dists = min_dists + m
            r0 = r0 - m
            occupied = occupied + (m < 0.1)
        coll = np.zeros((B, A), np.int64)
        rew = np.zeros((B, A))
        for i in range(A):
            r = r0.copy()
            for a in range(A):
                if a != i:
                    hit = _is_collision(pos, a, i, spec.size)
                    r = r - hit
                    coll[:, i] += hit
            rew[:, i] = r
        return {"rew": rew, "collisions": coll, "min_dists": min_dists, "occupied": occupied}
    coll = np.zeros((B, A), np.int64)
    good = [i for i in range(A) if not spec.adversary[i]]
    for i in range(A):
        if spec.adversary[i]:
            for g in good:
                coll[:, i] += _is_collision(pos, g, i, spec.size)
    return {"collisions": coll}


def share(spec: Spec, rew):
    """environment.py:103-105: with World.collaborative every agent receives np.sum(reward_n); for
    fewer than 8 values numpy adds them left to right (its pairwise blocking starts at 8)."""
    if not spec.shared:
        return rew
    assert spec.A < 8
    total = rew[:, 0]
    for i in range(1, spec.A):
        total = total + rew[:, i]
    return np.repeat(total[:, None], spec.A, axis=1)


def env_step(spec: Spec, pos, vel, actions, goal=None, noise_u=None, noise_c=None):
    """MultiAgentEnv.step, multiagent/environment.py:83-115, for B envs.

    Returns (pos, vel, comm, obs_list, rew [B, A], done [B] bool); rew is what step returns, i.e.
    after sharing when the world is collaborative.
    ``done`` is always False: no in-scope scenario supplies a done callback
    (environment.py:143-146).
    """
    u, c = decode_actions(spec, actions)
    pos, vel, comm = world_step(spec, pos, vel, u, c, noise_u, noise_c)
    obs = observe(spec, pos, vel, comm, goal)
    rew = share(spec, reward(spec, pos, goal, comm))
    return pos, vel, comm, obs, rew, np.zeros(pos.shape[0], bool)


def reset(spec: Spec, rng: np.random.RandomState, B: int = 1):
    """reset_world distributions (simple_spread.py:48-54, simple_tag.py:47-54,
    simple_world_comm.py:101-113): agents U(-1,1)^2; landmarks U(-1,1)^2 for
    simple_spread and U(-0.9,0.9)^2 otherwise; zero velocity and comm.

    The draw order of a single reference env is reproduced for B == 1 (the
    simple_world_comm food/forest re-draws included); for B > 1 envs are drawn
    one after another.
    """
    A, L = spec.A, spec.L
    pos = np.zeros((B, spec.E, 2))
    lim = 0.9 if spec.scenario in (SCN_TAG, SCN_WORLD_COMM) else 1.0
    for b in range(B):
        for i in range(A):
            pos[b, i] = rng.uniform(-1, +1, 2)
        for l in range(L):
            pos[b, A + l] = rng.uniform(-lim, +lim, 2)
        if spec.scenario == SCN_WORLD_COMM:
            for l in spec.food + spec.forests:
                pos[b, A + l] = rng.uniform(-0.9, +0.9, 2)
    return pos, np.zeros((B, A, 2)), np.zeros((B, A, spec.dim_c))


def reset_with_goal(spec: Spec, rng: np.random.RandomState, B: int = 1):
    """reset_world of simple_adversary (:35-54) / simple_push (:34-53) / simple_speaker_listener
    (:33-57): the goal landmark is drawn first (np.random.choice over the landmark list), then the
    positions as in ``reset``; simple_reference and simple_crypto draw two landmark indices.
    Returns (pos, vel, comm, goal [B] or [B, 2])."""
    two = spec.scenario in (SCN_REFERENCE, SCN_CRYPTO)      # simple_reference.py:33-35, simple_crypto.py:61-63
    goal = np.zeros((B, 2) if two else B, np.int64)
    pos = np.zeros((B, spec.E, 2))
    for b in range(B):
        goal[b] = [rng.choice(spec.L), rng.choice(spec.L)] if two else rng.choice(spec.L)
        pos[b] = reset(spec, rng, 1)[0][0]
    return pos, np.zeros((B, spec.A, 2)), np.zeros((B, spec.A, spec.dim_c)), goal


# --------------------------------------------------------------------------
# simple_predator_prey (multiagent/scenarios/simple_predator_prey.py), on the new_core physics.
# Beside pos / vel the world carries alive [B, 3] (food) and holding [B, 3] (the members, agents 4-6).
# The scenario reads distances from the cache World.step fills (new_core.py:147-170):
# cached_dist_vect[a, b] = p_a - p_b and its norm, the same bits as a direct evaluation.
# --------------------------------------------------------------------------
PP_MEMBERS = (4, 5, 6)
PP_LEADER_GOOD = 3


def pp_observe(spec: Spec, pos, vel, holding) -> List[np.ndarray]:
    """observation(), simple_predator_prey.py:300-349 (cast to float32 at :349)."""
    A, L = spec.A, spec.L
    B = pos.shape[0]
    f1, f2 = A + spec.forests[0], A + spec.forests[1]
    inf = np.stack([np.stack([_is_collision(pos, i, f, spec.size) for f in (f1, f2)], axis=1) for i in range(A)], axis=1)
    hold_val = np.where(holding, 1.0, -1.0)
    out = []
    for i in range(A):
        parts = [vel[:, i, :], pos[:, i, :]]
        if i in PP_MEMBERS:
            parts.append(hold_val[:, PP_MEMBERS.index(i), None])
        other_pos, other_vel, other_holding = [], [], []
        for j in range(A):
            if j == i:
                continue
            vis = (inf[:, i, 0] & inf[:, j, 0]) | (inf[:, i, 1] & inf[:, j, 1]) | \
                  (~inf[:, i, 0] & ~inf[:, j, 0] & ~inf[:, i, 1] & ~inf[:, j, 1])
            if spec.leader[i]:
                vis = np.ones(B, bool)
            other_pos.append(np.where(vis[:, None], pos[:, i, :] - pos[:, j, :], 0.0))     # cached_dist_vect[agent, other]
            other_vel.append(np.where(vis[:, None], vel[:, j, :], 0.0))
            if j in PP_MEMBERS:
                other_holding.append(np.where(vis, hold_val[:, PP_MEMBERS.index(j)], 0.0)[:, None])
        entity_pos = [pos[:, i, :] - pos[:, A + l, :] for l in range(L)]
        in_forest = np.where(inf[:, i, :], 1.0, -1.0)
        out.append(np.concatenate(parts + other_pos + other_vel + other_holding + entity_pos + [in_forest], axis=1).astype(np.float32))
    return out


def pp_reward(spec: Spec, pos, holding) -> np.ndarray:
    """reward(), simple_predator_prey.py:171-275."""
    A = spec.A
    B = pos.shape[0]
    advs = [i for i in range(A) if spec.adversary[i]]
    good = [i for i in range(A) if not spec.adversary[i]]
    foods = [A + f for f in spec.food]
    dist = lambda a, b: _dist(pos[:, a, :] - pos[:, b, :])
    hit = lambda a, b: _is_collision(pos, a, b, spec.size)
    rew = np.zeros((B, A))
    for i in range(A):
        r = np.zeros(B)
        if spec.adversary[i]:                                           # :243-275
            r = r - 0.1 * np.stack([dist(i, g) for g in good], axis=0).min(axis=0)
            n = np.zeros(B)
            for g in good:
                n = n + hit(i, g)
            r = r + 5 * n
            if spec.leader[i]:
                for g in good:
                    if g not in PP_MEMBERS:
                        continue                                        # only members ever hold food
                    n = np.zeros(B)
                    for a in advs:
                        n = n + hit(a, g)
                    r = np.where(holding[:, PP_MEMBERS.index(g)], r + 10 * n, r)
        else:                                                           # :183-241
            s = np.zeros(B)
            for a in advs:
                s = s + dist(a, i)
            r = r + 0.1 * s
            n = np.zeros(B)
            for a in advs:
                n = n + hit(a, i)
            r = r - 5 * n
            if not spec.leader[i]:
                h = holding[:, PP_MEMBERS.index(i)]
                to_food = np.stack([dist(f, i) for f in foods], axis=0).min(axis=0)
                r = np.where(h, r - 0.1 * dist(PP_LEADER_GOOD, i), r - 0.1 * to_food)
            else:
                d = np.stack([np.where(holding[:, k], dist(i, m), np.inf) for k, m in enumerate(PP_MEMBERS)], axis=0).min(axis=0)
                r = np.where(holding.any(axis=1), r - 0.1 * d, r - 0)
                for k, m in enumerate(PP_MEMBERS):
                    nf = np.zeros(B)
                    for f in foods:
                        nf = nf + hit(f, m)
                    r = np.where(holding[:, k], r + 5 * (hit(i, m) + 0.0), r + 5 * nf)
        rew[:, i] = r
    return rew


def pp_info(spec: Spec, pos):
    """benchmark_data(), simple_predator_prey.py:148-156: an adversary's collisions with good agents."""
    A = spec.A
    coll = np.zeros((pos.shape[0], A), np.int64)
    for i in range(A):
        if spec.adversary[i]:
            for g in range(A):
                if not spec.adversary[g]:
                    coll[:, i] += _is_collision(pos, g, i, spec.size)
    return {"collisions": coll}


def pp_post_step(spec: Spec, pos, alive, holding, respawn_pos):
    """post_step(), simple_predator_prey.py:83-107, with the distances World.step cached (i.e. of the
    positions in ``pos`` on entry).  respawn_pos [B, 3, 2] is where a dead food reappears (the
    reference draws U(-0.95, 0.95)^2; respawn_prob is 1).  Returns new (pos, alive, holding)."""
    A = spec.A
    pos = pos.copy()
    alive = alive.copy()
    holding = holding.copy()
    hit_food = {(f, m): _is_collision(pos, A + spec.food[f], m, spec.size) for f in range(3) for m in PP_MEMBERS}
    hit_leader = {m: _is_collision(pos, m, PP_LEADER_GOOD, spec.size) for m in PP_MEMBERS}
    for f in range(3):
        was_alive = alive[:, f].copy()
        taken = np.zeros(pos.shape[0], bool)
        for k, m in enumerate(PP_MEMBERS):
            pick = was_alive & ~taken & ~holding[:, k] & hit_food[(f, m)]
            holding[:, k] |= pick
            taken |= pick
        alive[:, f] = (was_alive & ~taken) | ~was_alive
        e = A + spec.food[f]
        pos[:, e, :] = np.where(taken[:, None], -999.0, np.where(was_alive[:, None], pos[:, e, :], respawn_pos[:, f, :]))
    for k, m in enumerate(PP_MEMBERS):
        holding[:, k] &= ~hit_leader[m]
    return pos, alive, holding


def pp_env_step(spec: Spec, pos, vel, actions, alive, holding, respawn_pos):
    """MultiAgentEnv.step (environment.py:83-115) with the scenario's post_step as post_step_callback
    (:107-108, run after the observations and rewards are taken).  Returns
    (pos, vel, obs_list, rew, alive, holding)."""
    u, c = decode_actions(spec, actions)
    pos, vel, _ = world_step(spec, pos, vel, u, c)
    obs = pp_observe(spec, pos, vel, holding)
    rew = pp_reward(spec, pos, holding)
    pos, alive, holding = pp_post_step(spec, pos, alive, holding, respawn_pos)
    return pos, vel, obs, rew, alive, holding


def done_outside(spec: Spec, pos, bound) -> np.ndarray:
    """MultiAgentEnv.step's done for the done_callback `lambda agent, world: any(|agent.state.p_pos| > bound)`:
    _get_done per agent (environment.py:98,143-146), then any(done_n) (:110-113).  pos [B, E, 2] after World.step."""
    done_n = [np.any(np.abs(pos[:, i, :]) > bound, axis=1) for i in range(spec.A)]
    return np.any(np.stack(done_n, axis=1), axis=1)


# --------------------------------------------------------------------------
# The counter-based generator of the on-device reset and respawn: Philox-4x32-10 (Salmon et al.,
# "Parallel random numbers: as easy as 1, 2, 3", SC'11), restated in numpy so that the device's
# draws can be checked value for value.
# --------------------------------------------------------------------------
def philox4x32(key, ctr_lo, ctr_hi):
    """Blocks for uint64 arrays key / ctr_lo / ctr_hi (broadcast together) -> uint32 [..., 4]."""
    key, ctr_lo, ctr_hi = np.broadcast_arrays(np.asarray(key, np.uint64), np.asarray(ctr_lo, np.uint64), np.asarray(ctr_hi, np.uint64))
    m32 = np.uint64(0xFFFFFFFF)
    c = [ctr_lo & m32, ctr_lo >> np.uint64(32), ctr_hi & m32, ctr_hi >> np.uint64(32)]
    k0, k1 = key & m32, key >> np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=-1).astype(np.uint32)


def _u01(a, b):
    """53-bit uniform in [0, 1) from two 32-bit words (numpy's random_sample construction)."""
    return ((a.astype(np.uint64) >> np.uint64(5)) * np.uint64(67108864) + (b.astype(np.uint64) >> np.uint64(6))).astype(np.float64) / 9007199254740992.0


def device_reset_draw(seed, b, episode, e, lim, attempt=0):
    """Position of entity e of world(s) b: block (key seed, counter (b, episode * 256 + e + attempt << 48)),
    mapped like numpy's uniform(-lim, lim)."""
    r = philox4x32(seed, b, np.uint64(episode * 256 + e) + (np.uint64(attempt) << np.uint64(48)))
    return np.stack([-lim + (lim + lim) * _u01(r[..., 0], r[..., 1]), -lim + (lim + lim) * _u01(r[..., 2], r[..., 3])], axis=-1)


def device_aux_words(seed, b, episode):
    """The block the landmark indices / stream key of world(s) b come from: counter (b, episode * 256 + 255)."""
    return philox4x32(seed, b, np.uint64(episode * 256 + 255))


def device_noise_pair(seed, b, step, k):
    """Standard-normal pair k of world b at step number `step` of the device's noise stream: Box-Muller on the
    Philox block with key seed ^ "noise" and counter (b, step * 256 + k).  b may be an array."""
    r = philox4x32(np.uint64(0x6E6F697365000000) ^ np.uint64(seed), b, np.uint64(step) * np.uint64(256) + np.uint64(k))
    u1 = 1.0 - _u01(r[..., 0], r[..., 1])         # (0, 1]
    u2 = _u01(r[..., 2], r[..., 3])
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)


def device_noise(spec: Spec, seed, step, B, env_base=0):
    """(noise_u [B, A, 2], noise_c [B, A, dim_c]) of step number `step`: pair i is agent i's force noise, the
    comm values of the speakers follow in agent order, packed two to a pair from pair A on.  World b of the batch
    draws as world env_base + b of the whole job."""
    b = np.arange(B, dtype=np.uint64) + np.uint64(env_base)
    nu = np.zeros((B, spec.A, 2))
    nc = np.zeros((B, spec.A, spec.dim_c))
    for i in range(spec.A):
        nu[:, i, 0], nu[:, i, 1] = device_noise_pair(seed, b, step, i)
    j = 0
    for i in range(spec.A):
        if spec.silent[i]:
            continue
        for c in range(spec.dim_c):
            pair = device_noise_pair(seed, b, step, spec.A + j // 2)
            nc[:, i, c] = pair[j % 2]
            j += 1
    return nu, nc


def device_respawn_draw(key, b, n, lim=0.95):
    """n-th respawn position of world(s) b with stream key `key` (int32 as stored in aux row 1)."""
    k = np.uint64(0x5EED5EED00000000) | (np.asarray(key).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    r = philox4x32(k, b, np.asarray(n).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    return np.stack([-lim + (lim + lim) * _u01(r[..., 0], r[..., 1]), -lim + (lim + lim) * _u01(r[..., 2], r[..., 3])], axis=-1)


def pp_device_reset(spec: Spec, seed, episode, B, env_base=0):
    """simple_predator_prey worlds as the device reset draws them (foods redrawn until clear of the
    barrier and the forests, simple_predator_prey.py:139-144).  Returns (pos [B, E, 2], stream key [B] int32)."""
    A = spec.A
    b = np.arange(B, dtype=np.uint64) + np.uint64(env_base)
    pos = np.stack([device_reset_draw(seed, b, episode, e, 1.0 if e < A else 0.9) for e in range(spec.E)], axis=1)
    obst = [A, A + spec.forests[0], A + spec.forests[1]]
    for f in spec.food:
        e = A + f
        for attempt in range(1, 65):
            bad = np.zeros(B, bool)
            for o in obst:
                bad |= _dist(pos[:, e, :] - pos[:, o, :]) < spec.size[e] + spec.size[o]
            if not bad.any():
                break
            pos[bad, e, :] = device_reset_draw(seed, b[bad], episode, e, 0.9, attempt)
    key = device_aux_words(seed, b, episode)[:, 0].astype(np.int32)
    return pos, key


# --------------------------------------------------------------------------
# consensus (multiagent/scenarios/consensus.py) on the new_core physics.  Beside pos / vel the world
# carries his_pos / his_vel [B, A, 10, 2]: the last ten positions and velocities of every agent,
# oldest first (state.p_his_pos / p_his_vel, slid by World.integrate_state, new_core.py:252-254).
# --------------------------------------------------------------------------
def consensus_neighbors(A: int, i: int):
    """neighbors(), consensus.py:79-83: the agents sharing an edge of the ring comm_matrix (:15-22), ascending."""
    return sorted({(i - 1) % A, (i + 1) % A} - {i})


def consensus_observe(spec: Spec, his_pos, his_vel) -> List[np.ndarray]:
    """observation(), consensus.py:122-140: neighbours' velocity-history offsets, then position-history
    offsets, flattened; cast to float32."""
    B = his_pos.shape[0]
    out = []
    for i in range(spec.A):
        nbs = consensus_neighbors(spec.A, i)
        parts = [his_vel[:, n] - his_vel[:, i] for n in nbs] + [his_pos[:, n] - his_pos[:, i] for n in nbs]
        out.append(np.concatenate(parts, axis=1).reshape(B, -1).astype(np.float32))
    return out


def consensus_reward(spec: Spec, pos) -> np.ndarray:
    """reward(), consensus.py:91-113 (before sharing)."""
    B = pos.shape[0]
    rew = np.zeros((B, spec.A))
    for i in range(spec.A):
        d = [_dist(pos[:, n, :] - pos[:, i, :]) for n in consensus_neighbors(spec.A, i)]
        r = 0 - (np.abs(d[0] - 0.8) + np.abs(d[1] - 0.8)) * 0.1
        if i == 0:
            r = r - np.abs(_dist(pos[:, 0, :]) - 0.8) * 0.1
        for p in range(2):
            r = r - 2 * _bound(np.abs(pos[:, i, p]))
        rew[:, i] = r
    return rew


def consensus_env_step(spec: Spec, pos, vel, his_pos, his_vel, actions):
    """MultiAgentEnv.step for consensus.  Returns (pos, vel, his_pos, his_vel, obs_list, shared rew, own rew)."""
    u, c = decode_actions(spec, actions)
    pos, vel, _ = world_step(spec, pos, vel, u, c)
    his_pos = np.concatenate([his_pos[:, :, 1:], pos[:, :spec.A, None, :]], axis=2)
    his_vel = np.concatenate([his_vel[:, :, 1:], vel[:, :, None, :]], axis=2)
    own = consensus_reward(spec, pos)
    return pos, vel, his_pos, his_vel, consensus_observe(spec, his_pos, his_vel), share(spec, own), own


# --------------------------------------------------------------------------
# collector (multiagent/scenarios/collector.py) on the new_core physics: collectors 0-5 pick up
# treasures of two types and carry them to the deposit of that type (agents 6, 7).  Beside pos / vel
# the world carries holding [B, 6] (-1: nothing, else the type), ttype [B, 6] and alive [B, 6] of the
# treasures.  The callbacks are restated per world in scalar form, following the reference's own
# loops (data-dependent sorting and branching); distances are those of the cache World.step fills.
# --------------------------------------------------------------------------
SCN_COLLECTOR = "collector"
COL_N_COLLECTORS = 6


def spec_collector() -> Spec:
    """collector.make_world, collector.py:8-71: 6 collectors (size .05, mass 1) and 2 deposits (size
    .075, mass 2.25), all colliding ghosts with accel 1.5 and max_speed 1; 6 treasures (size .025) that
    neither collide nor move; no walls (num_walls = 0)."""
    A, L = 8, 6
    col = np.arange(A) < COL_N_COLLECTORS
    return Spec(SCN_COLLECTOR, A, L, 2, np.concatenate([np.where(col, 0.05, 0.075), np.full(L, 0.025)]),
                np.array([True] * A + [False] * L), np.array([True] * A + [False] * L), np.concatenate([np.where(col, 1.0, 2.25), np.ones(L)]),
                sens=np.full(A, 1.5), max_speed=np.ones(A), silent=np.ones(A, bool), adversary=np.zeros(A, bool), leader=np.zeros(A, bool),
                ghost=np.ones(A, bool), new_core=True, accel=np.full(A, 1.5))


def _col_cache(spec, pos1):
    """cached_dist_vect / cached_dist_mag of one world (new_core.py:147-170): vect[a, b] = p_a - p_b."""
    vect = pos1[:, None, :] - pos1[None, :, :]
    mag = np.sqrt(vect[..., 0] * vect[..., 0] + vect[..., 1] * vect[..., 1])
    return vect, mag


def _col_hit(spec, mag, a, b):
    return bool(mag[a, b] < spec.size[a] + spec.size[b])


def collector_observe1(spec, pos1, vel1, holding1, ttype1):
    """observation() of every agent of one world, collector.py:257-295 (cast to float32)."""
    A, L, NC = spec.A, spec.L, COL_N_COLLECTORS
    vect, mag = _col_cache(spec, pos1)
    onehot = lambda v: (np.arange(2) == v)

    def encoding(j):
        return np.concatenate([np.zeros(2), onehot(holding1[j])]) if j < NC else np.concatenate([onehot(j - NC), np.zeros(2)])

    out = []
    for i in range(A):
        others = [j for j in range(A) if j != i]
        closest = sorted(zip(mag[others, i], others))
        treasures = list(range(A, A + L))
        closest_t = sorted(zip(mag[treasures, i], treasures))
        obs = [pos1[i], vel1[i]]
        if i < NC:
            obs.append(onehot(holding1[i]))
        for _, j in closest:
            obs += [vect[j, i], vel1[j], encoding(j)]
        for _, t in closest_t:
            obs += [vect[t, i], onehot(ttype1[t - A])]
        out.append(np.concatenate(obs).astype(np.float32))
    return out


def collector_reward1(spec, pos1, holding1, alive1):
    """reward() of every agent of one world, collector.py:165-255: everything goes to collector 0."""
    A, L, NC = spec.A, spec.L, COL_N_COLLECTORS
    vect, mag = _col_cache(spec, pos1)
    collectors, deposits, treasures = list(range(NC)), list(range(NC, A)), list(range(A, A + L))
    gdr = 0
    for d in deposits:                                                    # :248-255
        gdr += 5 * sum(_col_hit(spec, mag, d, a) for a in collectors if holding1[a] == d - NC)
    gcr = 0
    for t in treasures:                                                   # :239-246
        if alive1[t - A]:
            gcr += 5 * sum(_col_hit(spec, mag, a, t) for a in collectors if holding1[a] < 0)
    glob = gdr + gcr

    def deposit_reward(d):                                                # :177-213
        rew = 0
        dh = [mag[d, a] for a in collectors if holding1[a] == d - NC]
        if len(dh) > 0:
            rew -= 0.1 * min(dh)
        else:
            closest = sorted(zip(mag[collectors, d], collectors))[:A - 1]
            inds = [i for _, i in closest]
            rew -= 0.1 * np.linalg.norm(vect[inds, d].mean(axis=0))
        rew += gdr
        for a in collectors:
            if holding1[a] < 0:
                rew -= 0.1 * min([mag[t, a] for t in treasures if alive1[t - A]])
            else:
                rew -= 0.1 * min([mag[a, dd] for dd in deposits if dd - NC == holding1[a]])
            rew += glob
        return rew

    def collector_reward(a):                                              # :215-229
        return 0 - 5 * sum(_col_hit(spec, mag, a, o) for o in collectors if o != a)

    main = 0.0
    main += sum([deposit_reward(d) for d in deposits])
    main += sum([collector_reward(a) for a in collectors])
    return np.array([main] + [0.0] * (A - 1))


def collector_info1(spec, pos1, holding1):
    """benchmark_data() of every agent of one world, collector.py:132-155: (collected, deposited, collisions)."""
    A, L, NC = spec.A, spec.L, COL_N_COLLECTORS
    _, mag = _col_cache(spec, pos1)
    out = np.zeros((A, 3), np.int64)
    for i in range(A):
        if i < NC:
            if holding1[i] < 0:
                out[i, 0] = sum(_col_hit(spec, mag, t, i) for t in range(A, A + L))
            out[i, 2] = sum(_col_hit(spec, mag, a, i) for a in range(NC) if a != i)
        else:
            out[i, 1] = sum(_col_hit(spec, mag, a, i) for a in range(NC) if holding1[a] == i - NC)
    return out


def collector_post_step1(spec, pos1, holding1, ttype1, alive1, respawn_pos1, respawn_type1):
    """post_step(), collector.py:84-110, on one world (arrays are modified in place): a live treasure is
    taken by the first empty-handed collector touching it and parked at the origin; a dead one respawns
    (respawn_prob 1) at respawn_pos1 with respawn_type1; a collector touching the deposit of the type it
    holds hands over.  Distances are those cached before any treasure moved."""
    A, L, NC = spec.A, spec.L, COL_N_COLLECTORS
    _, mag = _col_cache(spec, pos1)
    for l in range(L):
        if alive1[l]:
            for a in range(NC):
                if holding1[a] < 0 and _col_hit(spec, mag, A + l, a):
                    alive1[l] = False
                    holding1[a] = ttype1[l]
                    pos1[A + l] = (-0.0, -0.0)
                    break
        else:
            pos1[A + l] = respawn_pos1[l]
            ttype1[l] = respawn_type1[l]
            alive1[l] = True
    for a in range(NC):
        if holding1[a] >= 0:
            for d in range(NC, A):
                if d - NC == holding1[a] and _col_hit(spec, mag, a, d):
                    holding1[a] = -1


def collector_env_step(spec: Spec, pos, vel, actions, holding, ttype, alive, respawn_pos, respawn_type):
    """MultiAgentEnv.step with post_step as post_step_callback, for B worlds.  Returns
    (pos, vel, obs [B, 684] float32, rew [B, A], info [B, A, 3], holding, ttype, alive)."""
    u, c = decode_actions(spec, actions)
    pos, vel, _ = world_step(spec, pos, vel, u, c)
    pos, holding, ttype, alive = pos.copy(), holding.copy(), ttype.copy(), alive.copy()
    B = pos.shape[0]
    obs = np.stack([np.concatenate(collector_observe1(spec, pos[b], vel[b], holding[b], ttype[b])) for b in range(B)])
    rew = np.stack([collector_reward1(spec, pos[b], holding[b], alive[b]) for b in range(B)])
    info = np.stack([collector_info1(spec, pos[b], holding[b]) for b in range(B)])
    for b in range(B):
        collector_post_step1(spec, pos[b], holding[b], ttype[b], alive[b], respawn_pos[b], respawn_type[b])
    return pos, vel, obs, rew, info, holding, ttype, alive


def device_respawn_type(key, b, n):
    """Type bit drawn with the n-th respawn of world(s) b: bit 0 of the block's first word."""
    k = np.uint64(0x5EED5EED00000000) | (np.asarray(key).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    r = philox4x32(k, b, np.asarray(n).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))
    return (r[..., 0] & np.uint32(1)).astype(np.int64)


def collector_device_reset(spec: Spec, seed, episode, B, env_base=0):
    """collector worlds as the device reset draws them: agents U(-1,1)^2, treasures U(-0.95,0.95)^2, treasure
    types = bits 0..5 of word 1 of the aux block, stream key = word 0.  Returns (pos, ttype [B, 6], key [B])."""
    b = np.arange(B, dtype=np.uint64) + np.uint64(env_base)
    pos = np.stack([device_reset_draw(seed, b, episode, e, 1.0 if e < spec.A else 0.95) for e in range(spec.E)], axis=1)
    w = device_aux_words(seed, b, episode)
    ttype = ((w[:, 1, None].astype(np.int64) >> np.arange(6)) & 1)
    return pos, ttype, w[:, 0].astype(np.int32)


# --------------------------------------------------------------------------
# Generalised advantage estimation as the reference's runner computes it
# (multiagent/backup/runner_old.py:58-80).
# --------------------------------------------------------------------------
def gae_reference(rewards, values, dones, last_values, last_dones, gamma, lam):
    """runner_old.py:66-80 as written.  rewards / values float32 [T, B, A], dones bool [T, B] (the flag
    step t returned), last_values float32 [B, A], last_dones bool [B] (the env's current flags).  One
    `lastgaelam` runs through the agent loop, the mask of step t comes from dones[t+1], the scan is
    float64 (1.0 - bool array) and the results are stored as float32.  Returns (advs, returns)."""
    rewards = np.asarray(rewards, np.float32)
    values = np.asarray(values, np.float32)
    T, B, A = rewards.shape
    advs = np.zeros_like(rewards)
    rets = np.zeros_like(rewards)
    lastgaelam = 0
    for t in reversed(range(T)):
        for i in range(A):
            if t == T - 1:
                nextnonterminal = 1.0 - np.asarray(last_dones)
                nextvalues = np.asarray(last_values, np.float32)[:, i]
            else:
                nextnonterminal = 1.0 - np.asarray(dones[t + 1])
                nextvalues = values[t + 1][:, i]
            delta = rewards[t][:, i] + gamma * nextvalues * nextnonterminal - values[t][:, i]
            advs[t][:, i] = lastgaelam = delta + gamma * lam * nextnonterminal * lastgaelam
            rets[t][:, i] = advs[t][:, i] + values[t][:, i]
    return advs, rets


def gae_standard(rewards, values, dones, last_values, gamma, lam):
    """Per-agent scan with the mask of step t taken from dones[t] (the flag returned with that step's
    reward); same float32 inputs / float64 scan / float32 outputs."""
    rewards = np.asarray(rewards, np.float32)
    values = np.asarray(values, np.float32)
    T, B, A = rewards.shape
    advs = np.zeros_like(rewards)
    rets = np.zeros_like(rewards)
    last = np.zeros((B, A))
    for t in reversed(range(T)):
        nnt = (1.0 - np.asarray(dones[t]))[:, None]
        nv = np.asarray(last_values, np.float32) if t == T - 1 else values[t + 1]
        delta = rewards[t] + (np.float32(gamma) * nv) * nnt - values[t]
        last = delta + gamma * lam * nnt * last
        advs[t] = last
        rets[t] = advs[t] + values[t]
    return advs, rets
