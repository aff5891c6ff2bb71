"""simple_predator_prey: three heavy adversaries hunt four good agents, whose members carry food to
their leader; forests hide agents, walls close the arena (multiagent/scenarios/simple_predator_prey.py,
on the physics of multiagent/new_core.py).

The batched form (``BatchedEnv("simple_predator_prey", ...)``) runs the whole step, ``post_step``
included, in one kernel.  This single-world plugin keeps the reference's contract: the env calls
``post_step`` only if it was given as ``post_step_callback`` (``make_env`` does not pass it).  Given, the
whole step is one fused ``mpe_step`` launch; not given, the step is World.step on the GPU and the
callbacks through one ``mpe_observe`` launch.  ``post_step`` itself is always the kernel's (``mpe_post_step``
when called on its own): this file only maps the kernel's state word onto the objects.
"""
import zlib

import numpy as np

from .. import _lib
from ..new_core import Agent, Landmark, Wall, World
from ..scenario import HipScenario


class Scenario(HipScenario):
    scenario_id = _lib.SCN_PREDATOR_PREY
    hip_name = "simple_predator_prey"
    obs_dtype = np.float32       # simple_predator_prey.py:349
    dynamic = True

    def make_world(self):
        world = World()
        world.cache_dists = True
        world.dim_c = 3
        n_adv, n_good = 3, 4
        world.comm_matrix = np.array([[1, -1, 0, 0, 0, 0, 0], [0, 1, -1, 0, 0, 0, 0], [0, 0, 0, 1, -1, 0, 0],
                                      [0, 0, 0, 0, 1, -1, 0], [0, 0, 0, 0, 0, 1, -1]], dtype=np.float32)
        for i in range(n_adv + n_good):
            a = Agent()
            a.i, a.name, a.collide, a.silent, a.holding = i, "agent %d" % i, True, True, False
            a.leader = i == 0 or i == n_adv
            a.adversary = i < n_adv
            a.size, a.initial_mass, a.max_speed = (0.065, 2.25, 1.0) if a.adversary else (0.045, 1.0, 1.3)
            world.agents.append(a)

        def landmarks(n, first, name, size, collide):
            out = []
            for k in range(n):
                lm = Landmark()
                lm.i, lm.name, lm.collide, lm.movable, lm.size, lm.boundary = first + k, "%s %d" % (name, k), collide, False, size, False
                out.append(lm)
            return out

        A = len(world.agents)
        world.barrier = landmarks(1, A, "landmark", 0.2, True)
        world.foods = landmarks(3, A + 1, "food", 0.03, False)
        for f in world.foods:
            f.alive, f.respawn_prob = True, 1.0
        world.forests = landmarks(2, A + 4, "forest", 0.25, False)
        world.landmarks = world.barrier + world.foods + world.forests
        for i in range(4):
            w = Wall(orient="H" if i % 2 == 0 else "V", axis_pos=-1.2 if i < 2 else 1.2, endpoints=(-1.2, 1.2), width=0.4)
            w.name = "wall %d" % i
            world.walls.append(w)
        self.reset_world(world, np.random)
        return world

    def reset_world(self, world, np_random=None):
        """simple_predator_prey.py:109-146: agents U(-1,1)^2, barrier and forests U(-0.9,0.9)^2, every
        food redrawn in U(-0.9,0.9)^2 until it touches neither; all food alive, nobody holds any."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = (np.array([0.95, 0.45, 0.45]) if a.adversary else np.array([0.85, 0.85, 0.85])) - (0.3 if a.leader else 0.0)
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
            a.holding = False
        for group, color in ((world.barrier, [0.25, 0.25, 0.25]), (world.forests, [0.6, 0.9, 0.6])):
            for lm in group:
                lm.color = np.array(color)
                lm.state.p_pos = rng.uniform(-0.9, +0.9, world.dim_p)
                lm.state.p_vel = np.zeros(world.dim_p)

        def touches(a, b):
            return np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos))) < a.size + b.size

        for f in world.foods:
            f.color = np.array([0.15, 0.15, 0.65])
            f.state.p_pos = rng.uniform(-0.9, +0.9, world.dim_p)
            f.state.p_vel = np.zeros(world.dim_p)
            f.alive = True
            while any(touches(f, o) for o in world.barrier + world.forests):
                f.state.p_pos = rng.uniform(-0.9, +0.9, world.dim_p)
        world.calculate_distances()
        # the device's respawn stream of this episode, keyed by the drawn state (no extra draw from rng)
        world.respawn_stream = [zlib.crc32(np.array([e.state.p_pos for e in world.entities]).tobytes()) & 0x7fffffff, 0]

    def is_collision(self, agent1, agent2, world):
        return bool(world.cached_dist_mag[agent1.i, agent2.i] < agent1.size + agent2.size)

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]

    # post_step (simple_predator_prey.py:83-107) is HipScenario.post_step: the mpe_post_step kernel applies
    # the pick-up / respawn / hand-over rules; absorb() below only copies its result onto the objects.
    def absorb(self, world, aux, landmark_pos, np_random=None):
        flags = aux[0]
        for k, f in enumerate(world.foods):
            was_alive, f.alive = f.alive, bool(flags >> k & 1)
            f.state.p_pos = np.array(landmark_pos[1 + k])           # landmarks: barrier, foods, forests
            if f.alive and not was_alive and np_random is not None:
                np_random.uniform()                                 # the respawn_prob test (1.0: always passes)
                f.state.p_pos = np_random.uniform(low=-0.95, high=0.95, size=world.dim_p)
        for k, m in enumerate(world.agents[4:]):
            m.holding = bool(flags >> (3 + k) & 1)
            m.color = np.array([0.55, 0.55, 0.85] if m.holding else [0.85, 0.85, 0.85])     # display only
        world.respawn_stream = [aux[1], aux[2]]

    def goal_index(self, world):
        """The aux rows of the one-world batch: the flags the kernel reads (bit f: food f alive, bit 3+k:
        member k holds food), then key and position of the world's respawn stream."""
        flags = sum(int(bool(f.alive)) << k for k, f in enumerate(world.foods))
        flags |= sum(int(bool(m.holding)) << (3 + k) for k, m in enumerate(world.agents[4:]))
        return [flags] + list(getattr(world, "respawn_stream", [0, 0]))
