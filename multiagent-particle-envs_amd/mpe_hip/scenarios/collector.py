"""collector: six collectors pick up treasures of two types and carry them to the deposit agent of that
type (multiagent/scenarios/collector.py, on the physics of multiagent/new_core.py).

The batched form (``BatchedEnv("collector", ...)``) runs the whole step, ``post_step`` included, in one
kernel.  This single-world plugin keeps the reference's contract: the env calls ``post_step`` only if
it was given as ``post_step_callback``.  Given, the whole step is one fused ``mpe_step`` launch; not given,
the step is World.step on the GPU and the callbacks of all agents through one ``mpe_observe`` launch.
``post_step`` itself is always the kernel's (``mpe_post_step`` when called on its own): this file only maps
the kernel's state word onto the objects.
"""
import zlib

import numpy as np

from .. import _lib
from ..new_core import Agent, Landmark, World
from ..scenario import HipScenario


class Scenario(HipScenario):
    scenario_id = _lib.SCN_COLLECTOR
    hip_name = "collector"
    obs_dtype = np.float32       # collector.py:295
    dynamic = True

    def make_world(self):
        world = World()
        world.cache_dists = True
        world.dim_c = 2
        n_agents, n_collectors = 8, 6
        world.treasure_types = list(range(n_agents - n_collectors))
        world.treasure_colors = np.array([[0.12, 0.47, 0.71], [1.0, 0.5, 0.05]])      # display only
        ring = np.zeros((n_agents, n_agents), np.float32)           # collector.py:21-28
        for e in range(n_agents - 1):
            ring[e, e], ring[e, e + 1] = 1, -1
        ring[n_agents - 1, 0], ring[n_agents - 1, n_agents - 1] = -1, 1
        world.comm_matrix = ring
        for i in range(n_agents):
            a = Agent()
            a.i, a.name, a.collector = i, "agent %d" % i, i < n_collectors
            if not a.collector:
                a.d_i = i - n_collectors
            a.collide, a.silent, a.ghost, a.holding = True, True, True, None
            a.size, a.initial_mass = (0.05, 1.0) if a.collector else (0.075, 2.25)
            a.accel, a.max_speed = 1.5, 1.0
            world.agents.append(a)
        for i in range(n_collectors):
            lm = Landmark()
            lm.i, lm.name, lm.respawn_prob = i + n_agents, "treasure %d" % i, 1.0
            lm.type, lm.alive, lm.collide, lm.movable, lm.size, lm.boundary = 0, True, False, False, 0.025, False
            world.landmarks.append(lm)
        self.reset_world(world, np.random)
        return world

    def collectors(self, world):
        return [a for a in world.agents if a.collector]

    def deposits(self, world):
        return [a for a in world.agents if not a.collector]

    def treasures(self, world):
        return world.landmarks

    def reset_world(self, world, np_random=None):
        """collector.py:112-130: agents U(-1,1)^2 at rest with empty hands; every treasure gets a type and a
        place in U(-0.95,0.95)^2 and is alive."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.state.p_pos = rng.uniform(low=-1, high=1, size=world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
            a.holding = None
            a.color = np.array([0.85, 0.85, 0.85]) if a.collector else world.treasure_colors[a.d_i] * 0.35
        for lm in world.landmarks:
            lm.type = int(rng.choice(world.treasure_types))
            lm.color = world.treasure_colors[lm.type]
            lm.state.p_pos = rng.uniform(low=-0.95, high=0.95, size=world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)
            lm.alive = True
        world.calculate_distances()
        # the device's respawn stream of this episode, keyed by the drawn state (no extra draw from rng)
        world.respawn_stream = [zlib.crc32(np.array([e.state.p_pos for e in world.entities]).tobytes()) & 0x7fffffff, 0]

    def is_collision(self, agent1, agent2, world):
        return bool(world.cached_dist_mag[agent1.i, agent2.i] < agent1.size + agent2.size)

    # post_step (collector.py:84-110) is HipScenario.post_step: the mpe_post_step kernel applies the
    # pick-up / respawn / deposit rules; absorb() below only copies its result onto the objects.
    def absorb(self, world, aux, landmark_pos, np_random=None):
        word = aux[0]
        for l, t in enumerate(world.landmarks):
            was_alive = t.alive
            t.alive = t.visible = bool(word >> (18 + l) & 1)
            t.type = int(word >> (12 + l) & 1)
            t.state.p_pos = np.array(landmark_pos[l])
            if t.alive and not was_alive and np_random is not None:
                # the caller's generator supplies the respawn, in the reference's order of draws
                np_random.uniform()                                 # the respawn_prob test (1.0: always passes)
                t.state.p_pos = np_random.uniform(low=-0.95, high=0.95, size=world.dim_p)
                t.type = int(np_random.choice(world.treasure_types))
            t.color = world.treasure_colors[t.type]
        for k, a in enumerate(self.collectors(world)):
            held = word >> (2 * k) & 3
            a.holding = None if held == 0 else held - 1
            a.color = np.array([0.85, 0.85, 0.85]) if a.holding is None else 0.85 * world.treasure_colors[a.holding]     # display only
        world.respawn_stream = [aux[1], aux[2]]

    def goal_index(self, world):
        """The aux rows of the one-world batch: the state word the kernel reads (2 bits per collector: 0
        nothing, 1 + type held; bit 12+l: type of treasure l; bit 18+l: treasure l alive), then key and
        position of the world's respawn stream."""
        w = 0
        for k, a in enumerate(self.collectors(world)):
            w |= (0 if a.holding is None else 1 + int(a.holding)) << (2 * k)
        for l, t in enumerate(world.landmarks):
            w |= (int(t.type) & 1) << (12 + l) | int(bool(t.alive)) << (18 + l)
        return [w] + list(getattr(world, "respawn_stream", [0, 0]))

    def benchmark_data(self, agent, world):
        """collector.py:132-155: (collected_treasures, deposited_treasures, collisions)."""
        i = world.agents.index(agent)
        return tuple(int(x) for x in self.evaluate(world)[3][3 * i:3 * i + 3])
