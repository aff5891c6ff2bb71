"""simple_world_comm: 4 adversaries (agent 0 leads and speaks) hunt 2 good agents among an
obstacle, 2 food items and 2 forests that hide agents (multiagent/scenarios/simple_world_comm.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_WORLD_COMM
    hip_name = "simple_world_comm"

    def make_world(self):
        world = World()
        world.dim_c = 4
        for i in range(6):
            a = Agent()
            a.name, a.collide = "agent %d" % i, True
            a.leader, a.silent, a.adversary = i == 0, i > 0, i < 4
            a.size, a.accel, a.max_speed = (0.075, 3.0, 1.0) if a.adversary else (0.045, 4.0, 1.3)  # simple_world_comm.py:26-29
            world.agents.append(a)

        def group(kind, n, size, collide):
            out = []
            for i in range(n):
                lm = Landmark()
                lm.name, lm.collide, lm.movable, lm.size, lm.boundary = "%s %d" % (kind, i), collide, False, size, False
                out.append(lm)
            return out

        obstacles = group("landmark", 1, 0.2, True)
        world.food = group("food", 2, 0.03, False)
        world.forests = group("forest", 2, 0.3, False)
        world.landmarks = obstacles + world.food + world.forests        # order of simple_world_comm.py:52-53
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_world_comm.py:101-113: agents U(-1,1)^2; all 5 landmarks U(-0.9,0.9)^2, then food
        and forests are drawn again (same draw count and order as the reference)."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = (np.array([0.95, 0.45, 0.45]) if a.adversary else np.array([0.45, 0.95, 0.45])) - (0.3 if a.leader else 0.0)
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for group, color in ((world.landmarks, [0.25, 0.25, 0.25]), (world.food, [0.15, 0.15, 0.65]), (world.forests, [0.6, 0.9, 0.6])):
            for lm in group:
                lm.color = np.array(color)
                lm.state.p_pos = rng.uniform(-0.9, +0.9, world.dim_p)
                lm.state.p_vel = np.zeros(world.dim_p)

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]
