"""simple_tag: 3 adversaries chase 1 good agent around 2 obstacles
(multiagent/scenarios/simple_tag.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_TAG
    hip_name = "simple_tag"

    def make_world(self):
        world = World()
        world.dim_c = 2
        for i in range(4):
            a = Agent()
            a.name, a.collide, a.silent = "agent %d" % i, True, True
            a.adversary = i < 3
            a.size, a.accel, a.max_speed = (0.075, 3.0, 1.0) if a.adversary else (0.05, 4.0, 1.3)   # simple_tag.py:22-25
            world.agents.append(a)
        for i in range(2):
            lm = Landmark()
            lm.name, lm.collide, lm.movable, lm.size, lm.boundary = "landmark %d" % i, True, False, 0.2, False
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_tag.py:47-54: agents U(-1,1)^2, obstacles U(-0.9,0.9)^2.  The reference draws from
        the global numpy generator; a generator passed by MultiAgentEnv.reset is used when given."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.85, 0.35, 0.35]) if a.adversary else np.array([0.35, 0.85, 0.35])
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.color = np.array([0.25, 0.25, 0.25])
            lm.state.p_pos = rng.uniform(-0.9, +0.9, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]
