"""simple: one agent is rewarded for reaching one landmark (multiagent/scenarios/simple.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_SIMPLE
    hip_name = "simple"
    has_info = False         # simple.py defines no benchmark_data

    def make_world(self):
        world = World()
        a = Agent()
        a.name, a.collide, a.silent = "agent 0", False, True
        world.agents.append(a)
        lm = Landmark()
        lm.name, lm.collide, lm.movable = "landmark 0", False, False
        world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple.py:24-39: agent and landmark U(-1,1)^2.  The reference draws from the global numpy
        generator; a generator passed by MultiAgentEnv.reset is used when given."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.25, 0.25, 0.25])
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.color = np.array([0.75, 0.75, 0.75])
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)
        world.landmarks[0].color = np.array([0.75, 0.25, 0.25])
