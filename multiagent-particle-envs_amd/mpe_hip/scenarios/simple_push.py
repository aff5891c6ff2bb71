"""simple_push: the good agent heads for the goal landmark while the adversary, which is not told
which landmark it is, pushes it away (multiagent/scenarios/simple_push.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_PUSH
    hip_name = "simple_push"
    has_info = False         # simple_push.py defines no benchmark_data

    def make_world(self):
        world = World()
        world.dim_c = 2
        for i in range(2):
            a = Agent()
            a.name, a.collide, a.silent = "agent %d" % i, True, True
            a.adversary = i < 1
            world.agents.append(a)
        for i in range(2):
            lm = Landmark()
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_push.py:34-53: landmark colours, the goal draw, the goal-dependent colour of the good
        agent, then agents and landmarks U(-1,1)^2.  The reference draws from the global numpy
        generator; a generator passed by MultiAgentEnv.reset is used when given."""
        rng = np_random if np_random is not None else np.random
        for i, lm in enumerate(world.landmarks):
            lm.color = np.array([0.1, 0.1, 0.1])
            lm.color[i + 1] += 0.8
            lm.index = i
        goal = world.landmarks[int(rng.choice(len(world.landmarks)))]
        for a in world.agents:
            a.goal_a = goal
            a.color = np.array([0.25, 0.25, 0.25])
            if a.adversary:
                a.color = np.array([0.75, 0.25, 0.25])
            else:
                a.color[goal.index + 1] += 0.5
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)
