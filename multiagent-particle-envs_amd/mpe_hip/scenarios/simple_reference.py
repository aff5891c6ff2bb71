"""simple_reference: each of two agents knows which landmark the other one should reach and has to
tell it (multiagent/scenarios/simple_reference.py).  Both agents move and speak, so each takes a
``(move, comm)`` action pair."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_REFERENCE
    hip_name = "simple_reference"
    has_info = False         # simple_reference.py defines no benchmark_data
    fused_info = True        # the agents' own rewards come back beside the shared one

    def make_world(self):
        world = World()
        world.dim_c = 10
        world.collaborative = True
        for i in range(2):
            a = Agent()
            a.name, a.collide = "agent %d" % i, False
            world.agents.append(a)
        for i in range(3):
            lm = Landmark()
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_reference.py:26-53: the two target landmarks are drawn first (agent 0's, then agent
        1's), then agents and landmarks U(-1,1)^2."""
        rng = np_random if np_random is not None else np.random
        ag = world.agents
        ag[0].goal_a, ag[0].goal_b = ag[1], world.landmarks[int(rng.choice(len(world.landmarks)))]
        ag[1].goal_a, ag[1].goal_b = ag[0], world.landmarks[int(rng.choice(len(world.landmarks)))]
        for i, lm in enumerate(world.landmarks):
            lm.color = np.full(3, 0.25)
            lm.color[i] = 0.75
        ag[0].goal_a.color = ag[0].goal_b.color
        ag[1].goal_a.color = ag[1].goal_b.color
        for a in ag:
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def goal_index(self, world):
        return [world.landmarks.index(a.goal_b) for a in world.agents]
