"""simple_speaker_listener: the speaker sees the colour of the goal landmark and cannot move, the
listener hears it and has to go there (multiagent/scenarios/simple_speaker_listener.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_SPEAKER_LISTENER
    hip_name = "simple_speaker_listener"
    has_info = False         # the reference's benchmark_data raises NameError (simple_speaker_listener.py:61)
    fused_info = True        # the agents' own rewards come back beside the shared one

    def make_world(self):
        world = World()
        world.dim_c = 3
        world.collaborative = True
        for i in range(2):
            a = Agent()
            a.name, a.collide, a.size = "agent %d" % i, False, 0.075
            world.agents.append(a)
        world.agents[0].movable = False      # speaker
        world.agents[1].silent = True        # listener
        for i in range(3):
            lm = Landmark()
            lm.name, lm.collide, lm.movable, lm.size = "landmark %d" % i, False, False, 0.04
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_speaker_listener.py:33-57: the listener's target landmark is drawn first, then agents
        and landmarks U(-1,1)^2.  The reference draws from the global numpy generator; a generator
        passed by MultiAgentEnv.reset is used when given."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.goal_a = None
            a.goal_b = None
            a.color = np.array([0.25, 0.25, 0.25])
        world.agents[0].goal_a = world.agents[1]
        world.agents[0].goal_b = world.landmarks[int(rng.choice(len(world.landmarks)))]
        for i, lm in enumerate(world.landmarks):
            lm.color = np.full(3, 0.15)
            lm.color[i] = 0.65
        world.agents[0].goal_a.color = world.agents[0].goal_b.color + np.array([0.45, 0.45, 0.45])
        for a in world.agents:
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def goal_index(self, world):
        return world.landmarks.index(world.agents[0].goal_b)
