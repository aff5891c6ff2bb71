"""simple_adversary: two good agents cover two landmarks so that the adversary cannot tell which one
is the goal (multiagent/scenarios/simple_adversary.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class Scenario(HipScenario):
    scenario_id = _lib.SCN_ADVERSARY
    hip_name = "simple_adversary"

    def make_world(self):
        world = World()
        world.dim_c = 2
        world.num_agents = 3
        for i in range(world.num_agents):
            a = Agent()
            a.name, a.collide, a.silent, a.size = "agent %d" % i, False, True, 0.15
            a.adversary = i < 1
            world.agents.append(a)
        for i in range(world.num_agents - 1):
            lm = Landmark()
            lm.name, lm.collide, lm.movable, lm.size = "landmark %d" % i, False, False, 0.08
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_adversary.py:35-54: the goal landmark is drawn first, then agents and landmarks
        U(-1,1)^2.  The reference draws from the global numpy generator; a generator passed by
        MultiAgentEnv.reset is used when given."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.85, 0.35, 0.35]) if a.adversary else np.array([0.35, 0.35, 0.85])
        for lm in world.landmarks:
            lm.color = np.array([0.15, 0.15, 0.15])
        goal = world.landmarks[int(rng.choice(len(world.landmarks)))]
        goal.color = np.array([0.15, 0.65, 0.15])
        for a in world.agents:
            a.goal_a = goal
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def benchmark_data(self, agent, world):
        """simple_adversary.py:56-65: the adversary's squared distance to the goal; for a good agent
        the tuple of its squared distances to every landmark, then to the goal."""
        info_f = self.evaluate(world)[4]
        L, k = len(world.landmarks), 0
        for a in world.agents:
            n = 1 if a.adversary else L + 1
            if a is agent:
                return float(info_f[k]) if a.adversary else tuple(float(x) for x in info_f[k:k + n])
            k += n
        raise ValueError("agent is not part of this world")

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]
