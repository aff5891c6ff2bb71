"""simple_crypto: Alice (agent 2) tells Bob (agent 1) which landmark is the goal using a key they
share, while Eve (agent 0) listens in (multiagent/scenarios/simple_crypto.py).  Nobody moves; every
agent's action is one comm word."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


class CryptoAgent(Agent):
    def __init__(self):
        super().__init__()
        self.key = None


class Scenario(HipScenario):
    scenario_id = _lib.SCN_CRYPTO
    hip_name = "simple_crypto"

    def make_world(self):
        world = World()
        world.dim_c = 4
        for i in range(3):
            a = CryptoAgent()
            a.name, a.collide, a.movable = "agent %d" % i, False, False
            a.adversary = i < 1
            a.speaker = i == 2
            world.agents.append(a)
        for i in range(2):
            lm = Landmark()
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
            world.landmarks.append(lm)
        self.reset_world(world)
        return world

    def reset_world(self, world, np_random=None):
        """simple_crypto.py:47-75: one-hot landmark colours, the goal landmark and the key landmark are
        drawn, then agents and landmarks U(-1,1)^2."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.75, 0.25, 0.25]) if a.adversary else np.array([0.25, 0.25, 0.25])
            a.key = None
        for i, lm in enumerate(world.landmarks):
            lm.color = np.zeros(world.dim_c)
            lm.color[i] += 1
        goal = world.landmarks[int(rng.choice(len(world.landmarks)))]
        world.agents[1].color = goal.color
        world.agents[2].key = world.landmarks[int(rng.choice(len(world.landmarks)))].color
        for a in world.agents:
            a.goal_a = goal
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def goal_index(self, world):
        return [world.landmarks.index(world.agents[0].goal_a), int(np.argmax(world.agents[2].key))]

    def benchmark_data(self, agent, world):
        """simple_crypto.py:78-80: (agent.state.c, agent.goal_a.color)."""
        info_f = self.evaluate(world)[4]
        d, i = world.dim_c, world.agents.index(agent)
        return info_f[2 * d * i:2 * d * i + d].copy(), info_f[2 * d * i + d:2 * d * (i + 1)].copy()

    def good_listeners(self, world):
        return [a for a in world.agents if not a.adversary and not a.speaker]

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]
