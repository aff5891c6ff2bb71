"""simple_spread: N agents cover N landmarks (multiagent/scenarios/simple_spread.py)."""
import numpy as np

from .. import _lib
from ..scenario import HipScenario
from ..world import Agent, Landmark, World


def ring_comm_matrix(n):
    """Incidence matrix of the directed ring 0->1->...->n-1->0 (trainer metadata,
    simple_spread.py:15-22): row k has +1 at k and -1 at k+1 (mod n)."""
    m = np.zeros((n, n), dtype=np.float32)
    for k in range(n):
        m[k, k] += 1.0
        m[k, (k + 1) % n] -= 1.0
    return m


class Scenario(HipScenario):
    scenario_id = _lib.SCN_SPREAD
    hip_name = "simple_spread"
    obs_dtype = np.float32          # the reference casts this observation (simple_spread.py:130)

    def __init__(self, num_agents=3, num_landmarks=3):
        super().__init__()
        self.num_agents, self.num_landmarks = num_agents, num_landmarks

    def make_world(self):
        world = World()
        world.dim_c = 2
        world.comm_matrix = ring_comm_matrix(self.num_agents)
        for i in range(self.num_agents):
            a = Agent()
            a.name, a.collide, a.silent, a.size = "agent %d" % i, True, True, 0.15
            world.agents.append(a)
        for i in range(self.num_landmarks):
            lm = Landmark()
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
            world.landmarks.append(lm)
        self.reset_world(world, np.random)
        return world

    def reset_world(self, world, np_random=None):
        """Draw order of simple_spread.py:48-54: every agent, then every landmark, U(-1,1)^2."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.35, 0.35, 0.85])
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.color = np.array([0.25, 0.25, 0.25])
            lm.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)
        world.t = 0

    def benchmark_data(self, agent, world):
        """(rew, collisions, min_dists, occupied_landmarks), simple_spread.py:58-74."""
        _, _, _, ii, ff = self.evaluate(world)
        i, A = world.agents.index(agent), len(world.agents)
        return (float(ff[1 + i]), int(ii[i]), float(ff[0]), int(ii[A]))
