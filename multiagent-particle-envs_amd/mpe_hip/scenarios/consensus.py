"""consensus: six agents on a ring graph try to settle 0.8 apart from their two neighbours, each seeing
only the neighbours' last ten positions and velocities relative to its own
(multiagent/scenarios/consensus.py, on the physics of multiagent/new_core.py).

The batched form (``BatchedEnv("consensus", ...)``) keeps the history as a ring in device memory and
runs the whole step in one kernel.  So does this single-world plugin through ``MultiAgentEnv.step``: the
world's state and history go up, one fused ``mpe_step`` launch, then one ``mpe_observe`` launch on the new
state for the agents' own rewards (a step leaves only their sum), and everything comes down in one copy.
"""
import numpy as np

from .. import _lib
from ..new_core import Agent, Wall, World
from ..scenario import HipScenario


class Scenario(HipScenario):
    scenario_id = _lib.SCN_CONSENSUS
    hip_name = "consensus"
    obs_dtype = np.float32       # consensus.py:140
    has_info = True              # benchmark_data is defined (and constant: no landmarks, no collisions)
    fused_info = False
    rescore_after_step = True    # a step's rew is the shared sum; info_n['r<i>'] wants the agents' own rewards

    def make_world(self):
        world = World()
        world.dim_c = 2
        n = 6
        ring = np.zeros((n, n), np.float32)             # edge e joins agents e and e + 1 (consensus.py:15-22)
        for e in range(n - 1):
            ring[e, e], ring[e, e + 1] = 1, -1
        ring[n - 1, 0], ring[n - 1, n - 1] = -1, 1
        world.comm_matrix = ring
        world.collaborative = True
        for i in range(n):
            a = Agent()
            a.id, a.name, a.collide, a.silent, a.ghost, a.size = i, "agent %d" % i, False, True, True, 0.03
            world.agents.append(a)
        for i in range(4):
            w = Wall(orient="H" if i % 2 == 0 else "V", axis_pos=-1.2 if i < 2 else 1.2, endpoints=(-1.2, 1.2), width=0.4)
            w.name = "wall %d" % i
            world.walls.append(w)
        self.reset_world(world, np.random)
        return world

    def reset_world(self, world, np_random=None):
        """consensus.py:46-59: agents U(-1,1)^2 at rest; every history row starts as the initial state."""
        rng = np_random if np_random is not None else np.random
        for a in world.agents:
            a.color = np.array([0.35, 0.35, 0.85])
            a.state.p_pos = rng.uniform(-1, +1, world.dim_p)
            a.state.p_his_pos = np.tile(a.state.p_pos, [10, 1])
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.p_his_vel = np.tile(a.state.p_vel, [10, 1])
            a.state.c = np.zeros(world.dim_c)

    def neighbors(self, agent, world):
        """consensus.py:79-83."""
        edges = np.where(world.comm_matrix[:, agent.id] != 0)[0]
        nbs = np.unique(np.hstack([np.where(world.comm_matrix[e] != 0)[0] for e in edges]))
        return [world.agents[nb] for nb in nbs if nb != agent.id]

    def benchmark_data(self, agent, world):
        """consensus.py:61-77 with no landmarks and non-colliding agents."""
        return (0, 0, 0, 0)

    # -- device plumbing: the history travels in the comm buffer as a ring whose oldest slot is aux row 0 ----
    def goal_index(self, world):
        return 0

    def state_rows(self, world):
        pos, vel, _ = world.pack_state()
        hp = np.array([a.state.p_his_pos for a in world.agents], dtype=np.float64)      # [A, 10, 2], oldest first
        hv = np.array([a.state.p_his_vel for a in world.agents], dtype=np.float64)
        ring = np.concatenate([hp, hv], axis=2).transpose(1, 0, 2).reshape(-1)          # rows (slot, agent, component)
        return pos, vel, ring, 0

    def _outputs(self, ow):
        # mpe_observe leaves what the reward callback returns (the agents' own rewards) in rew
        return ow.v["obs"][0].copy(), ow.v["rew"][0].copy(), None, None
