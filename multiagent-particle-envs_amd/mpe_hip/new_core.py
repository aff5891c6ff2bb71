"""The data model of the reference's ``multiagent/new_core.py`` with ``World.step`` on the GPU.

Differences from ``multiagent/core.py`` that the step kernel reproduces: the action force is
``(mass * accel, or mass when accel is None) * action.u`` (new_core.py:214) and a contact between
two movable entities is split by their mass ratio (:290-294).  Around the kernel, as in the
reference: entities carry an index ``i`` and a ``ghost`` flag (:50,60), a state with ``p_his_pos`` /
``p_his_vel`` keeps a sliding history of the last positions and velocities (:252-254), and with
``cache_dists`` the pairwise offsets, distances and contacts are stored after every step
(:147-170,204-205) for the scenario's callbacks.
"""
import numpy as np

from .world import Action, Agent, AgentState, Entity, EntityState, Landmark, Wall  # noqa: F401
from .world import World as _CoreWorld

__all__ = ["EntityState", "AgentState", "Action", "Entity", "Landmark", "Agent", "Wall", "World"]


class World(_CoreWorld):
    def __init__(self):
        super().__init__()
        self.new_core = True
        self.cache_dists = False
        self.cached_dist_vect = None
        self.cached_dist_mag = None

    def calculate_distances(self):
        """new_core.py:147-170: offsets p_a - p_b, their norms, and the contact table dist <= size_a + size_b."""
        ents = self.entities
        n = len(ents)
        if self.cached_dist_vect is None:
            self.cached_dist_vect = np.zeros((n, n, self.dim_p))
            size = np.array([e.size for e in ents])
            self.min_dists = size[:, None] + size[None, :]
            np.fill_diagonal(self.min_dists, 0.0)
        pos = np.array([e.state.p_pos for e in ents], dtype=np.float64)
        # every pair at once: pos[a] - pos[b] for a != b (the reference fills [a, b] with the offset and [b, a] with its
        # negation, and b - a == -(a - b) exactly in IEEE arithmetic); the diagonal stays zero
        np.subtract(pos[:, None, :], pos[None, :, :], out=self.cached_dist_vect)
        self.cached_dist_mag = np.linalg.norm(self.cached_dist_vect, axis=2)
        self.cached_collisions = self.cached_dist_mag <= self.min_dists

    def assign_agent_colors(self):
        """new_core.py:172-185 (display only; two fixed ramps stand in for seaborn's palettes)."""
        n_dummies = len([a for a in self.agents if getattr(a, "dummy", False)])
        n_adv = len([a for a in self.agents if getattr(a, "adversary", False)])
        n_good = len(self.agents) - n_adv - n_dummies
        ramp = lambda lo, hi, n: [tuple(np.asarray(lo) + (np.asarray(hi) - np.asarray(lo)) * (k + 1) / (n + 1)) for k in range(n)]
        colors = [(0, 0, 0)] * n_dummies + ramp((0.99, 0.8, 0.6), (0.5, 0.0, 0.0), n_adv) + ramp((0.8, 0.92, 0.77), (0.03, 0.25, 0.5), n_good)
        for color, agent in zip(colors, self.agents):
            agent.color = color

    def step(self):
        """new_core.py:188-205."""
        super().step()
        self.after_step()

    def after_step(self):
        """What new_core.py:204-205,252-254 do once the new state is known: slide the histories, refresh the cache."""
        for e in self.entities:
            if e.movable and hasattr(e.state, "p_his_pos"):
                e.state.p_his_pos = np.vstack([e.state.p_his_pos[1:], e.state.p_pos])
                e.state.p_his_vel = np.vstack([e.state.p_his_vel[1:], e.state.p_vel])
        if self.cache_dists:
            self.calculate_distances()
