"""``multiagent.new_core`` backed by mpe_hip: same classes, ``World.step`` on the GPU."""
from mpe_hip.new_core import Action, Agent, AgentState, Entity, EntityState, Landmark, Wall, World  # noqa: F401
