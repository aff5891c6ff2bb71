"""mpe_hip: MI355X-native batched multi-agent particle worlds."""
from ._lib import MpeError  # noqa: F401
from .spec import (WorldSpec, adversary_spec, collector_spec, consensus_spec, crypto_spec, make_spec, physics_spec, predator_prey_spec, push_spec, reference_spec,  # noqa: F401
                   simple_spec, speaker_listener_spec, spread_spec, tag_spec, world_comm_spec)
from .batch import BatchedEnv  # noqa: F401
from .world import Action, Agent, AgentState, Entity, EntityState, Landmark, Wall, World  # noqa: F401
from .scenario import BaseScenario, HipScenario  # noqa: F401
from .env import MultiAgentEnv, make_env  # noqa: F401
from .train import BatchedVecEnv, gae  # noqa: F401
