"""``multiagent.core`` names (World, Agent, Landmark, ...) backed by the GPU-stepped classes."""
from mpe_hip.world import *  # noqa: F401,F403
