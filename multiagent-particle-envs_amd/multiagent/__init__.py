"""Import-compatible alias of the reference's ``multiagent`` package, backed by mpe_hip."""
from mpe_hip.env import MultiAgentEnv  # noqa: F401
