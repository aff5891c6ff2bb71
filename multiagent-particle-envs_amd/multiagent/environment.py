"""``multiagent.environment.MultiAgentEnv`` backed by mpe_hip."""
from mpe_hip.env import MultiAgentEnv  # noqa: F401
