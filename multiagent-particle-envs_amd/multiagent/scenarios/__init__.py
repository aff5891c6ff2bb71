"""``multiagent.scenarios.load`` backed by mpe_hip's built-in scenarios."""
from mpe_hip.scenarios import load  # noqa: F401
