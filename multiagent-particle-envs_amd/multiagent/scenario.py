"""``multiagent.scenario.BaseScenario`` backed by mpe_hip."""
from mpe_hip.scenario import BaseScenario  # noqa: F401
