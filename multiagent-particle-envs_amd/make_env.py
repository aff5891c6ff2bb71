"""``from make_env import make_env`` as in the reference (make_env.py:15-44)."""
from mpe_hip.env import make_env  # noqa: F401
